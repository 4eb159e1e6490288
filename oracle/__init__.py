"""CPU oracle -- TEST INFRASTRUCTURE ONLY (see oracle/oc_oracle.c).  Never imported by
the product package `gym-comm_amd/`."""

#!/bin/bash
# GPU box: the HIP-vs-oracle soaks profiles/ keeps per kernel revision.   tools/soak_set.sh <tag>
# Output: gpurun_out/soak_<tag>/*.txt (last lines = the totals); copy into profiles/<prefix>_parity_soak_*.txt.
# ~6 minutes: base step (specialised, generic), fused wrapper step (specialised x2 seeds, generic),
# the one-launch closed loop against policy kernel + step.
TAG="$1"; OUT=gpurun_out/soak_$TAG; mkdir -p $OUT
cd "$(dirname "$0")/.."
s() { name=$1; shift; timeout -k 10 400 python "$@" > $OUT/$name.txt 2>&1 || echo "FAILED $name"; tail -1 $OUT/$name.txt; }
s spec tests/soak.py 1200 2000 spec
s generic tests/soak.py 600 1000
s wrapper_spec tests/soak_wrapper.py 600 2000 spec
s wrapper_spec_seed2 tests/soak_wrapper.py 600 2000 spec 7
s wrapper_generic tests/soak_wrapper.py 300 1000
s closed_loop tests/soak_closed_loop.py

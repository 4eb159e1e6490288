// Exhaustive check (4.3e9 pairs, ~6 s): the two-FMA quotient used by timestep_of() in
// gym-comm_amd/csrc/oc_kernels.hip equals the correctly rounded fp64 division t / T for every
// 0 <= t <= 65535, 1 <= T <= 65535.   gcc -O2 -ffp-contract=off tools/div_check.c -lm
#include <math.h>
#include <stdio.h>
#include <string.h>
int main() {
  long bad = 0;
  for (int T = 1; T <= 65535; T++) {
    const double dT = (double)T, y = 1.0 / dT;
    for (int t = 0; t <= 65535; t++) {
      const double dt = (double)t;
      const double q0 = dt * y;
      const double r = fma(-dT, q0, dt);
      const double q = fma(r, y, q0);
      const double ref = dt / dT;
      if (memcmp(&q, &ref, 8) != 0) { if (bad < 5) printf("mismatch t=%d T=%d\n", t, T); bad++; }
    }
  }
  printf("mismatches: %ld\n", bad);
  return 0;
}

// Microbenchmark (not product): what does a dependent chain of launches cost on MI355X when
// each launch only moves bytes (row loads + row stores per env, one lane per env, 64-thread
// workgroups, 256 launches captured in a hipGraph) and does no arithmetic?  Compare with
// k_multi_step's measured time per launch (12 row loads, 76 row stores).
//   hipcc --offload-arch=gfx950 -O3 tools/floor_probe.hip -o /tmp/floor_probe && /tmp/floor_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int STORES, int VEC>
__global__ void probe(const int *in, int *out, int n, int loads, int spin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int acc = 0;
  for (int r = 0; r < loads; r++) acc += in[r * n + i];
  for (int k = 0; k < spin; k++) acc = acc * 1664525 + 1013904223;
  if (VEC == 1) {
#pragma unroll
    for (int r = 0; r < STORES; r++) out[r * n + i] = acc + r;            // SoA rows, 256 B per wave store
  } else if (VEC == 0) {
    signed char *o8 = (signed char *)out;                                  // SoA byte rows, 64 B per wave store
#pragma unroll
    for (int r = 0; r < STORES; r++) o8[(size_t)r * n + i] = (signed char)(acc + r);
  } else {
    int4 *o = (int4 *)out + (size_t)i * (STORES / 4);                      // AoS: 16 B per lane per store
#pragma unroll
    for (int r = 0; r < STORES / 4; r++) o[r] = make_int4(acc, acc + r, acc, acc);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int STORES, int VEC>
int run(int n, int *in, int *out, hipStream_t s, int spin) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int k = 0; k < 256; k++)
    hipLaunchKernelGGL((probe<STORES, VEC>), dim3((n + 63) / 64), dim3(64), 0, s, in, out, n, 12, spin);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 4; w++) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a, s));
  for (int w = 0; w < 20; w++) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(b, s));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("n=%6d  12 row loads, %2d words stored per env as %s, %4d dependent VALU ops: %6.2f us per launch\n", n,
         STORES, VEC == 1 ? "row stores (SoA)  " : VEC == 0 ? "BYTE row stores (SoA)" : "dwordx4 stores (AoS)", 2 * spin,
         ms * 1e3 / (20 * 256));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 0;
}

int main() {
  const int sizes[] = {4096, 131072};
  for (int n : sizes) {
    int *in, *out;
    CK(hipMalloc(&in, sizeof(int) * 12 * n));
    CK(hipMalloc(&out, sizeof(int) * 76 * n));
    CK(hipMemset(in, 0, sizeof(int) * 12 * n));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    run<0, 1>(n, in, out, s, 0);
    run<8, 1>(n, in, out, s, 0);
    run<20, 1>(n, in, out, s, 0);
    run<40, 1>(n, in, out, s, 0);
    run<76, 1>(n, in, out, s, 0);
    run<76, 4>(n, in, out, s, 0);
    run<20, 4>(n, in, out, s, 0);
    run<76, 0>(n, in, out, s, 0);
    run<60, 0>(n, in, out, s, 0);
    run<31, 1>(n, in, out, s, 0);    // 16 state/scalar dword rows + 60 obs values packed 4 per dword
    run<76, 1>(n, in, out, s, 500);
    run<0, 1>(n, in, out, s, 500);
    CK(hipFree(in)); CK(hipFree(out));
  }
  return 0;
}

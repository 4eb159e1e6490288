// Microbenchmark (not product): where does the store phase of a step launch go on MI355X?
// Same dependent chain of graph-captured launches as floor_probe.hip (12 row loads, ~1200
// dependent VALU ops, 76 row stores per env), varying
//   LPW   active lanes (= envs) per 64-lane wave: fewer lanes per wave = the same batch spread
//         over more CUs, each store instruction touching fewer bytes;
//   AUX   cache policy bits of the buffer stores (0 default, 1 sc0, 2 nt, 16 sc1, 17 sc0+sc1).
//   hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int STORES, int LPW, int AUX>
__global__ void probe(const int *in, int *out, int n, int loads, int spin) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * LPW + lane;
  if (lane >= LPW || i >= n) return;
  int acc = 0;
  for (int r = 0; r < loads; r++) acc += in[r * n + i];
  for (int k = 0; k < spin; k++) acc = ((acc ^ k) + (acc >> 3)) | 1;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, n * STORES * 4, 0x00020000);
#pragma unroll
  for (int r = 0; r < STORES; r++) __builtin_amdgcn_raw_buffer_store_b32(acc + r, rsrc, i * 4, r * n * 4, AUX);
}

// the same words as [STORES/4] groups of 4 rows, 16 bytes per lane, lanes contiguous
// (layout [group][n][4]): a wave store covers 1 KiB of consecutive memory
template <int STORES, int AUX>
__global__ void probe4(const int *in, int *out, int n, int loads, int spin) {
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  if (i >= n) return;
  int acc = 0;
  for (int r = 0; r < loads; r++) acc += in[r * n + i];
  for (int k = 0; k < spin; k++) acc = ((acc ^ k) + (acc >> 3)) | 1;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, n * STORES * 4, 0x00020000);
  typedef int v4i __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int g = 0; g < STORES / 4; g++) {
    v4i v = {acc + g, acc, acc + 1, acc + 2};
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, i * 16, g * n * 16, AUX);
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int STORES, int LPW, int AUX>
int run(int n, int *in, int *out, hipStream_t s, int spin) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int k = 0; k < 256; k++)
    hipLaunchKernelGGL((probe<STORES, LPW, AUX>), dim3((n + LPW - 1) / LPW), dim3(64), 0, s, in, out, n, 12, spin);
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
  printf("n=%6d  lanes/wave=%2d  waves=%5d  stores=%2d  aux=%2d  valu~%4d: %6.2f us per launch\n", n, LPW,
         (n + LPW - 1) / LPW, STORES, AUX, 3 * spin, ms * 1e3 / (20 * 256));
  fflush(stdout);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 0;
}

template <int STORES, int AUX>
int run4(int n, int *in, int *out, hipStream_t s, int spin) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int k = 0; k < 256; k++)
    hipLaunchKernelGGL((probe4<STORES, AUX>), dim3((n + 63) / 64), dim3(64), 0, s, in, out, n, 12, spin);
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
  printf("n=%6d  [group][n][4] dwordx4 stores  words=%2d  aux=%2d  valu~%4d: %6.2f us per launch\n", n, STORES, AUX,
         3 * spin, ms * 1e3 / (20 * 256));
  fflush(stdout);
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 0;
}

template <int LPW>
int lanes(int n, int *in, int *out, hipStream_t s) {
  run<76, LPW, 0>(n, in, out, s, 0);
  run<76, LPW, 0>(n, in, out, s, 400);
  run<0, LPW, 0>(n, in, out, s, 400);
  return 0;
}

int main() {
  const int sizes[] = {4096, 32768, 131072};
  for (int n : sizes) {
    int *in, *out;
    CK(hipMalloc(&in, sizeof(int) * 12 * n));
    CK(hipMalloc(&out, sizeof(int) * 76 * n));
    CK(hipMemset(in, 0, sizeof(int) * 12 * n));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    lanes<64>(n, in, out, s);
    lanes<32>(n, in, out, s);
    lanes<16>(n, in, out, s);
    lanes<8>(n, in, out, s);
    if (n == 4096) lanes<4>(n, in, out, s);
    run<76, 64, 1>(n, in, out, s, 0);
    run<76, 64, 2>(n, in, out, s, 0);
    run<76, 64, 16>(n, in, out, s, 0);
    run<76, 64, 17>(n, in, out, s, 0);
    run<76, 64, 3>(n, in, out, s, 0);
    run4<76, 0>(n, in, out, s, 0);
    run4<76, 16>(n, in, out, s, 0);
    run4<76, 0>(n, in, out, s, 400);
    run4<76, 16>(n, in, out, s, 400);
    run<76, 64, 16>(n, in, out, s, 400);
    run<8, 64, 0>(n, in, out, s, 0);
    run<8, 64, 2>(n, in, out, s, 0);
    run<8, 64, 17>(n, in, out, s, 0);
    CK(hipFree(in)); CK(hipFree(out));
    CK(hipStreamDestroy(s));
  }
  return 0;
}

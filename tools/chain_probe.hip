// chain_probe.hip -- can the launch boundary of a chain of dependent launches be OVERLAPPED?
//
// bench.py --decompose: every step of the chained-launch graph pays a ~1.1 us launch boundary (store
// drain, end-of-kernel cache work, command processor, next dispatch), 38 % of a step at 4 096 envs.
// The step kernels' dependency is per WORKGROUP (workgroup b of launch k+1 needs what workgroup b
// of launch k wrote, nothing else), so a chain could in principle run as launches alternating between
// TWO streams, with workgroup b of launch k+1 polling an epoch word that workgroup b of launch k
// publishes behind its write-through stores (the sc1-store / sc1-load hand-off of
// MI355X_MICROARCH.md): every launch still does one step of every env.  This probe measures, with a
// stand-in body of ~the step's length, what that would buy before any of it is built:
//   (a) one stream, no flags           period = body + boundary             (what bench.py times today)
//   (b) two streams alternating, flags  period = ?   (overlap -> body + hand-off latency)
//   (c) the same captured into ONE hipGraph with two parallel branches (does the graph executor run
//       branches concurrently?)
// Every poll loop is BOUNDED (a wave that never sees its flag gives up, raises an error word and
// ends), so nothing here can hang the GPU.
//     hipcc --offload-arch=gfx950 -O2 tools/chain_probe.hip -o tools/_bin/chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int WG = 64;          // workgroups per launch (4 096 envs / 64)
constexpr int LIMIT = 20000;    // polls before a wave gives up (a few ms): nothing here can hang

// body: `iters` dependent VALU instructions (~4 cycles each) + a row of loads at the start and stores at the end
__global__ void __launch_bounds__(256) chain_k(unsigned *flags, unsigned epoch, int iters, int *data, unsigned *err,
                                               unsigned long long *stamps) {
  const int b = blockIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (epoch > 1u) {   // wait for workgroup b of the previous launch (every wave polls for itself)
    int it = 0;
    while (__hip_atomic_load(&flags[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch - 1u && ++it < LIMIT)
      __builtin_amdgcn_s_sleep(1);
    if (it >= LIMIT && threadIdx.x == 0) __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  int v = __hip_atomic_load(&data[b * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1 load of handed-off data
  for (int k = 0; k < iters; k++) v = v * 3 + k;
  __hip_atomic_store(&data[b * 256 + threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // sc1 store
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && epoch != 0u) __hip_atomic_store(&flags[b], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0 && stamps != nullptr) {
    stamps[(size_t)(epoch ? epoch : 1) * WG * 3 + b * 3 + 0] = t0;
    stamps[(size_t)(epoch ? epoch : 1) * WG * 3 + b * 3 + 1] = t1;
    stamps[(size_t)(epoch ? epoch : 1) * WG * 3 + b * 3 + 2] = __builtin_amdgcn_s_memrealtime();
  }
}

static double now_ms(hipEvent_t a, hipEvent_t b) {
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 600;     // ~600 x 4 cycles = 1 us of body at 2.4 GHz
  const int L = 2000;
  unsigned *flags, *err;
  int *data;
  unsigned long long *stamps;
  CHECK(hipMalloc(&flags, WG * 4));
  CHECK(hipMalloc(&err, 4));
  CHECK(hipMalloc(&data, WG * 256 * 4));
  CHECK(hipMalloc(&stamps, (size_t)(L + 2) * WG * 3 * 8));
  CHECK(hipMemset(data, 0, WG * 256 * 4));
  hipStream_t s[3];
  for (auto &x : s) CHECK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto reset = [&] { CHECK(hipMemset(flags, 0, WG * 4)); CHECK(hipMemset(err, 0, 4)); CHECK(hipDeviceSynchronize()); };
  auto errword = [&] { unsigned h; CHECK(hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost)); return h; };

  // (a) one stream, no flags
  for (int rep = 0; rep < 2; rep++) {
    reset();
    CHECK(hipEventRecord(e0, s[0]));
    for (int k = 0; k < L; k++) hipLaunchKernelGGL(chain_k, dim3(WG), dim3(256), 0, s[0], flags, 0u, iters, data, err, (unsigned long long *)nullptr);
    CHECK(hipEventRecord(e1, s[0]));
    CHECK(hipEventSynchronize(e1));
    if (rep) printf("(a) one stream, no flags, eager:                 %.3f us per launch\n", now_ms(e0, e1) * 1e3 / L);
  }
  // (a') the same as one graph
  {
    hipGraph_t g; hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < 240; k++) hipLaunchKernelGGL(chain_k, dim3(WG), dim3(256), 0, s[0], flags, 0u, iters, data, err, (unsigned long long *)nullptr);
    CHECK(hipStreamEndCapture(s[0], &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 20; w++) CHECK(hipGraphLaunch(ge, s[0]));
    CHECK(hipStreamSynchronize(s[0]));
    CHECK(hipEventRecord(e0, s[0]));
    for (int w = 0; w < 20; w++) CHECK(hipGraphLaunch(ge, s[0]));
    CHECK(hipEventRecord(e1, s[0]));
    CHECK(hipEventSynchronize(e1));
    printf("(a') one stream, no flags, 240-launch graph:      %.3f us per launch\n", now_ms(e0, e1) * 1e3 / (20 * 240));
  }
  // (a'') one stream WITH flags (the protocol's own cost when nothing overlaps)
  {
    reset();
    CHECK(hipEventRecord(e0, s[0]));
    for (int k = 0; k < L; k++) hipLaunchKernelGGL(chain_k, dim3(WG), dim3(256), 0, s[0], flags, (unsigned)(k + 1), iters, data, err, (unsigned long long *)nullptr);
    CHECK(hipEventRecord(e1, s[0]));
    CHECK(hipEventSynchronize(e1));
    printf("(a'') one stream, flags set and polled, eager:    %.3f us per launch (err %u)\n", now_ms(e0, e1) * 1e3 / L, errword());
  }
  // (b) two / three streams alternating, flags
  for (int ns = 2; ns <= 3; ns++) {
    for (int rep = 0; rep < 2; rep++) {
      reset();
      CHECK(hipEventRecord(e0, s[0]));
      for (int k = 0; k < L; k++)
        hipLaunchKernelGGL(chain_k, dim3(WG), dim3(256), 0, s[k % ns], flags, (unsigned)(k + 1), iters, data, err, rep ? stamps : nullptr);
      for (int q = 1; q < ns; q++) { CHECK(hipEventRecord(e1, s[q])); CHECK(hipStreamWaitEvent(s[0], e1, 0)); }
      CHECK(hipEventRecord(e1, s[0]));
      CHECK(hipEventSynchronize(e1));
      if (rep) {
        printf("(b) %d streams alternating, flags, eager:          %.3f us per launch (err %u)\n", ns, now_ms(e0, e1) * 1e3 / L, errword());
        std::vector<unsigned long long> h((size_t)(L + 2) * WG * 3);
        CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        // workgroup 0: start-to-start period, time spent polling, body, and flag latency (end of k -> poll matched in k+1)
        double per = 0, poll = 0, body = 0, lat = 0; int cnt = 0;
        for (int k = 200; k < L - 1; k++) {
          const unsigned long long *a = &h[(size_t)(k) * WG * 3], *c = &h[(size_t)(k + 1) * WG * 3];
          per += (double)(c[0] - a[0]); poll += (double)(c[1] - c[0]); body += (double)(a[2] - a[1]);
          lat += (double)((long long)(c[1] - a[2])); cnt++;
        }
        printf("    workgroup 0: start-to-start %.2f us, polling %.2f us, body %.2f us, end of k -> k+1 released %.2f us\n",
               per / cnt * 0.01, poll / cnt * 0.01, body / cnt * 0.01, lat / cnt * 0.01);
      }
    }
  }
  // (c) two parallel branches of ONE graph, flags; the chain words are reset by a memset node at the head
  {
    hipGraph_t g; hipGraphExec_t ge;
    const int GL = 240;
    CHECK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
    CHECK(hipMemsetAsync(flags, 0, WG * 4, s[0]));
    CHECK(hipEventRecord(e1, s[0]));
    CHECK(hipStreamWaitEvent(s[1], e1, 0));     // fork
    for (int k = 0; k < GL; k++)
      hipLaunchKernelGGL(chain_k, dim3(WG), dim3(256), 0, s[k % 2], flags, (unsigned)(k + 1), iters, data, err, (unsigned long long *)nullptr);
    CHECK(hipEventRecord(e1, s[1]));
    CHECK(hipStreamWaitEvent(s[0], e1, 0));     // join
    CHECK(hipStreamEndCapture(s[0], &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    reset();
    for (int w = 0; w < 10; w++) CHECK(hipGraphLaunch(ge, s[0]));
    CHECK(hipStreamSynchronize(s[0]));
    CHECK(hipEventRecord(e0, s[0]));
    for (int w = 0; w < 20; w++) CHECK(hipGraphLaunch(ge, s[0]));
    CHECK(hipEventRecord(e1, s[0]));
    CHECK(hipEventSynchronize(e1));
    printf("(c) ONE graph, two parallel branches, flags:      %.3f us per launch (err %u)\n", now_ms(e0, e1) * 1e3 / (20 * GL), errword());
  }
  return 0;
}

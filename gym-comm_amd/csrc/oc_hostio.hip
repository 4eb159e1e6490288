// oc_hostio.hip -- liboc_hostio.so: one step's worth of the numpy API, packed for one PCIe copy
// (include/oc_hostio.h).  One lane = one env; the [F][n] rows are read coalesced, the [n][k]
// blocks are written with a stride (a few hundred KB per step: the launch count, not the
// bandwidth, is what this kernel removes -- ~17 torch launches per step before it).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/oc_hostio.h"

namespace {

thread_local char g_err[256] = "";
int fail(const char *msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return -1;
}

struct Args {
  const void *obs;
  const int32_t *plan;
  const double *timestep, *reward, *ep_return;
  const int32_t *done, *ep_length;
  char *out;
  int64_t n;
  int32_t F, w64, w32, w8;
  int64_t off32, off_ts, off_rew, off_ret, off_done, off_len, off8;
};

template <int OT>
__global__ void __launch_bounds__(256) k_pack_host(const Args p) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  int64_t *b64 = (int64_t *)p.out + i * p.w64;
  float *b32 = (float *)(p.out + p.off32) + i * p.w32;
  int8_t *b8 = (int8_t *)(p.out + p.off8) + i * p.w8;
  for (int r = 0; r < p.F; r++) {
    const int e = p.plan[r];   // uniform
    const int64_t idx = (int64_t)r * p.n + i;
    int v;
    if (OT == 1) v = ((const int8_t *)p.obs)[idx];
    else if (OT == 2) v = (int)((const float *)p.obs)[idx];   // the rows hold integers
    else v = ((const int32_t *)p.obs)[idx];
    const int blk = e >> 16, col = e & 0xFFFF;
    if (blk == 0) b64[col] = v;
    else if (blk == 1) b32[col] = (float)v;
    else b8[col] = (int8_t)v;
  }
  ((float *)(p.out + p.off_ts))[i] = (float)p.timestep[i];
  if (p.reward) ((float *)(p.out + p.off_rew))[i] = (float)p.reward[i];
  if (p.ep_return) ((double *)(p.out + p.off_ret))[i] = p.ep_return[i];
  if (p.done) ((int32_t *)(p.out + p.off_done))[i] = p.done[i];
  if (p.ep_length) ((int32_t *)(p.out + p.off_len))[i] = p.ep_length[i];
}

void offsets(Args &a, bool rew, bool ret, bool done, bool len, int64_t &total) {
  int64_t o = a.n * a.w64 * 8;
  a.off_ret = o, o += ret ? a.n * 8 : 0;
  a.off32 = o, o += a.n * a.w32 * 4;
  a.off_ts = o, o += a.n * 4;
  a.off_rew = o, o += rew ? a.n * 4 : 0;
  a.off_done = o, o += done ? a.n * 4 : 0;
  a.off_len = o, o += len ? a.n * 4 : 0;
  a.off8 = o, o += a.n * a.w8;
  total = o;
}

}  // namespace

extern "C" {

int oc_hostio_abi_version(void) { return OC_HOSTIO_ABI_VERSION; }
const char *oc_hostio_last_error(void) { return g_err; }

int64_t oc_pack_host_bytes(int32_t w64, int32_t w32, int32_t w8, int32_t has_reward, int32_t has_ep_return,
                           int32_t has_done, int32_t has_ep_length, int64_t n) {
  if (w64 < 0 || w32 < 0 || w8 < 0 || n < 0) return -1;
  Args a{};
  a.n = n, a.w64 = w64, a.w32 = w32, a.w8 = w8;
  int64_t total;
  offsets(a, has_reward, has_ep_return, has_done, has_ep_length, total);
  return total;
}

int oc_pack_host(const void *obs_rows, int32_t obs_type, int32_t F, const int32_t *plan, int32_t w64, int32_t w32,
                 int32_t w8, const double *timestep, const double *reward, const double *ep_return,
                 const int32_t *done, const int32_t *ep_length, void *out, int64_t n, void *stream) {
  if (!obs_rows || !plan || !timestep || !out || F < 1 || w64 < 0 || w32 < 0 || w8 < 0 || n < 0 || obs_type < 0 ||
      obs_type > 2)
    return fail("oc_pack_host: bad argument");
  if (n == 0) return 0;
  Args a{};
  a.obs = obs_rows, a.plan = plan, a.timestep = timestep, a.reward = reward, a.ep_return = ep_return;
  a.done = done, a.ep_length = ep_length, a.out = (char *)out, a.n = n, a.F = F;
  a.w64 = w64, a.w32 = w32, a.w8 = w8;
  int64_t total;
  offsets(a, reward != nullptr, ep_return != nullptr, done != nullptr, ep_length != nullptr, total);
  const int bs = 256;
  const int64_t grid = (n + bs - 1) / bs;
  if (grid > 0x7FFFFFFF) return fail("oc_pack_host: n too large");
  const dim3 g((unsigned)grid), b(bs);
  if (obs_type == 1) hipLaunchKernelGGL(k_pack_host<1>, g, b, 0, (hipStream_t)stream, a);
  else if (obs_type == 2) hipLaunchKernelGGL(k_pack_host<2>, g, b, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(k_pack_host<0>, g, b, 0, (hipStream_t)stream, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "oc_pack_host: kernel launch: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // extern "C"

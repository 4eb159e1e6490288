// oc_policy.hip -- liboc_policy.so: a 64-unit tanh MLP policy on the observation rows of the
// batched Overcooked stepper, sampled into (move, comm) pairs (include/oc_policy.h).
//
// gfx950 only.  One wave = one PASS of 32 envs (oc_policy_device.h: both products on the matrix
// cores); two waves per workgroup, XCD-aware grid (k_policy_mlp below).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/oc_policy.h"
#include "oc_policy_device.h"

namespace {

thread_local char g_err[256] = "";
int fail(const char *msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return -1;
}

struct Args {
  oc_policy_player pl[2];
  const double *timestep;
  int32_t F, C, ksteps;
  int64_t n;
};

using namespace ocpol;

// Workgroup -> envs, XCD-aware.  WPB = 2 (the shipped mapping): a workgroup is 64 envs, workgroup x
// of a player = envs 64 x .. 64 x + 63 -- the step kernels' mapping -- and the grid is one-
// dimensional with every player's range padded to a multiple of 8 workgroups (`gx8`), so that
// workgroup x of EVERY player runs on XCD x % 8 (workgroups are dealt to the 8 XCDs round-robin):
// the observation rows a workgroup reads were written, a launch earlier, through the L2 of its
// own XCD.  With 32-env workgroups in launch order (WPB = 1, kept for the A/B) a closed-loop step
// cost more than its two kernels apart: 28.5 us against 21.3 at 131 072 envs; matched: 22.2.
template <int OT, int WPB, int CMAX>
__global__ void __launch_bounds__(64 * WPB) k_policy_mlp(const Args p, const int gx, const int gx8) {
  const int x = (int)(blockIdx.x % (unsigned)gx8), y = (int)(blockIdx.x / (unsigned)gx8);
  if (x >= gx) return;                       // padding workgroup (uniform)
  const oc_policy_player &P = p.pl[y];
  const int lane = threadIdx.x & 63, r = lane & 31;
  const int64_t env0 = ((int64_t)x * WPB + (threadIdx.x >> 6)) * 32 + r;
  const bool valid = env0 < p.n;
  const int64_t env = valid ? env0 : p.n - 1;   // lanes past the batch compute on the last env, store nothing
  policy_pass<OT, CMAX>(P.obs, (uint32_t)p.n, (uint32_t)env, valid, lane, P.w1, P.w2, P.b2, P.rng, P.pairs, P.logits,
                        (float)p.timestep[env], p.F, p.C, p.ksteps);
}

uint16_t f32_to_f16_bits(float f) {   // round to nearest even, host side
  _Float16 hv = (_Float16)f;
  uint16_t b;
  memcpy(&b, &hv, 2);
  return b;
}

}  // namespace

extern "C" {

int oc_policy_abi_version(void) { return OC_POLICY_ABI_VERSION; }
const char *oc_policy_last_error(void) { return g_err; }
int32_t oc_policy_ksteps(int32_t F) { return (F + 2 + 15) / 16; }

int oc_policy_pack_w1(const float *w1, const float *wt, const float *b1, int32_t F, uint16_t *out) {
  if (!w1 || !wt || !b1 || !out || F < 1) return fail("oc_policy_pack_w1: bad argument");
  const int ks = oc_policy_ksteps(F);
  for (int m = 0; m < 2; m++)
    for (int s = 0; s < ks; s++)
      for (int l = 0; l < 64; l++)
        for (int j = 0; j < 8; j++) {
          const int row = 32 * m + (l & 31), k = 16 * s + 8 * (l >> 5) + j;
          const float v = k < F ? w1[(size_t)row * F + k] : k == F ? wt[row] : k == F + 1 ? b1[row] : 0.0f;
          out[(((size_t)m * ks + s) * 64 + l) * 8 + j] = f32_to_f16_bits(2.0f * ocpol::K_LOG2E * v);
        }
  return 0;
}

// which logit (0..3 move, 4 + c comm) lives in row o of the second product, or -1
static int logit_of_row(int o, int C) {
  if (o < 4) return o;
  if (((o - 4) & 7) < 4) {
    const int c = ((o - 4) & 3) + 4 * ((o - 4) >> 3);
    if (c < C) return 4 + c;
  }
  return -1;
}
static float w2_folded(const float *w2, int logit, int hid) {   // the fp16 value the kernel multiplies r by
  const _Float16 hv = (_Float16)(-2.0f * ocpol::K_LOG2E * w2[(size_t)logit * OC_POLICY_HIDDEN + hid]);
  return (float)hv;
}

int oc_policy_pack_w2(const float *w2, int32_t C, uint16_t *out) {
  if (!w2 || !out || C < 1 || C > OC_POLICY_MAX_COMM) return fail("oc_policy_pack_w2: bad argument (1 <= C <= 16)");
  for (int s = 0; s < 4; s++)
    for (int l = 0; l < 64; l++)
      for (int j = 0; j < 8; j++) {
        const int logit = logit_of_row(l & 31, C), hid = 16 * s + 8 * (j >> 2) + 4 * (l >> 5) + (j & 3);
        out[((size_t)s * 64 + l) * 8 + j] = f32_to_f16_bits(logit >= 0 ? w2_folded(w2, logit, hid) : 0.0f);
      }
  return 0;
}

int oc_policy_pack_b2(const float *b2, const float *w2, int32_t C, float *out) {
  if (!b2 || !w2 || !out || C < 1 || C > OC_POLICY_MAX_COMM)
    return fail("oc_policy_pack_b2: bad argument (1 <= C <= 16)");
  for (int l = 0; l < 64; l++)
    for (int r = 0; r < 16; r++) {
      const int logit = logit_of_row((r & 3) + 8 * (r >> 2) + 4 * (l >> 5), C);
      float v = 0.0f;
      if (logit >= 0) {   // log2(e) (b2 + sum_j W2[j]), the sum taken over the ROUNDED folded weights
        float sum = 0.0f;
        for (int j = 0; j < OC_POLICY_HIDDEN; j++) sum += w2_folded(w2, logit, j);
        v = ocpol::K_LOG2E * b2[logit] - 0.5f * sum;
      }
      out[l * 16 + r] = v;
    }
  return 0;
}

int oc_policy_mlp(const oc_policy_player *players, int32_t num_players, const double *timestep, int32_t F,
                  int32_t C, int32_t obs_type, int64_t n, void *stream) {
  if (!players || num_players < 1 || num_players > 2 || !timestep || F < 1 || C < 1 || C > OC_POLICY_MAX_COMM ||
      obs_type < 0 || obs_type > 2 || n < 0)
    return fail("oc_policy_mlp: bad argument (1..2 players, 1 <= C <= 16, obs_type 0..2)");
  if (n == 0) return 0;
  if ((int64_t)F * n >= ((int64_t)1 << 31)) return fail("oc_policy_mlp: F * n must stay below 2^31; split the batch");
  Args a;
  memset(&a, 0, sizeof(a));
  for (int k = 0; k < num_players; k++) {
    if (!players[k].obs || !players[k].w1 || !players[k].w2 || !players[k].b2 || !players[k].pairs)
      return fail("oc_policy_mlp: a player needs obs, w1, w2, b2 and pairs");
    a.pl[k] = players[k];
  }
  a.timestep = timestep;
  a.F = F, a.C = C, a.ksteps = oc_policy_ksteps(F), a.n = n;
  const char *ev = getenv("OC_POLICY_WG32");   // tuning / A-B: 32-env workgroups in launch order
  const int wpb = (ev && ev[0] == '1') ? 1 : 2;
  const int64_t gx = (n + 32 * wpb - 1) / (32 * wpb), gx8 = (gx + 7) / 8 * 8;
  if (gx8 * num_players > 0x7FFFFFFF) return fail("oc_policy_mlp: n too large");
  const dim3 g((unsigned)(gx8 * num_players)), b(64 * wpb);
#define OC_PL2(OT_, CM_)                                                                                    \
  do {                                                                                                      \
    if (wpb == 2) hipLaunchKernelGGL((k_policy_mlp<OT_, 2, CM_>), g, b, 0, (hipStream_t)stream, a, (int)gx, (int)gx8); \
    else hipLaunchKernelGGL((k_policy_mlp<OT_, 1, CM_>), g, b, 0, (hipStream_t)stream, a, (int)gx, (int)gx8); \
  } while (0)
#define OC_PL(OT_)                 \
  do {                             \
    if (C <= 4) OC_PL2(OT_, 4);    \
    else if (C <= 8) OC_PL2(OT_, 8); \
    else OC_PL2(OT_, 16);          \
  } while (0)
  if (obs_type == 1) OC_PL(1);
  else if (obs_type == 2) OC_PL(2);
  else OC_PL(0);
#undef OC_PL2
#undef OC_PL
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "oc_policy_mlp: kernel launch: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // extern "C"

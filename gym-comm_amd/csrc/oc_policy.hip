// oc_policy.hip -- liboc_policy.so: a 64-unit tanh MLP policy on the observation rows of the
// batched Overcooked stepper, sampled into (move, comm) pairs (include/oc_policy.h).
//
// gfx950 only.  One wave = 32 envs (two waves per workgroup).  Both products run on the matrix cores:
//
//   H^T [64 hidden x 32 envs] = W1aug [64 x K] . X^T [K x 32 envs]      v_mfma_f32_32x32x16_f16
//       A = weights (fragment order, one 16-byte load per lane, M-tile and k-step),
//       B = the observation: lane (env = l & 31, half h = l >> 5) holds features
//           16 s + 8 h + 0..7 of ITS env -- eight coalesced row loads per k-step, converted to
//           fp16 in registers; feature F is the timestep, F + 1 the constant 1 (bias b1).
//   the accumulators hold H^T with the ENV ON THE LANE and the hidden units in the 16 registers
//   (row = (r & 3) + 8 (r >> 2) + 4 h), which is exactly the B-operand layout of a product that
//   sums over hidden units: registers 8 (s & 1) .. + 7 of M-tile s >> 1, tanh'ed and packed,
//   are the B fragment of k-step s -- no LDS, no lane movement; the weights' k index is
//   permuted on the host instead (oc_policy_pack_w2).
//   L^T [32 rows x 32 envs] = W2row [32 x 64] . tanh(H^T)
//       rows 0..3 = move logits -> registers 0..3 of the LOWER half-wave's lanes,
//       comm logit c sits in row 4 + (c & 3) + 8 (c >> 2) -> register c of the UPPER half's
//       lanes: lane l samples the move of env l & 31, lane l + 32 its comm -- each from its own
//       PCG32 stream -- and the wave's 64 results are 256 contiguous bytes of the pairs tensor.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/oc_policy.h"

namespace {

thread_local char g_err[256] = "";
int fail(const char *msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return -1;
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

struct Args {
  oc_policy_player pl[2];
  const double *timestep;
  int32_t F, C, ksteps;
  int64_t n;
};

__device__ __forceinline__ uint32_t pcg32(uint32_t &state) {   // the stepper's generator (oc_kernels.hip)
  state = state * 747796405u + 2891336453u;
  const uint32_t w = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (w >> 22u) ^ w;
}

// The activations are folded into the weights on the host (oc_policy_pack_*): with W1, wt, b1
// scaled by 2 log2(e) the first product delivers a = 2 log2(e) h, and
//     tanh(h) = 1 - 2 r,   r = 1 / (2^a + 1)                      (v_exp_f32, v_add, v_rcp_f32)
// so the second product takes r itself with W2' = -2 log2(e) W2 and the accumulator started at
// b2' = log2(e) (b2 + sum_j W2[.][j]): it delivers the logits in BASE 2 (logit * log2(e)), which
// is what the sampler's 2^x wants.  Saturates cleanly: 2^a -> inf gives r = 0 (tanh 1), 2^a -> 0
// gives r = 1 (tanh -1).
constexpr float K_LOG2E = 1.4426950408889634f, K_LN2 = 0.6931471805599453f;
__device__ __forceinline__ float sigmoid_complement(float a) {
  return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a) + 1.0f);
}

// element `idx` of the rows (32-bit unsigned index: scalar base + 32-bit vector offset addressing;
// oc_policy_mlp checks F * n < 2^31)
template <int OT>
__device__ __forceinline__ float obs_at(const void *obs, uint32_t idx) {
  if (OT == 1) return (float)((const int8_t *)obs)[idx];
  if (OT == 2) return ((const float *)obs)[idx];
  return (float)((const int32_t *)obs)[idx];
}

// One action out of `count` candidates whose base-2 logits sit in registers 0..count-1 of `v`
// (count <= 16, uniform per half-wave).  Sampling: inverse CDF of the softmax with ONE uniform
// draw from the lane's PCG32 stream -- p_c = 2^(l_c - max) / S, the action is the number of
// cumulative sums that do not exceed u S.  Greedy (`sample` false): the first maximum
// (torch.argmax's rule on these sizes).
// `count` differs between the two half-waves (4 moves below, C comms above): candidates past a
// lane's count are given the logit -inf (probability 0, never the maximum) instead of a branch;
// CMAX (4, 8 or 16 >= max(4, C)) bounds the unrolled loops at compile time.
template <int CMAX>
__device__ __forceinline__ int pick(const f32x16 &v, int count, bool sample, uint32_t &state) {
  float x[CMAX], m = -3.0e38f;
#pragma unroll
  for (int c = 0; c < CMAX; c++) {
    x[c] = c < count ? v[c] : -3.0e38f;
    m = fmaxf(m, x[c]);
  }
  int arg = 0;
  if (sample) {   // uniform
    float total = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; c++) {
      x[c] = __builtin_amdgcn_exp2f(x[c] - m);   // 2^-huge = 0 for the padding candidates
      total += x[c];
    }
    const float u = ((float)(pcg32(state) >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0, 1)
    const float t = u * total;
    float cum = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX - 1; c++) {
      cum += x[c];
      arg += (cum <= t && c < count - 1) ? 1 : 0;
    }
  } else {
#pragma unroll
    for (int c = CMAX - 1; c >= 0; c--) arg = (x[c] == m) ? c : arg;   // the first maximum wins
  }
  return arg;
}

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
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int64_t env0 = ((int64_t)x * WPB + (threadIdx.x >> 6)) * 32 + r;
  const bool valid = env0 < p.n;
  const int64_t env = valid ? env0 : p.n - 1;   // lanes past the batch compute on the last env, store nothing
  const int F = p.F;
  const float ts = (float)p.timestep[env];

  // ---- H^T = W1aug . X^T ---------------------------------------------------------------
  f32x16 acc0, acc1;
#pragma unroll
  for (int q = 0; q < 16; q++) acc0[q] = 0.0f, acc1[q] = 0.0f;
  const half8 *w1 = (const half8 *)P.w1;
  const uint32_t n32 = (uint32_t)p.n, e32 = (uint32_t)env;
  for (int s = 0; s < p.ksteps; s++) {
    half8 b;
    const uint32_t k0 = 16 * s + 8 * h, off0 = k0 * n32 + e32;   // this lane's first feature of the k-step
    if (16 * s + 16 <= F) {   // uniform: every feature of this k-step is an observation row
#pragma unroll
      for (int j = 0; j < 8; j++) b[j] = (_Float16)obs_at<OT>(P.obs, off0 + (uint32_t)j * n32);
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int k = (int)k0 + j;
        const float xv = obs_at<OT>(P.obs, k < F ? off0 + (uint32_t)j * n32 : e32);   // (always a readable element)
        b[j] = (_Float16)(k < F ? xv : k == F ? ts : k == F + 1 ? 1.0f : 0.0f);
      }
    }
    const half8 a0 = w1[(size_t)(0 * p.ksteps + s) * 64 + lane];
    const half8 a1 = w1[(size_t)(1 * p.ksteps + s) * 64 + lane];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b, acc1, 0, 0, 0);
  }

  // ---- log2(e) L^T = W2' . r + b2'  (see sigmoid_complement) -----------------------------
  f32x16 out;
  {
    const float4 *b2 = (const float4 *)P.b2 + lane * 4;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 v = b2[q];
      out[4 * q + 0] = v.x, out[4 * q + 1] = v.y, out[4 * q + 2] = v.z, out[4 * q + 3] = v.w;
    }
  }
  const half8 *w2 = (const half8 *)P.w2;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    half8 b;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float av = (s >> 1) ? acc1[8 * (s & 1) + j] : acc0[8 * (s & 1) + j];
      b[j] = (_Float16)sigmoid_complement(av);
    }
    out = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[s * 64 + lane], b, out, 0, 0, 0);
  }

  // ---- sample and store ------------------------------------------------------------------
  // lower half-wave: move of env r from registers 0..3; upper: comm of env r from registers 0..C-1
  const int count = h ? p.C : 4;
  const bool sample = P.rng != nullptr;
  uint32_t state = 0;
  if (sample) state = P.rng[(int64_t)h * p.n + env];
  const int choice = pick<CMAX>(out, count, sample, state);
  if (valid) {
    P.pairs[env * 2 + h] = choice;
    if (sample) P.rng[(int64_t)h * p.n + env] = state;
    if (P.logits != nullptr) {
#pragma unroll
      for (int c = 0; c < CMAX; c++)
        if (c < count) P.logits[(int64_t)((h ? 4 : 0) + c) * p.n + env] = out[c] * K_LN2;   // natural-log logits
    }
  }
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
          out[(((size_t)m * ks + s) * 64 + l) * 8 + j] = f32_to_f16_bits(2.0f * K_LOG2E * v);
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
  const _Float16 hv = (_Float16)(-2.0f * K_LOG2E * w2[(size_t)logit * OC_POLICY_HIDDEN + hid]);
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
        v = K_LOG2E * b2[logit] - 0.5f * sum;
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

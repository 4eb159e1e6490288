// oc_policy_device.h -- device code of the MLP policy (include/oc_policy.h), shared by the
// policy kernel (oc_policy.hip) and by the fused step kernel when it evaluates the policies
// itself (oc_kernels.hip, oc_step_opts.policy).  gfx950 only.
//
// One PASS = one wave evaluating the network for 32 envs:
//
//   H^T [64 hidden x 32 envs] = W1aug [64 x K] . X^T [K x 32 envs]      v_mfma_f32_32x32x16_f16
//       A = weights (fragment order, one 16-byte load per lane, M-tile and k-step),
//       B = the observation: lane (env = l & 31, half h = l >> 5) holds features
//           16 s + 8 h + 0..7 of ITS env -- eight coalesced row loads per k-step, converted to
//           fp16 in registers; feature F is the timestep, F + 1 the constant 1 (bias b1).
//   the accumulators hold H^T with the ENV ON THE LANE and the hidden units in the 16 registers
//   (row = (r & 3) + 8 (r >> 2) + 4 h), which is exactly the B-operand layout of a product that
//   sums over hidden units: registers 8 (s & 1) .. + 7 of M-tile s >> 1, through the activation and
//   packed, are the B fragment of k-step s -- no LDS, no lane movement; the weights' k index is
//   permuted on the host instead (oc_policy_pack_w2).
//   L^T [32 rows x 32 envs] = W2row [32 x 64] . act(H^T)
//       rows 0..3 = move logits -> registers 0..3 of the LOWER half-wave's lanes,
//       comm logit c sits in row 4 + (c & 3) + 8 (c >> 2) -> register c of the UPPER half's
//       lanes: lane l samples the move of env l & 31, lane l + 32 its comm -- each from its own
//       PCG32 stream -- and the wave's 64 results are 256 contiguous bytes of the pairs tensor.
#ifndef OC_POLICY_DEVICE_H
#define OC_POLICY_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ocpol {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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
// the launchers check F * n < 2^31).  OT: 0 int32, 1 int8, 2 float32.
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

// A lane's share of one player's packed weights (include/oc_policy.h): the A fragments of both
// products and the second product's start value -- 14 x 16 bytes.  The policy kernel loads them
// where it needs them; the fused step kernel loads them at its start, under the wait for the
// env state, so that the pass behind the step finds them in registers.
constexpr int MAX_KSTEPS = 3;   // F + 2 <= 48 features (the fused step: C <= 4, S <= 14)
struct Weights {
  half8 w1[2][MAX_KSTEPS];
  half8 w2[4];
  float4 b2[4];
};
__device__ __forceinline__ void load_weights(Weights &w, const uint16_t *w1_, const uint16_t *w2_,
                                             const float *b2_, int lane, int ksteps) {
  const half8 *w1 = (const half8 *)w1_, *w2 = (const half8 *)w2_;
#pragma unroll
  for (int s = 0; s < MAX_KSTEPS; s++)
    if (s < ksteps) {   // uniform
      w.w1[0][s] = w1[(size_t)(0 * ksteps + s) * 64 + lane];
      w.w1[1][s] = w1[(size_t)(1 * ksteps + s) * 64 + lane];
    }
#pragma unroll
  for (int s = 0; s < 4; s++) w.w2[s] = w2[s * 64 + lane];
#pragma unroll
  for (int q = 0; q < 4; q++) w.b2[q] = ((const float4 *)b2_ + lane * 4)[q];
}

// One pass: the calling wave evaluates the network for 32 envs -- lane l (r = l & 31, h = l >> 5)
// works for env `env` (the same for lanes l and l + 32; `valid` false = a lane past the batch,
// which computes on a clamped env and stores nothing) -- and writes pairs[env][h].
//   obs: the viewer's rows [F][n]; ts: the env's timestep; rng: uint32 [2][n] or NULL (greedy)
//   LDSSRC: the features come from an LDS image instead -- `lds` float [F][64] (one column per env
//   of the workgroup), this lane's column `col` -- as the fused step kernel hands them over
//   PRE: the weights are already in registers (`pre`, load_weights), ksteps <= MAX_KSTEPS
template <int OT, int CMAX, bool LDSSRC = false, bool PRE = false>
__device__ __forceinline__ void policy_pass(const void *obs, uint32_t n32, uint32_t env, bool valid, int lane,
                                            const uint16_t *w1_, const uint16_t *w2_, const float *b2_,
                                            uint32_t *rng, int32_t *pairs, float *logits, float ts, int F,
                                            int C, int ksteps, const float *lds = nullptr, int col = 0,
                                            const Weights *pre = nullptr) {
  const int h = lane >> 5;
  // ---- H^T = W1aug . X^T ---------------------------------------------------------------
  f32x16 acc0, acc1;
#pragma unroll
  for (int q = 0; q < 16; q++) acc0[q] = 0.0f, acc1[q] = 0.0f;
  const half8 *w1 = (const half8 *)w1_;
  auto kstep = [&](int s, const half8 &a0, const half8 &a1) {
    half8 b;
    const uint32_t k0 = 16 * s + 8 * h, off0 = k0 * n32 + env;   // this lane's first feature of the k-step
    if (16 * s + 16 <= F) {   // uniform: every feature of this k-step is an observation row
#pragma unroll
      for (int j = 0; j < 8; j++)
        b[j] = (_Float16)(LDSSRC ? lds[(k0 + j) * 64 + col] : obs_at<OT>(obs, off0 + (uint32_t)j * n32));
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int k = (int)k0 + j;
        const float xv = LDSSRC ? lds[(k < F ? k : 0) * 64 + col]
                                : obs_at<OT>(obs, k < F ? off0 + (uint32_t)j * n32 : env);   // (always a readable element)
        b[j] = (_Float16)(k < F ? xv : k == F ? ts : k == F + 1 ? 1.0f : 0.0f);
      }
    }
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b, acc1, 0, 0, 0);
  };
  if constexpr (PRE) {
#pragma unroll
    for (int s = 0; s < MAX_KSTEPS; s++)
      if (s < ksteps) kstep(s, pre->w1[0][s], pre->w1[1][s]);   // uniform; static register indices
  } else {
    for (int s = 0; s < ksteps; s++)
      kstep(s, w1[(size_t)(0 * ksteps + s) * 64 + lane], w1[(size_t)(1 * ksteps + s) * 64 + lane]);
  }

  // ---- log2(e) L^T = W2' . r + b2'  (see sigmoid_complement) -----------------------------
  f32x16 out;
  {
    const float4 *b2 = (const float4 *)b2_ + lane * 4;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const float4 v = PRE ? pre->b2[q] : b2[q];
      out[4 * q + 0] = v.x, out[4 * q + 1] = v.y, out[4 * q + 2] = v.z, out[4 * q + 3] = v.w;
    }
  }
  const half8 *w2 = (const half8 *)w2_;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    half8 b;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float av = (s >> 1) ? acc1[8 * (s & 1) + j] : acc0[8 * (s & 1) + j];
      b[j] = (_Float16)sigmoid_complement(av);
    }
    out = __builtin_amdgcn_mfma_f32_32x32x16_f16(PRE ? pre->w2[s] : w2[s * 64 + lane], b, out, 0, 0, 0);
  }

  // ---- sample and store ------------------------------------------------------------------
  // lower half-wave: move of env r from registers 0..3; upper: comm of env r from registers 0..C-1
  const int count = h ? C : 4;
  const bool sample = rng != nullptr;
  uint32_t state = 0;
  if (sample) state = rng[(size_t)h * n32 + env];
  const int choice = pick<CMAX>(out, count, sample, state);
  if (valid) {
    pairs[(size_t)env * 2 + h] = choice;
    if (sample) rng[(size_t)h * n32 + env] = state;
    if (logits != nullptr) {
#pragma unroll
      for (int c = 0; c < CMAX; c++)
        if (c < count) logits[(size_t)((h ? 4 : 0) + c) * n32 + env] = out[c] * K_LN2;   // natural-log logits
    }
  }
}

}  // namespace ocpol
#endif

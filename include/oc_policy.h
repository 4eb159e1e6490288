/* oc_policy.h -- C ABI of liboc_policy.so: a small MLP policy evaluated on the observation
 * rows oc_multi_step / oc_obs leave in HBM, its action sampled and written as the [n][2]
 * (move, comm) pairs oc_step_opts.ego_pairs / alt_pairs consume (include/oc_hip.h).
 *
 * What it stands in for: the per-step partner / ego forward of the reference's training loop --
 * pantheonrl's OnPolicyAgent.get_action (pantheonrl/common/agents.py:112-194: one
 * policy.forward on ONE observation per env step) behind MultiAgentEnv.step
 * (pantheonrl/common/multiagentenv.py:149-215) -- for the stable-baselines3 default MlpPolicy
 * shape: one hidden layer of 64 tanh units per head group.  SURVEY section 8 (f) 2: "the partner
 * policy batched".  It is NOT part of the environment's semantics: no oracle, a PyTorch fp32
 * reference of the same network in the tests (fp16 operands, fp32 accumulation: logits within
 * 2e-2 of it).  Everything else (any torch module) runs through vec_env.TorchPolicyPartner.
 *
 * The network (gym-comm_amd/vec_env.py: MLPPolicy), per env, F = oc_obs_rows() features x:
 *     h      = tanh(W1 x + wt * timestep + b1)              64 hidden units
 *     logits = W2 h + b2                                    4 move logits, C <= 16 comm logits
 *     move ~ softmax(logits[0:4]),  comm ~ softmax(logits[4:4+C])   (from the env's two PCG32
 *     streams), or the argmax of each (rng == NULL).
 *
 * Both GEMMs run on the matrix cores (v_mfma_f32_32x32x16_f16): one wave = 32 envs; the hidden
 * layer comes out with the env on the lane and the hidden units in the accumulator registers,
 * which IS the B-operand layout of the second product -- no LDS, no lane movement.  Weights are
 * passed pre-arranged in MFMA fragment order with the activation's constants folded in
 * (`oc_policy_pack_*` below do it; host only).  With a = 2 log2(e) h, tanh(h) = 1 - 2 r where
 * r = 1 / (2^a + 1), so the kernel evaluates r (exp2, add, rcp) and the second product yields the
 * logits in base 2, which is what the sampler's 2^x wants:
 *   w1   fp16 [2][ksteps][64 lanes][8]  A fragments of 2 log2(e) [W1 | wt | b1 | 0...]
 *        (64 x 16*ksteps), ksteps = ceil((F + 2) / 16): element j of lane l, M-tile m, k-step s =
 *        2 log2(e) W1aug[32 m + (l & 31)][16 s + 8 (l >> 5) + j]
 *   w2   fp16 [4][64 lanes][8]           A fragments of -2 log2(e) W2 padded to 32 rows, k permuted
 *        to the accumulator order: element j of lane l, k-step s = -2 log2(e) W2row[l & 31][16 s +
 *        8 (j >> 2) + 4 (l >> 5) + (j & 3)], where row o holds move logit o (o < 4) and comm
 *        logit c sits in row 4 + (c & 3) + 8 (c >> 2)
 *   b2   fp32 [64 lanes][16]             the second product's initial accumulator: log2(e) (b2 +
 *        sum_j W2[.][j]) of row (r & 3) + 8 (r >> 2) + 4 (l >> 5) in register r of lane l (the sum
 *        over the ROUNDED fp16 weights, so that the fold is exact)
 * Sampling: inverse CDF of the softmax, one uniform draw per head and step from the env's stream.
 * Every pointer but the pack functions' is a DEVICE pointer of a caller-owned tensor; nothing is
 * allocated, freed or synchronised; calls are ordered by `stream`. */
#ifndef OC_POLICY_H
#define OC_POLICY_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#ifndef OC_API
#define OC_API __attribute__((visibility("default")))
#endif

#define OC_POLICY_ABI_VERSION 1
#define OC_POLICY_HIDDEN 64
#define OC_POLICY_MAX_COMM 16

typedef struct {
  const void *obs;      /* this player's observation rows: [F][n], element type `obs_type` */
  const uint16_t *w1;   /* fp16 bits, fragment order (above) */
  const uint16_t *w2;
  const float *b2;
  uint32_t *rng;        /* uint32 [2][n] PCG32 states (move stream, comm stream), advanced in
                           place; NULL = greedy (argmax) */
  int32_t *pairs;       /* out: int32 [n][2] = (move 0..3, comm 0..C-1) */
  float *logits;        /* optional out (tests): float [4 + C][n]; NULL = not written */
} oc_policy_player;

OC_API int oc_policy_abi_version(void);
OC_API const char *oc_policy_last_error(void);

/* number of 16-feature k-steps of the first product for F observation rows (+ timestep + bias) */
OC_API int32_t oc_policy_ksteps(int32_t F);

/* Host-side packing (plain row-major fp32 in, fragment order out).
 *   w1 [64][F], wt [64], b1 [64]  ->  out fp16 [2][ksteps][64][8]
 *   w2 [4 + C][64]                ->  out fp16 [4][64][8]
 *   b2 [4 + C], w2 [4 + C][64]    ->  out fp32 [64][16] */
OC_API int oc_policy_pack_w1(const float *w1, const float *wt, const float *b1, int32_t F, uint16_t *out);
OC_API int oc_policy_pack_w2(const float *w2, int32_t C, uint16_t *out);
OC_API int oc_policy_pack_b2(const float *b2, const float *w2, int32_t C, float *out);

/* One launch: `num_players` (1 or 2) policies, each on its own observation rows, for n envs.
 *   timestep  double [n] (oc_multi_step's / oc_obs's timestep tensor)
 *   F         observation rows per viewer (oc_obs_rows());  C  comm channels (1..16)
 *   obs_type  element type of the rows: 0 int32, 1 int8, 2 float32 (oc_obs_cfg.obs_int8) */
OC_API int oc_policy_mlp(const oc_policy_player *players, int32_t num_players, const double *timestep,
                         int32_t F, int32_t C, int32_t obs_type, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif

/* oc_hip.h -- C ABI of the MI355X-native batched Overcooked stepper (liboc_hip.so).
 *
 * The reference has no FFI: its hot path is a Python object API.  These entry
 * points are what a binding for that path calls instead of the Python loops
 * (INTEGRATION.md shows the ctypes stub).  Each one cites the reference code it
 * replaces; paths are relative to the reference root.
 *
 * Conventions
 *   - Every env of a call shares one level (oc_level_t).  N envs are stepped by one
 *     launch, one lane per env.
 *   - All tensors are env-major struct-of-arrays: row r of a [R][n] tensor is the n
 *     consecutive values of field r, so a wave's 64 lanes read 256 contiguous bytes.
 *   - Pointers are raw DEVICE pointers owned by the caller (PyTorch-ROCm tensors in
 *     our host code).  The library never allocates or frees tensor memory and never
 *     synchronises: calls are asynchronous on `stream` (a hipStream_t passed as
 *     void*, NULL = the default stream).  The caller serialises calls that touch the
 *     same state.
 *   - Return value: 0 on success, otherwise a hipError_t value or one of the
 *     OC_E_* codes below; oc_last_error() describes the most recent failure of the
 *     calling thread.
 */
#ifndef OC_HIP_H
#define OC_HIP_H

#include <stdint.h>

#include "oc_level.h"

#ifdef __cplusplus
extern "C" {
#endif

#define OC_ABI_VERSION 6
#define OC_API __attribute__((visibility("default")))

enum {
  OC_OK = 0,
  OC_E_BADARG = -1,  /* NULL pointer, negative n, bad blob, unsupported A/M */
  OC_E_NODEVICE = -2 /* no HIP device / runtime failure before launch */
};

typedef struct oc_level oc_level_t; /* opaque; device-resident static tables */

/* ---- state tensor ----------------------------------------------------------
 * int32 [oc_state_words()][n].  Rows (A >= 2 agents, M items, S <= 32 subtasks):
 *   row a           agent a : x | y<<4 | (held_group+1)<<8           (0 = empty hands)
 *                             row 0 also carries t<<16, row 1 merge_counter<<16 | error_flags<<24
 *   row A+i         item i  : x | y<<4 | chopped<<8 | group<<9 | (holder+1)<<12 | seq<<16 | tset<<24
 *                             group  = smallest item id in the same Object
 *                             holder = agent holding that Object
 *                             seq    = rank of the Object in world.objects iteration
 *                                      order (initial items 0..M-1, the k-th merge of
 *                                      the episode gets M+k)
 *                             tset   = set of content types of the Object (bit t = type t)
 *   row A+M         completed_subtasks bitmask (bit oc_level_subtask_info().slot[s] = subtask s)
 *   row A+M+1       goal_objects_count: one bit per subtask (same slots; the goal object of the
 *                   subtask exists), or -- a level that repeats a content type -- two bits per
 *                   distinct goal object (goal_index[s]), the number of cells holding one
 *   rows A+M+2, +3  only in such a "dup" level: creation ranks of the merged object names
 * Replaces the World / SimAgent object graph (gym_cooking/utils/world.py:14-320,
 * utils/agent.py:258-314, utils/core.py:149-237). */

/* ---- observation tensor ----------------------------------------------------
 * int32 (or int8 / float32, see oc_obs_cfg.obs_int8) [2][oc_obs_rows()][n]: viewer 0 then viewer 1; rows per viewer, in the key
 * order of OvercookedMultiEnv.get_observation2
 * (gym_comm/envs/overcooked_env.py:145-157) minus `timestep`:
 *   object_encodings_x[4] object_encodings_y[4] state_encodings[4] is_hidden[4]
 *   completed_subtasks[S] agent1_location[2] agent2_location[2] agent_is_holding[2]
 *   agent1_comm[C] agent2_comm[C]                         => 22 + S + 2C rows
 * `timestep` (t / max_num_timesteps, fp64) is the same for both viewers and goes to
 * its own double[n] tensor. */

typedef struct {
  int32_t fow_radius;   /* arglist.fow_radius */
  int32_t blind_mask;   /* bit0: ego_config["BLIND"], bit1: partner_config["BLIND"] */
  int32_t num_comm;     /* arglist.num_communication (C) */
  int32_t obs_int8;     /* element type of the observation rows: 0 int32; 1 int8 (every value
                           fits: 4x fewer observation bytes, 8.1 -> 6.1 us per launch at 131072
                           envs); 2 float32 (the same integers converted -- what a policy
                           network's first layer takes as obs[v].T without a cast) */
} oc_obs_cfg;

typedef struct {
  oc_obs_cfg obs;
  int32_t communication_on; /* arglist.communication_on */
  int32_t ego_led;          /* arglist.ego_led */
  int32_t ego_agent_idx;    /* OvercookedMultiEnv(ego_agent_idx=...) */
  int32_t can_move_mask;    /* bit0: ego_config["CAN_MOVE"], bit1: partner_config["CAN_MOVE"] */
} oc_wrap_cfg;

/* Optional device pointers of oc_multi_step (a HOST struct, read at the call; every field may
 * be NULL):
 *   ep_return double[n], ep_length int32[n] (both or neither): per-env episode statistics kept
 *            by the kernel, what stable-baselines3's Monitor keeps around the reference's env
 *            (trainer.py:87-121 -> info["episode"] = {"r", "l"}): an env whose `done` row was
 *            set by the PREVIOUS step starts from zero, then the step's shaped reward / 1 is
 *            added.  After a step that returns done they hold the finished episode's return
 *            and length until the next step.  `done` must therefore be the same tensor from
 *            step to step (zero it together with the statistics).
 *   ego_pairs / alt_pairs  int32 [n][2] (int64 [n][2] when pairs_int64 != 0): a player's
 *            (move, comm) as an array of PAIRS -- the batched form of multi_step's ego_action /
 *            alt_action tuples (gym_comm/envs/overcooked_env.py:207-221), i.e. a policy's [n, 2]
 *            output as it lies (torch's argmax / sampling give int64) -- used instead of rows
 *            0,1 / 2,3 of `actions`.
 *   alt_rng  uint32 [n]: the partner plays uniformly at random (move 0..3, comm 0..C-1) from
 *            this per-env PCG32 stream, advanced in place; rows 2,3 / alt_pairs are ignored.
 *            No reference analogue (its partners are SB3 policies); the zero-launch partner
 *            for throughput runs.  alt_played int32 [2][n], if given, receives what was drawn.
 *   waves_per_64  launch hint, not semantics (results are identical): 0 = the library decides,
 *            1 = one wave computes the whole step of its 64 envs, 4 = "split launch": four waves
 *            per 64 envs (one workgroup = the four SIMDs of a CU), each producing one of the
 *            step's outputs -- state + metrics / shaped reward / viewer 0 / viewer 1; 2 = two
 *            waves (state + viewer 0 / shaping + viewer 1).  The library picks 4 up to 32 768
 *            envs (3.41 -> 2.82 us per step at 4 096 envs on an MI355X), 1 beyond.
 *            Any other value = 0.
 *   policy   NULL, or oc_step_policy[2] (HOST array, read at the call): the closed loop in ONE
 *            launch.  After the step, the kernel itself evaluates both players' MLP policies
 *            (include/oc_policy.h: the same network, packed weights and arithmetic as oc_policy_mlp)
 *            on the observations it has just produced -- policy[0] on viewer 0's rows, policy[1]
 *            on viewer 1's -- and OVERWRITES ego_pairs / alt_pairs with the sampled (move, comm) of
 *            the NEXT step.  Needs both pair tensors as int32, at most 4 comm channels, and a
 *            specialised library; in a split launch each of the four waves evaluates one
 *            (viewer, half of the 64 envs) behind a second barrier.  Results are identical to
 *            oc_policy_mlp followed by oc_multi_step.
 * `actions` may be NULL when ego_pairs and one of alt_pairs / alt_rng are given. */
typedef struct {
  const uint16_t *w1;   /* fp16 fragments, include/oc_policy.h */
  const uint16_t *w2;
  const float *b2;
  uint32_t *rng;        /* uint32 [2][n] PCG32 states (move, comm), or NULL = greedy */
} oc_step_policy;

typedef struct {
  double *ep_return;
  int32_t *ep_length;
  const int32_t *ego_pairs;
  const int32_t *alt_pairs;
  uint32_t *alt_rng;
  int32_t *alt_played;
  int32_t pairs_int64;
  int32_t waves_per_64;
  const oc_step_policy *policy;
} oc_step_opts;

/* metrics accumulated by the step kernels when `metrics` != NULL: a device tensor
 * int64 [oc_metrics_slots(n)][8] -- one 64-byte slot per wave (64 envs); lanes 0..5 of the
 * wave add their counter with a no-return atomic each (no two waves share a slot, so there
 * is no contention); the totals are the column sums.  Zero it to start a new rollout. */
enum {
  OC_MET_ENV_STEPS = 0,
  OC_MET_EPISODES = 1,      /* steps that returned done */
  OC_MET_SUCCESSES = 2,     /* done because every delivery was made */
  OC_MET_REWARD_SUM = 3,    /* sum of the sparse integer reward */
  OC_MET_COMPLETED_SUM = 4, /* sum over finished episodes of completed subtasks */
  OC_MET_ERRORS = 5,        /* env-steps that raised an OC_ERR_* flag (include/oc_level.h) */
  OC_MET_COUNT = 8
};

OC_API int oc_abi_version(void);
OC_API const char *oc_last_error(void);

/* Upload one compiled level (include/oc_level.h blob, HOST pointer) to the current
 * device.  Replaces the per-reset static work of OvercookedEnvironment.reset():
 * load_level, run_recipes, make_reachability_graph, cache_distances
 * (gym_cooking/envs/overcooked_environment.py:180-206). */
OC_API int oc_level_create(const int32_t *blob, int32_t n_words, oc_level_t **out);
OC_API int oc_level_destroy(oc_level_t *lv);
/* Specialisation (optional, see gym-comm_amd/specialize.py).  The same source compiled with
 * -DOC_SPECIALIZED and a generated header yields a library whose kernels have a level's static
 * data folded in as compile-time constants; it exports this same ABI.  Two flavours:
 *   "structure" library  folds what the recipes, the item multiset, the agent count and the
 *       border kind fix; the map itself (size, tiles, positions) stays a run-time argument.
 *       oc_level_create() accepts ANY map of that structure.
 *   "level" library (additionally -DOC_SPEC_GEOMETRY)  folds the map as well: fastest, one map.
 *   oc_level_spec_source: write the generated header (C++ text) for a blob -- with_geometry != 0
 *   for a level library; returns its length, or a negative OC_E_* code.  Host only.
 *   oc_is_specialized: 0 generic, 1 structure library, 2 level library. */
OC_API int oc_level_spec_source(const int32_t *blob, int32_t n_words, int32_t with_geometry, char *buf,
                                int32_t buf_size);
OC_API int oc_is_specialized(void);
/* Where the state tensor keeps a subtask's bits.  The kernels order subtasks canonically
 * (Chop / Merge sorted by kind, goal object and food, then the Deliver subtasks in the blob's
 * order), so that one specialised library serves every subtask order of a level -- the
 * reference's order changes with PYTHONHASHSEED (recipe_planner/stripsworld.py:72-77).  For the
 * blob's subtask s: slot[s] = its bit in the completed_subtasks / goal_objects_count words;
 * goal_index[s] = the index of its distinct goal object (a level in dup mode stores
 * goal_objects_count as two bits per distinct goal); *dup = the level repeats a content type.
 * The completed_subtasks observation rows are already in the blob's order.  Host only. */
OC_API int oc_level_subtask_info(const int32_t *blob, int32_t n_words, int32_t *slot, int32_t *goal_index,
                                 int32_t *dup);
OC_API int64_t oc_metrics_slots(int64_t n);                         /* 4 * ceil(n / 256): a slot per wave of 64 envs, whole workgroups */
OC_API int32_t oc_state_words(const oc_level_t *lv);                 /* A + M + 2 */
OC_API int32_t oc_obs_rows(const oc_level_t *lv, int32_t num_comm);  /* 22 + S + 2C */

/* Random item placement (random-* levels; overcooked_environment.py:157-173 scatters the
 * level's "plt" items over random Counter tiles at every reset).  Two ways to supply the
 * start cells of a fresh episode, used by oc_reset and by the auto-reset inside the step
 * kernels (both NULL is an error for such a level, both ignored for fixed levels):
 *   placement  int32 [M][n]   x | y<<4 per item (world order): the caller's draw, e.g. the
 *                             reference's own for parity tests
 *   rng        uint32 [n]     per-env PCG32 state, advanced in place: the kernel draws
 *                             distinct Counters uniformly (same distribution as the
 *                             reference's rejection loop, not CPython's MT sequence)
 * `rng` wins when both are given. */

/* OvercookedEnvironment.reset() (overcooked_environment.py:180-206) for every env
 * whose mask[n] != 0 (mask NULL = all). */
OC_API int oc_reset(const oc_level_t *lv, int32_t *state, const int32_t *mask, const int32_t *placement,
                    uint32_t *rng, int64_t n, void *stream);

/* OvercookedEnvironment.step() (overcooked_environment.py:211-241): check_collisions
 * (:578-613), execute_navigation -> interact (:615-618, utils/interact.py:4-75),
 * done (:243-270), reward (:399-432), calculate_reward_shaping for sim agents 0 and 1
 * (:272-397).
 *   actions  int32 [A][n]  action codes 0..4 (OC_ACT_*), one row per sim agent
 *   reward   int32 [n]     the sparse reward
 *   done     int32 [n]
 *   shaping  double[2][n]  info["agent_0_reward_shaping"], info["agent_1_reward_shaping"]
 *   auto_reset != 0: an env that returns done is reset in the same launch (its state
 *   tensor then holds the fresh episode; reward/done/shaping are the terminal step's).
 *   metrics  int64 [oc_metrics_slots(n)][8] or NULL */
OC_API int oc_step(const oc_level_t *lv, int32_t *state, const int32_t *actions, int32_t *reward,
            int32_t *done, double *shaping, int32_t auto_reset, int64_t *metrics,
            const int32_t *placement, uint32_t *rng, int64_t n, void *stream);

/* OvercookedMultiEnv.get_observation2 for both viewers
 * (gym_comm/envs/overcooked_env.py:105-159).
 *   comm     int32 [2][n]  per_agent_communications as the index of the set bit, -1 = zeros */
OC_API int oc_obs(const oc_level_t *lv, const int32_t *state, const int32_t *comm, const oc_obs_cfg *cfg,
           void *obs, double *timestep, int64_t n, void *stream);

/* OvercookedMultiEnv.get_partial_observability_FOW for both viewers
 * (gym_comm/envs/overcooked_env.py:161-202), the image-style fog-of-war observation.
 * Plane k of a viewer at cell (x, y) -- the reference's map[k][x][y] -- is int8 byte x*H + y of
 * the plane, -1 = fogged; four consecutive cells of a plane of one env travel in one dword:
 *   out      int32 [2][7][Q][n], Q = ceil(W*H / 4) = oc_image_words() / 7; byte b of word q =
 *            cell 4q + b (zero past W*H)
 *   holding  int8 [2][n]                     (agent 0 holds, agent 1 holds) */
OC_API int32_t oc_image_words(const oc_level_t *lv);                 /* 7 * ceil(W*H / 4) */
OC_API int oc_obs_image(const oc_level_t *lv, const int32_t *state, int32_t radius, int32_t *out,
                        int8_t *holding, int64_t n, void *stream);

/* OvercookedMultiEnv.multi_step (gym_comm/envs/overcooked_env.py:207-282) in ONE
 * launch: action decoding + CAN_MOVE gating + comm update (:220-262), the base step,
 * both observations, and the shaped reward (r - s0) - s1 (:282).  2-agent levels only
 * (the reference's wrapper only drives agent-0 and agent-1).
 *   actions  int32 [4][n]  ego move (0..3), ego comm, alt move, alt comm
 *   comm     int32 [2][n]  in/out, persists across resets (:89-91,284-297)
 *   reward   double[n]     shaped reward, identical for both agents
 *   sparse   int32 [n] or NULL  the unshaped integer reward
 *   opts     optional inputs / outputs, see oc_step_opts (NULL = none) */
OC_API int oc_multi_step(const oc_level_t *lv, int32_t *state, int32_t *comm, const int32_t *actions,
                  const oc_wrap_cfg *cfg, void *obs, double *timestep, double *reward,
                  int32_t *done, int32_t *sparse, int32_t auto_reset, int64_t *metrics,
                  const int32_t *placement, uint32_t *rng, const oc_step_opts *opts,
                  int64_t n, void *stream);

/* A PREPARED oc_multi_step for callers that make the same call every step with, at most, another
 * ego-action tensor (OvercookedVecEnv.step_tensors: a Python caller pays ~0.3 us per argument it has to
 * convert; 17 of them were most of the 5.7 us a one-launch step cost on the host).  prepare() copies
 * every argument BY VALUE (cfg, opts and opts.policy included; the device pointers must stay valid);
 * launch() = oc_multi_step with those arguments, `ego_pairs` (if not NULL, with `pairs_int64`)
 * replacing opts.ego_pairs for this launch only.  Same checks, same kernels, same results. */
typedef struct oc_call oc_call_t;
OC_API int oc_multi_step_prepare(const oc_level_t *lv, int32_t *state, int32_t *comm, const int32_t *actions,
                                 const oc_wrap_cfg *cfg, void *obs, double *timestep, double *reward,
                                 int32_t *done, int32_t *sparse, int32_t auto_reset, int64_t *metrics,
                                 const int32_t *placement, uint32_t *rng, const oc_step_opts *opts, int64_t n,
                                 oc_call_t **out);
OC_API int oc_call_launch(const oc_call_t *call, const void *ego_pairs, int32_t pairs_int64, void *stream);
OC_API int oc_call_destroy(oc_call_t *call);

/* Waves per 64 envs oc_multi_step WILL LAUNCH for a batch of n envs given the caller's hint
 * (oc_step_opts.waves_per_64) and the variant the call selects -- general_variant != 0: any of
 * oc_step_opts' action sources / episode statistics / policy, or a non-standard wrapper
 * configuration; 0: the plain step.  1, 2 or 4: the general variant splits four ways or not at
 * all, the generic (unspecialised) library splits the plain variant four ways only; OC_LAUNCH
 * (a measurement knob, csrc/oc_kernels.hip: launch_policy) overrides the policy, never the
 * results.  Host only; for reports and tests. */
OC_API int32_t oc_multi_step_waves(int64_t n, int32_t hint, int32_t general_variant);

/* Measurement hook of the TIMELINE build flavour (the same source compiled with -DOC_TIMELINE=1;
 * gym-comm_amd/specialize.py, variant="timeline"; every other build returns OC_E_BADARG).  In such a
 * build every wave of the step kernels reads the chip-wide constant-rate counter (s_memrealtime,
 * 100 MHz) at its start and after its last instruction (-DOC_TIMELINE=2, variant
 * "timeline-drain": also after an added wait for its last store); the next `count` calls of
 * oc_step / oc_multi_step -- eager or captured into a hipGraph -- each get one record of `records`
 * (DEVICE memory, uint32 [count][stride][4]) into which wave w of the launch (w = workgroup * waves
 * per workgroup + wave, < 4 * ceil(n / 64) <= stride) stores its 16 bytes with ONE write-through store:
 *   [w][0], [w][1]  start, low and high word (64-bit tick count)
 *   [w][2]          issue-end - start in bits 0..15, drain-end - start in bits 16..31 (ticks, saturated
 *                   at 65 535 = 655 us; drain = issue unless OC_TIMELINE=2)
 *   [w][3]          shader-clock cycles (s_memtime) from start to the last stamp: against the realtime
 *                   span it gives the shader clock the wave ran at
 * A launch with fewer waves leaves the rest untouched, so the caller fills the records with 0xFF
 * before every run (a start of ~0 = "no wave") and reduces min start / max end over w.  Per launch,
 * max issue-end - min start is the span with waves on the chip ("kernel-active"), the next launch's
 * min start - max issue-end the launch boundary (store drain, end-of-kernel cache work, command
 * processor, dispatch); bench.py --decompose reports both beside the unchanged headline.  No
 * reference analogue.  records = NULL, count = 0 stops the recording. */
OC_API int oc_timeline_begin(uint64_t *records, int64_t count, int64_t stride);

/* Uniform random (move, comm) indices for one player of every env, written straight into two
 * rows of the [4][n] action tensor: move in 0..3, comm in 0..num_comm-1, from the env's own
 * PCG32 stream (`rng`, uint32 [n], advanced in place).  No reference analogue -- the
 * reference's partners are SB3 policies (pantheonrl/common/agents.py:60-194); this is the
 * cheapest partner for throughput runs, one launch instead of two torch generator ops. */
OC_API int oc_random_actions(uint32_t *rng, int32_t *move_row, int32_t *comm_row, int32_t num_comm,
                             int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif

// oc_kernels.hip -- hand-written CDNA4 (gfx950) kernels + the C ABI of liboc_hip.so.
//
// One lane = one environment; the env's whole dynamic state (A + M + 2 packed int32
// words, include/oc_hip.h) lives in VGPRs for the step.  All global tensors are
// env-major SoA, so every wave load/store touches 256 contiguous bytes.
//
// Everything that is the same for all envs of a launch (map bit-planes, subtask goal
// sets, item types, lookup programs for reward shaping) travels BY VALUE in the kernel
// argument block: it is read with scalar loads into SGPRs, costs no VGPRs, no LDS and
// no workgroup barrier, and makes every test on it a scalar branch.  The only tables a
// lane indexes with its own data -- the cell-to-cell path-distance table (u8) and the
// table of fp64 quotients k / MAX_PATH -- are read straight from global memory (2.4 KB +
// 2 KB, resident in every CU's vector L1 after first touch); the kernels were measured
// with those tables staged in LDS first (round-1 "v1", profiles/r01_v1_*) and the
// staging loop + barrier + LDS byte reads dominated the critical path at the BASELINE
// batch sizes, where only 1-2 waves per SIMD exist to hide latency.
//
// Control flow inside a step is branch-free on per-lane data: interact() is a decision
// phase (which of move / deliver / merge / chop / drop / pick fires) followed by
// predicated per-item updates; done/reward use a bitmask of "which goal object exists"
// instead of loops over subtasks.  Pure integer / indexing work; the fp64 shaping terms
// are table lookups plus adds in the reference's order (bit-exact).  No MFMA.
//
// Semantics follow the reference line by line (citations = paths relative to the
// reference root); the data model is our own: an Object is an equivalence class over M
// base items, each item carrying {cell, chopped, group, holder, world-order rank,
// type-set of its Object} in one register.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>

#include "../../include/oc_hip.h"
#include "oc_policy_device.h"

namespace {

constexpr int MAX_GOALS = 16;   // distinct goal type-sets (all 15 non-empty subsets of 4 types fit)
constexpr int MAX_DELS = 4;     // Deliver subtasks
constexpr int MAX_PAIRLK = 12;  // item-pair distance lookups of the shaping pair term
constexpr int MAX_NAMES = 16;   // dup mode: distinct merged names (4 bits of key rank each, two state words)

// Uniform per-level data.  Generic build: passed by value in the kernel arguments
// (scalar loads).  Specialised build (-DOC_SPECIALIZED, one .so per level, see
// gym-comm_amd/specialize.py): a constexpr object, so every loop bound, type test and
// bit-plane below folds at compile time and the kernels become straight-line code.
struct LevelHdr {
  int32_t W, H, ncells, max_path, S, A, M;
  uint32_t item_types;   // nibble i = content type of item i
  uint64_t nonfloor[2];  // bit c: cell c (= y*W + x) is not Floor
  uint64_t cell_lo[2], cell_hi[2];  // cell type bit-planes
  uint32_t nondeliver_mask, deliver_mask;  // subtask bitmasks by kind
  uint32_t chop_mask[3];  // per food type: Chop(food) subtasks
  uint32_t food_item[3];  // per food type: index of its item (255 = absent)
  uint32_t ngoal;
  uint32_t goal_tset[MAX_GOALS];  // distinct goal type-sets (bit t = type t present)
  uint32_t goal_nd[MAX_GOALS];    // Chop/Merge subtasks whose goal is that set
  uint32_t goal_dl[MAX_GOALS];    // Deliver subtasks whose goal is that set
  uint32_t ndel;
  uint32_t del_tset[MAX_DELS], del_bit[MAX_DELS];  // Deliver subtasks in subtask order
  uint32_t npairlk;
  uint32_t pairlk[MAX_PAIRLK];  // i | j<<4 | last_of_group<<8
  uint32_t pair_static_max;     // name pairs with an absent type: each appends MAX_PATH
  uint32_t ndeliv;
  uint32_t deliv_pos[OC_MAX_DELIV];  // x | y<<4, world order
  int32_t init_words[OC_MAX_AGENTS + OC_MAX_ITEMS + 4];
  uint32_t nquot;  // entries in the quotient table
  // random-* levels: items placed on random Counter tiles at every reset
  uint32_t nscatter, ncounters;
  uint32_t scatter_item[4];  // world-order item id of each scattered letter, file order
  // levels that repeat a content type ("dup" mode; no shipped level does): an Object is then a
  // MULTISET of types, and the kernels are instantiated with DUP = true
  uint32_t has_dup;
  uint32_t goal_sig[MAX_GOALS];  // per distinct goal: its content counts (sig7: T | L<<2 | O<<4 | P<<6)
  uint32_t del_sig[MAX_DELS];    // the same for the Deliver subtasks, subtask order
  uint32_t food_items[3];        // per food type: bit i = item i is of that type
  uint32_t nnames;               // merged names (multisets with >= 2 contents) a merge can create
  uint32_t name_sig[MAX_NAMES];
  // two facts about the MAP that select code paths at compile time in a specialised build
  uint32_t closed_border;        // every border cell is a non-Floor tile (see OC_BORDER_CLOSED)
  uint32_t planes128;            // more than 64 cells: the tile bit-planes need both 64-bit words
};

// per-run settings that do not select a specialisation
struct RunCfg {
  int32_t T;          // arglist.max_num_timesteps (0 = no limit)
  uint32_t allergic;  // bit a: agent a is ALLERGIC
  double inv_T;       // 1.0 / T, correctly rounded on the host (0 when T == 0)
  double inv_max_path;  // 1.0 / MAX_PATH, correctly rounded on the host
  // The caller's subtask order is run-time data too.  Inside the kernels subtask bits sit in a
  // CANONICAL order (Chop / Merge subtasks sorted by kind, goal object and food; Deliver subtasks
  // after them in the caller's order, which the fp64 shaping sum follows), so one specialised
  // library serves every order of a level -- the reference's own order is `set` iteration order
  // and changes with PYTHONHASHSEED (recipe_planner/stripsworld.py:72-77).  slot4: byte s = the
  // bit of the caller's subtask s; only the completed_subtasks observation rows need it.
  uint32_t slot_identity;
  uint32_t slot4[OC_MAX_SUBTASKS / 4];
  uint32_t play;   // arglist.play: the "playable" branches of interact() (utils/interact.py:44-47,52,66-67)
};

#ifdef OC_STAMPS
// Diagnostic build only (never shipped, never timed): s_memtime stamps of the phases of
// k_multi_step, written by lane 0 of every wave to a debug buffer nothing else reads
// (cdna_hip_programming.md section 7, "In-kernel stamps").
#define OC_STAMP(k)                                                                  \
  do {                                                                               \
    unsigned long long t_;                                                           \
    __builtin_amdgcn_sched_barrier(0);                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
    __builtin_amdgcn_sched_barrier(0);                                               \
    oc_tt[(k)] = t_;                                                                 \
  } while (0)
#define OC_STAMP_PARAM , unsigned long long (&oc_tt)[16]
#define OC_STAMP_PASS , oc_tt
#else
#define OC_STAMP(k) do { } while (0)
#define OC_STAMP_PARAM
#define OC_STAMP_PASS
#endif

#ifdef OC_TIMELINE
// Diagnostic build flavour (never the product, never the headline): every wave of k_step /
// k_multi_step reads the constant-rate 100 MHz counter (s_memrealtime: the same clock on every XCD,
// unlike the per-XCD shader clock of s_memtime) when it starts and when its last instruction has
// been issued, and lane 0 writes them -- and the wave's lifetime in shader-clock cycles -- to the
// wave's OWN 16 bytes of the launch's record, uint32 [stride][4].  A graph of chained launches
// replayed on such a build yields, per launch, the span in which the kernel had waves on the chip
// ("kernel-active") and the gap to the next launch's first wave (the launch boundary: store drain,
// end-of-kernel cache work, the command processor, the next dispatch) -- the split of ms_per_step
// that bench.py --decompose reports, without a profiler attached (include/oc_hip.h:
// oc_timeline_begin).  -DOC_TIMELINE=2 additionally waits for the wave's stores (s_waitcnt
// vmcnt(0)) and stamps that too: how much of the boundary is store drain.
// Earlier forms perturbed what they measured: same-address atomics (4 x 256 waves on one line:
// 14.6 us per launch instead of 3.1); a store wait + default-policy stamp stores in every wave
// (+0.45 us: the wave outlives its stores and leaves dirty lines for the end-of-kernel write-back);
// four 8-byte write-through stores per wave (+0.2 us at 4 096 envs, +1 us at 131 072).  Now: ONE
// 16-byte write-through store per wave.
#define OC_TL_BEGIN()                                                            \
  const unsigned long long oc_tl0_ = __builtin_amdgcn_s_memrealtime();           \
  const unsigned long long oc_tc0_ = __builtin_amdgcn_s_memtime()
#define OC_TL_END(ptr_, stride_)                                                                 \
  do {                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                           \
    const unsigned long long oc_tl1_ = __builtin_amdgcn_s_memrealtime();                         \
    unsigned long long oc_tl2_ = oc_tl1_;                                                        \
    if (OC_TIMELINE >= 2) {                                                                      \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
      oc_tl2_ = __builtin_amdgcn_s_memrealtime();                                                \
    }                                                                                            \
    const unsigned long long oc_tc2_ = __builtin_amdgcn_s_memtime();                             \
    unsigned long long *tl_ = (ptr_);                                                            \
    if (tl_ != nullptr && (threadIdx.x & 63) == 0) {                                             \
      /* ONE 16-byte write-through store per wave: {start (64 bits), issue-end - start | (drain-end  \
         - start) << 16, shader cycles}; spans are < 65 536 ticks (655 us) */                     \
      typedef int v4i_ __attribute__((ext_vector_type(4)));                                      \
      const int64_t w_ = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);           \
      const __amdgpu_buffer_rsrc_t r_ = __builtin_amdgcn_make_buffer_rsrc(tl_, 0, 0x7FFFFFFF, 0x00020000); \
      const unsigned d1_ = (unsigned)min((unsigned long long)0xFFFF, oc_tl1_ - oc_tl0_);         \
      const unsigned d2_ = (unsigned)min((unsigned long long)0xFFFF, oc_tl2_ - oc_tl0_);         \
      v4i_ q_;                                                                                   \
      q_.x = (int)(unsigned)oc_tl0_;                                                             \
      q_.y = (int)(unsigned)(oc_tl0_ >> 32);                                                     \
      q_.z = (int)(d1_ | (d2_ << 16));                                                           \
      q_.w = (int)(unsigned)(oc_tc2_ - oc_tc0_);                                                 \
      __builtin_amdgcn_raw_buffer_store_b128(q_, r_, (int)w_ * 16, 0, 16);                       \
    }                                                                                            \
  } while (0)
#else
#define OC_TL_BEGIN() do { } while (0)
#define OC_TL_END(ptr_, stride_) do { } while (0)
#endif

// The kernels read the level through ACCESSORS (L.W(), L.goal_tset(g), ...), one list of fields
// (OC_HDR_FIELDS) for two header classes.  The fields come in two kinds:
//   STRUCTURE  what the recipes and the item multiset fix -- subtask masks, goal objects, item
//              types, the shaping lookup programs, the two map flags above;
//   GEOMETRY   the map itself -- size, tile bit-planes, Delivery positions, start cells (the
//              distance and Counter tables are device buffers anyway).
//   HdrC  specialised build: STRUCTURE accessors return fields of the constexpr OC_SPEC_HDR, so
//         loop bounds, type tests and masks fold at compile time.  GEOMETRY accessors come in
//         two flavours of library:
//           -DOC_SPEC_GEOMETRY  ("level" library) constexpr as well: everything folds, the
//               fastest code (3.64 us per step, tomato-2 x 4096), valid for ONE map;
//           otherwise           ("structure" library) they read the by-value kernel argument (a
//               dozen scalar loads; 3.82 us): the library is keyed by the STRUCTURE alone, so
//               every map with the same recipes, item multiset, agent count and border kind
//               runs on it -- all `*_tomato` levels share one, and so does a user-made map with
//               those recipes on a box that has no hipcc (the generic library takes 7.3 us).
//   HdrK  generic build: both kinds are scalar loads from the kernel arguments.
// (Tried and dropped in round 2: the header spread over the lanes of three VGPRs, one
// v_readlane per access.  Slower -- 7.68 us per step at 4 096 envs against 6.97 us with kernarg
// loads: ~290 readlanes with their SGPR-hazard wait states cost more than the scalar-cache hits
// they replace -- and unsafe: the compiler may copy such a VGPR under a partial EXEC mask, which
// loses the words parked in inactive lanes.)
#define OC_HDR_FIELDS(FLD, ARR, ARR64, GFLD, GARR, GARR64)                                           \
  GFLD(int32_t, W) GFLD(int32_t, H) GFLD(int32_t, ncells) GFLD(int32_t, max_path) FLD(int32_t, S)  \
  FLD(int32_t, A) FLD(int32_t, M) FLD(uint32_t, item_types)                                        \
  GARR64(nonfloor) GARR64(cell_lo) GARR64(cell_hi)                                                 \
  FLD(uint32_t, nondeliver_mask) FLD(uint32_t, deliver_mask)                                       \
  ARR(uint32_t, chop_mask) ARR(uint32_t, food_item)                                                \
  FLD(uint32_t, ngoal) ARR(uint32_t, goal_tset) ARR(uint32_t, goal_nd) ARR(uint32_t, goal_dl)     \
  FLD(uint32_t, ndel) ARR(uint32_t, del_tset) ARR(uint32_t, del_bit)                               \
  FLD(uint32_t, npairlk) ARR(uint32_t, pairlk) FLD(uint32_t, pair_static_max)                      \
  FLD(uint32_t, ndeliv) GARR(uint32_t, deliv_pos) GARR(int32_t, init_words) GFLD(uint32_t, nquot) \
  FLD(uint32_t, nscatter) GFLD(uint32_t, ncounters) ARR(uint32_t, scatter_item)                    \
  FLD(uint32_t, has_dup) ARR(uint32_t, goal_sig) ARR(uint32_t, del_sig) ARR(uint32_t, food_items) \
  FLD(uint32_t, nnames) ARR(uint32_t, name_sig) FLD(uint32_t, closed_border) FLD(uint32_t, planes128)

#ifdef OC_SPECIALIZED
#include OC_SPEC_FILE  // constexpr LevelHdr OC_SPEC_HDR = {...};
struct HdrC {
  const LevelHdr &k;   // the by-value kernel argument: the map's geometry
#define OC_F(T, name) __device__ __forceinline__ constexpr T name() const { return OC_SPEC_HDR.name; }
#define OC_A(T, name) __device__ __forceinline__ constexpr T name(int i) const { return OC_SPEC_HDR.name[i]; }
#define OC_A64(name) __device__ __forceinline__ constexpr uint64_t name(int i) const { return OC_SPEC_HDR.name[i]; }
#ifdef OC_SPEC_GEOMETRY
#define OC_GF(T, name) OC_F(T, name)
#define OC_GA(T, name) OC_A(T, name)
#define OC_GA64(name) OC_A64(name)
#else
#define OC_GF(T, name) __device__ __forceinline__ T name() const { return k.name; }
#define OC_GA(T, name) __device__ __forceinline__ T name(int i) const { return k.name[i]; }
#define OC_GA64(name) __device__ __forceinline__ uint64_t name(int i) const { return k.name[i]; }
#endif
  OC_HDR_FIELDS(OC_F, OC_A, OC_A64, OC_GF, OC_GA, OC_GA64)
#undef OC_F
#undef OC_A
#undef OC_A64
#undef OC_GF
#undef OC_GA
#undef OC_GA64
};
using Hdr = HdrC;
#define OC_HDR_LOAD(args) const HdrC L {(args).L}
#else
struct HdrK {
  const LevelHdr &k;   // the by-value kernel argument
#define OC_F(T, name) __device__ __forceinline__ T name() const { return k.name; }
#define OC_A(T, name) __device__ __forceinline__ T name(int i) const { return k.name[i]; }
#define OC_A64(name) __device__ __forceinline__ uint64_t name(int i) const { return k.name[i]; }
  OC_HDR_FIELDS(OC_F, OC_A, OC_A64, OC_F, OC_A, OC_A64)
#undef OC_F
#undef OC_A
#undef OC_A64
};
using Hdr = HdrK;
#define OC_HDR_LOAD(args) const HdrK L {(args).L}
#endif

}  // namespace

struct oc_level {
  LevelHdr hdr;
  RunCfg run;
  int32_t slot[OC_MAX_SUBTASKS];        // canonical bit of the caller's subtask s
  int32_t goal_index[OC_MAX_SUBTASKS];  // index of its distinct goal object (dup mode: where its count lives)
  void *dev_tables;    // [nquot] fp64 quotients k / max_path, then [ncells*ncells] u8 distances
  int32_t n16;         // table bytes / 16 (rounded up)
  int32_t quot_bytes;
  int device;
};

// A prepared oc_multi_step (oc_multi_step_prepare): every argument by value
struct oc_call {
  const oc_level *lv;
  int32_t *state, *comm;
  const int32_t *actions;
  oc_wrap_cfg cfg;
  void *obs;
  double *timestep, *reward;
  int32_t *done, *sparse;
  int32_t auto_reset;
  int64_t *metrics;
  const int32_t *placement;
  uint32_t *rng;
  oc_step_opts opts;
  oc_step_policy pol[2];
  bool has_pol;
  int64_t n;
};

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char *msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
int fail_hip(hipError_t e, const char *what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return (int)e;
}

// ---------------------------------------------------------------------------
// per-env registers
// ---------------------------------------------------------------------------
// Items stay PACKED in registers exactly as they sit in the state tensor (include/oc_hip.h):
//   x | y<<4 | chopped<<8 | group<<9 | (holder+1)<<12 | seq<<16 | tset<<24
// Every item of one Object carries the same group / seq / tset, and on a non-Delivery cell at
// most one unheld Object exists, so "the held Object" and "the Object on the target cell" are
// plain ORs over the matching item words, a field test is one AND + compare on the word, and
// an update of several fields is one bit-field insert (v_bfi_b32).
constexpr int IW_POS = 0x000000FF, IW_CHOP = 0x00000100, IW_GRP = 0x00000E00, IW_HOLD = 0x00007000,
              IW_SEQ = 0x00FF0000, IW_TSET = 0x0F000000;
// "dup" mode (template bool DUP; a level that repeats a content type): bits 24..30 hold the
// Object's content COUNTS instead of its type set -- T | L<<2 | O<<4 (two bits each, at most
// three of a food) | P<<6 -- and bits 16..23 hold kseq<<4 | seq: seq = the Object's insertion
// number as before (4 bits are enough: at most M - 1 <= 7 merges per episode), kseq = the seq of
// the FIRST Object ever inserted under the same name this episode, i.e. the creation rank of its
// key in the reference's dict of lists (utils/world.py:21,236-237).  world.objects iterates key
// by key, so the composite is the world-order rank -- and equals seq<<4 | seq whenever every
// name is created at most once, which is why non-dup levels never needed it.
template <bool DUP>
constexpr int TS = DUP ? 0x7F000000 : IW_TSET;
template <bool DUP>
constexpr int OBJ = IW_GRP | IW_SEQ | TS<DUP>;   // what a merge rewrites
template <bool DUP>
__host__ __device__ constexpr int sig_of_type(int t) { return DUP ? (1 << (24 + 2 * t)) : (1 << (24 + t)); }

template <int A, int M, bool DUP>
struct Env {
  int ap[A], ahp[A];   // agent cell (x | y<<4); held group + 1 (0 = empty hands)
  int iw[M];           // packed item words
  int t, completed, goalcnt, mctr, err;
  int kn[2];           // DUP: 4 bits per merged name = 1 + seq of the first Object created under it (0 = never)
};
template <int A, int M, bool DUP>
constexpr int state_words() { return A + M + 2 + (DUP ? 2 : 0); }

__device__ __forceinline__ int ipos(int w) { return w & IW_POS; }
__device__ __forceinline__ int ichop(int w) { return (w >> 8) & 1; }
__device__ __forceinline__ int igrp(int w) { return (w >> 9) & 7; }
__device__ __forceinline__ int iseq(int w) { return (w >> 16) & 255; }
__device__ __forceinline__ int itset(int w) { return (w >> 24) & 15; }
__device__ __forceinline__ int bfi(int mask, int a, int b) { return (a & mask) | (b & ~mask); }  // v_bfi_b32

template <int A, int M, bool DUP>
__device__ __forceinline__ void unpack(Env<A, M, DUP> &e, const int32_t *w) {
#pragma unroll
  for (int a = 0; a < A; a++) {
    e.ap[a] = w[a] & 255;
    e.ahp[a] = (w[a] >> 8) & 15;
  }
#pragma unroll
  for (int i = 0; i < M; i++) e.iw[i] = w[A + i];
  e.t = (w[0] >> 16) & 0xFFFF;
  e.mctr = (w[1] >> 16) & 255;
  e.err = (w[1] >> 24) & 255;
  e.completed = w[A + M];
  e.goalcnt = w[A + M + 1];
  if constexpr (DUP) e.kn[0] = w[A + M + 2], e.kn[1] = w[A + M + 3];
}

template <int A, int M, bool DUP>
__device__ __forceinline__ void pack(const Env<A, M, DUP> &e, int32_t *w) {
#pragma unroll
  for (int a = 0; a < A; a++) w[a] = e.ap[a] | (e.ahp[a] << 8);
  w[0] |= e.t << 16;
  w[1] |= (e.mctr << 16) | (e.err << 24);
#pragma unroll
  for (int i = 0; i < M; i++) w[A + i] = e.iw[i];
  w[A + M] = e.completed;
  w[A + M + 1] = e.goalcnt;
  if constexpr (DUP) w[A + M + 2] = e.kn[0], w[A + M + 3] = e.kn[1];
}

// ---------------------------------------------------------------------------
// per-lane predicates as VALU words
// ---------------------------------------------------------------------------
// A per-lane `bool` of the compiler lives in an SGPR pair (v_cmp -> s[n:n+1]) and its logic runs on
// the scalar unit (s_and_b64 / s_or_b64 ...), to come back through v_cndmask.  On gfx950 a scalar
// instruction that reads an SGPR a VECTOR instruction has just written stalls a lone wave ~16
// cycles (tools/issue_probe.hip, one wave on a SIMD: v_cmp + s_and = 24 cycles, the chain v_cmp ->
// s_and -> v_cndmask 28; the same decision on 0 / -1 words in VGPRs -- v_cmp/v_cndmask or
// v_bfe_i32 to make the word, v_and / v_or / v_bitop3 for the logic, v_bfi to select -- 4 per
// instruction), and interact()'s decision tree is a few hundred such hops: the step kernels ran
// at ~7 cycles per instruction where 4 is the issue rate.  So every per-lane predicate of the hot
// path is a P: an int that is 0 or -1, made opaque to the optimiser where it is born (`hide`:
// an empty asm; otherwise InstCombine folds and/or of sign-extended compares back into i1 logic
// and the backend selects the scalar unit again).
typedef int P;
__device__ __forceinline__ int hide(int v) {
  asm("" : "+v"(v));
  return v;
}
// How a P is born.  v_cmp + v_cndmask(0, -1) is two instructions, but every v_cmp writes vcc -- one
// register for all of them, so the pairs cannot interleave -- and a v_cndmask reading it needs two
// wait states behind the compare (the 3-agent kernel carried 80 s_nop).  For operands in
// [0, 2^31) -- every field of the packed state -- the sign of a difference is the predicate:
// two plain, independent VALU instructions (v_xad_u32 / v_sub + v_ashrrev_i32), no vcc, no nop.
__device__ __forceinline__ P p_z(int a) { return hide((int)((unsigned)a - 1u) >> 31); }              // a == 0   (a >= 0)
__device__ __forceinline__ P p_nz(int a) { return hide((int)(0u - (unsigned)a) >> 31); }             // a != 0   (a >= 0)
__device__ __forceinline__ P p_eq(int a, int b) { return hide((int)((unsigned)(a ^ b) - 1u) >> 31); }  // (bit 31 of a, b equal)
__device__ __forceinline__ P p_ne(int a, int b) { return ~p_eq(a, b); }
__device__ __forceinline__ P p_lt(int a, int b) { return hide((a - b) >> 31); }                      // a < b    (0 <= a, b < 2^31)
__device__ __forceinline__ P p_gt(int a, int b) { return p_lt(b, a); }
__device__ __forceinline__ P p_ge(int a, int b) { return ~p_lt(a, b); }
__device__ __forceinline__ P p_le(int a, int b) { return ~p_lt(b, a); }
// ... and for operands of any value (caller-supplied action indices, 32-bit subtask masks): the compare
__device__ __forceinline__ P p_eq_any(int a, int b) { return hide(a == b ? -1 : 0); }
__device__ __forceinline__ P p_gtu_any(unsigned a, unsigned b) { return hide(a > b ? -1 : 0); }
__device__ __forceinline__ P p_geu_any(unsigned a, unsigned b) { return hide(a >= b ? -1 : 0); }
__device__ __forceinline__ P p_ltu_any(unsigned a, unsigned b) { return hide(a < b ? -1 : 0); }
__device__ __forceinline__ P p_bit(int w, unsigned k) { return __builtin_amdgcn_sbfe(w, k, 1u); }  // bit k as 0 / -1: one v_bfe_i32
__device__ __forceinline__ P p_of(bool uniform) { return uniform ? -1 : 0; }                       // a wave-uniform condition
__device__ __forceinline__ int sel(P m, int a, int b) { return (a & m) | (b & ~m); }               // v_bfi_b32

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }
// |a - b| + c on unsigned operands in one instruction (hipcc expands __sad() into compare,
// two subtracts, select and add)
__device__ __forceinline__ unsigned sad_u32(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ int manhattan(int p, int q) {   // packed cells x | y<<4
  return (int)sad_u32((unsigned)(p & 15), (unsigned)(q & 15), sad_u32((unsigned)(p >> 4), (unsigned)(q >> 4), 0u));
}
__device__ __forceinline__ int px(int p) { return p & 15; }
__device__ __forceinline__ int py(int p) { return p >> 4; }
__device__ __forceinline__ int dense(const Hdr &L, int p) {
  // y * W + x from p = x + 16 y without unpacking x: p - (16 - W) * y  (v_lshrrev + v_mad_i32_i24)
  return __mul24(py(p), L.W() - 16) + p;
}
// bit c of a 128-bit plane held as two 64-bit words
__device__ __forceinline__ int bit128(uint64_t w0, uint64_t w1, int c) {
  const uint64_t v = (c & 64) ? w1 : w0;
  return (int)((v >> (c & 63)) & 1);
}
__device__ __forceinline__ int item_type(const Hdr &L, int i) { return (L.item_types() >> (4 * i)) & 15; }
// tile type (OC_FLOOR / COUNTER / CUTBOARD / DELIVERY) of dense cell c
__device__ __forceinline__ int cell_type(const Hdr &L, int c) {
  if (!L.planes128())  // uniform (compile-time in specialised builds): one 64-bit plane each
    return (int)((L.cell_lo(0) >> c) & 1) | ((int)((L.cell_hi(0) >> c) & 1) << 1);
  return bit128(L.cell_lo(0), L.cell_lo(1), c) | (bit128(L.cell_hi(0), L.cell_hi(1), c) << 1);
}

// Every border cell of the map is a non-Floor tile: agents (always on Floor) can then never
// propose a cell outside the map, so the proposal needs no bounds test, no clamp and no
// OC_ERR_OOB path.  True for all fixed levels of the reference; decided at compile time in a
// per-level specialised build, never assumed by the generic library.
constexpr bool border_closed(const LevelHdr &L) {
  for (int y = 0; y < L.H; y++)
    for (int x = 0; x < L.W; x++)
      if (x == 0 || y == 0 || x == L.W - 1 || y == L.H - 1) {
        const int c = y * L.W + x;
        if (!((L.nonfloor[c >> 6] >> (c & 63)) & 1)) return false;
      }
  return true;
}
#ifdef OC_SPECIALIZED
constexpr bool OC_BORDER_CLOSED = OC_SPEC_HDR.closed_border != 0;   // (set by build_header from the map)
#else
constexpr bool OC_BORDER_CLOSED = false;
#endif

// t / T as CPython computes it (overcooked_env.py:146: int / int, correctly rounded fp64)
// without the 11-instruction fp64 division: q0 = t * RN(1/T), one FMA for the exact
// residual, one FMA to correct.  Equal to the division for every 0 <= t, 1 <= T <= 65535
// (all 4.3e9 pairs compared bit for bit: tests/test_host_cpu.py, tools/div_check.c).
__device__ __forceinline__ double timestep_of(int t, const RunCfg &R) {
  const double dt = (double)t;
  if (R.T == 0) return t == 0 ? __builtin_nan("") : __builtin_inf();  // uniform; no time limit: t / 0.0
  const double q0 = dt * R.inv_T;
  const double r = __builtin_fma(-(double)R.T, q0, dt);
  return __builtin_fma(r, R.inv_T, q0);
}


// An [R][n] tensor of 4-byte (or 8-byte) elements addressed through a buffer resource:
// the per-lane part of the address is one 32-bit byte offset (voffset), the row offset is
// a scalar (soffset), so a row access costs no vector address arithmetic at all.  The
// descriptor's record count bounds the whole tensor; the tail lanes of the last wave are
// still masked by the caller because rows are contiguous.
//
// AUX = cache-policy bits of the stores (bit 0 sc0, bit 1 nt, bit 4 sc1).  With sc1 (agent
// scope) the L2 writes a line through to the fabric at once instead of holding it dirty
// until the end-of-kernel write-back, so the write-back overlaps the rest of the launch
// instead of trailing it.  tools/store_probe.hip (76 row stores per env): 10.4 -> 7.9 us
// per launch at n = 131072, 4.4 -> 3.7 us at 32768; sc0 / nt change nothing.  In the
// kernels (round 1, v5): salad-2 x 32768 6.84 -> 5.88 us, tomato-2 x 131072 11.1 -> 10.0 us, but
// 4.56 -> 4.66 us at n = 4096 -- back then every wave still ended on a wait for its own stores
// (the metrics slot was a load + store), and a lone wave per CU waited longer for a written-
// through one.  Since v11 no wave waits for its stores, and round 2 re-measured
// (tools/wt_threshold.sh): write-through is never slower, 3.64 -> 3.51 us at n = 4096, 3.53 ->
// 3.46 us at 512, 3.44 -> 3.43 us at 64, salad-2 x 4096 3.89 -> 3.74 us.  The launcher picks the
// write-through variant (template bool WT) at every batch size; OC_LAUNCH=wt=0 restores
// write-back stores.
template <int AUX>
struct RowsT {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff;      // lane byte offset inside a row
  int rowbytes;  // n * element size
  __device__ __forceinline__ RowsT(const void *base, int64_t n, int rows, int64_t i, int elem = 4) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)(n * rows * elem), 0x00020000);
    voff = (int)i * elem;
    rowbytes = (int)n * elem;
  }
  __device__ __forceinline__ int ld(int row) const {
    return __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, row * rowbytes, 0);
  }
  __device__ __forceinline__ void st(int row, int v) const {
    __builtin_amdgcn_raw_buffer_store_b32(v, rsrc, voff, row * rowbytes, AUX);
  }
  __device__ __forceinline__ void st8(int row, int v) const {   // rows of 1-byte elements
    __builtin_amdgcn_raw_buffer_store_b8((char)v, rsrc, voff, row * rowbytes, AUX);
  }
  __device__ __forceinline__ void st_f64(int row, double v) const {
    typedef int v2i __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v), rsrc, voff, row * rowbytes, AUX);
  }
};
// RowsT that also leaves every stored value, as a float, in an LDS image [row - rowbase][64 lanes]:
// how a split workgroup's observation waves hand a viewer's rows to the waves that evaluate the
// policies (multi_step_body, POL) without a trip through memory.  OT: 0 int32, 1 int8, 2 float32 bits.
template <int AUX, int OT>
struct RowsLdsT : RowsT<AUX> {
  float *lds;
  int rowbase, lane;
  __device__ __forceinline__ RowsLdsT(const RowsT<AUX> &rows, float *lds_, int rowbase_, int lane_)
      : RowsT<AUX>(rows), lds(lds_), rowbase(rowbase_), lane(lane_) {}
  __device__ __forceinline__ void st(int row, int v) const {
    RowsT<AUX>::st(row, v);
    lds[(row - rowbase) * 64 + lane] = OT == 2 ? __builtin_bit_cast(float, v) : (float)v;
  }
  __device__ __forceinline__ void st8(int row, int v) const {
    RowsT<AUX>::st8(row, v);
    lds[(row - rowbase) * 64 + lane] = (float)v;
  }
};
using Rows = RowsT<0>;          // loads and default-policy stores
#ifndef OC_AUX_WT
#define OC_AUX_WT 16            // sc1; -DOC_AUX_WT=<bits> via OC_HIP_EXTRA_FLAGS to try other store policies
#endif
constexpr int AUX_WT = OC_AUX_WT;

// calculate_reward_shaping for sim agents 0 and 1 (overcooked_environment.py:272-397),
// given the agents' cells, the item cells, the completed flags and, per Deliver subtask, the
// cell of its (all-chopped) object if one exists.  The int/int divisions of the reference
// are entries of the quotient table k / MAX_PATH; sums run left to right in fp64.
//
// Split so the kernels can order their memory operations: shaping_issue_pos/_del() form the
// addresses and issue the path-distance loads; shaping_lookup() consumes the distances and
// issues the quotient loads -- still ahead of the observation stores, because vmcnt retires
// loads and stores in issue order; shaping_sum() does the fp64 adds after the stores.
template <int B>
struct ShapeIn {   // what shaping_lookup() / shaping_sum() still need of the pre-reset env
  int ap[B];
  int completed;
  int del_has[MAX_DELS], del_p[MAX_DELS];
  int chop_p[3];   // DUP: the cell of "the" fresh food of each type (the set's element [0])
};

// ---------------------------------------------------------------------------
// dup mode: which location does list(set(locations))[0] return?
// ---------------------------------------------------------------------------
// World.get_all_object_locs is list(set(held_locs + unheld_locs)) (utils/world.py:290-291) and
// calculate_reward_shaping walks to element [0] of it (overcooked_environment.py:287,374-379).
// With several matching objects that is the location in the LOWEST SLOT of CPython's 8-slot set
// table: slot = hash((x, y)) & 7, a taken slot sends the newcomer along its probe sequence
// i <- (5 i + 1 + (perturb >>= 5)) & 7 (Objects/setobject.c; no linear probing in an 8-slot
// table; the table only grows at the fifth element and at most three objects can match).  The
// host stores each cell's first eight probe slots, 3 bits each, in a u32 table (`probe`) and
// checks at level creation that eight are enough for every triple of cells.
// cand[i]: item i is the representative of a matching Object.  Insertion order: held Objects
// first, then unheld ones, each in the order of the name's list = ascending seq field.
template <int M>
__device__ __forceinline__ int pyset_first(const Hdr &L, const uint32_t *__restrict__ probe,
                                           const int (&iw)[M], const bool (&cand)[M], int &has) {
  constexpr int BIG = 1 << 20;
  int ord[M];
#pragma unroll
  for (int i = 0; i < M; i++)
    ord[i] = cand[i] ? (((iw[i] & IW_HOLD) ? 0 : 256) | ((iw[i] >> 16) & 255)) : BIG;
  int cell[3], code[3];
  bool valid[3];
  int prev = -1;
#pragma unroll
  for (int k = 0; k < 3; k++) {   // the k-th candidate in insertion order
    int cur = BIG, c = 0;
#pragma unroll
    for (int i = 0; i < M; i++) {
      const bool better = ord[i] > prev && ord[i] < cur;
      cur = better ? ord[i] : cur;
      c = better ? ipos(iw[i]) : c;
    }
    valid[k] = cur != BIG;
    cell[k] = c;
    prev = valid[k] ? cur : BIG;
  }
  has = valid[0];
  if (__ballot(valid[1]) == 0) return cell[0];   // one match in every env of the wave: no set order to ask for
  // a location already in the set is not inserted again
  valid[1] = valid[1] && cell[1] != cell[0];
  valid[2] = valid[2] && cell[2] != cell[0] && !(valid[1] && cell[2] == cell[1]);
#pragma unroll
  for (int k = 0; k < 3; k++) code[k] = valid[k] ? (int)probe[dense(L, cell[k])] : 0;
  const int s0 = code[0] & 7;
  int s1 = 8, s2 = 8;
  // first free slot along each newcomer's probe sequence (scanned backwards, so the earliest
  // probe that is free is the one that stays)
#pragma unroll
  for (int t = 7; t >= 0; t--) {
    const int q1 = (code[1] >> (3 * t)) & 7;
    s1 = (q1 != s0) ? q1 : s1;
  }
  s1 = valid[1] ? s1 : 8;
#pragma unroll
  for (int t = 7; t >= 0; t--) {
    const int q2 = (code[2] >> (3 * t)) & 7;
    s2 = (q2 != s0 && q2 != s1) ? q2 : s2;
  }
  s2 = valid[2] ? s2 : 8;
  int best = cell[0], bs = s0;
  best = s1 < bs ? cell[1] : best;
  bs = min(bs, s1);
  best = s2 < bs ? cell[2] : best;
  return best;
}
template <int B>
struct ShapeLoads {   // raw path distances, in flight until shaping_lookup()
  int d_chop[3][B];
  int d_pair[MAX_PAIRLK];
  int d_del[MAX_DELS][B];
  int d_tile[OC_MAX_DELIV][B];
};

// The lookups that only need positions (Chop, pair and Delivery-tile terms): issued right after
// interact(), a hundred instructions before the rest, so they are back when shaping_lookup()
// wants them.  Table offsets are unsigned 24-bit products: full-rate v_mul_u32_u24 /
// v_mad_u32_u24 and a 32-bit offset on a scalar base (no 64-bit address arithmetic per lookup).
template <int B, int M, bool DUP>
__device__ __forceinline__ void shaping_issue_pos(const Hdr &L, const uint8_t *__restrict__ dist,
                                                  const ShapeIn<B> &in, const int (&ipos)[M], ShapeLoads<B> &ld) {
  const unsigned nc = (unsigned)L.ncells();
  unsigned arow[B];
#pragma unroll
  for (int b = 0; b < B; b++) arow[b] = __umul24((unsigned)dense(L, in.ap[b]), nc);
  unsigned ic[M];
#pragma unroll
  for (int i = 0; i < M; i++) ic[i] = (unsigned)dense(L, ipos[i]);
#pragma unroll
  for (int f = 0; f < 3; f++) {
#pragma unroll
    for (int b = 0; b < B; b++) ld.d_chop[f][b] = 0;
    if (L.chop_mask(f) != 0) {  // uniform
      unsigned fc = 0;
      if constexpr (DUP) {
        fc = (unsigned)dense(L, in.chop_p[f]);
      } else {
#pragma unroll
        for (int i = 0; i < M; i++) fc = ((int)L.food_item(f) == i) ? ic[i] : fc;
      }
#pragma unroll
      for (int b = 0; b < B; b++) ld.d_chop[f][b] = dist[arow[b] + fc];
    }
  }
#pragma unroll
  for (int k = 0; k < MAX_PAIRLK; k++) {
    ld.d_pair[k] = 0;
    if (k < (int)L.npairlk()) {  // uniform
      const int li = L.pairlk(k) & 15, lj = (L.pairlk(k) >> 4) & 15;
      unsigned ci = 0, cj = 0;
#pragma unroll
      for (int i = 0; i < M; i++) {
        ci = (li == i) ? ic[i] : ci;
        cj = (lj == i) ? ic[i] : cj;
      }
      ld.d_pair[k] = dist[__umul24(ci, nc) + cj];
    }
  }
#pragma unroll
  for (int k = 0; k < OC_MAX_DELIV; k++) {
#pragma unroll
    for (int b = 0; b < B; b++) ld.d_tile[k][b] = 0;
    if (k < (int)L.ndeliv()) {  // uniform
      const unsigned dc = (unsigned)dense(L, (int)L.deliv_pos(k));
#pragma unroll
      for (int b = 0; b < B; b++) ld.d_tile[k][b] = dist[arow[b] + dc];
    }
  }
}

// The Deliver-term lookups need the cell of each Deliver subtask's object (known after
// done/reward).
template <int B>
__device__ __forceinline__ void shaping_issue_del(const Hdr &L, const uint8_t *__restrict__ dist,
                                                  const ShapeIn<B> &in, ShapeLoads<B> &ld) {
  const unsigned nc = (unsigned)L.ncells();
#pragma unroll
  for (int k = 0; k < MAX_DELS; k++) {
#pragma unroll
    for (int b = 0; b < B; b++) ld.d_del[k][b] = 0;
    if (k < (int)L.ndel()) {  // uniform
      const unsigned mc = (unsigned)dense(L, in.del_p[k]);
#pragma unroll
      for (int b = 0; b < B; b++) ld.d_del[k][b] = dist[__umul24((unsigned)dense(L, in.ap[b]), nc) + mc];
    }
  }
}

// quotient values between shaping_lookup() and shaping_sum()
template <int B>
struct ShapeQ {
  double q_chop[B], q_pair, q_del[MAX_DELS][B];
  int nchop, npairs;
  bool del_direct[MAX_DELS][B];
};

// second part: consume the path distances (integer min / select logic) and form the
// quotients.  (Until round-1 v11 the quotients were table lookups, issued here -- ahead of the
// observation stores, because vmcnt retires loads and stores in issue order.)
// k / MAX_PATH as CPython computes it (int / int, correctly rounded fp64) by the two-FMA
// construction of timestep_of(): exact for every 0 <= k <= 65535, 1 <= MAX_PATH <= 65535
// (tools/div_check.c).  Replaces a table lookup -- five to seven global loads per env-step
// that had to be ordered around the stores.
__device__ __forceinline__ double quotient(int k, const Hdr &L, double inv_max_path) {
  const double dk = (double)k;
  const double q0 = dk * inv_max_path;
  const double r = __builtin_fma(-(double)L.max_path(), q0, dk);
  return __builtin_fma(r, inv_max_path, q0);
}

template <int B>
__device__ __forceinline__ void shaping_lookup(const Hdr &L, double inv_max_path, const ShapeIn<B> &in,
                                               const ShapeLoads<B> &ld, ShapeQ<B> &q OC_STAMP_PARAM) {
  const int MAXP = L.max_path();
  const int completed = in.completed;
  int d_tile[B];  // min over Delivery tiles of path distance + manhattan (:382-388)
#pragma unroll
  for (int b = 0; b < B; b++) d_tile[b] = 1 << 20;
#pragma unroll
  for (int k = 0; k < OC_MAX_DELIV; k++)
    if (k < (int)L.ndeliv()) {  // uniform
#pragma unroll
      for (int b = 0; b < B; b++)
        d_tile[b] = min(d_tile[b], ld.d_tile[k][b] + manhattan(in.ap[b], (int)L.deliv_pos(k)));
    }
  // Chop term (:278-304)
  int nchop = 0;
  int mind[B];
#pragma unroll
  for (int b = 0; b < B; b++) mind[b] = 1 << 20;
#pragma unroll
  for (int f = 0; f < 3; f++)
    if (L.chop_mask(f) != 0) {  // uniform
      const int open = __popc((int)L.chop_mask(f) & ~completed);
      nchop += open;
#pragma unroll
      for (int b = 0; b < B; b++) mind[b] = open ? min(mind[b], ld.d_chop[f][b]) : mind[b];
    }
  // pair term (:319-363): agent independent
  int npairs = (int)L.pair_static_max();
  int minpair = npairs ? MAXP : (1 << 20);
  {
    int cur = MAXP;
#pragma unroll
    for (int k = 0; k < MAX_PAIRLK; k++)
      if (k < (int)L.npairlk()) {  // uniform
        cur = min(cur, ld.d_pair[k]);
        if ((L.pairlk(k) >> 8) & 1) {  // uniform: last lookup of this name pair
          const bool keep = cur != 0;    // a zero distance is not appended (:351-352)
          npairs += keep ? 1 : 0;
          minpair = keep ? min(minpair, cur) : minpair;
          cur = MAXP;
        }
      }
  }
  // the quotients
  int kq_chop[B], kq_pair;
#pragma unroll
  for (int b = 0; b < B; b++) kq_chop[b] = nchop ? (mind[b] + MAXP) + (nchop - 1) * 2 * MAXP : 0;
  kq_pair = nchop ? npairs * MAXP : (npairs ? minpair + (npairs - 1) * MAXP : 0);
  int kq_del[MAX_DELS][B];
#pragma unroll
  for (int k = 0; k < MAX_DELS; k++)
#pragma unroll
    for (int b = 0; b < B; b++) {
      kq_del[k][b] = 0;
      q.del_direct[k][b] = false;
      if (k < (int)L.ndel()) {
        const int d = ld.d_del[k][b] + manhattan(in.ap[b], in.del_p[k]);
        q.del_direct[k][b] = d == 0;                 // the agent holds it (:381)
        kq_del[k][b] = d == 0 ? d_tile[b] : d;
      }
    }
#pragma unroll
  for (int b = 0; b < B; b++) q.q_chop[b] = quotient(kq_chop[b], L, inv_max_path);
  q.q_pair = quotient(kq_pair, L, inv_max_path);
#pragma unroll
  for (int k = 0; k < MAX_DELS; k++)
#pragma unroll
    for (int b = 0; b < B; b++) q.q_del[k][b] = (k < (int)L.ndel()) ? quotient(kq_del[k][b], L, inv_max_path) : 0.0;
  q.nchop = nchop;
  q.npairs = npairs;
  OC_STAMP(4);   // distances consumed, quotients formed
}

// last third: the fp64 sums in the reference's order
template <int B>
__device__ __forceinline__ void shaping_sum(const Hdr &L, const ShapeIn<B> &in, const ShapeQ<B> &q,
                                            double &s0, double &s1 OC_STAMP_PARAM) {
  double tot[B];
#pragma unroll
  for (int b = 0; b < B; b++) {
    // `tot = 0; tot += x` of the reference is x itself (the quotients are never -0.0), so the
    // first term is selected, not added to zero
    tot[b] = q.nchop ? q.q_chop[b] : 0.0;
    tot[b] = q.npairs ? tot[b] + q.q_pair : tot[b];
  }
#pragma unroll
  for (int k = 0; k < MAX_DELS; k++)
    if (k < (int)L.ndel()) {  // uniform; Deliver term in subtask order (:370-395)
      const bool open = !((in.completed >> L.del_bit(k)) & 1);
#pragma unroll
      for (int b = 0; b < B; b++) {
        const double add = !in.del_has[k] ? 2.0 : (q.del_direct[k][b] ? q.q_del[k][b] : q.q_del[k][b] + 1.0);
        tot[b] = open ? tot[b] + add : tot[b];
      }
    }
  s0 = tot[0];
  s1 = B > 1 ? tot[1] : 0.0;
  OC_STAMP(6);   // shaping done
}

// ---------------------------------------------------------------------------
// one environment tick: OvercookedEnvironment.step
// (gym_cooking/envs/overcooked_environment.py:211-241)
// ---------------------------------------------------------------------------
// Everything up to done/reward, plus the address formation and the loads of the reward
// shaping (shaping_issue_*); the caller stores what it has to store and then calls
// shaping_lookup(L, quot, sin, sld, sq) and, after its stores, shaping_sum(L, sin, sq, ...).
// PM: arglist.play known at compile time (0 = off, 1 = on) or a run-time flag (2: R.play)
template <int A, int M, bool DUP, int PM>
__device__ __forceinline__ void env_step(const Hdr &L, const RunCfg &R, const uint8_t *__restrict__ dist,
                                         const uint32_t *__restrict__ probe,
                                         Env<A, M, DUP> &e, const int (&act_in)[A], int &reward, int &done,
                                         int &success, ShapeIn<(A < 2 ? A : 2)> &sin,
                                         ShapeLoads<(A < 2 ? A : 2)> &sld OC_STAMP_PARAM) {
  const int W = L.W(), H = L.H();
  e.t = min(e.t + 1, 0xFFFF);  // :213 (16-bit field: saturates; max_num_timesteps <= 65535 is enforced)
  static_assert(OC_ACT_NOOP == 4 && OC_FLOOR == 0 && OC_COUNTER == 1 && OC_CUTBOARD == 2 && OC_DELIVERY == 3, "codes");

  // ---- check_collisions (:578-613) on the ORIGINAL actions -------------------
  // (per-lane predicates are P words, 0 / -1: see `hide`)
  int act[A], np[A], tgt_p[A];   // action, proposed cell, interact()'s target cell
  P moving[A], t_nonfloor[A], t_deliv[A], t_cutb[A];   // action != (0,0); tile type of the target cell
#pragma unroll
  for (int a = 0; a < A; a++) {
    int c = act_in[a];
    e.err |= p_gtu_any((unsigned)c, 4u) & OC_ERR_ACTION;   // no such NAV action: flagged, executed as (0, 0)
    c = (int)min((unsigned)c, 4u);
    act[a] = c;
    moving[a] = p_ne(c, OC_ACT_NOOP);
    int cell;   // dense index of the target cell
    P oob = 0;
    if constexpr (OC_BORDER_CLOSED) {
      // packed cell += {+16, -16, -1, +1, 0}: one signed byte per action code
      constexpr uint64_t STEP = 0x0001FFF010ull;  // NOOP 00 | RIGHT 01 | LEFT ff | UP f0 | DOWN 10
      const int q = e.ap[a] + (int)(int8_t)(STEP >> (8 * c));
      tgt_p[a] = q;
      cell = dense(L, q);
    } else {
      const int dx = (c == OC_ACT_RIGHT) - (c == OC_ACT_LEFT);
      const int dy = (c == OC_ACT_DOWN) - (c == OC_ACT_UP);
      const int qx = px(e.ap[a]) + dx, qy = py(e.ap[a]) + dy;
      oob = ~(p_ltu_any((unsigned)qx, (unsigned)W) & p_ltu_any((unsigned)qy, (unsigned)H));
      e.err |= oob & OC_ERR_OOB;  // get_gridsquare_at asserts (utils/world.py:310-315)
      const int tx = min(max(qx, 0), W - 1), ty = min(max(qy, 0), H - 1);  // world.inbounds (world.py:317-320)
      tgt_p[a] = tx | (ty << 4);
      cell = ty * W + tx;
    }
    // tile type of the target cell as three predicates, straight from the bit-planes
    P lo, hi;
    if (!L.planes128()) {   // uniform (compile-time in specialised builds)
      lo = p_bit((int)(L.cell_lo(0) >> cell), 0);
      hi = p_bit((int)(L.cell_hi(0) >> cell), 0);
    } else {
      lo = -bit128(L.cell_lo(0), L.cell_lo(1), cell);
      hi = -bit128(L.cell_hi(0), L.cell_hi(1), cell);
    }
    t_nonfloor[a] = lo | hi;
    t_deliv[a] = lo & hi;
    t_cutb[a] = hi & ~lo;
    np[a] = sel(t_nonfloor[a] | oob, e.ap[a], tgt_p[a]);  // :551-559
  }
  P ex[A];
#pragma unroll
  for (int a = 0; a < A; a++) ex[a] = -1;
#pragma unroll
  for (int i = 0; i < A; i++)
#pragma unroll
    for (int j = i + 1; j < A; j++) {
      const P same = p_eq(np[i], np[j]);  // :562-569
      const P i_stays = p_eq(np[i], e.ap[i]) & moving[i];
      const P j_stays = p_eq(np[j], e.ap[j]) & moving[j];
      const P swap = p_eq(e.ap[i], np[j]) & p_eq(e.ap[j], np[i]);  // :572-575
      const P block_i = sel(same, ~i_stays, swap);
      const P block_j = sel(same, i_stays | ~j_stays, swap);
      ex[i] &= ~block_i;
      ex[j] &= ~block_j;
    }

  // ---- execute_navigation (:615-618): interact(), sequential in agent order ---
  // decision phase + one bit-field insert per item (utils/interact.py:4-75)
  // arglist.play (uniform): a merge is put straight onto the counter (:44-47), a fresh food is
  // put DOWN on a Cutboard (:52) and chopped where it lies by an empty-handed press (:66-67)
  const P play = p_of(PM == 2 ? R.play != 0 : PM == 1);
#pragma unroll
  for (int a = 0; a < A; a++) {
    const P acting = ex[a] & moving[a];  // blocked -> (0,0) (:610-612); interact.py:12
    // the agent's own cell has not changed since the proposal phase (only its own interact()
    // moves it), so the target cell and its tile type computed there still hold
    const int pa = e.ap[a];
    const int tp = tgt_p[a];
    const P holding = p_nz(e.ahp[a]);
    const int hold_code = (a + 1) << 12;
    // the held Object (items with holder == a) and the unheld Object on the target cell, as
    // ORs of their item words with bit 8 turned into "a food that is still fresh"
    int held_or = 0, tgt_or = 0;
    P mine[M], tgt[M];
#pragma unroll
    for (int i = 0; i < M; i++) {
      const int w = e.iw[i];
      const int u = item_type(L, i) != OC_PLATE ? (w ^ IW_CHOP) : w;  // uniform choice; a Plate is never chopped
      // (two agents: holder + 1 is 0, 1 or 2, so "held by agent a" is ONE bit of the word)
      mine[i] = A == 2 ? p_bit(w, 12 + a) : p_eq(w & IW_HOLD, hold_code);
      tgt[i] = p_eq(w & (IW_HOLD | IW_POS), tp);                       // unheld and on the target cell
      held_or |= mine[i] & u;
      tgt_or |= tgt[i] & u;
    }
    const P tgt_any = p_nz(tgt_or);                                     // tset of an item is never empty
    P held_multi;                                                       // > 1 content
    if constexpr (DUP) {   // counts: two of one type are two contents
      int nm = 0;
#pragma unroll
      for (int i = 0; i < M; i++) nm -= mine[i];
      held_multi = p_gt(nm, 1);
    } else {
      held_multi = p_nz(held_or & IW_TSET & ((held_or & IW_TSET) - (1 << 24)));
    }
    const P held_fresh = p_bit(held_or, 8);
    const P any_fresh = p_bit(held_or | tgt_or, 8);
    const P two_plates = p_nz((held_or & tgt_or) & sig_of_type<DUP>(OC_PLATE));
    const P at_deliv = t_deliv[a];
    const P nf = acting & t_nonfloor[a];
    const P do_move = acting & ~t_nonfloor[a];                                        // interact.py:19-20
    const P nfh = nf & holding & ~at_deliv, nfe = nf & ~holding & ~at_deliv;
    const P do_deliver = nf & holding & at_deliv & held_multi & ~held_fresh;          // :25-30, core.py:232-237
    const P mergeable = ~(two_plates | any_fresh);                                    // core.py:240-257
    const P do_merge = nfh & tgt_any & mergeable;                                     // :33-46
    const P chop_here = t_cutb[a] & ~held_multi & held_fresh & ~play;                 // :52
    const P do_chop = nfh & ~tgt_any & chop_here;                                     // :52-54
    const P do_drop = nfh & ~tgt_any & ~chop_here;                                    // :56-57
    const P chop_there = nfe & tgt_any & play & t_cutb[a] & p_bit(tgt_or, 8);         // :66-67 (a fresh food is always alone)
    const P do_pick = nfe & tgt_any & ~chop_there & p_of(!((R.allergic >> a) & 1));   // :62-71, agent.py:296-298
    const P put = do_deliver | do_drop | (do_merge & play);
    const P take = (do_merge & ~play) | do_pick;
    const int newg = min(sel(holding, e.ahp[a] - 1, 7), igrp(tgt_or));  // only used when `take` (then tgt_any)
    // the merged Object: smallest item id as group, re-inserted under a new name = last in
    // world order (world.py:236-237), union of the type sets
    int objf;
    if constexpr (DUP) {
      // the merged name = the two multisets added; its key rank: looked up / entered in the
      // per-env name table (see Env::kn)
      const int newsig = (held_or & TS<true>) + (tgt_or & TS<true>);
      const int seq4 = M + e.mctr;
      int kf = 0;
#pragma unroll
      for (int j = 0; j < MAX_NAMES; j++)
        if (j < (int)L.nnames()) {   // uniform
          const P hit = p_eq(newsig >> 24, (int)L.name_sig(j));
          const int sh = 4 * (j & 7);
          const int cur = (e.kn[j >> 3] >> sh) & 15;
          kf = sel(hit, cur, kf);
          const P enter = hit & p_z(cur);   // first Object of this name this episode: the key is created now
          e.kn[j >> 3] |= enter & do_merge & ((seq4 + 1) << sh);
        }
      const int kseq = sel(p_nz(kf), kf - 1, seq4);
      objf = newsig | (newg << 9) | (((kseq << 4) | seq4) << 16);
    } else {
      objf = ((held_or | tgt_or) & IW_TSET) | (newg << 9) | ((M + e.mctr) << 16);
    }
    if (A > 2) {
      // World.remove(agent.holding) deletes by (name, location), last match
      // (world.py:239-247): with another agent on the same cell holding a same-named
      // Object that sits later in world order it removes the wrong one and the
      // reference's store is corrupt from here on.  Flag it.
      // Only reachable when another agent stands on this agent's cell (the 3-agent overlap
      // quirk of check_collisions) while this one merges: one ballot skips the item scan for
      // the whole wave in every other step.
      // (Tried: ONE ballot per step on "two agents may come to share a cell" -- any pair with equal
      // old or proposed cells -- instead of one per agent: true in nearly every wave of 64 three-agent
      // envs, so the scan below ran always: tl-3 x 65 536 3.82 -> 4.16 us.  The ballot per agent costs
      // a scalar read of a vector-written mask, ~16 cycles, and is almost never taken.)
      P shared = 0;
#pragma unroll
      for (int b = 0; b < A; b++)
        if (b != a) shared |= p_eq(e.ap[b], pa);
      if (__ballot((shared & do_merge) != 0) != 0) {
        P alias = 0;
#pragma unroll
        for (int j = 0; j < M; j++) {
          const int w = e.iw[j];
          alias |= p_nz(w & IW_HOLD) & p_ne(w & IW_HOLD, hold_code) & p_eq(ipos(w), pa) &
                   p_z((w ^ held_or) & TS<DUP>) & p_gt(w & IW_SEQ, held_or & IW_SEQ);
        }
        e.err |= shared & do_merge & alias & OC_ERR_ALIAS;
      }
    }
    // held items: cell <- target (move / put down), holder <- none (put down), object
    // fields (merge), chopped (chop); target-cell items: cell <- agent, holder <- agent
    // (merge / pick up), object fields (merge)
    const int mask_m = ((do_move | put) & IW_POS) | (put & IW_HOLD) | (do_merge & OBJ<DUP>) | (do_chop & IW_CHOP);
    const int mask_t = (take & (IW_POS | IW_HOLD)) | (do_merge & OBJ<DUP>) | (chop_there & IW_CHOP);
    const int val_m = tp | IW_CHOP | objf;
    const int val_t = pa | hold_code | objf | IW_CHOP;
#pragma unroll
    for (int i = 0; i < M; i++) {
      // (an item is held by this agent or lies unheld on the target cell, never both)
      const int selm = (mine[i] & mask_m) | (tgt[i] & mask_t);
      e.iw[i] = bfi(selm, sel(mine[i], val_m, val_t), e.iw[i]);
    }
    e.ap[a] = sel(do_move, tp, pa);  // agent.py:311-314
    e.ahp[a] = sel(take, newg + 1, e.ahp[a]) & ~put;
    e.mctr -= do_merge;
  }

  // ---- calculate_reward_shaping (:272-397): the position-only distance lookups go out now ----
  constexpr int B = A < 2 ? A : 2;
#pragma unroll
  for (int b = 0; b < B; b++) sin.ap[b] = e.ap[b];
  if constexpr (DUP) {
    // Chop(X): get_all_object_locs(fresh X)[0] (:287) -- with two fresh X, the set's element [0]
#pragma unroll
    for (int f = 0; f < 3; f++) {
      sin.chop_p[f] = 0;
      if (L.chop_mask(f) != 0) {   // uniform
        bool cand[M];
#pragma unroll
        for (int i = 0; i < M; i++) cand[i] = ((L.food_items(f) >> i) & 1) && !(e.iw[i] & IW_CHOP);
        int has;
        sin.chop_p[f] = pyset_first<M>(L, probe, e.iw, cand, has);
      }
    }
  }
  {
    int ipb[M];
#pragma unroll
    for (int i = 0; i < M; i++) ipb[i] = ipos(e.iw[i]);
    shaping_issue_pos<B, M, DUP>(L, dist, sin, ipb, sld);
    // keep the lookups HERE: left alone, the scheduler sinks them below done/reward, ~40
    // instructions ahead of their first use
    __builtin_amdgcn_sched_barrier(0);
  }
  OC_STAMP(2);
  // ---- done (:243-270) and reward (:399-432) ---------------------------------
  // present / at_delivery: bit s set iff an Object with type-set s and every food chopped
  // exists (anywhere / on the first Delivery tile).  A multi-item Object is all-chopped
  // by construction (mergeable() required it).
  const int d0 = (int)L.deliv_pos(0);  // first Delivery tile only (:259,:402)
  int present = 0, at_delivery = 0;
  P rep_ok[M];   // item i represents its Object (group == i) and the Object is all-chopped
#pragma unroll
  for (int i = 0; i < M; i++) {
    const int w = e.iw[i];
    const P rep = p_eq(w & IW_GRP, i << 9);
    // a lone fresh food is the only Object that is not all-chopped
    const P lone_fresh = item_type(L, i) != OC_PLATE   // uniform
                             ? p_eq(w & (TS<DUP> | IW_CHOP), sig_of_type<DUP>(item_type(L, i))) : 0;
    rep_ok[i] = rep & ~lone_fresh;
    if constexpr (!DUP) {
      const int b = rep_ok[i] & (1 << itset(w));
      present |= b;
      at_delivery |= p_eq(ipos(w), d0) & b;
    }
  }
  int cnt_mask = 0, del_mask = 0, newly;
  if constexpr (DUP) {
    // a goal object may exist several times: its count = the number of DISTINCT cells that hold
    // one (get_all_object_locs is a set of locations, world.py:290-291), two bits per goal
    int rose = 0;
#pragma unroll
    for (int g = 0; g < MAX_GOALS; g++)
      if (g < (int)L.ngoal()) {  // uniform
        int cnt = 0;
        P hasd = 0;
        P m[M];
#pragma unroll
        for (int i = 0; i < M; i++) {
          m[i] = rep_ok[i] & p_eq(e.iw[i] & TS<true>, (int)L.goal_sig(g) << 24);
          P seen = 0;
#pragma unroll
          for (int j = 0; j < i; j++) seen |= m[j] & p_eq(ipos(e.iw[j]), ipos(e.iw[i]));
          cnt -= m[i] & ~seen;
          hasd |= m[i] & p_eq(ipos(e.iw[i]), d0);
        }
        cnt = min(cnt, 3);
        const int old = (e.goalcnt >> (2 * g)) & 3;
        rose |= p_gt(cnt, old) & (int)L.goal_nd(g);
        cnt_mask |= cnt << (2 * g);
        del_mask |= hasd & (int)L.goal_dl(g);
      }
    newly = rose;
  } else {
#pragma unroll
    for (int g = 0; g < MAX_GOALS; g++) {
      if (g < (int)L.ngoal()) {  // uniform
        cnt_mask |= p_bit(present, L.goal_tset(g)) & (int)L.goal_nd(g);
        del_mask |= p_bit(at_delivery, L.goal_tset(g)) & (int)L.goal_dl(g);
      }
    }
    newly = cnt_mask & ~e.goalcnt;  // goal count rose above goal_objects_count (:408-415)
  }
  reward = __popc(newly) + 3 * __popc(del_mask);  // Deliver pays +3 every step (:400-406)
  e.completed |= newly | del_mask;
  e.goalcnt = cnt_mask;
  const P timeout = R.T != 0 ? p_ge(e.t, R.T) : 0;  // checked first (:245-249)
  const P all_delivered = p_eq_any(del_mask, (int)L.deliver_mask());   // (32-bit subtask masks)
  done = (timeout | all_delivered) & 1;
  success = (~timeout & all_delivered) & 1;

  // ---- calculate_reward_shaping for sim agents 0 and 1 (:272-397): the rest of the inputs ----
  sin.completed = e.completed;
#pragma unroll
  for (int k = 0; k < MAX_DELS; k++) {
    sin.del_has[k] = 0;
    sin.del_p[k] = 0;
    if (k < (int)L.ndel()) {  // uniform
      if constexpr (DUP) {   // get_all_object_locs(goal object)[0] (:374-379): the set's element [0]
        bool cand[M];
#pragma unroll
        for (int i = 0; i < M; i++)
          cand[i] = (rep_ok[i] & p_eq(e.iw[i] & TS<true>, (int)L.del_sig(k) << 24)) != 0;
        sin.del_p[k] = pyset_first<M>(L, probe, e.iw, cand, sin.del_has[k]);
      } else {
#pragma unroll
        for (int i = 0; i < M; i++) {
          const P ok = rep_ok[i] & p_eq(e.iw[i] & IW_TSET, (int)L.del_tset(k) << 24);
          sin.del_has[k] |= ok & 1;
          sin.del_p[k] = sel(ok, ipos(e.iw[i]), sin.del_p[k]);
        }
      }
    }
  }
  shaping_issue_del<B>(L, dist, sin, sld);
  __builtin_amdgcn_sched_barrier(0);   // ... and these ahead of the caller's stores
  OC_STAMP(3);   // done/reward computed, distance loads issued
}

// get_observation2 (gym_comm/envs/overcooked_env.py:105-159) for one viewer;
// writes F = 22 + S + 2C rows with stride n.
// OT = element type of the observation rows: 0 int32, 1 int8, 2 float32 (the same integers,
// converted; what a policy network's first layer consumes as obs[v].T without a cast)
template <int A, int M, bool DUP, int OT, typename OutRows>
__device__ __forceinline__ void env_obs(const Hdr &L, const RunCfg &R, const Env<A, M, DUP> &e, int viewer, int radius,
                                        bool viewer_blind, bool ego_blind, int C, int comm0, int comm1,
                                        const OutRows &out, int row0) {
  const int vp = viewer == 0 ? e.ap[0] : e.ap[1];
  const int vhp = viewer == 0 ? e.ahp[0] : e.ahp[1];
  const int vx = px(vp), vy = py(vp);
  int ddx[4], ddy[4], st[4], hid[4];
  int loc[4];   // agent1_location, agent2_location
  if (viewer_blind) {  // uniform: deltas and locations 0, everything hidden (:109,:139-143)
#pragma unroll
    for (int ch = 0; ch < 4; ch++) ddx[ch] = ddy[ch] = st[ch] = loc[ch] = 0, hid[ch] = 1;
  } else {
#pragma unroll
    for (int ch = 0; ch < 4; ch++) {
      // last writer in world.objects order wins (:121-131): the item of this type
      // whose Object has the highest rank
      int bw = 0;
      bool any = false;
#pragma unroll
      for (int i = 0; i < M; i++)
        if (item_type(L, i) == ch) {  // uniform
          bw = (!any || (e.iw[i] & IW_SEQ) > (bw & IW_SEQ)) ? e.iw[i] : bw;
          any = true;
        }
      // an absent type keeps delta (0,0)
      const int bx = any ? px(ipos(bw)) : vx, by = any ? py(ipos(bw)) : vy;
      const bool within = (int)sad_u32(bx, vx, sad_u32(by, vy, 0u)) <= radius;   // |dx| + |dy|
      hid[ch] = within ? 0 : 1;                       // :133
      ddx[ch] = within ? 0 : bx - vx;                 // :135 (sic: zeroed when visible)
      ddy[ch] = within ? 0 : by - vy;
      st[ch] = (any && ch != OC_PLATE) ? ichop(bw) : 0;
    }
    loc[0] = px(e.ap[0]), loc[1] = py(e.ap[0]), loc[2] = px(e.ap[1]), loc[3] = py(e.ap[1]);
  }
#define OUT(r_, v_)                                                              \
  do {                                                                           \
    if (OT == 1) out.st8((r_), (v_));                                            \
    else if (OT == 2) out.st((r_), __builtin_bit_cast(int, (float)(v_)));        \
    else out.st((r_), (v_));                                                     \
  } while (0)
  int row = row0;
#pragma unroll
  for (int ch = 0; ch < 4; ch++) OUT(row++, ddx[ch]);
#pragma unroll
  for (int ch = 0; ch < 4; ch++) OUT(row++, ddy[ch]);
#pragma unroll
  for (int ch = 0; ch < 4; ch++) OUT(row++, st[ch]);
#pragma unroll
  for (int ch = 0; ch < 4; ch++) OUT(row++, hid[ch]);
  if (R.slot_identity) {   // uniform: the caller's order is the canonical one
    for (int s = 0; s < L.S(); s++) OUT(row++, (e.completed >> s) & 1);
  } else {
    for (int s = 0; s < L.S(); s++) OUT(row++, (e.completed >> ((R.slot4[s >> 2] >> (8 * (s & 3))) & 31)) & 1);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) OUT(row++, loc[k]);
  OUT(row++, ego_blind ? 0 : (vhp != 0 ? 1 : 0));  // :154, gated on the EGO's BLIND flag
  OUT(row++, 0);
  if (C == 2) {  // uniform; the BASELINE configuration: straight-line instead of four scalar loops
    OUT(row++, comm0 == 0 ? 1 : 0);
    OUT(row++, comm0 == 1 ? 1 : 0);
    OUT(row++, comm1 == 0 ? 1 : 0);
    OUT(row++, comm1 == 1 ? 1 : 0);
  } else {
    for (int c = 0; c < C; c++) OUT(row++, comm0 == c ? 1 : 0);
    for (int c = 0; c < C; c++) OUT(row++, comm1 == c ? 1 : 0);
  }
#undef OUT
}

// Sum of a per-lane integer over the 64 lanes of the wave, left in lane 63: an inclusive scan
// inside each row of 16 lanes (row_shr 1/2/4/8, zero fill), then row 0 -> 1, 2 -> 3
// (row_bcast:15) and rows 0-1 -> 2-3 (row_bcast:31).  Six DPP adds, no LDS, no SALU loop.
__device__ __forceinline__ int wave_sum_lane63(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2, 3
  return v;
}

// Per-wave metric accumulation: the six counters of a step are packed into two words per
// lane (every field wide enough for a sum over 64 lanes), both words are summed across the
// wave with DPP adds, the totals are read from lane 63 into SGPRs, split by scalar bit-field
// extracts and written back into lanes 0..5, and lane k adds counter k to the wave's OWN
// 64-byte slot of the metrics tensor (one no-return atomic per lane, no two waves share a
// slot).  Must be called with all 64 lanes active.
struct MetricsSlot {
  unsigned long long *p;
  // thread_index: global thread id; one slot per 64 envs
  __device__ __forceinline__ MetricsSlot(int64_t *metrics, int64_t thread_index) {
    const int lane = threadIdx.x & 63;
    p = (metrics != nullptr && lane < 6)
            ? (unsigned long long *)metrics + (thread_index >> 6) * OC_MET_COUNT + lane
            : nullptr;
  }
  __device__ __forceinline__ void add(bool has_metrics, bool valid, int done, int success, int reward,
                                      int completed_bits, bool err) {
    if (!has_metrics) return;  // uniform
    // (P words, see `hide`: valid / done / success / err combine on the vector unit)
    const P ok = hide(valid ? -1 : 0), fin = ok & -done;
    // word A: reward (<= 32 + 3 * MAX_DELS < 64 -> 12-bit sum) | completed subtasks of a finished
    // episode (<= OC_MAX_SUBTASKS = 32 per lane, 64 lanes -> a 12-bit sum)
    // word B: valid | done | success | error, 7 bits each (a count up to 64)
    const int a = (ok & reward) | ((fin & __popc(completed_bits)) << 12);
    const int b = (ok & 1) | (fin & (1 << 7)) | (ok & -success & (1 << 14)) | (ok & hide(err ? -1 : 0) & (1 << 21));
    const unsigned ta = (unsigned)__builtin_amdgcn_readlane(wave_sum_lane63(a), 63);
    const unsigned tb = (unsigned)__builtin_amdgcn_readlane(wave_sum_lane63(b), 63);
    // lane k picks counter k out of the two totals: a per-lane (word, offset, width) from
    // three packed constants -- plain VALU selects, no divergent control flow
    constexpr unsigned OFF = 0u | 7u << 5 | 14u << 10 | 0u << 15 | 12u << 20 | 21u << 25;    // 5 bits per lane
    constexpr unsigned WID = 7u | 7u << 5 | 7u << 10 | 12u << 15 | 12u << 20 | 7u << 25;
    static_assert(64 * OC_MAX_SUBTASKS < (1 << 12) && 64 * (OC_MAX_SUBTASKS + 3 * MAX_DELS) < (1 << 12),
                  "wave sums must fit their 12-bit fields");
    static_assert(OC_MET_ENV_STEPS == 0 && OC_MET_EPISODES == 1 && OC_MET_SUCCESSES == 2 &&
                  OC_MET_REWARD_SUM == 3 && OC_MET_COMPLETED_SUM == 4 && OC_MET_ERRORS == 5, "slot order");
    const unsigned lane = threadIdx.x & 63, k5 = (lane < 6 ? lane : 0) * 5;
    const unsigned off = (OFF >> k5) & 31, wid = (WID >> k5) & 31;
    const unsigned src = (lane == OC_MET_REWARD_SUM || lane == OC_MET_COMPLETED_SUM) ? ta : tb;
    const int v = (int)((src >> off) & ((1u << wid) - 1u));
    // Fire-and-forget atomic add by lanes 0..5, each to its own word of the wave's OWN slot (no
    // contention).  A load at kernel start + a plain store here had to be consumed behind the
    // 60+ stores of the step: vmcnt retires in issue order, so the wave ended on an
    // s_waitcnt vmcnt(0) -- a wait for every store it had issued.
    if (p) __hip_atomic_fetch_add(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
// The two lane-indexed tables (fp64 quotients, then the u8 distance table) sit in one
// device buffer.  Two variants of every step kernel exist (template bool LDS):
//   LDS = false  the tables are read straight from global memory (vector L1/L2): no
//                staging pass, no barrier.  The default: fastest at every batch size with
//                64- or 128-thread workgroups (round-1 v3 sweep: n = 4096: 4.7 us vs
//                5.7 us per step; n = 131072: 11.1 us vs 12.7 us).
//   LDS = true   every workgroup copies them to LDS with 16-byte loads issued BEFORE the
//                state loads, so both round trips overlap.  Only ahead with 256-thread
//                workgroups (n = 131072: 12.1 us vs 13.3 us), which lose overall.
// Full sweep: profiles/r01_v3_block_lds_sweep.txt.
struct Tables {
  const uint8_t *dist;
  const uint8_t *counters;  // Counter tiles (x | y<<4), world order, 64 bytes
  const uint32_t *probe;    // dup mode: per cell, its first eight set-table probe slots (pyset_first)
};

template <bool LDS>
__device__ __forceinline__ Tables stage_tables(const void *__restrict__ tables, int n16, int quot_bytes) {
  Tables tb;
  if constexpr (LDS) {
    extern __shared__ uint4 oc_lds[];
    const uint4 *src = (const uint4 *)tables;
    // four named registers, not an array: a conditionally written array went to scratch
    const int i0 = threadIdx.x, i1 = i0 + blockDim.x, i2 = i1 + blockDim.x, i3 = i2 + blockDim.x;
    uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0, t2 = t0, t3 = t0;
    if (i0 < n16) t0 = src[i0];
    if (i1 < n16) t1 = src[i1];
    if (i2 < n16) t2 = src[i2];
    if (i3 < n16) t3 = src[i3];
    for (int idx = threadIdx.x + 4 * blockDim.x; idx < n16; idx += blockDim.x) oc_lds[idx] = src[idx];
    if (i0 < n16) oc_lds[i0] = t0;
    if (i1 < n16) oc_lds[i1] = t1;
    if (i2 < n16) oc_lds[i2] = t2;
    if (i3 < n16) oc_lds[i3] = t3;
    __syncthreads();
    tb.dist = (const uint8_t *)oc_lds + quot_bytes;
    tb.counters = (const uint8_t *)oc_lds + (n16 * 16 - OC_MAX_COUNTERS);
    tb.probe = (const uint32_t *)((const uint8_t *)oc_lds + (n16 * 16 - OC_MAX_COUNTERS - 4 * OC_MAX_CELLS));
  } else {
    tb.dist = (const uint8_t *)tables;   // quot_bytes is 0 since the quotient table went (v12)
    tb.counters = (const uint8_t *)tables + (n16 * 16 - OC_MAX_COUNTERS);
    tb.probe = (const uint32_t *)((const uint8_t *)tables + (n16 * 16 - OC_MAX_COUNTERS - 4 * OC_MAX_CELLS));
  }
  return tb;
}

// Start cells of the items for a fresh episode of a random-* level
// (overcooked_environment.py:157-173: for every scattered letter, random.choice over ALL
// Counter tiles until one not yet taken by this phase comes up).  Either read from the
// caller's `placement` tensor ([M][n], x | y<<4; parity mode: the reference's own draws)
// or drawn here from the env's own PCG32 stream (`rng`, uint32 [n]; production mode --
// same distribution, not CPython's Mersenne Twister sequence).
__device__ __forceinline__ uint32_t pcg32(uint32_t &state) {
  state = state * 747796405u + 2891336453u;
  const uint32_t w = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (w >> 22u) ^ w;
}

template <int A, int M, int WS>
__device__ __forceinline__ void place_items_from(const Hdr &L, const Tables &tb, const int32_t *placement,
                                                 bool use_rng, uint32_t &st, int64_t n, int64_t i,
                                                 int32_t (&w)[WS]) {
  if (L.nscatter() == 0) return;  // uniform (compile-time in specialised builds)
  int pos[M];
#pragma unroll
  for (int k = 0; k < M; k++) pos[k] = w[A + k] & 255;
  if (use_rng) {
    unsigned long long taken = 0;
    for (int k = 0; k < (int)L.nscatter(); k++) {
      int idx = 0;
      bool ok = false;
      for (int attempt = 0; attempt < 64 && !ok; attempt++) {
        idx = (int)__umulhi(pcg32(st), L.ncounters());
        ok = !((taken >> idx) & 1);
      }
      for (int c = 0; c < (int)L.ncounters() && !ok; c++) {  // practically unreachable
        idx = c;
        ok = !((taken >> idx) & 1);
      }
      taken |= 1ull << idx;
      const int cell = tb.counters[idx];
      const int item = (int)L.scatter_item(k & 3);
#pragma unroll
      for (int m = 0; m < M; m++) pos[m] = (item == m) ? cell : pos[m];
    }
  } else if (placement != nullptr) {
#pragma unroll
    for (int k = 0; k < M; k++) pos[k] = placement[(int64_t)k * n + i] & 255;
  }
#pragma unroll
  for (int k = 0; k < M; k++) w[A + k] = (w[A + k] & ~255) | pos[k];
}

// read-modify-write form: the env's PCG32 state lives in rng[i]
template <int A, int M, int WS>
__device__ __forceinline__ void place_items(const Hdr &L, const Tables &tb, const int32_t *placement,
                                            uint32_t *rng, int64_t n, int64_t i, int32_t (&w)[WS]) {
  if (L.nscatter() == 0) return;
  const bool use_rng = rng != nullptr;
  uint32_t st = use_rng ? rng[i] : 0u;
  place_items_from<A, M, WS>(L, tb, placement, use_rng, st, n, i, w);
  if (use_rng) rng[i] = st;
}

struct StepArgs {
  LevelHdr L;
  RunCfg R;
  const void *tables;
  int32_t n16, quot_bytes;
  int32_t *state;
  const int32_t *actions;
  int32_t *reward;
  int32_t *done;
  double *shaping;
  int64_t *metrics;
  const int32_t *placement;
  uint32_t *rng;
  int64_t n;
  int32_t auto_reset;
  unsigned long long *timeline;   // OC_TIMELINE builds: this launch's record (else NULL, never read)
  int64_t timeline_stride;
};

// (leading scalars: preloaded kernel arguments, see k_multi_step)
// DUTY (see multi_step_body below): DUTY_STATE = reward, done, the state rows, metrics;
// DUTY_SHAPE = the reward shaping of sim agents 0 and 1.  Both = the whole step in one wave.
template <int A, int M, bool LDS, bool WT, bool DUP, bool PLAY, int DUTY, bool SPLIT>
__device__ __forceinline__ void step_body(int32_t *const state_, const int32_t *const actions_,
                                          int64_t *const metrics_, const int64_t n_, const int32_t launch_,
                                          const int32_t T_, const void *const tables_,
                                          const double inv_max_path_, const StepArgs &p) {
  constexpr bool D_STATE = (DUTY & 1) != 0, D_SHAPE = (DUTY & 2) != 0;
  using Out = RowsT<WT ? AUX_WT : 0>;
  OC_HDR_LOAD(p);
  const int block_ = launch_ & 0xFFFF;
  const bool auto_reset_ = (launch_ >> 16) & 1;
  // n < 2^31 / (4 * rows): fits_buffer().  Split: one workgroup = two waves over the same 64 envs.
  const int i = SPLIT ? (int)blockIdx.x * 64 + (int)(threadIdx.x & 63) : (int)blockIdx.x * block_ + (int)threadIdx.x;
  const bool valid = i < (int)n_;
  Tables tb;   // (global variant: formed after the state loads are issued, see k_multi_step)
  if constexpr (LDS) tb = stage_tables<true>(p.tables, p.n16, p.quot_bytes);
  MetricsSlot slot(metrics_, i);
  int reward = 0, done = 0, success = 0, comp = 0;
  bool err = false;
  if (valid) {
    constexpr int WS = state_words<A, M, DUP>();
    const Out st(state_, n_, WS, i);
    const Rows ac(actions_, n_, A, i);
    int32_t w[WS];
#pragma unroll
    for (int r = 0; r < WS; r++) w[r] = st.ld(r);
    int act[A];
#pragma unroll
    for (int a = 0; a < A; a++) act[a] = ac.ld(a);
    if constexpr (!LDS) tb = stage_tables<false>(tables_, p.n16, p.quot_bytes);
    Env<A, M, DUP> e;
    unpack<A, M, DUP>(e, w);
    // split: no state row is stored before both waves hold their copy of the state
    if constexpr (SPLIT) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    const int err_before = e.err;
    constexpr int B = A < 2 ? A : 2;
    ShapeIn<B> sin;
    ShapeLoads<B> sld;
#ifdef OC_STAMPS
    unsigned long long oc_tt[16];
#endif
    RunCfg R = p.R;
    R.T = T_;   // the preloaded copy
    env_step<A, M, DUP, PLAY ? 1 : 0>(L, R, tb.dist, tb.probe, e, act, reward, done, success, sin, sld OC_STAMP_PASS);
    comp = e.completed;
    err = e.err != err_before;
    ShapeQ<B> sq;
    if constexpr (D_SHAPE) shaping_lookup<B>(L, inv_max_path_, sin, sld, sq OC_STAMP_PASS);
    if constexpr (D_STATE) {
      Out(p.reward, p.n, 1, i).st(0, reward);
      Out(p.done, p.n, 1, i).st(0, done);
      if (L.nscatter() == 0) {   // uniform: a fixed level -- the fresh episode is a constant, selected word by word
        pack<A, M, DUP>(e, w);
        const P fresh = -done & p_of(auto_reset_);
#pragma unroll
        for (int r = 0; r < WS; r++) w[r] = sel(fresh, L.init_words(r), w[r]);
      } else if (done && auto_reset_) {
#pragma unroll
        for (int r = 0; r < WS; r++) w[r] = L.init_words(r);
        place_items<A, M, WS>(L, tb, p.placement, p.rng, p.n, i, w);
      } else {
        pack<A, M, DUP>(e, w);
      }
#pragma unroll
      for (int r = 0; r < WS; r++) st.st(r, w[r]);
    }
    if constexpr (D_SHAPE) {
      double s0, s1;
      shaping_sum<B>(L, sin, sq, s0, s1 OC_STAMP_PASS);
      const Out sh(p.shaping, p.n, 2, i, 8);
      sh.st_f64(0, s0);
      sh.st_f64(1, s1);
    }
  }
  if constexpr (D_STATE) slot.add(metrics_ != nullptr, valid, done, success, reward, comp, err);
}

// OvercookedEnvironment.step for n envs.  SP = waves per 64 envs: 1, or 2 = split launch (128
// threads per workgroup: one wave steps and stores the state, the other computes the reward
// shaping -- the same idea as k_multi_step's four-way split, see multi_step_body).
template <int A, int M, bool LDS, bool WT, bool DUP, bool PLAY, int SP>
__global__ void __launch_bounds__(256) k_step(int32_t *const state_, const int32_t *const actions_,
                                              int64_t *const metrics_, const int64_t n_, const int32_t launch_,
                                              const int32_t T_, const void *const tables_,
                                              const double inv_max_path_, const StepArgs p) {
  // (tables_ / inv_max_path_ / T_ and, packed into launch_ = block | auto_reset << 16, the
  // auto-reset flag too: this kernel runs at the SGPR limit with 3-4 agents, and the compiler
  // otherwise loads each of them right before its first use and waits on the spot)
  static_assert(SP == 1 || (SP == 2 && !LDS), "waves per 64 envs");
  OC_TL_BEGIN();
  if constexpr (SP == 1) {
    step_body<A, M, LDS, WT, DUP, PLAY, 3, false>(state_, actions_, metrics_, n_, launch_, T_, tables_, inv_max_path_, p);
  } else {
    // (only launched by the specialised libraries, see oc_step)
    if (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) == 0)
      step_body<A, M, LDS, WT, DUP, PLAY, 1, true>(state_, actions_, metrics_, n_, launch_, T_, tables_, inv_max_path_, p);
    else
      step_body<A, M, LDS, WT, DUP, PLAY, 2, true>(state_, actions_, metrics_, n_, launch_, T_, tables_, inv_max_path_, p);
  }
  OC_TL_END(p.timeline, p.timeline_stride);
}

struct ObsArgs {
  LevelHdr L;
  RunCfg R;
  const int32_t *state;
  const int32_t *comm;
  void *obs;          // int32, int8 or float32 rows (cfg.obs_int8)
  double *timestep;
  int64_t n;
  oc_obs_cfg cfg;
};

template <int A, int M, int OT, bool WT, bool DUP>
__global__ void __launch_bounds__(256) k_obs(const ObsArgs p) {
  using Out = RowsT<WT ? AUX_WT : 0>;
  OC_HDR_LOAD(p);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  constexpr int WS = state_words<A, M, DUP>();
  const Rows st(p.state, p.n, WS, i);
  int32_t w[WS];
#pragma unroll
  for (int r = 0; r < WS; r++) w[r] = st.ld(r);
  Env<A, M, DUP> e;
  unpack<A, M, DUP>(e, w);
  const int C = p.cfg.num_comm;
  const int F = 22 + L.S() + 2 * C;
  const int c0 = p.comm[i], c1 = p.comm[p.n + i];
  const bool ego_blind = p.cfg.blind_mask & 1;
  const Out ob(p.obs, p.n, 2 * F, i, OT == 1 ? 1 : 4);
#pragma unroll
  for (int v = 0; v < 2; v++)
    env_obs<A, M, DUP, OT>(L, p.R, e, v, p.cfg.fow_radius, (p.cfg.blind_mask >> v) & 1, ego_blind, C, c0, c1, ob, v * F);
  Out(p.timestep, p.n, 1, i, 8).st_f64(0, timestep_of(e.t, p.R));  // overcooked_env.py:146
}

struct ImageArgs {
  LevelHdr L;
  const int32_t *state;
  int32_t *out;       // [2][7][ceil(W*H / 4)][n]: four consecutive cells of a plane per dword
  int8_t *holding;    // [2][n]
  int64_t n;
  int32_t radius;
};

// OvercookedMultiEnv.get_partial_observability_FOW for both viewers
// (gym_comm/envs/overcooked_env.py:161-202; the image-style observation the reference
// defines but does not call).  Plane k of viewer v at cell (x, y) -- the reference's
// map[k][x][y] -- is byte (x*H + y) of the plane: plane 0 the tile type, planes 1.. "agent i
// stands here" (:183-185 -- with 3+ agents these overwrite the content planes, as in the
// reference), planes 3 + channel the contents (Food: state_index + 1, Plate: 1); cells farther
// than `radius` (manhattan) from the viewer are -1 in every plane.  A lane packs FOUR consecutive
// cells of a plane of its env into one dword (little-endian, zero padded past the last cell), so
// a wave stores 256 contiguous bytes per instruction.
// Work per env is organised by QUAD, not by byte: the fog bytes of a quad are formed once per
// viewer and reused by its seven planes; an agent / item contributes to the one quad its cell
// falls into (a compare and a select per quad); a plane's dword is then one bit-field insert
// (fog bytes win) and one store.  ~1 100 VALU + 182 stores per wave at 7x7 -- the first version
// walked the 343 bytes of a viewer one by one, ~120 instructions each (56 us per launch at 4 096
// envs; this one: see DESIGN.md).
template <int A, int M, bool DUP>
__global__ void __launch_bounds__(256) k_obs_image(const ImageArgs p) {
  OC_HDR_LOAD(p);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  constexpr int WS = state_words<A, M, DUP>();
  const Rows st(p.state, p.n, WS, i);
  int32_t w[WS];
#pragma unroll
  for (int r = 0; r < WS; r++) w[r] = st.ld(r);
  Env<A, M, DUP> e;
  unpack<A, M, DUP>(e, w);
  const int W = L.W(), H = L.H();
  const int ncell = W * H, Q = (ncell + 3) >> 2;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)(2 * 7 * Q * p.n * 4), 0x00020000);
  // per agent / item: the quad its cell (x-major index x*H + y) falls into and its byte there
  int aq[A], abyte[A];
#pragma unroll
  for (int a = 0; a < A; a++) {
    const int c = (int)__umul24((unsigned)px(e.ap[a]), (unsigned)H) + py(e.ap[a]);
    aq[a] = c >> 2;
    abyte[a] = 8 * (c & 3);
  }
  int iq[M], ishift[M], ival[M];
#pragma unroll
  for (int m = 0; m < M; m++) {
    const int pos = ipos(e.iw[m]);
    const int c = (int)__umul24((unsigned)px(pos), (unsigned)H) + py(pos);
    iq[m] = c >> 2;
    ishift[m] = 8 * (c & 3);
    ival[m] = item_type(L, m) == OC_PLATE ? 1 : ichop(e.iw[m]) + 1;
    // the last writer in world order wins (:171-178): an item gives way to a later one of its
    // type on the same cell (only in levels that repeat a type)
    bool later = false;
#pragma unroll
    for (int o = 0; o < M; o++)
      if (o != m && item_type(L, o) == item_type(L, m))   // uniform
        later |= ipos(e.iw[o]) == pos && (e.iw[o] & IW_SEQ) > (e.iw[m] & IW_SEQ);
    iq[m] = later ? -1 : iq[m];
  }
  const int vx[2] = {px(e.ap[0]), px(e.ap[1])}, vy[2] = {py(e.ap[0]), py(e.ap[1])};
  int x = 0, y = 0;   // cell of the quad's first byte, advanced incrementally (uniform)
  for (int q = 0; q < Q; q++) {
    unsigned fog[2] = {0u, 0u}, tile = 0u;
#pragma unroll
    for (int b = 0; b < 4; b++)
      if (4 * q + b < ncell) {   // uniform
        tile |= (unsigned)cell_type(L, y * W + x) << (8 * b);   // uniform: SALU
#pragma unroll
        for (int v = 0; v < 2; v++)
          fog[v] |= (int)sad_u32((unsigned)x, (unsigned)vx[v], sad_u32((unsigned)y, (unsigned)vy[v], 0u)) > p.radius
                        ? 0xFFu << (8 * b) : 0u;
        if (++y == H) y = 0, x++;
      }
#pragma unroll
    for (int k = 0; k < 7; k++) {
      unsigned val = k == 0 ? tile : 0u;
      if (k >= 3) {
#pragma unroll
        for (int m = 0; m < M; m++)
          if (item_type(L, m) + 3 == k)   // uniform
            val |= iq[m] == q ? (unsigned)ival[m] << ishift[m] : 0u;
      }
      if (k >= 1 && k <= A) {   // agent k-1 stands here: 1, over whatever the plane held at that cell
        const int a = k - 1;
        val = aq[a] == q ? (val & ~(0xFFu << abyte[a])) | (1u << abyte[a]) : val;
      }
#pragma unroll
      for (int v = 0; v < 2; v++)
        __builtin_amdgcn_raw_buffer_store_b32((int)(val | fog[v]), rsrc, (int)i * 4,
                                              (int)(((v * 7 + k) * Q + q) * p.n * 4), AUX_WT);
    }
  }
  p.holding[i] = e.ahp[0] != 0;
  p.holding[p.n + i] = e.ahp[1] != 0;
}

struct ResetArgs {
  LevelHdr L;
  const void *tables;
  int32_t n16, quot_bytes;
  int32_t *state;
  const int32_t *mask;
  const int32_t *placement;
  uint32_t *rng;
  int64_t n;
};

// OvercookedEnvironment.reset() (overcooked_environment.py:180-206), masked
template <int A, int M, bool DUP>
__global__ void __launch_bounds__(256) k_reset(const ResetArgs p) {
  OC_HDR_LOAD(p);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  if (p.mask != nullptr && p.mask[i] == 0) return;
  constexpr int WS = state_words<A, M, DUP>();
  const Tables tb = stage_tables<false>(p.tables, p.n16, p.quot_bytes);
  int32_t w[WS];
#pragma unroll
  for (int r = 0; r < WS; r++) w[r] = L.init_words(r);
  place_items<A, M, WS>(L, tb, p.placement, p.rng, p.n, i, w);
#pragma unroll
  for (int r = 0; r < WS; r++) p.state[(int64_t)r * p.n + i] = w[r];
}

// Uniform random (move, comm) indices of one player (include/oc_hip.h: oc_random_actions)
__global__ void __launch_bounds__(256) k_random_actions(uint32_t *rng, int32_t *move_row, int32_t *comm_row,
                                                        uint32_t num_comm, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t st = rng[i];
  move_row[i] = (int32_t)__umulhi(pcg32(st), 4u);
  comm_row[i] = (int32_t)__umulhi(pcg32(st), num_comm);
  rng[i] = st;
}

struct MultiArgs {
  LevelHdr L;
  RunCfg R;
  const void *tables;
  int32_t n16, quot_bytes;
  int32_t *state;
  int32_t *comm;
  const int32_t *actions;
  void *obs;          // int32, int8 or float32 rows (cfg.obs.obs_int8)
  double *timestep;
  double *reward;
  int32_t *done;
  int32_t *sparse;
  int64_t *metrics;
  const int32_t *placement;
  uint32_t *rng;
  oc_step_opts opt;      // optional inputs / outputs (include/oc_hip.h), all NULL = off
  int64_t n;
  int32_t auto_reset;
  oc_wrap_cfg cfg;
  oc_step_policy pol[2]; // opt.policy by value (it is a host pointer), used by the POL variants
  int32_t pol_ksteps;
  unsigned long long *timeline;   // OC_TIMELINE builds: this launch's record (else NULL, never read)
  int64_t timeline_stride;
};

// OvercookedMultiEnv.multi_step (gym_comm/envs/overcooked_env.py:207-282), 2 agents.
// XO = 0: the plain step in the wrapper's standard configuration -- actions from the four
// rows, no episode statistics (`p.opt` is not even looked at), communication on, not ego-led,
// both players CAN_MOVE, ego = sim agent 0, nobody BLIND, arglist.play off (the reference's
// env_args*.json and BASELINE.md section 3) -- with those settings folded: no selects on them, no
// BLIND branch and none of the register copies its join costs.
// XO = 1: the same standard configuration, still folded, plus oc_step_opts' action sources
// ([n][2] pairs, the in-kernel partner), episode statistics and the fused policies: what
// OvercookedVecEnv launches with the reference's own run configurations (round 3: until then
// every option landed in ONE general variant and step_tensors paid for run-time `play` and
// wrapper-configuration selects it never uses -- specialised libraries only).
// XO = 2: the general variant -- any wrapper configuration, arglist.play at run time, and the options.
//
// DUTY: which of the step's OUTPUTS this wave produces.  Everything up to done/reward is needed by
// every output and is computed by every wave; what follows splits four ways:
//   DUTY_STATE  comm rows, done, sparse reward, the state rows, the metrics counters (and the
//               in-place words of the optional random streams)
//   DUTY_SHAPE  the reward shaping (distance lookups, fp64 sums), the shaped reward, episode statistics
//   DUTY_OBS0 / DUTY_OBS1  get_observation2 for viewer 0 (+ the timestep) / viewer 1
// DUTY_ALL is the whole step in one wave.  A *split* launch (k_multi_step<..., SP = 4>: four waves
// per 64 envs in one workgroup, i.e. one wave on each SIMD of a CU) gives each wave one duty.  Why:
// at the BASELINE batch sizes a step puts ONE wave on 64 ... 256 of the chip's 1 024 SIMDs, a lone
// wave issues an instruction every ~7 cycles whatever the other SIMDs do, and the step lasts as
// long as that wave's instruction stream (~950 instructions).  The stream is therefore cut where
// the data flow forks, and the three idle SIMDs next door run the branches side by side: ~620
// instructions per wave instead of ~950 (the part before the fork is recomputed by every wave --
// free while the batch leaves SIMDs idle; each arm is compiled on its own, so the loads and the
// arithmetic only another duty needs are gone from it).  The state is updated in place, so a split
// workgroup passes one s_barrier between "every wave holds its copy of the state" and the first
// store of anything a later-starting wave might still have to read.
// Measured (tools/split_sweep.sh, MI355X): tomato-2, 4 096 envs 3.55 -> 3.07 us per step; equal at
// 32 768 envs (every SIMD has a wave of its own by then), slower beyond: split_for().
constexpr int DUTY_STATE = 1, DUTY_SHAPE = 2, DUTY_OBS0 = 4, DUTY_OBS1 = 8, DUTY_ALL = 15;

// Split launch with the policies fused: both viewers' observation rows of the workgroup's 64 envs
// as floats, [viewer][row][env], and the timestep -- written by the observation waves, read by all
// four waves' policy passes behind the second barrier.  (Function templates of their own: ONE LDS
// array for the four arms of the kernel, which are four instantiations of multi_step_body.)
template <int ROWS>
__device__ __forceinline__ float *pol_lds_feat() {
  __shared__ float a[2 * ROWS * 64];
  return a;
}
__device__ __forceinline__ float *pol_lds_ts() {
  __shared__ float t[64];
  return t;
}

// POL (general variant only): the closed loop in one launch -- behind the step, the wave(s) evaluate
// both players' MLP policies (oc_policy_device.h) on the observation rows just written and put the
// NEXT step's (move, comm) pairs where this step read its own (oc_step_opts.policy).  One pass =
// one wave x 32 envs; a split workgroup gives each of its four waves one (viewer, half) pass
// behind a second barrier ("every observation row of these 64 envs is written"), a lone wave
// runs all four.
template <int M, bool LDS, int OT, bool WT, bool DUP, int XO, int DUTY, bool SPLIT, bool POL = false>
__device__ __forceinline__ void multi_step_body(int32_t *const state_, const int32_t *const actions_,
                                                int32_t *const comm_, int64_t *const metrics_,
                                                const int32_t n_, const int32_t block_, const void *const ego_src_,
                                                const void *const alt_src_, const MultiArgs &p) {
  constexpr int A = 2;
  constexpr bool D_STATE = (DUTY & DUTY_STATE) != 0, D_SHAPE = (DUTY & DUTY_SHAPE) != 0;
  using Out = RowsT<WT ? AUX_WT : 0>;
#ifdef OC_SPECIALIZED
  constexpr int POL_ROWS = (POL && SPLIT) ? 22 + OC_SPEC_HDR.S + 8 : 1;   // F at most: 4 comm channels
#else
  constexpr int POL_ROWS = 1;
#endif
  OC_HDR_LOAD(p);
  // n < 2^31 / (4 * rows): fits_buffer().  Split: one workgroup = SP waves over the same 64 envs.
  const int i = SPLIT ? (int)blockIdx.x * 64 + (int)(threadIdx.x & 63)
                      : (int)blockIdx.x * (block_ & 0xFFFF) + (int)threadIdx.x;
  const bool valid = i < (int)n_;
  const bool ego_from_pairs = XO != 0 && ((block_ >> 16) & 1), alt_from_pairs = XO != 0 && ((block_ >> 17) & 1),
             alt_from_rng = XO != 0 && ((block_ >> 18) & 1), pairs64 = XO != 0 && ((block_ >> 19) & 1);
#ifdef OC_STAMPS
  unsigned long long oc_tt[16];
  for (int k = 0; k < 16; k++) oc_tt[k] = 0;
#endif
  OC_STAMP(0);
  // LDS variant: staged first (its loads overlap the state loads).  Global variant: the table
  // base pointers are formed AFTER the state loads are issued -- formed first, their scalar
  // kernarg load was waited for before a single vector load had left.
  Tables tb;
  if constexpr (LDS) tb = stage_tables<true>(p.tables, p.n16, p.quot_bytes);
  MetricsSlot slot(metrics_, i);
  int reward = 0, done = 0, success = 0, comp = 0;
  bool err = false;
  // (POL, split: which (viewer, half) pass this wave runs behind the step -- viewer 0 / first half
  // on the OBS0 wave, viewer 1 / first half on OBS1, the second halves on the STATE and SHAPE waves)
  constexpr int POL_MINE = (DUTY == DUTY_OBS0) ? 0 : (DUTY == DUTY_OBS1) ? 2 : (DUTY == DUTY_STATE) ? 1 : 3;
  constexpr int POL_VIEWER = POL_MINE >> 1;
  [[maybe_unused]] ocpol::Weights pol_w;
  if constexpr (POL && SPLIT) {
    // This wave's pass is for viewer POL_VIEWER: its share of that player's weights is fetched
    // now, under the wait for the state -- by EVERY lane: a lane's fragments are rows of the weight
    // matrices, needed whether or not the lane has an env of its own.
    ocpol::load_weights(pol_w, p.pol[POL_VIEWER].w1, p.pol[POL_VIEWER].w2, p.pol[POL_VIEWER].b2,
                        (int)threadIdx.x & 63, p.pol_ksteps);
  }
  if (valid) {
    // (Tried: the state loads ahead of this tail-lane test -- v_cmp -> s_and_saveexec costs ~16 cycles
    // in front of the first load.  Nothing at 4 096 envs, and salad-2 x 32 768, two waves per SIMD,
    // went from 3.96 to 4.34 us: the shaping wave of a workgroup ended 0.2 us later.  Left as it was.)
    constexpr int WS = state_words<A, M, DUP>();
    const Out st(state_, n_, WS, i), cm(comm_, n_, 2, i);
    int32_t w[WS];
#pragma unroll
    for (int r = 0; r < WS; r++) w[r] = st.ld(r);
    // The two players' (move, comm): rows 0..3 of `actions` [4][n] -- or, per player, an
    // [n][2] array of pairs (the batched form of multi_step's ego_action / alt_action tuples: a
    // policy's [n, 2] output is consumed as it lies); the partner may also be drawn here,
    // uniformly from the env's own PCG32 stream (oc_step_opts).  All three tests are uniform.
    // ISSUE PHASE: every load of the step goes out before any loaded word is touched.  (Until
    // round 3 each source decoded its words inside its own branch -- the int64 range check, the
    // PCG32 draw -- which put an s_waitcnt vmcnt(0) for ALL loads issued so far, the state's
    // included, in front of the loads still to come: the episode statistics' three loads started
    // a second memory round trip.  ego pairs + in-kernel partner + statistics: 3.91 us per step at
    // 4 096 envs against 2.9 plain.)
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef int v4i __attribute__((ext_vector_type(4)));
    // raw words as they are loaded (no shuffling here: a move of a loaded word is a wait for it):
    // int64 pairs {move lo, move hi, comm lo, comm hi}; int32 pairs and rows {move, comm}.
    // XO != 0: NO BRANCH over the sources.  Per player ONE 16-byte load serves both [n][2] forms --
    // int64 pairs (lane offset 16 i: the pair), int32 pairs (8 i: the pair and the next env's) -- two
    // dword loads serve the rows and one the partner's PCG32 word, each through a buffer descriptor
    // whose length is 0 when the form is not in use: a load past the end of its buffer returns 0 and
    // touches no memory.  (A chain of uniform branches, one
    // load per arm, is laid out as consecutive `if`s with flag words and a value merged at every
    // join; the wait-count pass follows EVERY path through them, found a register written on one path
    // -- a zero default, a copy for the merge, an address formed in a destination register -- with a
    // load outstanding on another, and put an s_waitcnt vmcnt(0) into the common path in front of the
    // episode statistics' loads: a second memory round trip for int32 pairs + statistics and for the
    // fused policies.)
    v4i eq = {0, 0, 0, 0}, aq = {0, 0, 0, 0};
    v2i er = {0, 0}, ar = {0, 0};
    uint32_t alt_rs = 0;
    if constexpr (XO != 0) {
      const auto desc = [&](const void *base, bool on, int bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, on ? bytes : 0, 0x00020000);
      };
      const int sh = pairs64 ? 4 : 3;
      const __amdgpu_buffer_rsrc_t d_er = desc(actions_, !ego_from_pairs, (int)n_ * 16),
                                   d_ar = desc(actions_, !alt_from_pairs && !alt_from_rng, (int)n_ * 16),
                                   d_eq = desc(ego_src_, ego_from_pairs, (int)n_ << sh),
                                   d_aq = desc(alt_src_, alt_from_pairs && !alt_from_rng, (int)n_ << sh),
                                   d_rs = desc(alt_src_, alt_from_rng, (int)n_ * 4);
      const int off4 = i << 2, offq = i << sh, row = (int)n_ * 4;
      eq = (v4i)__builtin_amdgcn_raw_buffer_load_b128(d_eq, offq, 0, 0);
      aq = (v4i)__builtin_amdgcn_raw_buffer_load_b128(d_aq, offq, 0, 0);
      alt_rs = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(d_rs, off4, 0, 0);
      er.x = __builtin_amdgcn_raw_buffer_load_b32(d_er, off4, 0, 0);
      er.y = __builtin_amdgcn_raw_buffer_load_b32(d_er, off4, row, 0);
      ar.x = __builtin_amdgcn_raw_buffer_load_b32(d_ar, off4, 2 * row, 0);
      ar.y = __builtin_amdgcn_raw_buffer_load_b32(d_ar, off4, 3 * row, 0);
    } else {
      const Rows ac(actions_, n_, 4, i);
      er.x = ac.ld(0), er.y = ac.ld(1);
      ar.x = ac.ld(2), ar.y = ac.ld(3);
    }
    if constexpr (!LDS) tb = stage_tables<false>(p.tables, p.n16, p.quot_bytes);
    // the output pointers are needed hundreds of instructions from here, where the compiler
    // would place their scalar loads -- and a wait on them -- in the middle of the step; fetch
    // them now, under the wait for the state that has to be served anyway
    asm volatile("" ::"s"(tb.dist), "s"(p.obs), "s"(p.timestep), "s"(p.reward), "s"(p.done),
                 "s"(p.sparse), "s"(p.auto_reset), "s"(p.R.inv_T), "s"(p.R.inv_max_path));
    if constexpr (XO != 0) asm volatile("" ::"s"(p.opt.ep_return), "s"(p.opt.ep_length));
#if defined(OC_SPECIALIZED) && !defined(OC_SPEC_GEOMETRY)
    // structure library: the map's geometry is a kernel argument; the first things the step
    // needs of it -- row length, tile planes, the Delivery tile -- are fetched here as well
    asm volatile("" ::"s"(L.W()), "s"(L.ncells()), "s"(L.max_path()), "s"(L.cell_lo(0)), "s"(L.cell_hi(0)),
                 "s"(L.deliv_pos(0)));
#endif
    // episode statistics: the running return / length and the previous step's done flag are
    // loaded now, with the state, and consumed after the last store of the step.  (Loading the
    // totals behind the barrier instead -- they are the shaping wave's own -- was slower: vmcnt
    // retires in order, so the shaping's distance lookups then waited for them: 3.64 -> 3.90 us.)
    double ep_ret = 0.0;
    int ep_len = 0, prev_done = 0;
    if (XO != 0 && p.opt.ep_return != nullptr) {   // uniform
      const Rows er(p.opt.ep_return, n_, 1, i, 8);
      ep_ret = __builtin_bit_cast(double, (v2i)__builtin_amdgcn_raw_buffer_load_b64(er.rsrc, er.voff, 0, 0));
      ep_len = Rows(p.opt.ep_length, n_, 1, i).ld(0);
      prev_done = Rows(p.done, n_, 1, i).ld(0);
    }
    Env<A, M, DUP> e;
    unpack<A, M, DUP>(e, w);
    // split: nothing that is updated in place -- state rows, the words of the random streams, the
    // done row the episode statistics read -- may be stored before every wave of the workgroup
    // holds its copy
    uint32_t place_rs = 0;
    if constexpr (SPLIT) {
      if (L.nscatter() != 0 && p.rng != nullptr) place_rs = (uint32_t)Rows(p.rng, n_, 1, i).ld(0);   // uniform
      // (the raw action words pass THROUGH this statement: the optimiser otherwise threads the decode
      // below back into the branch that issued each load, and its wait in front of the later loads)
      if constexpr (XO != 0)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" : "+v"(eq), "+v"(aq), "+v"(er), "+v"(ar), "+v"(alt_rs)::"memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    } else if constexpr (XO != 0) {
      asm volatile("" : "+v"(eq), "+v"(aq), "+v"(er), "+v"(ar), "+v"(alt_rs));
    }
    // DECODE PHASE (every load is out; a split workgroup is past its barrier: branches are free of
    // waits here).  A form not in use read 0; an int64 outside int32 is no valid index: -1.
    int ego_mv = er.x, ego_cm = er.y, alt_mv = ar.x, alt_cm = ar.y;
    if constexpr (XO != 0) {
      const auto narrow = [](int lo, int hi) { return lo | ~p_eq_any(hi, lo >> 31); };
      if (pairs64) {   // uniform
        ego_mv |= narrow(eq.x, eq.y), ego_cm |= narrow(eq.z, eq.w);
        alt_mv |= narrow(aq.x, aq.y), alt_cm |= narrow(aq.z, aq.w);
      } else {
        ego_mv |= eq.x, ego_cm |= eq.y;
        alt_mv |= aq.x, alt_cm |= aq.y;
      }
    }
    if (alt_from_rng) {   // uniform: the partner's draw, two steps of the env's PCG32 stream
      alt_mv = (int)__umulhi(pcg32(alt_rs), 4u);
      alt_cm = (int)__umulhi(pcg32(alt_rs), (uint32_t)p.cfg.obs.num_comm);
      if (!SPLIT || D_STATE) {   // (split: stored by the wave that owns the state, behind the barrier)
        Rows(alt_src_, n_, 1, i).st(0, (int)alt_rs);
        if (p.opt.alt_played != nullptr) {
          const Rows ap(p.opt.alt_played, n_, 2, i);
          ap.st(0, alt_mv);
          ap.st(1, alt_cm);
        }
      }
    }
    OC_STAMP(1);   // state + actions arrived
    // comm one-hots (:227-246); an index the reference's one_hot[idx] = 1 would raise on is
    // flagged (OC_ERR_ACTION) and sends nothing
    // (XO == 2: any wrapper configuration; 0 and 1 run the standard one, folded)
    const bool cfg_comm_on = XO == 2 ? p.cfg.communication_on != 0 : true, cfg_ego_led = XO == 2 ? p.cfg.ego_led != 0 : false;
    const int cfg_can_move = XO == 2 ? p.cfg.can_move_mask : 3, cfg_ego_idx = XO == 2 ? p.cfg.ego_agent_idx : 0;
    const int cfg_blind = XO == 2 ? p.cfg.obs.blind_mask : 0;
    const unsigned NC = (unsigned)p.cfg.obs.num_comm;
    // (per-lane predicates are P words, 0 / -1, see `hide`; the cfg_* tests are wave-uniform)
    const bool ego_talks = cfg_comm_on, alt_talks = cfg_comm_on && !cfg_ego_led;
    const P ego_cm_bad = p_geu_any((unsigned)ego_cm, NC), alt_cm_bad = p_geu_any((unsigned)alt_cm, NC);
    const P bad_cm = (p_of(ego_talks) & ego_cm_bad) | (p_of(alt_talks) & alt_cm_bad);
    const int c0 = ego_talks ? (ego_cm | ego_cm_bad) : -1;    // the index, or -1 = nothing sent
    const int c1 = alt_talks ? (alt_cm | alt_cm_bad) : -1;
    if constexpr (D_STATE) {
      cm.st(0, c0);
      cm.st(1, c1);
    }
    // NAV_ACTIONS lookup (both indices, moved or not: :248) + CAN_MOVE gating + ego_agent_idx
    // (:250-262); NAV_ACTIONS[idx] raises for idx > 3: flagged, executed as (0, 0)
    const P bad_mv = p_gtu_any((unsigned)ego_mv, 3u) | p_gtu_any((unsigned)alt_mv, 3u);
    const int em = (cfg_can_move & 1) ? (int)min((unsigned)ego_mv, 4u) : OC_ACT_NOOP;   // (> 3 -> 4 = OC_ACT_NOOP)
    const int am = (cfg_can_move & 2) ? (int)min((unsigned)alt_mv, 4u) : OC_ACT_NOOP;
    int act[A];
    act[0] = cfg_ego_idx == 0 ? em : am;
    act[1] = cfg_ego_idx == 0 ? am : em;
    const int err_before = e.err;
    e.err |= (bad_mv | bad_cm) & OC_ERR_ACTION;
    ShapeIn<2> sin;
    ShapeLoads<2> sld;
    // (the plain variant is only launched for play == 0; the general one reads the flag)
    env_step<A, M, DUP, XO == 2 ? 2 : 0>(L, p.R, tb.dist, tb.probe, e, act, reward, done, success, sin, sld OC_STAMP_PASS);
    comp = e.completed;
    err = e.err != err_before;
    if constexpr (D_STATE) {
      Out(p.done, p.n, 1, i).st(0, done);
#ifndef OC_STAMPS
      if (p.sparse != nullptr) Out(p.sparse, p.n, 1, i).st(0, reward);
#endif
    }
    if constexpr (DUTY != DUTY_SHAPE) {   // (the shaping reads the pre-reset env only, through `sin`)
      if ((DUTY & (DUTY_OBS0 | DUTY_OBS1)) == 0 && L.nscatter() == 0) {
        // the state wave of a split launch, a fixed level (uniform; compile-time in specialised
        // builds): the fresh episode is a constant, selected word by word -- a branch on `done` is
        // v_cmp -> s_and_saveexec, a scalar read of a vector-written mask (see `hide`).  (A wave that
        // goes on to the observations keeps the branch: it would have to unpack the words again.)
        pack<A, M, DUP>(e, w);
        const P fresh = -done & p_of(p.auto_reset != 0);
#pragma unroll
        for (int r = 0; r < WS; r++) w[r] = sel(fresh, L.init_words(r), w[r]);
      } else if (done && p.auto_reset) {
#pragma unroll
        for (int r = 0; r < WS; r++) w[r] = L.init_words(r);
        if constexpr (SPLIT) {   // every wave draws the same cells from its copy of the stream's word
          place_items_from<A, M, WS>(L, tb, p.placement, p.rng != nullptr, place_rs, p.n, i, w);
          if (D_STATE && L.nscatter() != 0 && p.rng != nullptr) p.rng[i] = place_rs;
        } else {
          place_items<A, M, WS>(L, tb, p.placement, p.rng, p.n, i, w);
        }
        unpack<A, M, DUP>(e, w);
      } else {
        pack<A, M, DUP>(e, w);
      }
    }
    if constexpr (D_STATE) {
#pragma unroll
      for (int r = 0; r < WS; r++) st.st(r, w[r]);
    }
    ShapeQ<2> sq;
    if constexpr (D_SHAPE) shaping_lookup<2>(L, p.R.inv_max_path, sin, sld, sq OC_STAMP_PASS);
    const int C = p.cfg.obs.num_comm;
    const int F = 22 + L.S() + 2 * C;
    const bool ego_blind = cfg_blind & 1;
    const Out ob(p.obs, p.n, 2 * F, i, OT == 1 ? 1 : 4);
    if constexpr (POL && SPLIT) {
      // (split launch with the policies fused: the rows also go to the LDS image the policy passes read)
      const int ln = (int)threadIdx.x & 63;
      if constexpr ((DUTY & DUTY_OBS0) != 0) {
        const RowsLdsT<WT ? AUX_WT : 0, OT> obl(ob, pol_lds_feat<POL_ROWS>(), 0, ln);
        env_obs<A, M, DUP, OT>(L, p.R, e, 0, p.cfg.obs.fow_radius, cfg_blind & 1, ego_blind, C, c0, c1, obl, 0);
        const double tsd = timestep_of(e.t, p.R);
        Out(p.timestep, p.n, 1, i, 8).st_f64(0, tsd);
        pol_lds_ts()[ln] = (float)tsd;
      }
      if constexpr ((DUTY & DUTY_OBS1) != 0) {
        const RowsLdsT<WT ? AUX_WT : 0, OT> obl(ob, pol_lds_feat<POL_ROWS>() + POL_ROWS * 64, F, ln);
        env_obs<A, M, DUP, OT>(L, p.R, e, 1, p.cfg.obs.fow_radius, (cfg_blind >> 1) & 1, ego_blind, C, c0, c1, obl, F);
      }
    } else {
      if constexpr ((DUTY & DUTY_OBS0) != 0)
        env_obs<A, M, DUP, OT>(L, p.R, e, 0, p.cfg.obs.fow_radius, cfg_blind & 1, ego_blind, C, c0, c1, ob, 0);
      if constexpr ((DUTY & DUTY_OBS1) != 0)
        env_obs<A, M, DUP, OT>(L, p.R, e, 1, p.cfg.obs.fow_radius, (cfg_blind >> 1) & 1, ego_blind, C, c0, c1, ob, F);
      if constexpr ((DUTY & DUTY_OBS0) != 0) Out(p.timestep, p.n, 1, i, 8).st_f64(0, timestep_of(e.t, p.R));
    }
    OC_STAMP(5);   // observation stores issued
    if constexpr (D_SHAPE) {
      // ... and they drain while the shaping is summed
      double s0, s1;
      shaping_sum<2>(L, sin, sq, s0, s1 OC_STAMP_PASS);
      const double shaped = ((double)reward - s0) - s1;  // :282
      Out(p.reward, p.n, 1, i, 8).st_f64(0, shaped);
      if (XO != 0 && p.opt.ep_return != nullptr) {   // uniform
        Out(p.opt.ep_return, p.n, 1, i, 8).st_f64(0, prev_done ? shaped : ep_ret + shaped);
        Out(p.opt.ep_length, p.n, 1, i).st(0, prev_done ? 1 : ep_len + 1);
      }
    }
  }
  OC_STAMP(7);   // every store issued
  if constexpr (D_STATE) slot.add(metrics_ != nullptr, valid, done, success, reward, comp, err);
  OC_STAMP(8);
  if constexpr (POL) {
    static_assert(XO != 0, "the fused policies belong to the variants with action sources");
    // split: every observation row (and the timestep) of this workgroup's 64 envs is in LDS;
    // a lone wave re-reads its own rows from memory
    if constexpr (SPLIT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int lane = (int)threadIdx.x & 63;
    const int C = p.cfg.obs.num_comm, F = 22 + L.S() + 2 * C;
    const uint32_t n32 = (uint32_t)n_;
    constexpr int ELEM = OT == 1 ? 1 : 4;
#pragma unroll
    for (int v = 0; v < 2; v++)
#pragma unroll
      for (int q = 0; q < 2; q++) {
        // a split workgroup: this wave's one pass (POL_MINE); a lone wave: all four
        if (SPLIT && (2 * v + q) != POL_MINE) continue;
        const int64_t env0 = (int64_t)(i & ~63) + 32 * q + (lane & 31);
        const bool ok = env0 < n_;
        const uint32_t env = (uint32_t)(ok ? env0 : n_ - 1);
        const void *rows = (const char *)p.obs + (size_t)v * F * n_ * ELEM;
        int32_t *pairs = (int32_t *)const_cast<void *>(v == 0 ? ego_src_ : alt_src_);
        if constexpr (SPLIT) {
          const int col = 32 * q + (lane & 31);
          ocpol::policy_pass<OT, 4, true, true>(rows, n32, env, ok, lane, p.pol[v].w1, p.pol[v].w2, p.pol[v].b2,
                                                p.pol[v].rng, pairs, nullptr, pol_lds_ts()[col], F, C, p.pol_ksteps,
                                                pol_lds_feat<POL_ROWS>() + v * POL_ROWS * 64, col, &pol_w);
        } else {
          ocpol::policy_pass<OT, 4>(rows, n32, env, ok, lane, p.pol[v].w1, p.pol[v].w2, p.pol[v].b2, p.pol[v].rng,
                                    pairs, nullptr, (float)p.timestep[env], F, C, p.pol_ksteps);
        }
      }
  }
#ifdef OC_STAMPS
  // the sparse-reward pointer doubles as the debug buffer in this build: int64 [waves][16]
  if ((threadIdx.x & 63) == 0 && p.sparse != nullptr) {   // one record per wave, split or not
    long long *dbg = (long long *)p.sparse + ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;
    for (int k = 0; k < 16; k++) dbg[k] = (long long)oc_tt[k];
  }
#endif
}

// OvercookedMultiEnv.multi_step in one launch.  SP = waves per 64 envs: 1 = one wave does the
// whole step (block_ & 0xFFFF threads per workgroup); 4 = split launch, 256 threads per workgroup,
// the wave's index picks its duty (a uniform branch; each arm is its own instruction stream).
// The six leading scalars repeat fields of `p` (and the workgroup size, which would otherwise come
// from the hidden arguments): as plain leading arguments they are preloaded into SGPRs at wave
// launch (-mllvm -amdgpu-kernarg-preload-count, build.py), so the state and action loads are
// issued without first waiting for a scalar kernarg load.  (block_ = workgroup size | which
// optional action sources are in use << 16: the branches on them are taken on a preloaded SGPR,
// not on a pointer that a scalar load has yet to deliver; ego_src_ = opts.ego_pairs, alt_src_ =
// opts.alt_rng or opts.alt_pairs: preloaded as well (n_ is 32 bits wide so that the lot fits the
// 14 preloadable dwords), so the general variant issues its action loads with the state loads)
template <int M, bool LDS, int OT, bool WT, bool DUP, int XO, int SP, bool POL = false>
__global__ void __launch_bounds__(256) k_multi_step(int32_t *const state_, const int32_t *const actions_,
                                                    int32_t *const comm_, int64_t *const metrics_,
                                                    const int32_t n_, const int32_t block_,
                                                    const void *const ego_src_, const void *const alt_src_,
                                                    const MultiArgs p) {
  static_assert(SP == 1 || SP == 2 || SP == 4, "waves per 64 envs");
  static_assert(SP != 2 || !POL, "the fused policies need the four-wave split");
  static_assert(SP == 1 || !LDS, "the split launch reads the tables from global memory");
  OC_TL_BEGIN();
#ifdef OC_SPECIALIZED
  const MultiArgs &pk = p;
#else
  // Generic library: the split arms read the argument block through the kernarg segment pointer,
  // not through the by-value parameter.  With the body inlined four times the compiler no longer
  // forwarded `p` to the constant address space and kept a private copy instead -- 984 bytes of
  // scratch per lane here, where the header accessors index into the block dynamically (17 us per
  // step instead of 6.8; through the pointer 5.6).  The specialised libraries never had the copy
  // (tests/test_host_cpu.py checks every kernel of every built library) and keep the parameter:
  // loads through the pointer are not known to be invariant, and cost them 3 % (tomato-2) to 28 %
  // (a random-* level, whose map geometry is read at run time).
  struct KernArgs { int32_t *a; const int32_t *b; int32_t *c; int64_t *d; int32_t n, blk; const void *e, *f; MultiArgs p; };
  [[maybe_unused]] const MultiArgs &pk = *reinterpret_cast<const MultiArgs *>(
      reinterpret_cast<const char *>((const void *)__builtin_amdgcn_kernarg_segment_ptr()) + offsetof(KernArgs, p));
#endif
#define OC_BODY(duty) multi_step_body<M, LDS, OT, WT, DUP, XO, (duty), true, POL>(state_, actions_, comm_, metrics_, n_, block_, ego_src_, alt_src_, pk)
  if constexpr (SP == 1) {
    multi_step_body<M, LDS, OT, WT, DUP, XO, DUTY_ALL, false, POL>(state_, actions_, comm_, metrics_, n_, block_, ego_src_, alt_src_, p);
  } else {
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if constexpr (SP == 2) {   // two waves per 64 envs: state + viewer 0 | shaping + viewer 1
      if (role == 0) OC_BODY(DUTY_STATE | DUTY_OBS0);
      else OC_BODY(DUTY_SHAPE | DUTY_OBS1);
    } else {
      // The waves of a workgroup do not start together: waves 2 and 3 came ~240 cycles after waves
      // 0 and 1 in every stamped run, and all four leave the barrier behind the state loads at the
      // time of the LAST one.  The two long arms (the observations: ~30 row stores each) therefore
      // go to the early waves, which at least issue their prologue under that wait, and the two
      // short arms to the late ones.
      if (role == 0) OC_BODY(DUTY_OBS0);
      else if (role == 1) OC_BODY(DUTY_OBS1);
      else if (role == 2) OC_BODY(DUTY_STATE);
      else OC_BODY(DUTY_SHAPE);
    }
  }
#undef OC_BODY
  OC_TL_END(p.timeline, p.timeline_stride);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// OC_TIMELINE builds: the launches that follow oc_timeline_begin() take consecutive records
#ifdef OC_TIMELINE
unsigned long long *g_timeline = nullptr;
int64_t g_timeline_left = 0, g_timeline_stride = 0;
#endif
// `waves`: an upper bound of the waves the launch will have
unsigned long long *timeline_next(int64_t waves, int64_t &stride) {
  stride = 0;
#ifdef OC_TIMELINE
  if (g_timeline_left > 0 && waves <= g_timeline_stride) {
    unsigned long long *r = g_timeline;
    g_timeline += 2 * g_timeline_stride;   // 16 bytes per wave
    g_timeline_left--;
    stride = g_timeline_stride;
    return r;
  }
#endif
  (void)waves;
  return nullptr;
}

// ---- launch policy ---------------------------------------------------------------------------
// ONE knob for measurements and tests forces what the library otherwise decides per call:
//   OC_LAUNCH="split=4,step_split=1,wt=0,lds=1,block=128"      (any subset, comma separated)
//     split       waves per 64 envs of the fused step (oc_multi_step): 1, 2 or 4     [split_for]
//     step_split  waves per 64 envs of the base step (oc_step): 1 or 2                [step_split_for]
//     wt          1 = write-through (sc1) stores, 0 = write-back                      [write_through]
//     lds         1 = the lane-indexed tables staged in LDS (one wave per 64 envs)    [tables_in_lds]
//     block       threads per workgroup of an unsplit launch: 64, 128 or 256          [block_size_for]
// Read at EVERY call (a getenv and a string compare), so one process can run several policies --
// tests/test_hip_parity.py::test_forced_launch_policies does.  Results never depend on it.
// (Round 2 had six variables, three of them latched at first use; they are gone.)
struct LaunchPolicy {
  int split = 0, step_split = 0, wt = -1, lds = -1, block = 0;
};
LaunchPolicy launch_policy() {
  static thread_local char seen[128] = "\x01";
  static thread_local LaunchPolicy cur;
  const char *v = getenv("OC_LAUNCH");
  if (!v) v = "";
  if (strncmp(v, seen, sizeof(seen)) == 0) return cur;
  snprintf(seen, sizeof(seen), "%s", v);
  LaunchPolicy p;
  for (const char *q = v; *q;) {
    int val = 0;
    char key[16] = "";
    int used = 0;
    if (sscanf(q, " %15[a-z_]=%d%n", key, &val, &used) == 2) {
      if (!strcmp(key, "split") && (val == 1 || val == 2 || val == 4)) p.split = val;
      else if (!strcmp(key, "step_split") && (val == 1 || val == 2)) p.step_split = val;
      else if (!strcmp(key, "wt") && (val == 0 || val == 1)) p.wt = val;
      else if (!strcmp(key, "lds") && (val == 0 || val == 1)) p.lds = val;
      else if (!strcmp(key, "block") && (val == 64 || val == 128 || val == 256)) p.block = val;
      q += used;
    }
    while (*q && *q != ',') q++;
    if (*q == ',') q++;
  }
  cur = p;
  return cur;
}

int block_size_for(int64_t) {
  // One wave per workgroup spreads a batch over the most CUs and measured fastest at every
  // batch size from 4 096 to 524 288 envs (MI355X sweeps, profiles/r01_v3_block_lds_sweep.txt,
  // r01_v11_geometry_sweep.txt).
  const int forced = launch_policy().block;
  return forced ? forced : 64;
}

// rows are addressed with 32-bit byte offsets through a buffer descriptor
bool fits_buffer(int64_t n, int64_t rows, int elem) { return n * rows * elem < (int64_t)0x7FFFFFFF; }

template <typename K, typename... Args>
int launch_n(K kernel, int64_t n, void *stream, size_t lds_bytes, const Args &...args) {
  if (n == 0) return OC_OK;
  const int bs = block_size_for(n);
  const int64_t grid = (n + bs - 1) / bs;
  if (grid > 0x7FFFFFFF) return fail(OC_E_BADARG, "n too large");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(bs), lds_bytes, (hipStream_t)stream, args...);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "kernel launch");
  return OC_OK;
}
template <typename Args, typename K>
int launch(K kernel, const Args &args, int64_t n, void *stream, size_t lds_bytes = 0) {
  return launch_n(kernel, n, stream, lds_bytes, args);
}
// k_step / k_multi_step: hot scalars first (preloaded kernel arguments), then the full block
template <typename K>
int launch_st(K kernel, const StepArgs &a, int64_t n, void *stream, size_t lds_bytes = 0, int sp = 1) {
  if (sp > 1) {   // split launch: sp waves per 64 envs, one workgroup each
    const int64_t grid = (n + 63) / 64;
    if (grid > 0x7FFFFFFF) return fail(OC_E_BADARG, "n too large");
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(64 * sp), 0, (hipStream_t)stream, a.state, a.actions,
                       a.metrics, a.n, (int32_t)(64 | ((a.auto_reset ? 1 : 0) << 16)), a.R.T, a.tables,
                       a.R.inv_max_path, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return OC_OK;
  }
  return launch_n(kernel, n, stream, lds_bytes, a.state, a.actions, a.metrics, a.n,
                  (int32_t)(block_size_for(n) | ((a.auto_reset ? 1 : 0) << 16)), a.R.T, a.tables,
                  a.R.inv_max_path, a);
}
template <typename K>
int launch_ms(K kernel, const MultiArgs &a, int64_t n, void *stream, size_t lds_bytes = 0) {
  const int32_t src = (a.opt.ego_pairs ? 1 : 0) | (a.opt.alt_pairs ? 2 : 0) | (a.opt.alt_rng ? 4 : 0) |
                      (a.opt.pairs_int64 ? 8 : 0);
  return launch_n(kernel, n, stream, lds_bytes, a.state, a.actions, a.comm, a.metrics, (int32_t)a.n,
                  (int32_t)(block_size_for(n) | (src << 16)), (const void *)a.opt.ego_pairs,
                  a.opt.alt_rng ? (const void *)a.opt.alt_rng : (const void *)a.opt.alt_pairs, a);
}

// Split launch of the fused step (k_multi_step<..., SP = 4>): 256 threads per workgroup over
// ceil(n / 64) workgroups.
template <typename K>
int launch_ms_split(K kernel, int sp, const MultiArgs &a, int64_t n, void *stream) {
  const int64_t grid = (n + 63) / 64;
  if (grid > 0x7FFFFFFF) return fail(OC_E_BADARG, "n too large");
  const int32_t src = (a.opt.ego_pairs ? 1 : 0) | (a.opt.alt_pairs ? 2 : 0) | (a.opt.alt_rng ? 4 : 0) |
                      (a.opt.pairs_int64 ? 8 : 0);
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(64 * sp), 0, (hipStream_t)stream, a.state, a.actions,
                     a.comm, a.metrics, (int32_t)a.n, (int32_t)(64 | (src << 16)), (const void *)a.opt.ego_pairs,
                     a.opt.alt_rng ? (const void *)a.opt.alt_rng : (const void *)a.opt.alt_pairs, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "kernel launch");
  return OC_OK;
}

// Waves per 64 envs for the fused step (see multi_step_body): four up to 32 768 envs, one beyond.
// us per step, one / two / four waves per 64 envs (tools/ab_set.sh, MI355X, round-3 kernels --
// predicates as VALU words, ~4 cycles per instruction):
//   tomato-2  4 096 3.41 / - / 2.82   16 384 - / - / 2.95   32 768 - / 4.13 / 3.75   40 960 3.93 / - / 4.40
//             49 152 4.14 / - / 4.48   65 536 4.94 / 5.17 / 5.31   131 072 7.73 / - / -
//   salad-2   32 768 4.08 / 4.32 / 3.97
// (Round 2, at ~7 cycles per instruction: four up to 24 576, two up to 32 768 -- the two-way split
// no longer wins anywhere and is only launched on the caller's hint.)
// The caller's hint (oc_step_opts.waves_per_64 = 1 / 2 / 4) and OC_LAUNCH=split=... override.
int split_for(int64_t n, int hint) {
  const int forced = launch_policy().split;
  if (forced) return forced;
  if (hint == 1 || hint == 2 || hint == 4) return hint;
  return n <= 32768 ? 4 : 1;
}

// The same for the base step (k_step<..., SP = 2>: state wave + shaping wave): two up to 16 384
// envs (tl-3: 16 384 3.44 -> 3.00 us, 24 576 3.51 / 3.97, 32 768 3.58 / 4.06, 65 536 3.82 / 4.38).
int step_split_for(int64_t n) {
  const int forced = launch_policy().step_split;
  if (forced) return forced;
  return n <= 16384 ? 2 : 1;
}

bool write_through(int64_t n) {
  // sc1 stores (see RowsT) at every batch size (tools/wt_threshold.sh)
  (void)n;
  const int forced = launch_policy().wt;
  return forced >= 0 ? forced == 1 : true;
}

bool tables_in_lds(int64_t) {
  // Off by default: with one- or two-wave workgroups the staging pass + barrier never paid
  // in the sweep (it only wins with 256-thread workgroups, which lose overall).
  // OC_LAUNCH=lds=1 selects the LDS variant (kept compiled, tested, and measured).
  return launch_policy().lds == 1;
}

// X(A, M, DUP): one instantiation per (agents, items, dup mode).  `A_`, `M_`, `D_` are locals of
// the caller.
#ifdef OC_SPECIALIZED
constexpr bool OC_SPEC_DUP = OC_SPEC_HDR.has_dup != 0;
#define OC_FOR_AM(X)                                                                                  \
  if (A_ == OC_SPEC_HDR.A && M_ == OC_SPEC_HDR.M && D_ == OC_SPEC_DUP) { X(OC_SPEC_HDR.A, OC_SPEC_HDR.M, OC_SPEC_DUP); } \
  return fail(OC_E_BADARG, "specialised library built for another (num_agents, num_items)");
#else
#define OC_FOR_AM_D(X, DD)                        \
  if (A_ == 2 && M_ == 3) { X(2, 3, DD); }        \
  if (A_ == 2 && M_ == 4) { X(2, 4, DD); }        \
  if (A_ == 2 && M_ == 5) { X(2, 5, DD); }        \
  if (A_ == 3 && M_ == 3) { X(3, 3, DD); }        \
  if (A_ == 3 && M_ == 4) { X(3, 4, DD); }        \
  if (A_ == 3 && M_ == 5) { X(3, 5, DD); }        \
  if (A_ == 4 && M_ == 3) { X(4, 3, DD); }        \
  if (A_ == 4 && M_ == 4) { X(4, 4, DD); }        \
  if (A_ == 4 && M_ == 5) { X(4, 5, DD); }
#define OC_FOR_AM(X)                              \
  if (D_) { OC_FOR_AM_D(X, true) } else { OC_FOR_AM_D(X, false) } \
  return fail(OC_E_BADARG, "unsupported (num_agents, num_items): need A in 2..4, M in 3..5");
#endif

int tset_of_sig(int sig) {
  int ts = 0;
  for (int t = 0; t < OC_NTYPES; t++)
    if ((sig >> (4 * t)) & 15) ts |= 1 << t;
  return ts;
}
// nibble counts (include/oc_level.h goal_sig) -> the 7-bit form the dup-mode item words carry
// (two bits per food type, bit 6 the Plate); -1 when a count does not fit
int sig7_of_sig(int sig) {
  int out = 0;
  for (int t = 0; t < OC_NTYPES; t++) {
    const int c = (sig >> (4 * t)) & 15;
    if (c > (t == OC_PLATE ? 1 : 3)) return -1;
    out |= c << (2 * t);
  }
  return out;
}

// hash((x, y)) of CPython >= 3.8 for small non-negative ints (Objects/tupleobject.c: the
// xxHash-style tuplehash; hash(int) is the int) -- what orders list(set(locations)), see pyset_first
uint64_t py_hash_xy(int x, int y) {
  const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
  uint64_t acc = P5;
  const uint64_t lane[2] = {(uint64_t)x, (uint64_t)y};
  for (int k = 0; k < 2; k++) {
    acc += lane[k] * P2;
    acc = (acc << 31) | (acc >> 33);
    acc *= P1;
  }
  acc += 2ULL ^ (P5 ^ 3527539ULL);
  return acc == (uint64_t)-1 ? 1546275796ULL : acc;
}
// the first eight probe slots of a location in an 8-slot set table, 3 bits each
// (Objects/setobject.c set_add_entry: i = hash & 7, then i = (5 i + 1 + (perturb >>= 5)) & 7)
uint32_t probe_code(int x, int y) {
  const uint64_t h = py_hash_xy(x, y);
  uint64_t perturb = h;
  uint32_t i = (uint32_t)(h & 7), code = 0;
  for (int t = 0; t < 8; t++) {
    code |= i << (3 * t);
    perturb >>= 5;
    i = (uint32_t)((i * 5 + 1 + perturb) & 7);
  }
  return code;
}

// Level blob (include/oc_level.h) -> LevelHdr + RunCfg.  Host only, no device work.
// Returns NULL on success, else a message.
const char *build_header(const int32_t *b, int32_t n_words, LevelHdr &h, RunCfg &run,
                         int32_t *slot_out = nullptr, int32_t *goal_index_out = nullptr) {
  if (!b || n_words < OC_LV_HEADER_WORDS) return "null or short blob";
  if (b[OC_LV_MAGIC] != OC_LV_MAGIC_VALUE || b[OC_LV_VERSION] != OC_LV_VERSION_VALUE ||
      b[OC_LV_TOTAL] != n_words)
    return "bad magic/version/length";
  const int W = b[OC_LV_W], H = b[OC_LV_H], A = b[OC_LV_A], M = b[OC_LV_M], S = b[OC_LV_S];
  const int nc = W * H;
  if (W < 1 || H < 1 || W > 16 || H > 16 || nc > OC_MAX_CELLS || A < 2 || A > OC_MAX_AGENTS || M < 1 ||
      M > OC_MAX_ITEMS || S < 1 || S > OC_MAX_SUBTASKS || b[OC_LV_NPAIR] > OC_MAX_PAIR ||
      b[OC_LV_NDELIV] < 1 || b[OC_LV_NDELIV] > OC_MAX_DELIV || b[OC_LV_MAX_PATH] > 255 ||
      b[OC_LV_T] < 0 || b[OC_LV_T] > 0xFFFF)
    return "level dimensions out of range";
  memset(&h, 0, sizeof(h));
  h.W = W; h.H = H; h.A = A; h.M = M; h.S = S; h.max_path = b[OC_LV_MAX_PATH]; h.ncells = nc;
  run.T = b[OC_LV_T];
  run.inv_T = run.T ? 1.0 / (double)run.T : 0.0;
  run.inv_max_path = 1.0 / (double)b[OC_LV_MAX_PATH];
  run.allergic = (uint32_t)b[OC_LV_ALLERGIC];
  run.play = (uint32_t)(b[OC_LV_FLAGS] & OC_FLAG_PLAY);
  const int32_t *cells = b + b[OC_LV_OFF_CELLS];
  const int32_t *ag = b + b[OC_LV_OFF_AGENTS], *it = b + b[OC_LV_OFF_ITEMS];
  const int32_t *st = b + b[OC_LV_OFF_SUBTASKS], *pr = b + b[OC_LV_OFF_PAIR], *dl = b + b[OC_LV_OFF_DELIV];
  for (int c = 0; c < nc; c++) {
    const uint64_t bit = 1ull << (c & 63);
    if (cells[c] != OC_FLOOR) h.nonfloor[c >> 6] |= bit;
    if (cells[c] & 1) h.cell_lo[c >> 6] |= bit;
    if (cells[c] & 2) h.cell_hi[c >> 6] |= bit;
  }
  for (int f = 0; f < 3; f++) h.food_item[f] = 255;
  int type_count[OC_NTYPES] = {0, 0, 0, 0};
  for (int i = 0; i < M; i++) {
    const int t = it[3 * i];
    if (t < 0 || t >= OC_NTYPES) return "bad item type";
    if (i > 0 && t != it[3 * (i - 1)] && type_count[t] > 0) return "items must be grouped by type (world order)";
    type_count[t]++;
    if (t != OC_PLATE) {
      if (h.food_item[t] != 255) h.has_dup = 1;   // a food type occurs twice: dup mode
      else h.food_item[t] = (uint32_t)i;
      h.food_items[t] |= 1u << i;
    }
    h.item_types |= (uint32_t)t << (4 * i);
  }
  for (int t = 0; t < OC_NTYPES; t++)
    if (type_count[t] > 3) return "more than three items of one type";
  for (int s = 0; s < S; s++)
    for (int t = 0; t < OC_NTYPES; t++)
      if (((st[4 * s + 1] >> (4 * t)) & 15) > 1) h.has_dup = 1;   // a goal object repeats a content type
  if (h.has_dup) {
    // the names a merge can create: every multiset of >= 2 contents drawn from the level's
    // items with at most one Plate (mergeable(), utils/core.py:240-257, checks nothing else)
    const int cp = type_count[OC_PLATE] > 0 ? 1 : 0;
    for (int a = 0; a <= type_count[0]; a++)
      for (int b2 = 0; b2 <= type_count[1]; b2++)
        for (int c = 0; c <= type_count[2]; c++)
          for (int d = 0; d <= cp; d++)
            if (a + b2 + c + d >= 2) {
              if (h.nnames >= (uint32_t)MAX_NAMES) return "too many distinct merged object names";
              h.name_sig[h.nnames++] = (uint32_t)(a | (b2 << 2) | (c << 4) | (d << 6));
            }
  }
  // canonical subtask slots (see RunCfg): Chop / Merge subtasks sorted by (kind, goal object,
  // food), ties in the caller's order -- tied subtasks are indistinguishable to the kernels --
  // then the Deliver subtasks in the caller's order
  int order[OC_MAX_SUBTASKS], slot[OC_MAX_SUBTASKS], nord = 0;
  for (int pass = 0; pass < 2; pass++)
    for (int u = 0; u < S; u++)
      if ((st[4 * u] == OC_DELIVER) == (pass == 1)) order[nord++] = u;
  for (int i = 1; i < S; i++) {   // insertion sort of the non-Deliver prefix (stable)
    const int u = order[i];
    if (st[4 * u] == OC_DELIVER) break;
    int j = i;
    while (j > 0) {
      const int v = order[j - 1];
      const bool greater = st[4 * v] != st[4 * u] ? st[4 * v] > st[4 * u]
                           : st[4 * v + 1] != st[4 * u + 1] ? st[4 * v + 1] > st[4 * u + 1]
                                                            : st[4 * v + 2] > st[4 * u + 2];
      if (!greater) break;
      order[j] = v;
      j--;
    }
    order[j] = u;
  }
  run.slot_identity = 1;
  memset(run.slot4, 0, sizeof(run.slot4));
  for (int c = 0; c < S; c++) {
    slot[order[c]] = c;
    if (order[c] != c) run.slot_identity = 0;
  }
  for (int u = 0; u < S; u++) run.slot4[u >> 2] |= (uint32_t)slot[u] << (8 * (u & 3));
  for (int s = 0; s < S; s++) {   // s = canonical slot from here on
    const int u = order[s];
    const int kind = st[4 * u], sig = st[4 * u + 1], food = st[4 * u + 2];
    const int s7 = sig7_of_sig(sig);
    if (s7 < 0) return "goal object holds more than three of a food or two Plates";
    // goals are told apart by their multiset in dup mode, by their type set otherwise (the same
    // thing when nothing repeats)
    const int ts = h.has_dup ? s7 : tset_of_sig(sig);
    if (kind == OC_DELIVER) {
      h.deliver_mask |= 1u << s;
      if (h.ndel >= (uint32_t)MAX_DELS) return "too many Deliver subtasks";
      h.del_sig[h.ndel] = (uint32_t)s7;
      h.del_tset[h.ndel] = (uint32_t)ts;
      h.del_bit[h.ndel] = (uint32_t)s;
      h.ndel++;
    } else {
      h.nondeliver_mask |= 1u << s;
      if (kind == OC_CHOP) {
        if (food < 0 || food > 2 || h.food_item[food] == 255) return "Chop of an absent food";
        h.chop_mask[food] |= 1u << s;
      }
    }
    uint32_t g = 0;
    for (; g < h.ngoal; g++)
      if (h.goal_tset[g] == (uint32_t)ts) break;
    if (g == h.ngoal) {
      if (h.ngoal >= (uint32_t)MAX_GOALS) return "too many distinct goal objects";
      h.goal_sig[h.ngoal] = (uint32_t)s7;
      h.goal_tset[h.ngoal++] = (uint32_t)ts;
    }
    if (kind == OC_DELIVER) h.goal_dl[g] |= 1u << s; else h.goal_nd[g] |= 1u << s;
    if (slot_out) slot_out[u] = s;
    if (goal_index_out) goal_index_out[u] = (int32_t)g;
  }
  // pair term: Plate + recipe[0] ingredient names, every unordered pair in that order
  // (overcooked_environment.py:319-363); one distance lookup per item pair
  for (int p = 0; p < b[OC_LV_NPAIR]; p++)
    for (int q = p + 1; q < b[OC_LV_NPAIR]; q++) {
      int cnt = 0;
      for (int i = 0; i < M; i++)
        for (int j = 0; j < M; j++)
          if (it[3 * i] == pr[p] && it[3 * j] == pr[q]) {
            if (h.npairlk >= (uint32_t)MAX_PAIRLK) return "too many item pairs in the shaping pair term";
            h.pairlk[h.npairlk++] = (uint32_t)(i | (j << 4));
            cnt++;
          }
      if (cnt) h.pairlk[h.npairlk - 1] |= 1u << 8; else h.pair_static_max++;
    }
  h.ndeliv = (uint32_t)b[OC_LV_NDELIV];
  for (uint32_t k = 0; k < h.ndeliv; k++) h.deliv_pos[k] = (uint32_t)(dl[2 * k] | (dl[2 * k + 1] << 4));
  // initial state words: OvercookedEnvironment.reset() (overcooked_environment.py:180-206)
  for (int a = 0; a < A; a++) h.init_words[a] = ag[2 * a] | (ag[2 * a + 1] << 4);
  for (int i = 0; i < M; i++) {
    int first = i;   // first item of the same type: its key is created when that one is inserted
    while (first > 0 && it[3 * (first - 1)] == it[3 * i]) first--;
    const int seqf = h.has_dup ? ((first << 4) | i) : i;
    const int sigf = h.has_dup ? sig_of_type<true>(it[3 * i]) : sig_of_type<false>(it[3 * i]);
    h.init_words[A + i] = it[3 * i + 1] | (it[3 * i + 2] << 4) | (i << 9) | (seqf << 16) | sigf;
  }
  {
    int n_chop = 0, n_groups = (int)h.pair_static_max;
    for (int f = 0; f < 3; f++) n_chop += __builtin_popcount(h.chop_mask[f]);
    for (uint32_t k = 0; k < h.npairlk; k++) n_groups += (h.pairlk[k] >> 8) & 1;
    int kmax = h.max_path + 64;                                          // Deliver term: distance + manhattan
    if (2 * n_chop * h.max_path > kmax) kmax = 2 * n_chop * h.max_path;  // Chop term numerator
    if (n_groups * h.max_path > kmax) kmax = n_groups * h.max_path;      // pair term numerator
    h.nquot = (uint32_t)(kmax + 2);
  }
  h.closed_border = border_closed(h) ? 1u : 0u;
  h.planes128 = nc > 64 ? 1u : 0u;
  h.nscatter = (uint32_t)b[OC_LV_NSCATTER];
  h.ncounters = (uint32_t)b[OC_LV_NCOUNTERS];
  if (h.nscatter > 4 || h.ncounters > OC_MAX_COUNTERS || (h.nscatter > 0 && h.ncounters < h.nscatter))
    return "bad scatter / Counter counts";
  for (uint32_t k = 0; k < h.nscatter; k++) {
    const int item = (b + b[OC_LV_OFF_SCATTER])[k];
    if (item < 0 || item >= M) return "bad scatter item index";
    h.scatter_item[k] = (uint32_t)item;
  }
  return nullptr;
}

// The STRUCTURE of a level (see OC_HDR_FIELDS): the header with the map's geometry blanked.  A
// specialised library is generated from, and checks new levels against, this part only.
LevelHdr structure_of(const LevelHdr &h) {
  LevelHdr t = h;
  t.W = t.H = t.ncells = t.max_path = 0;
  memset(t.nonfloor, 0, sizeof(t.nonfloor));
  memset(t.cell_lo, 0, sizeof(t.cell_lo));
  memset(t.cell_hi, 0, sizeof(t.cell_hi));
  memset(t.deliv_pos, 0, sizeof(t.deliv_pos));
  memset(t.init_words, 0, sizeof(t.init_words));
  t.nquot = 0;
  t.ncounters = 0;
  return t;
}

}  // namespace

extern "C" {

int oc_abi_version(void) { return OC_ABI_VERSION; }
const char *oc_last_error(void) { return g_err; }

int oc_is_specialized(void) {
#if defined(OC_SPECIALIZED) && defined(OC_SPEC_GEOMETRY)
  return 2;   // a "level" library: structure and geometry folded
#elif defined(OC_SPECIALIZED)
  return 1;   // a "structure" library: geometry at run time
#else
  return 0;
#endif
}

int oc_level_spec_source(const int32_t *b, int32_t n_words, int32_t with_geometry, char *buf, int32_t buf_size) {
  LevelHdr full, h;
  RunCfg run;
  const char *msg = build_header(b, n_words, full, run);
  if (msg) {
    snprintf(g_err, sizeof(g_err), "oc_level_spec_source: %s", msg);
    return OC_E_BADARG;
  }
  // a "structure" library keeps the geometry a run-time argument: blank it in the header
  h = with_geometry ? full : structure_of(full);
  if (!buf || buf_size < 64) return fail(OC_E_BADARG, "oc_level_spec_source: buffer too small");
  // LevelHdr holds 32-bit words and three pairs of 64-bit planes; emit it field by field
  // in declaration order as one aggregate initialiser.
  int n = 0;
#define EMIT(...) do { n += snprintf(buf + n, n < buf_size ? (size_t)(buf_size - n) : 0, __VA_ARGS__); } while (0)
  EMIT("// generated by oc_level_spec_source -- do not edit\n");
  EMIT("constexpr LevelHdr OC_SPEC_HDR = {\n  %d, %d, %d, %d, %d, %d, %d,\n  0x%xu,\n", h.W, h.H, h.ncells,
       h.max_path, h.S, h.A, h.M, h.item_types);
  const uint64_t *planes[3] = {h.nonfloor, h.cell_lo, h.cell_hi};
  for (int k = 0; k < 3; k++)
    EMIT("  {0x%llxull, 0x%llxull},\n", (unsigned long long)planes[k][0], (unsigned long long)planes[k][1]);
  EMIT("  0x%xu, 0x%xu,\n", h.nondeliver_mask, h.deliver_mask);
#define EMIT_ARR(arr, cnt) do { EMIT("  {"); for (int k_ = 0; k_ < (cnt); k_++) EMIT("%s0x%xu", k_ ? ", " : "", (unsigned)(arr)[k_]); EMIT("},\n"); } while (0)
  EMIT_ARR(h.chop_mask, 3);
  EMIT_ARR(h.food_item, 3);
  EMIT("  %uu,\n", h.ngoal);
  EMIT_ARR(h.goal_tset, MAX_GOALS);
  EMIT_ARR(h.goal_nd, MAX_GOALS);
  EMIT_ARR(h.goal_dl, MAX_GOALS);
  EMIT("  %uu,\n", h.ndel);
  EMIT_ARR(h.del_tset, MAX_DELS);
  EMIT_ARR(h.del_bit, MAX_DELS);
  EMIT("  %uu,\n", h.npairlk);
  EMIT_ARR(h.pairlk, MAX_PAIRLK);
  EMIT("  %uu,\n  %uu,\n", h.pair_static_max, h.ndeliv);
  EMIT_ARR(h.deliv_pos, OC_MAX_DELIV);
  EMIT("  {");
  for (int k = 0; k < OC_MAX_AGENTS + OC_MAX_ITEMS + 4; k++) EMIT("%s%d", k ? ", " : "", h.init_words[k]);
  EMIT("},\n  %uu,\n  %uu, %uu,\n", h.nquot, h.nscatter, h.ncounters);
  EMIT_ARR(h.scatter_item, 4);
  EMIT("  %uu,\n", h.has_dup);
  EMIT_ARR(h.goal_sig, MAX_GOALS);
  EMIT_ARR(h.del_sig, MAX_DELS);
  EMIT_ARR(h.food_items, 3);
  EMIT("  %uu,\n", h.nnames);
  EMIT_ARR(h.name_sig, MAX_NAMES);
  EMIT("  %uu, %uu\n", h.closed_border, h.planes128);
  EMIT("};\n");
#undef EMIT_ARR
#undef EMIT
  if (n >= buf_size) return fail(OC_E_BADARG, "oc_level_spec_source: buffer too small");
  return n;
}

int oc_level_create(const int32_t *b, int32_t n_words, oc_level_t **out) {
  if (!out) return fail(OC_E_BADARG, "oc_level_create: null out pointer");
  oc_level *lv = new (std::nothrow) oc_level();
  if (!lv) return fail(OC_E_BADARG, "oc_level_create: out of memory");
  lv->dev_tables = nullptr;
  LevelHdr &h = lv->hdr;
  const char *msg = build_header(b, n_words, h, lv->run, lv->slot, lv->goal_index);
  if (msg) {
    delete lv;
    snprintf(g_err, sizeof(g_err), "oc_level_create: %s", msg);
    return OC_E_BADARG;
  }
#ifdef OC_SPECIALIZED
  {
#ifdef OC_SPEC_GEOMETRY
    const LevelHdr spec = OC_SPEC_HDR, mine = h;
#else
    const LevelHdr spec = OC_SPEC_HDR, mine = structure_of(h);
#endif
    if (memcmp(&spec, &mine, sizeof(LevelHdr)) != 0) {
      delete lv;
      return fail(OC_E_BADARG, "oc_level_create: this library is specialised for a different level (a \"level\" "
                               "library) or level structure (recipes, item multiset, agent count, border kind)");
    }
  }
#endif
  const int nc = h.ncells;
  const int32_t *dist = b + b[OC_LV_OFF_DIST];
  // tables buffer: the quotients k / MAX_PATH for every numerator the shaping formula can
  // form (correctly rounded fp64 division, as CPython's int / int), then the u8 distances
  lv->quot_bytes = 0;   // (until round-1 v11: a table of fp64 quotients k / max_path came first)
  // ... then, in the last 64 bytes, the Counter tiles (x | y<<4) for random placement
  // ... then (dup mode's set-order lookups) one u32 of probe slots per cell, 4 * OC_MAX_CELLS bytes
  const size_t bytes = (((size_t)lv->quot_bytes + (size_t)nc * nc + 15) & ~(size_t)15) + 4 * OC_MAX_CELLS +
                       OC_MAX_COUNTERS;
  lv->n16 = (int32_t)(bytes / 16);
  uint8_t *img = new (std::nothrow) uint8_t[bytes];
  if (!img) {
    delete lv;
    return fail(OC_E_BADARG, "oc_level_create: out of memory");
  }
  memset(img, 0, bytes);
  for (int i = 0; i < nc * nc; i++) img[lv->quot_bytes + i] = (uint8_t)dist[i];
  {
    uint32_t *pr = (uint32_t *)(img + bytes - OC_MAX_COUNTERS - 4 * OC_MAX_CELLS);
    for (int c = 0; c < nc; c++) pr[c] = probe_code(c % h.W, c / h.W);
    if (h.has_dup) {
      // eight stored probes must place a third location whatever two slots are taken
      for (int c = 0; c < nc; c++) {
        uint32_t seen = 0;
        for (int t = 0; t < 8; t++) seen |= 1u << ((pr[c] >> (3 * t)) & 7);
        if (__builtin_popcount(seen) < 3) {
          delete[] img;
          delete lv;
          return fail(OC_E_BADARG, "oc_level_create: a cell's set-table probe sequence is too short (dup mode)");
        }
      }
    }
  }
  {
    const int32_t *ct = b + b[OC_LV_OFF_COUNTERS];
    for (int k = 0; k < b[OC_LV_NCOUNTERS]; k++)
      img[bytes - OC_MAX_COUNTERS + k] = (uint8_t)(ct[2 * k] | (ct[2 * k + 1] << 4));
  }
  hipError_t e = hipGetDevice(&lv->device);
  if (e == hipSuccess) e = hipMalloc(&lv->dev_tables, bytes);
  if (e == hipSuccess) e = hipMemcpy(lv->dev_tables, img, bytes, hipMemcpyHostToDevice);
  delete[] img;
  if (e != hipSuccess) {
    oc_level_destroy(lv);
    fail_hip(e, "oc_level_create");
    return OC_E_NODEVICE;
  }
  *out = lv;
  return OC_OK;
}

int oc_level_destroy(oc_level_t *lv) {
  if (!lv) return OC_OK;
  if (lv->dev_tables) (void)hipFree(lv->dev_tables);
  delete lv;
  return OC_OK;
}

int oc_level_subtask_info(const int32_t *b, int32_t n_words, int32_t *slot, int32_t *goal_index, int32_t *dup) {
  LevelHdr h;
  RunCfg run;
  int32_t sl[OC_MAX_SUBTASKS], gi[OC_MAX_SUBTASKS];
  const char *msg = build_header(b, n_words, h, run, sl, gi);
  if (msg) {
    snprintf(g_err, sizeof(g_err), "oc_level_subtask_info: %s", msg);
    return OC_E_BADARG;
  }
  for (int s = 0; s < h.S; s++) {
    if (slot) slot[s] = sl[s];
    if (goal_index) goal_index[s] = gi[s];
  }
  if (dup) *dup = (int32_t)h.has_dup;
  return OC_OK;
}

// One slot per wave of 64 envs, rounded up to whole workgroups of four waves: under a forced
// 128- / 256-thread workgroup (OC_LAUNCH=block=...) the last workgroup may hold waves without an env,
// and those add zeros to THEIR slot.
int64_t oc_metrics_slots(int64_t n) { return n <= 0 ? 0 : (n + 255) / 256 * 4; }
int32_t oc_state_words(const oc_level_t *lv) {
  return lv ? lv->hdr.A + lv->hdr.M + 2 + (lv->hdr.has_dup ? 2 : 0) : 0;
}
int32_t oc_obs_rows(const oc_level_t *lv, int32_t num_comm) {
  return lv ? 22 + lv->hdr.S + 2 * num_comm : 0;
}

int oc_reset(const oc_level_t *lv, int32_t *state, const int32_t *mask, const int32_t *placement, uint32_t *rng,
             int64_t n, void *stream) {
  if (lv && n == 0) return OC_OK;
  if (!lv || !state || n < 0) return fail(OC_E_BADARG, "oc_reset: bad argument");
  if (lv->hdr.nscatter > 0 && !placement && !rng)
    return fail(OC_E_BADARG, "oc_reset: this level places items at random; pass `placement` or `rng`");
  ResetArgs a{lv->hdr, lv->dev_tables, lv->n16, lv->quot_bytes, state, mask, placement, rng, n};
  const int A_ = lv->hdr.A, M_ = lv->hdr.M;
  const bool D_ = lv->hdr.has_dup != 0;
#define OC_X(AA, MM, DD) return launch(k_reset<AA, MM, DD>, a, n, stream, 0)
  OC_FOR_AM(OC_X)
#undef OC_X
}

int oc_step(const oc_level_t *lv, int32_t *state, const int32_t *actions, int32_t *reward, int32_t *done,
            double *shaping, int32_t auto_reset, int64_t *metrics, const int32_t *placement, uint32_t *rng,
            int64_t n, void *stream) {
  if (lv && n == 0) return OC_OK;
  if (!lv || !state || !actions || !reward || !done || !shaping || n < 0)
    return fail(OC_E_BADARG, "oc_step: bad argument");
  if (!fits_buffer(n, oc_state_words(lv), 4) || !fits_buffer(n, 2, 8))
    return fail(OC_E_BADARG, "oc_step: n too large for one call (tensor rows are addressed with 32-bit offsets); split the batch");
  if (auto_reset && lv->hdr.nscatter > 0 && !placement && !rng)
    return fail(OC_E_BADARG, "oc_step: auto_reset on a random-placement level needs `placement` or `rng`");
  StepArgs a{lv->hdr, lv->run, lv->dev_tables, lv->n16, lv->quot_bytes, state, actions, reward, done, shaping,
             metrics, placement, rng, n, auto_reset, nullptr, 0};
  a.timeline = timeline_next(4 * ((n + 63) / 64), a.timeline_stride);
  const int A_ = lv->hdr.A, M_ = lv->hdr.M;
  const bool D_ = lv->hdr.has_dup != 0;
  const bool pl = lv->run.play != 0;
  const size_t lds = (size_t)lv->n16 * 16;
  if (tables_in_lds(n)) {
#define OC_X(AA, MM, DD) return pl ? launch_st(k_step<AA, MM, true, false, DD, true, 1>, a, n, stream, lds) \
                                   : launch_st(k_step<AA, MM, true, false, DD, false, 1>, a, n, stream, lds)
    OC_FOR_AM(OC_X)
#undef OC_X
  } else {
    if (write_through(n)) {
#ifdef OC_SPECIALIZED
      if (step_split_for(n) == 2) {   // (specialised libraries only: the generic library's build time)
#define OC_X(AA, MM, DD) return pl ? launch_st(k_step<AA, MM, false, true, DD, true, 2>, a, n, stream, 0, 2) \
                                   : launch_st(k_step<AA, MM, false, true, DD, false, 2>, a, n, stream, 0, 2)
        OC_FOR_AM(OC_X)
#undef OC_X
      }
#endif
#define OC_X(AA, MM, DD) return pl ? launch_st(k_step<AA, MM, false, true, DD, true, 1>, a, n, stream, 0) \
                                   : launch_st(k_step<AA, MM, false, true, DD, false, 1>, a, n, stream, 0)
      OC_FOR_AM(OC_X)
#undef OC_X
    }
#define OC_X(AA, MM, DD) return pl ? launch_st(k_step<AA, MM, false, false, DD, true, 1>, a, n, stream, 0) \
                                   : launch_st(k_step<AA, MM, false, false, DD, false, 1>, a, n, stream, 0)
    OC_FOR_AM(OC_X)
#undef OC_X
  }
}

int oc_obs(const oc_level_t *lv, const int32_t *state, const int32_t *comm, const oc_obs_cfg *cfg,
           void *obs, double *timestep, int64_t n, void *stream) {
  if (lv && cfg && n == 0) return OC_OK;
  if (!lv || !state || !comm || !cfg || !obs || !timestep || n < 0 || cfg->num_comm < 0 || cfg->num_comm > 128)
    return fail(OC_E_BADARG, "oc_obs: bad argument");
  if (!fits_buffer(n, 2 * (22 + lv->hdr.S + 2 * cfg->num_comm), 4))
    return fail(OC_E_BADARG, "oc_obs: n too large for one call (tensor rows are addressed with 32-bit offsets); split the batch");
  ObsArgs a{lv->hdr, lv->run, state, comm, obs, timestep, n, *cfg};
  const int A_ = lv->hdr.A, M_ = lv->hdr.M;
  const bool D_ = lv->hdr.has_dup != 0;
  const bool wt = write_through(n);
  if (cfg->obs_int8 < 0 || cfg->obs_int8 > 2) return fail(OC_E_BADARG, "oc_obs: obs_int8 must be 0 (int32), 1 (int8) or 2 (float32)");
  if (cfg->obs_int8 == 1) {
#define OC_X(AA, MM, DD) return wt ? launch(k_obs<AA, MM, 1, true, DD>, a, n, stream, 0) \
                                    : launch(k_obs<AA, MM, 1, false, DD>, a, n, stream, 0)
    OC_FOR_AM(OC_X)
#undef OC_X
  } else if (cfg->obs_int8 == 2) {
#define OC_X(AA, MM, DD) return wt ? launch(k_obs<AA, MM, 2, true, DD>, a, n, stream, 0) \
                                    : launch(k_obs<AA, MM, 2, false, DD>, a, n, stream, 0)
    OC_FOR_AM(OC_X)
#undef OC_X
  } else {
#define OC_X(AA, MM, DD) return wt ? launch(k_obs<AA, MM, 0, true, DD>, a, n, stream, 0) \
                                    : launch(k_obs<AA, MM, 0, false, DD>, a, n, stream, 0)
    OC_FOR_AM(OC_X)
#undef OC_X
  }
}

int32_t oc_image_words(const oc_level_t *lv) { return lv ? 7 * ((lv->hdr.ncells + 3) / 4) : 0; }

int oc_obs_image(const oc_level_t *lv, const int32_t *state, int32_t radius, int32_t *out, int8_t *holding,
                 int64_t n, void *stream) {
  if (lv && n == 0) return OC_OK;
  if (!lv || !state || !out || !holding || n < 0) return fail(OC_E_BADARG, "oc_obs_image: bad argument");
  if (!fits_buffer(n, 2 * oc_image_words(lv), 4))
    return fail(OC_E_BADARG, "oc_obs_image: n too large for one call; split the batch");
  ImageArgs a{lv->hdr, state, out, holding, n, radius};
  const int A_ = lv->hdr.A, M_ = lv->hdr.M;
  const bool D_ = lv->hdr.has_dup != 0;
#define OC_X(AA, MM, DD) return launch(k_obs_image<AA, MM, DD>, a, n, stream, 0)
  OC_FOR_AM(OC_X)
#undef OC_X
}

int oc_multi_step(const oc_level_t *lv, int32_t *state, int32_t *comm, const int32_t *actions,
                  const oc_wrap_cfg *cfg, void *obs, double *timestep, double *reward, int32_t *done,
                  int32_t *sparse, int32_t auto_reset, int64_t *metrics, const int32_t *placement, uint32_t *rng,
                  const oc_step_opts *opts, int64_t n, void *stream) {
  if (lv && cfg && n == 0) return OC_OK;
  oc_step_opts o;
  memset(&o, 0, sizeof(o));
  if (opts) o = *opts;
  if ((o.ep_return == nullptr) != (o.ep_length == nullptr))
    return fail(OC_E_BADARG, "oc_multi_step: pass both ep_return and ep_length, or neither");
  // the action rows may only be absent when both players' actions come from somewhere else
  if (!actions && !(o.ego_pairs && (o.alt_pairs || o.alt_rng)))
    return fail(OC_E_BADARG, "oc_multi_step: no `actions` and no complete replacement in `opts`");
  if (!lv || !state || !comm || !cfg || !obs || !timestep || !reward || !done || n < 0 ||
      cfg->obs.num_comm < 0 || cfg->obs.num_comm > 128)
    return fail(OC_E_BADARG, "oc_multi_step: bad argument");
  if (lv->hdr.A != 2)
    return fail(OC_E_BADARG, "oc_multi_step: the gym_comm wrapper drives exactly 2 agents");
  if (!fits_buffer(n, 2 * (22 + lv->hdr.S + 2 * cfg->obs.num_comm), 4) || !fits_buffer(n, 1, 16))
    return fail(OC_E_BADARG, "oc_multi_step: n too large for one call (tensor rows are addressed with 32-bit offsets); split the batch");
  if (auto_reset && lv->hdr.nscatter > 0 && !placement && !rng)
    return fail(OC_E_BADARG, "oc_multi_step: auto_reset on a random-placement level needs `placement` or `rng`");
  MultiArgs a{lv->hdr, lv->run, lv->dev_tables, lv->n16, lv->quot_bytes, state, comm, actions, obs, timestep,
              reward, done, sparse, metrics, placement, rng, o, n, auto_reset, *cfg, {}, 0, nullptr, 0};
  if (o.policy) {   // the closed loop in one launch (oc_step_opts.policy)
#ifndef OC_SPECIALIZED
    return fail(OC_E_BADARG, "oc_multi_step: opts.policy needs a specialised library (this is the generic one)");
#endif
    if (!o.ego_pairs || !o.alt_pairs || o.pairs_int64 || o.alt_rng)
      return fail(OC_E_BADARG, "oc_multi_step: opts.policy needs ego_pairs and alt_pairs (int32) and no alt_rng");
    if (cfg->obs.num_comm < 1 || cfg->obs.num_comm > 4)
      return fail(OC_E_BADARG, "oc_multi_step: opts.policy samples at most 4 comm channels; use oc_policy_mlp");
    for (int k = 0; k < 2; k++) {
      if (!o.policy[k].w1 || !o.policy[k].w2 || !o.policy[k].b2)
        return fail(OC_E_BADARG, "oc_multi_step: opts.policy[k] needs w1, w2 and b2");
      a.pol[k] = o.policy[k];
    }
    a.pol_ksteps = (22 + lv->hdr.S + 2 * cfg->obs.num_comm + 2 + 15) / 16;
    if (a.pol_ksteps > 3)
      return fail(OC_E_BADARG, "oc_multi_step: opts.policy handles at most 46 observation rows; use oc_policy_mlp");
    a.opt.policy = nullptr;   // (a host pointer: nothing on the device may look at it)
  }
  a.timeline = timeline_next(4 * ((n + 63) / 64), a.timeline_stride);
  const size_t lds = (size_t)lv->n16 * 16;
  const bool in_lds = tables_in_lds(n);
  const int ot = cfg->obs.obs_int8;   // 0 int32, 1 int8, 2 float32
  if (ot < 0 || ot > 2) return fail(OC_E_BADARG, "oc_multi_step: obs_int8 must be 0 (int32), 1 (int8) or 2 (float32)");
  const bool wt = write_through(n);
  const bool std_cfg = cfg->communication_on && !cfg->ego_led && cfg->can_move_mask == 3 &&
                       cfg->ego_agent_idx == 0 && cfg->obs.blind_mask == 0 && !lv->run.play;
  const bool opts_used = o.ep_return || o.ego_pairs || o.alt_pairs || o.alt_rng;
  const bool xo = opts_used || !std_cfg;
#ifdef OC_SPECIALIZED
  const bool x1 = opts_used && std_cfg;      // XO = 1: the options on the folded standard configuration
#else
  [[maybe_unused]] const bool x1 = false;    // (the generic library: its build time)
#endif
  if (o.policy && !std_cfg)
    return fail(OC_E_BADARG, "oc_multi_step: opts.policy runs on the wrapper's standard configuration (communication on, "
                             "not ego-led, both CAN_MOVE, ego_agent_idx 0, nobody BLIND, play off); use oc_policy_mlp + oc_multi_step");
  const int sp = oc_multi_step_waves(n, o.waves_per_64, xo ? 1 : 0);   // (the launch actually taken, see there)
#define OC_MS_X(MM, DD, XX)                                                           \
  do {                                                                                \
    if (in_lds && ot == 0) return launch_ms(k_multi_step<MM, true, 0, false, DD, XX, 1>, a, n, stream, lds);  \
    if (ot == 1) return wt ? launch_ms(k_multi_step<MM, false, 1, true, DD, XX, 1>, a, n, stream, 0)          \
                           : launch_ms(k_multi_step<MM, false, 1, false, DD, XX, 1>, a, n, stream, 0);        \
    if (ot == 2) return wt ? launch_ms(k_multi_step<MM, false, 2, true, DD, XX, 1>, a, n, stream, 0)          \
                           : launch_ms(k_multi_step<MM, false, 2, false, DD, XX, 1>, a, n, stream, 0);        \
    return wt ? launch_ms(k_multi_step<MM, false, 0, true, DD, XX, 1>, a, n, stream, 0)                       \
              : launch_ms(k_multi_step<MM, false, 0, false, DD, XX, 1>, a, n, stream, 0);                     \
  } while (0)
#define OC_MS_SPLIT(MM, DD, XX)                                                                            \
  do {                                                                                                     \
    if (ot == 1) return launch_ms_split(k_multi_step<MM, false, 1, true, DD, XX, 4>, 4, a, n, stream);    \
    if (ot == 2) return launch_ms_split(k_multi_step<MM, false, 2, true, DD, XX, 4>, 4, a, n, stream);    \
    return launch_ms_split(k_multi_step<MM, false, 0, true, DD, XX, 4>, 4, a, n, stream);                 \
  } while (0)
// (the generic library splits the plain variant only: its build time)
#define OC_MS_SPLIT2(MM, DD)                                                                                 \
  do {                                                                                                       \
    if (ot == 1) return launch_ms_split(k_multi_step<MM, false, 1, true, DD, 0, 2>, 2, a, n, stream);    \
    if (ot == 2) return launch_ms_split(k_multi_step<MM, false, 2, true, DD, 0, 2>, 2, a, n, stream);    \
    return launch_ms_split(k_multi_step<MM, false, 0, true, DD, 0, 2>, 2, a, n, stream);                 \
  } while (0)
#ifdef OC_SPECIALIZED
// general variant + both policies evaluated behind the step (oc_step_opts.policy)
#define OC_MS_POL(MM, DD)                                                                                       \
  do {                                                                                                          \
    if (!wt || in_lds) return fail(OC_E_BADARG, "oc_multi_step: opts.policy needs write-through stores and tables in global memory"); \
    if (sp == 4) {                                                                                              \
      if (ot == 1) return launch_ms_split(k_multi_step<MM, false, 1, true, DD, 1, 4, true>, 4, a, n, stream); \
      if (ot == 2) return launch_ms_split(k_multi_step<MM, false, 2, true, DD, 1, 4, true>, 4, a, n, stream); \
      return launch_ms_split(k_multi_step<MM, false, 0, true, DD, 1, 4, true>, 4, a, n, stream);             \
    }                                                                                                           \
    if (ot == 1) return launch_ms(k_multi_step<MM, false, 1, true, DD, 1, 1, true>, a, n, stream, 0);        \
    if (ot == 2) return launch_ms(k_multi_step<MM, false, 2, true, DD, 1, 1, true>, a, n, stream, 0);        \
    return launch_ms(k_multi_step<MM, false, 0, true, DD, 1, 1, true>, a, n, stream, 0);                     \
  } while (0)
#define OC_MS_SPLIT_BOTH(MM, DD)                   \
  if (o.policy) OC_MS_POL(MM, DD);                 \
  if (x1 && sp == 4) OC_MS_SPLIT(MM, DD, 1);       \
  if (xo && sp == 4) OC_MS_SPLIT(MM, DD, 2);       \
  if (!xo && sp == 4) OC_MS_SPLIT(MM, DD, 0);      \
  if (!xo && sp == 2) OC_MS_SPLIT2(MM, DD);        \
  if (x1) OC_MS_X(MM, DD, 1)
#else
#define OC_MS_SPLIT_BOTH(MM, DD) if (!xo && sp == 4) OC_MS_SPLIT(MM, DD, 0)
#endif
#define OC_MS(MM, DD)                            \
  do {                                           \
    OC_MS_SPLIT_BOTH(MM, DD);                    \
    if (xo) OC_MS_X(MM, DD, 2);                  \
    OC_MS_X(MM, DD, 0);                          \
  } while (0)
#ifdef OC_SPECIALIZED
  OC_MS(OC_SPEC_HDR.M, OC_SPEC_DUP);
#else
  if (lv->hdr.has_dup) {
    if (lv->hdr.M == 3) OC_MS(3, true);
    if (lv->hdr.M == 4) OC_MS(4, true);
    if (lv->hdr.M == 5) OC_MS(5, true);
  } else {
    if (lv->hdr.M == 3) OC_MS(3, false);
    if (lv->hdr.M == 4) OC_MS(4, false);
    if (lv->hdr.M == 5) OC_MS(5, false);
  }
  return fail(OC_E_BADARG, "oc_multi_step: unsupported number of items");
#endif
#undef OC_MS
#undef OC_MS_SPLIT_BOTH
#ifdef OC_SPECIALIZED
#undef OC_MS_POL
#endif
#undef OC_MS_SPLIT
#undef OC_MS_X
}

int oc_multi_step_prepare(const oc_level_t *lv, int32_t *state, int32_t *comm, const int32_t *actions,
                          const oc_wrap_cfg *cfg, void *obs, double *timestep, double *reward, int32_t *done,
                          int32_t *sparse, int32_t auto_reset, int64_t *metrics, const int32_t *placement,
                          uint32_t *rng, const oc_step_opts *opts, int64_t n, oc_call_t **out) {
  if (!out || !lv || !cfg) return fail(OC_E_BADARG, "oc_multi_step_prepare: bad argument");
  oc_call *c = new (std::nothrow) oc_call();
  if (!c) return fail(OC_E_BADARG, "oc_multi_step_prepare: out of memory");
  c->lv = lv, c->state = state, c->comm = comm, c->actions = actions, c->cfg = *cfg, c->obs = obs;
  c->timestep = timestep, c->reward = reward, c->done = done, c->sparse = sparse, c->auto_reset = auto_reset;
  c->metrics = metrics, c->placement = placement, c->rng = rng, c->n = n;
  memset(&c->opts, 0, sizeof(c->opts));
  if (opts) c->opts = *opts;
  c->has_pol = c->opts.policy != nullptr;
  if (c->has_pol) {   // (a host array of the caller: copied, so that it need not outlive this call)
    c->pol[0] = c->opts.policy[0], c->pol[1] = c->opts.policy[1];
    c->opts.policy = c->pol;
  }
  *out = c;
  return OC_OK;
}

int oc_call_launch(const oc_call_t *c, const void *ego_pairs, int32_t pairs_int64, void *stream) {
  if (!c) return fail(OC_E_BADARG, "oc_call_launch: null call");
  oc_step_opts o = c->opts;
  if (ego_pairs) {
    o.ego_pairs = (const int32_t *)ego_pairs;
    o.pairs_int64 = pairs_int64;
  }
  return oc_multi_step(c->lv, c->state, c->comm, c->actions, &c->cfg, c->obs, c->timestep, c->reward, c->done,
                       c->sparse, c->auto_reset, c->metrics, c->placement, c->rng, &o, c->n, stream);
}

int oc_call_destroy(oc_call_t *c) {
  delete c;
  return OC_OK;
}

int oc_timeline_begin(uint64_t *records, int64_t count, int64_t stride) {
#ifdef OC_TIMELINE
  if ((records == nullptr) != (count == 0) || count < 0 || (count > 0 && stride < 1))
    return fail(OC_E_BADARG, "oc_timeline_begin: bad argument");
  g_timeline = (unsigned long long *)records;
  g_timeline_left = count;
  g_timeline_stride = stride;
  return OC_OK;
#else
  (void)records;
  (void)count;
  (void)stride;
  return fail(OC_E_BADARG, "oc_timeline_begin: not a timeline build of the library (-DOC_TIMELINE)");
#endif
}

int32_t oc_multi_step_waves(int64_t n, int32_t hint, int32_t general_variant) {
  // the policy's wish ...
  const int sp = (write_through(n) && !tables_in_lds(n)) ? split_for(n, hint) : 1;
  // ... and what this library has kernels for (the same predicate oc_multi_step dispatches on):
#ifdef OC_SPECIALIZED
  return general_variant ? (sp == 4 ? 4 : 1) : sp;   // the general variant splits four ways or not at all
#else
  return (!general_variant && sp == 4) ? 4 : 1;      // the generic library splits the plain variant four ways only
#endif
}

int oc_random_actions(uint32_t *rng, int32_t *move_row, int32_t *comm_row, int32_t num_comm, int64_t n,
                      void *stream) {
  if (n == 0 && rng && move_row && comm_row) return OC_OK;
  if (!rng || !move_row || !comm_row || n < 0 || num_comm < 1)
    return fail(OC_E_BADARG, "oc_random_actions: bad argument");
  const int bs = 256;
  hipLaunchKernelGGL(k_random_actions, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, (hipStream_t)stream, rng,
                     move_row, comm_row, (uint32_t)num_comm, n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "kernel launch");
  return OC_OK;
}

}  // extern "C"

// oc_kernels.hip -- hand-written CDNA4 (gfx950) kernels + the C ABI of liboc_hip.so.
//
// One lane = one environment.  The whole dynamic state of an env (A + M + 2 packed
// int32 words, include/oc_hip.h) lives in VGPRs for the duration of a step; the
// level's static tables (cell grid, path-distance table, subtask descriptors) are
// staged once per workgroup into LDS.  All global tensors are env-major SoA, so each
// wave load/store touches 256 contiguous bytes.  Pure integer / indexing work plus a
// handful of fp64 divisions and adds for reward shaping: no MFMA.
//
// Semantics follow the reference line by line (cited below, paths relative to the
// reference root) but the data model is our own: objects are not heap nodes in a
// dict of lists, they are equivalence classes over M base items, each item carrying
// {cell, chopped, group, holder, world-order rank} in one register.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "../../include/oc_hip.h"

namespace {

// ---------------------------------------------------------------------------
// device-resident level image (copied verbatim into LDS by every workgroup)
// ---------------------------------------------------------------------------
struct DevLevel {
  int32_t W, H, A, M, S, T, max_path, allergic, npair, ndeliv, ncells, image_words;
  int32_t init_words[OC_MAX_AGENTS + OC_MAX_ITEMS + 2];
  uint16_t sub_sig[OC_MAX_SUBTASKS];
  uint8_t sub_kind[OC_MAX_SUBTASKS];
  int8_t sub_food[OC_MAX_SUBTASKS];
  uint8_t cells[OC_MAX_CELLS];
  uint8_t item_type[OC_MAX_ITEMS];
  uint8_t pair_type[OC_MAX_PAIR];
  uint8_t deliv_x[OC_MAX_DELIV], deliv_y[OC_MAX_DELIV];
  // followed by ncells*ncells bytes: dist[a*ncells + b]
};
static_assert(sizeof(DevLevel) % 4 == 0, "DevLevel must be word sized");

}  // namespace

struct oc_level {
  DevLevel host;
  void *dev;          // device copy of DevLevel + dist
  size_t image_bytes; // multiple of 4
  int device;
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
template <int A, int M>
struct Env {
  int ax[A], ay[A], ah[A];                               // ah: held group, -1 none
  int ix[M], iy[M], ist[M], ig[M], iho[M], isq[M];       // iho: holder agent, -1 none
  int t, completed, goalcnt, mctr, err;
};

template <int A, int M>
__device__ __forceinline__ void unpack(Env<A, M> &e, const int32_t *w) {
#pragma unroll
  for (int a = 0; a < A; a++) {
    e.ax[a] = w[a] & 15;
    e.ay[a] = (w[a] >> 4) & 15;
    e.ah[a] = ((w[a] >> 8) & 15) - 1;
  }
#pragma unroll
  for (int i = 0; i < M; i++) {
    int v = w[A + i];
    e.ix[i] = v & 15;
    e.iy[i] = (v >> 4) & 15;
    e.ist[i] = (v >> 8) & 1;
    e.ig[i] = (v >> 9) & 7;
    e.iho[i] = ((v >> 12) & 7) - 1;
    e.isq[i] = (v >> 16) & 255;
  }
  e.t = w[A + M] & 0xFFFF;
  e.completed = (w[A + M] >> 16) & 0xFFFF;
  e.goalcnt = w[A + M + 1] & 0xFFFF;
  e.mctr = (w[A + M + 1] >> 16) & 255;
  e.err = (w[A + M + 1] >> 24) & 255;
}

template <int A, int M>
__device__ __forceinline__ void pack(const Env<A, M> &e, int32_t *w) {
#pragma unroll
  for (int a = 0; a < A; a++) w[a] = e.ax[a] | (e.ay[a] << 4) | ((e.ah[a] + 1) << 8);
#pragma unroll
  for (int i = 0; i < M; i++)
    w[A + i] = e.ix[i] | (e.iy[i] << 4) | (e.ist[i] << 8) | (e.ig[i] << 9) | ((e.iho[i] + 1) << 12) |
               (e.isq[i] << 16);
  w[A + M] = e.t | (e.completed << 16);
  w[A + M + 1] = e.goalcnt | (e.mctr << 16) | (e.err << 24);
}

// LDS view of the level
struct Lv {
  const DevLevel *h;
  const uint8_t *dist;
  int W, H, S, T, maxp, ncells;
  __device__ __forceinline__ int cell(int x, int y) const { return h->cells[y * W + x]; }
  __device__ __forceinline__ int D(int ax, int ay, int bx, int by) const {
    return dist[(ay * W + ax) * ncells + (by * W + bx)];
  }
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Stage the level image into LDS (one copy per workgroup) and build the view.
__device__ __forceinline__ Lv stage_level(const uint32_t *__restrict__ img, uint32_t *lds) {
  const int words = uni((int)((const DevLevel *)img)->image_words);
  for (int w = threadIdx.x; w < words; w += blockDim.x) lds[w] = img[w];
  __syncthreads();
  Lv lv;
  lv.h = (const DevLevel *)lds;
  lv.dist = (const uint8_t *)lds + sizeof(DevLevel);
  lv.W = uni(lv.h->W);
  lv.H = uni(lv.h->H);
  lv.S = uni(lv.h->S);
  lv.T = uni(lv.h->T);
  lv.maxp = uni(lv.h->max_path);
  lv.ncells = uni(lv.h->ncells);
  return lv;
}

__device__ __forceinline__ int iabs(int v) { return v < 0 ? -v : v; }

// ---------------------------------------------------------------------------
// one environment tick
// ---------------------------------------------------------------------------
// Per-object aggregates, indexed by item: signature (content-type counts, nibbles)
// and "every food chopped" of the Object the item belongs to.
template <int M>
struct Agg {
  int sig[M];
  int chopped[M];
};

template <int A, int M>
__device__ __forceinline__ void aggregate(const Env<A, M> &e, const int (&type)[M], Agg<M> &g) {
#pragma unroll
  for (int i = 0; i < M; i++) {
    int s = 0, c = 1;
#pragma unroll
    for (int j = 0; j < M; j++) {
      const bool same = e.ig[j] == e.ig[i];
      s += same ? (1 << (4 * type[j])) : 0;
      c &= (same && type[j] != OC_PLATE) ? e.ist[j] : 1;
    }
    g.sig[i] = s;
    g.chopped[i] = c;
  }
}

// reward_shaping for sim agents 0 and 1 together
// (gym_cooking/envs/overcooked_environment.py:272-397).  Divisions are int/int in
// Python = one correctly rounded fp64 division each; sums run left to right.
template <int A, int M>
__device__ __forceinline__ void shaping2(const Lv &lv, const Env<A, M> &e, const int (&type)[M],
                                         const Agg<M> &g, double &s0, double &s1) {
  const int MAXP = lv.maxp;
  const double fmax = (double)MAXP;
  const int npair = uni(lv.h->npair);
  const int ndeliv = uni(lv.h->ndeliv);
  constexpr int B = A < 2 ? A : 2;
  double tot[2] = {0.0, 0.0};

  // Chop term (:278-304): incomplete Chop(X) -> distance from the agent to the fresh X
  int nchop = 0;
  int mind[2] = {1 << 20, 1 << 20};
  for (int s = 0; s < lv.S; s++) {
    if (lv.h->sub_kind[s] != OC_CHOP || ((e.completed >> s) & 1)) continue;
    const int food = lv.h->sub_food[s];
    int fx = 0, fy = 0;
#pragma unroll
    for (int i = 0; i < M; i++)
      if (type[i] == food) {
        fx = e.ix[i];
        fy = e.iy[i];
      }
#pragma unroll
    for (int b = 0; b < B; b++) mind[b] = min(mind[b], lv.D(e.ax[b], e.ay[b], fx, fy));
    nchop++;
  }
  const bool zero = nchop == 0;   // Python `total_penalty == 0` after the Chop term
  if (nchop > 0) {
#pragma unroll
    for (int b = 0; b < B; b++)
      tot[b] += (double)((mind[b] + MAXP) + (nchop - 1) * 2 * MAXP) / fmax;
  }

  // pair term (:319-363): agent independent
  int npairs = 0, minpair = 1 << 20;
  for (int p = 0; p < npair; p++)
    for (int q = p + 1; q < npair; q++) {
      const int tp = lv.h->pair_type[p], tq = lv.h->pair_type[q];
      int m = MAXP;
      bool hasp = false, hasq = false;
#pragma unroll
      for (int i = 0; i < M; i++) {
        hasp |= type[i] == tp;
        hasq |= type[i] == tq;
        if (type[i] != tp) continue;
#pragma unroll
        for (int j = 0; j < M; j++)
          if (type[j] == tq) m = min(m, lv.D(e.ix[i], e.iy[i], e.ix[j], e.iy[j]));
      }
      int val;
      if (hasp && hasq) {
        if (m == 0) continue;
        val = m;
      } else {
        val = MAXP;
      }
      minpair = min(minpair, val);
      npairs++;
    }
  if (npairs > 0) {
    const double add = zero ? (double)(minpair + (npairs - 1) * MAXP) / fmax
                            : (double)(npairs * MAXP) / fmax;
#pragma unroll
    for (int b = 0; b < B; b++) tot[b] += add;
  }

  // Deliver term (:370-395)
  for (int s = 0; s < lv.S; s++) {
    if (lv.h->sub_kind[s] != OC_DELIVER || ((e.completed >> s) & 1)) continue;
    const int sig = lv.h->sub_sig[s];
    bool match = false;
    int mx = 0, my = 0;
#pragma unroll
    for (int i = 0; i < M; i++)
      if (e.ig[i] == i && g.sig[i] == sig && g.chopped[i]) {
        match = true;
        mx = e.ix[i];
        my = e.iy[i];
      }
#pragma unroll
    for (int b = 0; b < B; b++) {
      if (!match) {
        tot[b] += 2.0;
      } else {
        const int d = lv.D(e.ax[b], e.ay[b], mx, my) + iabs(e.ax[b] - mx) + iabs(e.ay[b] - my);
        if (d == 0) {
          int best = 1 << 20;
          for (int k = 0; k < ndeliv; k++) {
            const int dx = lv.h->deliv_x[k], dy = lv.h->deliv_y[k];
            best = min(best, lv.D(e.ax[b], e.ay[b], dx, dy) + iabs(e.ax[b] - dx) + iabs(e.ay[b] - dy));
          }
          tot[b] += (double)best / fmax;
        } else {
          tot[b] += (double)d / fmax + 1.0;
        }
      }
    }
  }
  s0 = tot[0];
  s1 = B > 1 ? tot[1] : 0.0;
}

template <int A, int M>
__device__ __forceinline__ void env_step(const Lv &lv, Env<A, M> &e, const int (&type)[M],
                                         const int (&act_in)[A], int &reward, int &done,
                                         int &success, double &s0, double &s1) {
  const int W = lv.W, H = lv.H;
  e.t += 1;                                            // overcooked_environment.py:213

  // ---- check_collisions (:578-613) on the ORIGINAL actions -------------------
  int act[A], dx[A], dy[A], nx[A], ny[A];
#pragma unroll
  for (int a = 0; a < A; a++) {
    int c = act_in[a];
    c = (c < 0 || c > 4) ? OC_ACT_NOOP : c;
    act[a] = c;
    dx[a] = (c == OC_ACT_RIGHT) - (c == OC_ACT_LEFT);
    dy[a] = (c == OC_ACT_DOWN) - (c == OC_ACT_UP);
    const int px = e.ax[a] + dx[a], py = e.ay[a] + dy[a];
    const bool inb = (unsigned)px < (unsigned)W && (unsigned)py < (unsigned)H;
    if (!inb && A > 1) e.err |= OC_ERR_OOB;            // get_gridsquare_at asserts (world.py:310-315)
    const bool blocked = !inb || lv.cell(inb ? px : e.ax[a], inb ? py : e.ay[a]) != OC_FLOOR;
    nx[a] = blocked ? e.ax[a] : px;                    // :551-559
    ny[a] = blocked ? e.ay[a] : py;
  }
  bool ex[A];
#pragma unroll
  for (int a = 0; a < A; a++) ex[a] = true;
#pragma unroll
  for (int i = 0; i < A; i++)
#pragma unroll
    for (int j = i + 1; j < A; j++) {
      if (nx[i] == nx[j] && ny[i] == ny[j]) {          // :562-569
        const bool i_stays = nx[i] == e.ax[i] && ny[i] == e.ay[i] && act[i] != OC_ACT_NOOP;
        const bool j_stays = nx[j] == e.ax[j] && ny[j] == e.ay[j] && act[j] != OC_ACT_NOOP;
        if (i_stays) {
          ex[j] = false;
        } else if (j_stays) {
          ex[i] = false;
        } else {
          ex[i] = false;
          ex[j] = false;
        }
      } else if (e.ax[i] == nx[j] && e.ay[i] == ny[j] && e.ax[j] == nx[i] && e.ay[j] == ny[i]) {
        ex[i] = false;                                 // swap (:572-575)
        ex[j] = false;
      }
    }

  // ---- execute_navigation (:615-618): interact(), sequential in agent order ---
  const int allergic = uni(lv.h->allergic);
#pragma unroll
  for (int a = 0; a < A; a++) {
    if (!ex[a] || act[a] == OC_ACT_NOOP) continue;     // blocked -> (0,0) (:610-612); interact.py:12
    const int tx = min(max(e.ax[a] + dx[a], 0), W - 1);  // world.inbounds (world.py:317-320)
    const int ty = min(max(e.ay[a] + dy[a], 0), H - 1);
    const int c = lv.cell(tx, ty);
    if (c == OC_FLOOR) {                               // interact.py:19-20, agent.py:311-314
      e.ax[a] = tx;
      e.ay[a] = ty;
#pragma unroll
      for (int i = 0; i < M; i++)
        if (e.iho[i] == a) {
          e.ix[i] = tx;
          e.iy[i] = ty;
        }
    } else if (e.ah[a] >= 0) {                         // holding (:23)
      const int g = e.ah[a];
      int n = 0, plates = 0, chopped = 1, lone_fresh_food = 0;
#pragma unroll
      for (int i = 0; i < M; i++)
        if (e.ig[i] == g) {
          n++;
          if (type[i] == OC_PLATE) {
            plates++;
          } else {
            chopped &= e.ist[i];
            lone_fresh_food = !e.ist[i];
          }
        }
      if (c == OC_DELIVERY) {                          // :25-30, is_deliverable core.py:232-237
        if (n > 1 && chopped) {
#pragma unroll
          for (int i = 0; i < M; i++)
            if (e.ig[i] == g) {
              e.ix[i] = tx;
              e.iy[i] = ty;
              e.iho[i] = -1;
            }
          e.ah[a] = -1;
        }
      } else {
        int og = -1;                                   // unheld Object on the target cell
#pragma unroll
        for (int i = 0; i < M; i++)
          if (og < 0 && e.iho[i] < 0 && e.ix[i] == tx && e.iy[i] == ty) og = e.ig[i];
        if (og >= 0) {                                 // :33-46
#pragma unroll
          for (int i = 0; i < M; i++)
            if (e.ig[i] == og) {
              if (type[i] == OC_PLATE)
                plates++;
              else
                chopped &= e.ist[i];
            }
          if (plates <= 1 && chopped) {                // mergeable (core.py:240-257)
            if (A > 2) {
              // World.remove(agent.holding) deletes by (name, location), last match
              // (world.py:239-247): with a second agent on the same cell holding a
              // same-named Object that sits later in world order it removes the wrong
              // one and the reference's store is corrupt from here on.  Flag it.
              int my_sig = 0, my_seq = 0;
#pragma unroll
              for (int i = 0; i < M; i++)
                if (e.ig[i] == g) {
                  my_sig += 1 << (4 * type[i]);
                  my_seq = e.isq[i];
                }
#pragma unroll
              for (int j = 0; j < M; j++) {
                if (e.ig[j] == g || e.iho[j] < 0 || e.ig[j] != j) continue;
                if (e.ix[j] != e.ax[a] || e.iy[j] != e.ay[a]) continue;
                int sj = 0;
#pragma unroll
                for (int k = 0; k < M; k++)
                  if (e.ig[k] == j) sj += 1 << (4 * type[k]);
                if (sj == my_sig && e.isq[j] > my_seq) e.err |= OC_ERR_ALIAS;
              }
            }
            const int newg = min(g, og);
            const int seq = M + e.mctr;                // re-inserted under a new name: last in world order
            e.mctr += 1;
#pragma unroll
            for (int i = 0; i < M; i++)
              if (e.ig[i] == g || e.ig[i] == og) {
                e.ig[i] = newg;
                e.ix[i] = e.ax[a];
                e.iy[i] = e.ay[a];
                e.iho[i] = a;
                e.isq[i] = seq;
              }
            e.ah[a] = newg;
          }
        } else if (c == OC_CUTBOARD && n == 1 && lone_fresh_food) {   // :52-54 chop in hand
#pragma unroll
          for (int i = 0; i < M; i++)
            if (e.ig[i] == g) e.ist[i] = 1;
        } else {                                       // :56-57 put down
#pragma unroll
          for (int i = 0; i < M; i++)
            if (e.ig[i] == g) {
              e.ix[i] = tx;
              e.iy[i] = ty;
              e.iho[i] = -1;
            }
          e.ah[a] = -1;
        }
      }
    } else if (c != OC_DELIVERY) {                     // empty hands (:62-71)
      int og = -1;
#pragma unroll
      for (int i = 0; i < M; i++)
        if (og < 0 && e.iho[i] < 0 && e.ix[i] == tx && e.iy[i] == ty) og = e.ig[i];
      if (og >= 0 && !((allergic >> a) & 1)) {         // ALLERGIC: acquire is a no-op (agent.py:296-298)
#pragma unroll
        for (int i = 0; i < M; i++)
          if (e.ig[i] == og) {
            e.iho[i] = a;
            e.ix[i] = e.ax[a];
            e.iy[i] = e.ay[a];
          }
        e.ah[a] = og;
      }
    }
  }

  // ---- done (:243-270) and reward (:399-432) ---------------------------------
  Agg<M> g;
  aggregate<A, M>(e, type, g);
  const int d0x = uni(lv.h->deliv_x[0]), d0y = uni(lv.h->deliv_y[0]);  // first Delivery tile only (:259,:402)
  bool all_delivered = true;
  int r = 0;
  for (int s = 0; s < lv.S; s++) {
    const int kind = lv.h->sub_kind[s];
    const int sig = lv.h->sub_sig[s];
    bool match = false, on_delivery = false;
#pragma unroll
    for (int i = 0; i < M; i++)
      if (e.ig[i] == i && g.sig[i] == sig && g.chopped[i]) {
        match = true;
        on_delivery |= e.ix[i] == d0x && e.iy[i] == d0y;
      }
    if (kind == OC_DELIVER) {
      all_delivered &= on_delivery;
      if (on_delivery) {
        r += 3;
        e.completed |= 1 << s;
      }
    } else {
      const int cnt = match ? 1 : 0;                   // #distinct cells holding the goal object
      if (cnt > ((e.goalcnt >> s) & 1)) {
        r += 1;
        e.completed |= 1 << s;
      }
      e.goalcnt = (e.goalcnt & ~(1 << s)) | (cnt << s);
    }
  }
  const bool timeout = lv.T != 0 && e.t >= lv.T;       // checked first (:245-249)
  done = (timeout || all_delivered) ? 1 : 0;
  success = (!timeout && all_delivered) ? 1 : 0;
  reward = r;
  shaping2<A, M>(lv, e, type, g, s0, s1);              // after reward(): uses the updated completed flags
}

// get_observation2 (gym_comm/envs/overcooked_env.py:105-159) for one viewer;
// writes F = 22 + S + 2C rows with stride n.
template <int A, int M>
__device__ __forceinline__ void env_obs(const Lv &lv, const Env<A, M> &e, const int (&type)[M],
                                        int viewer, int radius, bool viewer_blind, bool ego_blind,
                                        int C, int comm0, int comm1, int32_t *__restrict__ out,
                                        int64_t n) {
  const int vx = viewer == 0 ? e.ax[0] : e.ax[1];
  const int vy = viewer == 0 ? e.ay[0] : e.ay[1];
  const int vh = viewer == 0 ? e.ah[0] : e.ah[1];
  int ddx[4], ddy[4], st[4], hid[4];
#pragma unroll
  for (int ch = 0; ch < 4; ch++) {
    // last writer in world.objects order wins (:121-131): the item of this type
    // whose Object has the highest rank
    int best = -1, bx = 0, by = 0, bs = 0;
#pragma unroll
    for (int i = 0; i < M; i++)
      if (type[i] == ch && e.isq[i] > best) {
        best = e.isq[i];
        bx = e.ix[i] - vx;
        by = e.iy[i] - vy;
        bs = ch == OC_PLATE ? 0 : e.ist[i];
      }
    const bool have = best >= 0 && !viewer_blind;
    ddx[ch] = have ? bx : 0;
    ddy[ch] = have ? by : 0;
    st[ch] = have ? bs : 0;
    const bool within = iabs(ddx[ch]) + iabs(ddy[ch]) <= radius;
    hid[ch] = viewer_blind ? 1 : (within ? 0 : 1);     // :109,:133
    if (within) {                                      // :135 (sic: zeroed when visible)
      ddx[ch] = 0;
      ddy[ch] = 0;
    }
  }
  int row = 0;
#pragma unroll
  for (int ch = 0; ch < 4; ch++) out[(row++) * n] = ddx[ch];
#pragma unroll
  for (int ch = 0; ch < 4; ch++) out[(row++) * n] = ddy[ch];
#pragma unroll
  for (int ch = 0; ch < 4; ch++) out[(row++) * n] = st[ch];
#pragma unroll
  for (int ch = 0; ch < 4; ch++) out[(row++) * n] = hid[ch];
  for (int s = 0; s < lv.S; s++) out[(row++) * n] = (e.completed >> s) & 1;
  out[(row++) * n] = viewer_blind ? 0 : e.ax[0];       // :139-143
  out[(row++) * n] = viewer_blind ? 0 : e.ay[0];
  out[(row++) * n] = viewer_blind ? 0 : e.ax[1];
  out[(row++) * n] = viewer_blind ? 0 : e.ay[1];
  out[(row++) * n] = ego_blind ? 0 : (vh >= 0 ? 1 : 0);  // :154, gated on the EGO's BLIND flag
  out[(row++) * n] = 0;
  for (int c = 0; c < C; c++) out[(row++) * n] = comm0 == c ? 1 : 0;
  for (int c = 0; c < C; c++) out[(row++) * n] = comm1 == c ? 1 : 0;
}

// wave-level metric accumulation: ballot/popcount for the flags, a butterfly sum for
// the integers, then one atomic per wave.
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ void accumulate_metrics(int64_t *metrics, bool valid, int done, int success,
                                                   int reward, int completed_bits, bool err) {
  if (metrics == nullptr) return;
  const unsigned long long vmask = __ballot(valid);
  const unsigned long long dmask = __ballot(valid && done);
  const unsigned long long smask = __ballot(valid && success);
  const unsigned long long emask = __ballot(valid && err);
  const int rsum = wave_sum(valid ? reward : 0);
  const int csum = wave_sum((valid && done) ? __popc(completed_bits) : 0);
  if ((threadIdx.x & 63) == 0) {
    unsigned long long *m = (unsigned long long *)metrics;
    if (vmask) atomicAdd(&m[OC_MET_ENV_STEPS], (unsigned long long)__popcll(vmask));
    if (dmask) atomicAdd(&m[OC_MET_EPISODES], (unsigned long long)__popcll(dmask));
    if (smask) atomicAdd(&m[OC_MET_SUCCESSES], (unsigned long long)__popcll(smask));
    if (rsum) atomicAdd(&m[OC_MET_REWARD_SUM], (unsigned long long)(long long)rsum);
    if (csum) atomicAdd(&m[OC_MET_COMPLETED_SUM], (unsigned long long)(long long)csum);
    if (emask) atomicAdd(&m[OC_MET_ERRORS], (unsigned long long)__popcll(emask));
  }
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------
template <int M>
__device__ __forceinline__ void load_types(const Lv &lv, int (&type)[M]) {
#pragma unroll
  for (int i = 0; i < M; i++) type[i] = uni(lv.h->item_type[i]);
}

struct StepArgs {
  const uint32_t *level;
  int32_t *state;
  const int32_t *actions;
  int32_t *reward;
  int32_t *done;
  double *shaping;
  int64_t *metrics;
  int64_t n;
  int32_t auto_reset;
};

template <int A, int M>
__global__ void __launch_bounds__(256) k_step(StepArgs p) {
  extern __shared__ uint32_t lds[];
  const Lv lv = stage_level(p.level, lds);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < p.n;
  int reward = 0, done = 0, success = 0, comp = 0;
  bool err = false;
  if (valid) {
    constexpr int WS = A + M + 2;
    int type[M];
    load_types<M>(lv, type);
    int32_t w[WS];
#pragma unroll
    for (int r = 0; r < WS; r++) w[r] = p.state[(int64_t)r * p.n + i];
    Env<A, M> e;
    unpack<A, M>(e, w);
    int act[A];
#pragma unroll
    for (int a = 0; a < A; a++) act[a] = p.actions[(int64_t)a * p.n + i];
    const int err_before = e.err;
    double s0, s1;
    env_step<A, M>(lv, e, type, act, reward, done, success, s0, s1);
    comp = e.completed;
    err = e.err != err_before;
    p.reward[i] = reward;
    p.done[i] = done;
    p.shaping[i] = s0;
    p.shaping[p.n + i] = s1;
    if (done && p.auto_reset) {
#pragma unroll
      for (int r = 0; r < WS; r++) w[r] = lv.h->init_words[r];
    } else {
      pack<A, M>(e, w);
    }
#pragma unroll
    for (int r = 0; r < WS; r++) p.state[(int64_t)r * p.n + i] = w[r];
  }
  accumulate_metrics(p.metrics, valid, done, success, reward, comp, err);
}

struct ObsArgs {
  const uint32_t *level;
  const int32_t *state;
  const int32_t *comm;
  int32_t *obs;
  double *timestep;
  int64_t n;
  oc_obs_cfg cfg;
};

template <int A, int M>
__global__ void __launch_bounds__(256) k_obs(ObsArgs p) {
  extern __shared__ uint32_t lds[];
  const Lv lv = stage_level(p.level, lds);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  constexpr int WS = A + M + 2;
  int type[M];
  load_types<M>(lv, type);
  int32_t w[WS];
#pragma unroll
  for (int r = 0; r < WS; r++) w[r] = p.state[(int64_t)r * p.n + i];
  Env<A, M> e;
  unpack<A, M>(e, w);
  const int C = p.cfg.num_comm;
  const int F = 22 + lv.S + 2 * C;
  const int c0 = p.comm[i], c1 = p.comm[p.n + i];
  const bool ego_blind = p.cfg.blind_mask & 1;
#pragma unroll
  for (int v = 0; v < 2; v++)
    env_obs<A, M>(lv, e, type, v, p.cfg.fow_radius, (p.cfg.blind_mask >> v) & 1, ego_blind, C, c0, c1,
                  p.obs + (int64_t)v * F * p.n + i, p.n);
  p.timestep[i] = (double)e.t / (double)lv.T;          // overcooked_env.py:146
}

struct ResetArgs {
  const uint32_t *level;
  int32_t *state;
  const int32_t *mask;
  int64_t n;
};

__global__ void __launch_bounds__(256) k_reset(ResetArgs p) {
  const DevLevel *h = (const DevLevel *)p.level;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  if (p.mask != nullptr && p.mask[i] == 0) return;
  const int ws = h->A + h->M + 2;
  for (int r = 0; r < ws; r++) p.state[(int64_t)r * p.n + i] = h->init_words[r];
}

struct MultiArgs {
  const uint32_t *level;
  int32_t *state;
  int32_t *comm;
  const int32_t *actions;
  int32_t *obs;
  double *timestep;
  double *reward;
  int32_t *done;
  int32_t *sparse;
  int64_t *metrics;
  int64_t n;
  int32_t auto_reset;
  oc_wrap_cfg cfg;
};

// OvercookedMultiEnv.multi_step (gym_comm/envs/overcooked_env.py:207-282), 2 agents.
template <int M>
__global__ void __launch_bounds__(256) k_multi_step(MultiArgs p) {
  constexpr int A = 2;
  extern __shared__ uint32_t lds[];
  const Lv lv = stage_level(p.level, lds);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < p.n;
  int reward = 0, done = 0, success = 0, comp = 0;
  bool err = false;
  if (valid) {
    constexpr int WS = A + M + 2;
    int type[M];
    load_types<M>(lv, type);
    int32_t w[WS];
#pragma unroll
    for (int r = 0; r < WS; r++) w[r] = p.state[(int64_t)r * p.n + i];
    Env<A, M> e;
    unpack<A, M>(e, w);
    const int ego_mv = p.actions[i], ego_cm = p.actions[p.n + i];
    const int alt_mv = p.actions[2 * p.n + i], alt_cm = p.actions[3 * p.n + i];
    // comm one-hots (:227-246)
    const int c0 = p.cfg.communication_on ? ego_cm : -1;
    const int c1 = (p.cfg.communication_on && !p.cfg.ego_led) ? alt_cm : -1;
    p.comm[i] = c0;
    p.comm[p.n + i] = c1;
    // NAV_ACTIONS lookup + CAN_MOVE gating + ego_agent_idx (:248-262)
    const int em = (p.cfg.can_move_mask & 1) ? (ego_mv & 3) : OC_ACT_NOOP;
    const int am = (p.cfg.can_move_mask & 2) ? (alt_mv & 3) : OC_ACT_NOOP;
    int act[A];
    act[0] = p.cfg.ego_agent_idx == 0 ? em : am;
    act[1] = p.cfg.ego_agent_idx == 0 ? am : em;
    const int err_before = e.err;
    double s0, s1;
    env_step<A, M>(lv, e, type, act, reward, done, success, s0, s1);
    comp = e.completed;
    err = e.err != err_before;
    p.reward[i] = ((double)reward - s0) - s1;          // :282
    p.done[i] = done;
    if (p.sparse != nullptr) p.sparse[i] = reward;
    if (done && p.auto_reset) {
#pragma unroll
      for (int r = 0; r < WS; r++) w[r] = lv.h->init_words[r];
      unpack<A, M>(e, w);
    } else {
      pack<A, M>(e, w);
    }
#pragma unroll
    for (int r = 0; r < WS; r++) p.state[(int64_t)r * p.n + i] = w[r];
    const int C = p.cfg.obs.num_comm;
    const int F = 22 + lv.S + 2 * C;
    const bool ego_blind = p.cfg.obs.blind_mask & 1;
#pragma unroll
    for (int v = 0; v < 2; v++)
      env_obs<A, M>(lv, e, type, v, p.cfg.obs.fow_radius, (p.cfg.obs.blind_mask >> v) & 1, ego_blind, C,
                    c0, c1, p.obs + (int64_t)v * F * p.n + i, p.n);
    p.timestep[i] = (double)e.t / (double)lv.T;
  }
  accumulate_metrics(p.metrics, valid, done, success, reward, comp, err);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
int block_size_for(int64_t n) {
  // small batches: spread over more CUs (one wave per workgroup); large batches:
  // amortise the per-workgroup level staging over four waves
  return n >= 256 * 256 ? 256 : 64;
}

template <typename Args, typename K>
int launch(K kernel, const oc_level *lv, const Args &args, int64_t n, void *stream) {
  if (n == 0) return OC_OK;
  const int bs = block_size_for(n);
  const int64_t grid = (n + bs - 1) / bs;
  if (grid > 0x7FFFFFFF) return fail(OC_E_BADARG, "n too large");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(bs), lv->image_bytes, (hipStream_t)stream, args);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "kernel launch");
  return OC_OK;
}

#define OC_DISPATCH_AM(KERNEL, A_, M_, ...)                                      \
  do {                                                                           \
    if (A_ == 2 && M_ == 3) return launch(KERNEL<2, 3>, __VA_ARGS__);            \
    if (A_ == 2 && M_ == 4) return launch(KERNEL<2, 4>, __VA_ARGS__);            \
    if (A_ == 3 && M_ == 3) return launch(KERNEL<3, 3>, __VA_ARGS__);            \
    if (A_ == 3 && M_ == 4) return launch(KERNEL<3, 4>, __VA_ARGS__);            \
    if (A_ == 4 && M_ == 3) return launch(KERNEL<4, 3>, __VA_ARGS__);            \
    if (A_ == 4 && M_ == 4) return launch(KERNEL<4, 4>, __VA_ARGS__);            \
    return fail(OC_E_BADARG, "unsupported (num_agents, num_items): need A in 2..4, M in 3..4"); \
  } while (0)

}  // namespace

extern "C" {

int oc_abi_version(void) { return OC_ABI_VERSION; }
const char *oc_last_error(void) { return g_err; }

int oc_level_create(const int32_t *b, int32_t n_words, oc_level_t **out) {
  if (!b || !out || n_words < OC_LV_HEADER_WORDS) return fail(OC_E_BADARG, "oc_level_create: null or short blob");
  if (b[OC_LV_MAGIC] != OC_LV_MAGIC_VALUE || b[OC_LV_VERSION] != OC_LV_VERSION_VALUE ||
      b[OC_LV_TOTAL] != n_words)
    return fail(OC_E_BADARG, "oc_level_create: bad magic/version/length");
  const int W = b[OC_LV_W], H = b[OC_LV_H], A = b[OC_LV_A], M = b[OC_LV_M], S = b[OC_LV_S];
  const int nc = W * H;
  if (W < 1 || H < 1 || W > 16 || H > 16 || nc > OC_MAX_CELLS || A < 2 || A > OC_MAX_AGENTS || M < 1 ||
      M > OC_MAX_ITEMS || S < 1 || S > OC_MAX_SUBTASKS || b[OC_LV_NPAIR] > OC_MAX_PAIR ||
      b[OC_LV_NDELIV] < 1 || b[OC_LV_NDELIV] > OC_MAX_DELIV || b[OC_LV_MAX_PATH] > 255 ||
      b[OC_LV_T] < 0 || b[OC_LV_T] > 0xFFFF)
    return fail(OC_E_BADARG, "oc_level_create: level dimensions out of range");
  oc_level *lv = new (std::nothrow) oc_level();
  if (!lv) return fail(OC_E_BADARG, "oc_level_create: out of memory");
  DevLevel &h = lv->host;
  memset(&h, 0, sizeof(h));
  h.W = W; h.H = H; h.A = A; h.M = M; h.S = S; h.T = b[OC_LV_T];
  h.max_path = b[OC_LV_MAX_PATH]; h.allergic = b[OC_LV_ALLERGIC];
  h.npair = b[OC_LV_NPAIR]; h.ndeliv = b[OC_LV_NDELIV]; h.ncells = nc;
  const int32_t *cells = b + b[OC_LV_OFF_CELLS], *dist = b + b[OC_LV_OFF_DIST];
  const int32_t *ag = b + b[OC_LV_OFF_AGENTS], *it = b + b[OC_LV_OFF_ITEMS];
  const int32_t *st = b + b[OC_LV_OFF_SUBTASKS], *pr = b + b[OC_LV_OFF_PAIR], *dl = b + b[OC_LV_OFF_DELIV];
  for (int i = 0; i < nc; i++) h.cells[i] = (uint8_t)cells[i];
  int food_seen[OC_NTYPES] = {0, 0, 0, 0};
  for (int i = 0; i < M; i++) {
    const int t = it[3 * i];
    if (t < 0 || t >= OC_NTYPES) { delete lv; return fail(OC_E_BADARG, "oc_level_create: bad item type"); }
    if (t != OC_PLATE && food_seen[t]++) {
      delete lv;
      return fail(OC_E_BADARG, "oc_level_create: a food type occurs twice (unsupported by the HIP path)");
    }
    h.item_type[i] = (uint8_t)t;
  }
  for (int s = 0; s < S; s++) {
    h.sub_kind[s] = (uint8_t)st[4 * s];
    h.sub_sig[s] = (uint16_t)st[4 * s + 1];
    h.sub_food[s] = (int8_t)st[4 * s + 2];
  }
  for (int p = 0; p < h.npair; p++) h.pair_type[p] = (uint8_t)pr[p];
  for (int k = 0; k < h.ndeliv; k++) { h.deliv_x[k] = (uint8_t)dl[2 * k]; h.deliv_y[k] = (uint8_t)dl[2 * k + 1]; }
  // initial state words: OvercookedEnvironment.reset() (overcooked_environment.py:180-206)
  for (int a = 0; a < A; a++) h.init_words[a] = ag[2 * a] | (ag[2 * a + 1] << 4);
  for (int i = 0; i < M; i++) h.init_words[A + i] = it[3 * i + 1] | (it[3 * i + 2] << 4) | (i << 9) | (i << 16);
  h.init_words[A + M] = 0;
  h.init_words[A + M + 1] = 0;
  const size_t bytes = (sizeof(DevLevel) + (size_t)nc * nc + 3) & ~(size_t)3;
  h.image_words = (int32_t)(bytes / 4);
  lv->image_bytes = bytes;
  uint8_t *img = new (std::nothrow) uint8_t[bytes];
  if (!img) { delete lv; return fail(OC_E_BADARG, "oc_level_create: out of memory"); }
  memset(img, 0, bytes);
  memcpy(img, &h, sizeof(DevLevel));
  for (int i = 0; i < nc * nc; i++) img[sizeof(DevLevel) + i] = (uint8_t)dist[i];
  hipError_t e = hipGetDevice(&lv->device);
  if (e == hipSuccess) e = hipMalloc(&lv->dev, bytes);
  if (e == hipSuccess) e = hipMemcpy(lv->dev, img, bytes, hipMemcpyHostToDevice);
  delete[] img;
  if (e != hipSuccess) {
    delete lv;
    fail_hip(e, "oc_level_create");
    return OC_E_NODEVICE;
  }
  *out = lv;
  return OC_OK;
}

int oc_level_destroy(oc_level_t *lv) {
  if (!lv) return OC_OK;
  if (lv->dev) (void)hipFree(lv->dev);
  delete lv;
  return OC_OK;
}

int32_t oc_state_words(const oc_level_t *lv) { return lv ? lv->host.A + lv->host.M + 2 : 0; }
int32_t oc_obs_rows(const oc_level_t *lv, int32_t num_comm) {
  return lv ? 22 + lv->host.S + 2 * num_comm : 0;
}

int oc_reset(const oc_level_t *lv, int32_t *state, const int32_t *mask, int64_t n, void *stream) {
  if (!lv || !state || n < 0) return fail(OC_E_BADARG, "oc_reset: bad argument");
  if (n == 0) return OC_OK;
  ResetArgs a{(const uint32_t *)lv->dev, state, mask, n};
  const int bs = 256;
  hipLaunchKernelGGL(k_reset, dim3((unsigned)((n + bs - 1) / bs)), dim3(bs), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? OC_OK : fail_hip(e, "oc_reset");
}

int oc_step(const oc_level_t *lv, int32_t *state, const int32_t *actions, int32_t *reward, int32_t *done,
            double *shaping, int32_t auto_reset, int64_t *metrics, int64_t n, void *stream) {
  if (!lv || !state || !actions || !reward || !done || !shaping || n < 0)
    return fail(OC_E_BADARG, "oc_step: bad argument");
  StepArgs a{(const uint32_t *)lv->dev, state, actions, reward, done, shaping, metrics, n, auto_reset};
  OC_DISPATCH_AM(k_step, lv->host.A, lv->host.M, lv, a, n, stream);
}

int oc_obs(const oc_level_t *lv, const int32_t *state, const int32_t *comm, const oc_obs_cfg *cfg,
           int32_t *obs, double *timestep, int64_t n, void *stream) {
  if (!lv || !state || !comm || !cfg || !obs || !timestep || n < 0 || cfg->num_comm < 0 || cfg->num_comm > 64)
    return fail(OC_E_BADARG, "oc_obs: bad argument");
  ObsArgs a{(const uint32_t *)lv->dev, state, comm, obs, timestep, n, *cfg};
  OC_DISPATCH_AM(k_obs, lv->host.A, lv->host.M, lv, a, n, stream);
}

int oc_multi_step(const oc_level_t *lv, int32_t *state, int32_t *comm, const int32_t *actions,
                  const oc_wrap_cfg *cfg, int32_t *obs, double *timestep, double *reward, int32_t *done,
                  int32_t *sparse, int32_t auto_reset, int64_t *metrics, int64_t n, void *stream) {
  if (!lv || !state || !comm || !actions || !cfg || !obs || !timestep || !reward || !done || n < 0 ||
      cfg->obs.num_comm < 0 || cfg->obs.num_comm > 64)
    return fail(OC_E_BADARG, "oc_multi_step: bad argument");
  if (lv->host.A != 2)
    return fail(OC_E_BADARG, "oc_multi_step: the gym_comm wrapper drives exactly 2 agents");
  MultiArgs a{(const uint32_t *)lv->dev, state, comm, actions, obs, timestep, reward, done, sparse, metrics,
              n, auto_reset, *cfg};
  if (lv->host.M == 3) return launch(k_multi_step<3>, lv, a, n, stream);
  if (lv->host.M == 4) return launch(k_multi_step<4>, lv, a, n, stream);
  return fail(OC_E_BADARG, "oc_multi_step: unsupported number of items");
}

}  // extern "C"

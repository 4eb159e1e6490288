/* oc_oracle.c -- CPU restatement of the reference's Overcooked step/reset/obs path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP path is compared
 * against; it is never linked into, imported by or called from the product
 * package.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it.
 *
 * Parity pin: the reference has no tests or fixtures for this path (SURVEY.md
 * section 4), so this restatement is pinned by golden vectors recorded by running
 * the reference itself in the build container (tests/golden/make_golden.py ->
 * the .npz files under tests/golden; checked by tests/test_oracle_golden.py).
 *
 * It deliberately keeps the reference's data model -- a dict of object lists in
 * insertion order, objects with an ordered contents list -- instead of the packed
 * per-item words the HIP kernels use, so the two are independent derivations of
 * the same semantics.  Every function cites the reference lines it follows
 * (paths relative to the reference root).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/oc_level.h"

#define MAXOBJ 16
#define MAXKEY 32
#define MAXCONT OC_MAX_ITEMS

typedef struct {
  int type;  /* OC_TOMATO.. */
  int state; /* Food.state_index: 0 fresh, 1 chopped; Plate: 0 */
} Item;

/* gym_cooking/utils/core.py:149-237 class Object */
typedef struct {
  int x, y;
  int is_held;
  int n;
  int c[MAXCONT]; /* item ids, in contents-list order */
} Obj;

/* one entry of World.objects (a dict name -> list, utils/world.py:21), movable objects
 * only; GridSquares live in the static cell table */
typedef struct {
  int sig; /* the name: counts of content types, nibble-packed */
  int n;
  int objs[MAXOBJ];
} Key;

typedef struct {
  int kind, sig, food, ncont;
} Sub;

typedef struct {
  /* static */
  int W, H, A, M, S, T, max_path, allergic, npair, ndeliv, play;
  const int32_t *cells, *dist, *agent0, *item0, *pair, *deliv;
  Sub sub[OC_MAX_SUBTASKS];
  int32_t *blob;
  /* dynamic */
  Item items[OC_MAX_ITEMS];
  Obj objs[MAXOBJ];
  int nobjs;
  Key keys[MAXKEY];
  int nkeys;
  int ax[OC_MAX_AGENTS], ay[OC_MAX_AGENTS], ahold[OC_MAX_AGENTS]; /* obj id or -1 */
  int adx[OC_MAX_AGENTS], ady[OC_MAX_AGENTS];
  int t;
  int completed[OC_MAX_SUBTASKS];
  int goalcnt[OC_MAX_SUBTASKS];
  int err;
  int successful;
  /* random-* levels: start cell (x | y<<4) of every item for the NEXT reset, -1 = the
   * blob's default (overcooked_environment.py:157-173 draws them with random.choice) */
  int next_place[OC_MAX_ITEMS];
} Env;

static const int NAV_DX[5] = {0, 0, -1, 1, 0};
static const int NAV_DY[5] = {1, -1, 0, 0, 0};

/* ------------------------------------------------------------------ helpers */
static int cell_at(const Env *e, int x, int y) { return e->cells[y * e->W + x]; }
static int in_map(const Env *e, int x, int y) { return x >= 0 && x < e->W && y >= 0 && y < e->H; }
static int D(const Env *e, int ax, int ay, int bx, int by) {
  int n = e->W * e->H;
  return e->dist[(ay * e->W + ax) * n + (by * e->W + bx)];
}
static int iabs(int v) { return v < 0 ? -v : v; }

/* Object.update_names (core.py:182-186): name = sorted content names */
static int obj_sig(const Env *e, const Obj *o) {
  int s = 0;
  for (int i = 0; i < o->n; i++) s += OC_SIG_OF_TYPE(e->items[o->c[i]].type);
  return s;
}

/* world.insert (world.py:236-237) */
static void world_insert(Env *e, int oid) {
  int sig = obj_sig(e, &e->objs[oid]);
  for (int k = 0; k < e->nkeys; k++)
    if (e->keys[k].sig == sig) {
      e->keys[k].objs[e->keys[k].n++] = oid;
      return;
    }
  Key *nk = &e->keys[e->nkeys++];
  nk->sig = sig;
  nk->n = 1;
  nk->objs[0] = oid;
}

/* world.remove (world.py:239-247): by NAME and LOCATION, last match wins.
 * Returns the object id actually removed, -1 if the assert would fire. */
static int world_remove(Env *e, int oid) {
  const Obj *o = &e->objs[oid];
  int sig = obj_sig(e, o);
  for (int k = 0; k < e->nkeys; k++) {
    Key *key = &e->keys[k];
    if (key->sig != sig) continue;
    int index = -1;
    for (int i = 0; i < key->n; i++) {
      const Obj *c = &e->objs[key->objs[i]];
      if (c->x == o->x && c->y == o->y) index = i;
    }
    if (index < 0) return -1;
    int removed = key->objs[index];
    for (int i = index; i + 1 < key->n; i++) key->objs[i] = key->objs[i + 1];
    key->n--;
    return removed;
  }
  return -1;
}

/* iterate World.objects.values() flattened (world.py:249-253), movable objects only */
static int world_list(const Env *e, int out[MAXOBJ * 2]) {
  int n = 0;
  for (int k = 0; k < e->nkeys; k++)
    for (int i = 0; i < e->keys[k].n; i++) out[n++] = e->keys[k].objs[i];
  return n;
}

/* world.is_occupied (world.py:217-222) */
static int is_occupied(const Env *e, int x, int y) {
  int lst[MAXOBJ * 2];
  int n = world_list(e, lst);
  for (int i = 0; i < n; i++) {
    const Obj *o = &e->objs[lst[i]];
    if (o->x == x && o->y == y && !o->is_held) return 1;
  }
  return 0;
}

/* world.get_object_at(location, None, find_held_objects=False) (world.py:293-308) */
static int get_unheld_object_at(Env *e, int x, int y) {
  int lst[MAXOBJ * 2];
  int n = world_list(e, lst), found = -1, cnt = 0;
  for (int i = 0; i < n; i++) {
    const Obj *o = &e->objs[lst[i]];
    if (o->x == x && o->y == y && !o->is_held) {
      if (cnt == 0) found = lst[i];
      cnt++;
    }
  }
  if (cnt != 1) e->err |= OC_ERR_ALIAS; /* the reference asserts len(objs) == 1 */
  return found;
}

/* Object.needs_chopped (core.py:191-193) + Food.needs_chopped (:302-303) */
static int needs_chopped(const Env *e, const Obj *o) {
  if (o->n > 1) return 0;
  const Item *it = &e->items[o->c[0]];
  return it->type != OC_PLATE && it->state == 0;
}

/* Object.is_deliverable (core.py:232-237) */
static int is_deliverable(const Env *e, const Obj *o) {
  for (int i = 0; i < o->n; i++) {
    const Item *it = &e->items[o->c[i]];
    if (!(it->type == OC_PLATE || it->state == 1)) return 0;
  }
  return o->n > 1;
}

/* mergeable (core.py:240-257): at most one Plate in the union, every Food done */
static int mergeable(const Env *e, const Obj *a, const Obj *b) {
  int plates = 0;
  const Obj *two[2] = {a, b};
  for (int k = 0; k < 2; k++)
    for (int i = 0; i < two[k]->n; i++)
      if (e->items[two[k]->c[i]].type == OC_PLATE) plates++;
  if (plates > 1) return 0;
  for (int k = 0; k < 2; k++)
    for (int i = 0; i < two[k]->n; i++) {
      const Item *it = &e->items[two[k]->c[i]];
      if (it->type != OC_PLATE && it->state != 1) return 0;
    }
  return 1;
}

/* ------------------------------------------------------------------ reset */
/* OvercookedEnvironment.reset / load_level (overcooked_environment.py:100-206):
 * a fresh World; items inserted in world order; counters etc. are static. */
static void env_reset(Env *e) {
  e->nobjs = 0;
  e->nkeys = 0;
  for (int i = 0; i < e->M; i++) {
    e->items[i].type = e->item0[3 * i];
    e->items[i].state = 0;
    Obj *o = &e->objs[e->nobjs];
    o->x = e->next_place[i] >= 0 ? (e->next_place[i] & 15) : e->item0[3 * i + 1];
    o->y = e->next_place[i] >= 0 ? (e->next_place[i] >> 4) : e->item0[3 * i + 2];
    o->is_held = 0;
    o->n = 1;
    o->c[0] = i;
    world_insert(e, e->nobjs);
    e->nobjs++;
  }
  for (int a = 0; a < e->A; a++) {
    e->ax[a] = e->agent0[2 * a];
    e->ay[a] = e->agent0[2 * a + 1];
    e->ahold[a] = -1;
    e->adx[a] = e->ady[a] = 0;
  }
  e->t = 0;
  for (int s = 0; s < e->S; s++) e->completed[s] = e->goalcnt[s] = 0;
  e->err = 0;
  e->successful = 0;
}

/* ------------------------------------------------------------------ collisions */
/* is_collision (overcooked_environment.py:543-576) */
static void is_collision(Env *e, int i, int j, int exec_[2]) {
  exec_[0] = exec_[1] = 1;
  int nix = e->ax[i] + e->adx[i], niy = e->ay[i] + e->ady[i];
  if (!in_map(e, nix, niy)) {
    e->err |= OC_ERR_OOB; /* get_gridsquare_at asserts; treat as collidable */
    nix = e->ax[i];
    niy = e->ay[i];
  } else if (cell_at(e, nix, niy) != OC_FLOOR) {
    nix = e->ax[i];
    niy = e->ay[i];
  }
  int njx = e->ax[j] + e->adx[j], njy = e->ay[j] + e->ady[j];
  if (!in_map(e, njx, njy)) {
    e->err |= OC_ERR_OOB;
    njx = e->ax[j];
    njy = e->ay[j];
  } else if (cell_at(e, njx, njy) != OC_FLOOR) {
    njx = e->ax[j];
    njy = e->ay[j];
  }
  if (nix == njx && niy == njy) {
    if (nix == e->ax[i] && niy == e->ay[i] && (e->adx[i] != 0 || e->ady[i] != 0))
      exec_[1] = 0;
    else if (njx == e->ax[j] && njy == e->ay[j] && (e->adx[j] != 0 || e->ady[j] != 0))
      exec_[0] = 0;
    else
      exec_[0] = exec_[1] = 0;
  } else if (e->ax[i] == njx && e->ay[i] == njy && e->ax[j] == nix && e->ay[j] == niy) {
    exec_[0] = exec_[1] = 0;
  }
}

/* check_collisions (:578-613): every unordered pair, on the ORIGINAL actions */
static void check_collisions(Env *e) {
  int execute[OC_MAX_AGENTS];
  for (int a = 0; a < e->A; a++) execute[a] = 1;
  for (int i = 0; i < e->A; i++)
    for (int j = i + 1; j < e->A; j++) {
      int ex[2];
      is_collision(e, i, j, ex);
      if (!ex[0]) execute[i] = 0;
      if (!ex[1]) execute[j] = 0;
    }
  for (int a = 0; a < e->A; a++)
    if (!execute[a]) e->adx[a] = e->ady[a] = 0;
}

/* ------------------------------------------------------------------ interact */
/* SimAgent.move_to (utils/agent.py:311-314) */
static void agent_move_to(Env *e, int a, int x, int y) {
  e->ax[a] = x;
  e->ay[a] = y;
  if (e->ahold[a] >= 0) {
    e->objs[e->ahold[a]].x = x;
    e->objs[e->ahold[a]].y = y;
  }
}

/* interact (utils/interact.py:4-75), both settings of arglist.play */
static void interact(Env *e, int a) {
  if (e->adx[a] == 0 && e->ady[a] == 0) return; /* :12 */
  int tx = e->ax[a] + e->adx[a], ty = e->ay[a] + e->ady[a];
  /* world.inbounds (world.py:317-320) */
  if (tx < 0) tx = 0;
  if (tx > e->W - 1) tx = e->W - 1;
  if (ty < 0) ty = 0;
  if (ty > e->H - 1) ty = e->H - 1;
  int gs = cell_at(e, tx, ty);

  if (gs == OC_FLOOR) { /* :19-20 */
    agent_move_to(e, a, tx, ty);
  } else if (e->ahold[a] >= 0) { /* :23 */
    int hid = e->ahold[a];
    Obj *held = &e->objs[hid];
    if (gs == OC_DELIVERY) { /* :25-30 */
      if (is_deliverable(e, held)) {
        held->x = tx; /* Delivery.acquire (core.py:123-125) */
        held->y = ty;
        held->is_held = 0; /* agent.release (agent.py:307-309) */
        e->ahold[a] = -1;
      }
    } else if (is_occupied(e, tx, ty)) { /* :33-46 */
      int oid = get_unheld_object_at(e, tx, ty);
      Obj *obj = &e->objs[oid];
      if (mergeable(e, held, obj)) {
        if (world_remove(e, oid) != oid) e->err |= OC_ERR_ALIAS;
        if (world_remove(e, hid) != hid) e->err |= OC_ERR_ALIAS;
        /* agent.acquire(obj) -> holding.merge(obj) (agent.py:305, core.py:210-218) */
        for (int i = 0; i < obj->n; i++) held->c[held->n++] = obj->c[i];
        world_insert(e, hid);
        if (e->play) { /* :44-47 "if playable version, merge onto counter first" */
          held->x = tx; /* gs.acquire(agent.holding) */
          held->y = ty;
          held->is_held = 0; /* agent.release() */
          e->ahold[a] = -1;
        }
      }
    } else { /* :50-59 */
      if (gs == OC_CUTBOARD && needs_chopped(e, held) && !e->play) { /* :52 */
        e->items[held->c[0]].state += 1; /* Object.chop -> Food.update_state */
      } else {
        held->x = tx; /* gs.acquire(obj) */
        held->y = ty;
        held->is_held = 0; /* agent.release() */
        e->ahold[a] = -1;
      }
    }
  } else { /* :62-75 */
    if (is_occupied(e, tx, ty) && gs != OC_DELIVERY) {
      int oid = get_unheld_object_at(e, tx, ty);
      if (gs == OC_CUTBOARD && needs_chopped(e, &e->objs[oid]) && e->play) {
        e->items[e->objs[oid].c[0]].state += 1; /* :66-67: chopped where it lies, not picked up */
      } else
      /* gs.release(); agent.acquire(obj): no-op for an ALLERGIC agent (agent.py:296-298) */
      if (!((e->allergic >> a) & 1)) {
        Obj *obj = &e->objs[oid];
        e->ahold[a] = oid;
        obj->is_held = 1;
        obj->x = e->ax[a];
        obj->y = e->ay[a];
      }
    }
  }
}

/* ------------------------------------------------------------------ goals */
/* Object.__eq__ against the goal object of subtask s (core.py:164-169,
 * navigation_planner/utils.py:161-209): same names, same length, every food chopped */
static int matches_goal(const Env *e, const Obj *o, const Sub *s) {
  if (obj_sig(e, o) != s->sig || o->n != s->ncont) return 0;
  for (int i = 0; i < o->n; i++) {
    const Item *it = &e->items[o->c[i]];
    if (it->type != OC_PLATE && it->state != 1) return 0;
  }
  return 1;
}

/* ---- list(set(locations)) as CPython builds and iterates it -------------------------------
 * World.get_all_object_locs returns list(set(held_locs + unheld_locs)) (world.py:290-291) and
 * calculate_reward_shaping takes element [0] of it (overcooked_environment.py:287,374-379).
 * With one matching object that is that object's location; with several (levels that repeat a
 * content type) it is whichever location sits in the LOWEST SLOT of the set's hash table --
 * deterministic (int and tuple hashes are not randomised), but neither world order nor
 * sorted.  Restated here for CPython >= 3.8 (tuple hash: Objects/tupleobject.c, the
 * xxHash-style tuplehash; set table: Objects/setobject.c, set_add_entry / set_table_resize /
 * set_insert_clean with LINEAR_PROBES = 9, PERTURB_SHIFT = 5, an 8-slot table that grows 4x
 * when fill * 5 >= mask * 3).  Pinned twice: against the running interpreter
 * (tests/test_oracle_invariants.py, random location lists) and through the shaping bits of
 * the tests/golden/c*_dup_* fixtures recorded from the reference. */
static uint64_t py_hash_xy(int x, int y) { /* hash((x, y)) for small non-negative ints */
  const uint64_t P1 = 11400714785074694791ULL, P2 = 14029467366897019727ULL, P5 = 2870177450012600261ULL;
  uint64_t acc = P5;
  const uint64_t lane[2] = {(uint64_t)x, (uint64_t)y}; /* hash(int) == int here */
  for (int k = 0; k < 2; k++) {
    acc += lane[k] * P2;
    acc = (acc << 31) | (acc >> 33);
    acc *= P1;
  }
  acc += 2ULL ^ (P5 ^ 3527539ULL);
  if (acc == (uint64_t)-1) return 1546275796ULL;
  return acc;
}

#define PYSET_MAX 64 /* table slots we ever need: <= 16 objects -> an 8-, 32- or 64-slot table */
typedef struct {
  int used, mask;
  int key[PYSET_MAX]; /* x | y<<4, -1 = empty */
  uint64_t hash[PYSET_MAX];
} PySet;

static void pyset_insert_clean(PySet *t, int key, uint64_t h) { /* set_insert_clean */
  uint64_t perturb = h;
  size_t i = (size_t)h & (size_t)t->mask;
  for (;;) {
    if (t->key[i] < 0) break;
    if (i + 9 <= (size_t)t->mask) {
      int found = 0;
      for (size_t j = 1; j <= 9; j++)
        if (t->key[i + j] < 0) {
          i += j;
          found = 1;
          break;
        }
      if (found) break;
    }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & (size_t)t->mask;
  }
  t->key[i] = key;
  t->hash[i] = h;
}

static void pyset_add(PySet *t, int x, int y) { /* set_add_entry; no deletions, so no dummies */
  const int key = x | (y << 4);
  const uint64_t h = py_hash_xy(x, y);
  uint64_t perturb = h;
  size_t i = (size_t)h & (size_t)t->mask;
  for (;;) {
    size_t probes = (i + 9 <= (size_t)t->mask) ? 9 : 0, j = i;
    int placed = 0;
    do {
      if (t->key[j] < 0) {
        t->key[j] = key;
        t->hash[j] = h;
        t->used++;
        placed = 1;
        break;
      }
      if (t->key[j] == key) return; /* already in the set */
      j++;
    } while (probes--);
    if (placed) break;
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & (size_t)t->mask;
  }
  if (t->used * 5 >= t->mask * 3) { /* set_table_resize(so, used * 4) */
    PySet old = *t;
    int newsize = 8;
    while (newsize <= old.used * 4) newsize <<= 1;
    t->mask = newsize - 1;
    for (int k = 0; k < PYSET_MAX; k++) t->key[k] = -1;
    for (int k = 0; k <= old.mask; k++)
      if (old.key[k] >= 0) pyset_insert_clean(t, old.key[k], old.hash[k]);
  }
}

/* list(set(locs)): the distinct locations in the set's iteration order; returns the count */
static int pyset_order(const int *xs, const int *ys, int n, int *ox, int *oy) {
  PySet t;
  t.used = 0;
  t.mask = 7;
  for (int k = 0; k < PYSET_MAX; k++) t.key[k] = -1;
  for (int i = 0; i < n; i++) pyset_add(&t, xs[i], ys[i]);
  int cnt = 0;
  for (int k = 0; k <= t.mask; k++)
    if (t.key[k] >= 0) {
      ox[cnt] = t.key[k] & 15;
      oy[cnt] = t.key[k] >> 4;
      cnt++;
    }
  return cnt;
}

/* exported for the test that pins the emulation against the running interpreter */
__attribute__((visibility("default"))) int oc_oracle_pyset_order(const int32_t *xy, int n, int32_t *out_xy) {
  int xs[MAXOBJ], ys[MAXOBJ], ox[MAXOBJ], oy[MAXOBJ];
  if (n > MAXOBJ) n = MAXOBJ;
  for (int i = 0; i < n; i++) xs[i] = xy[2 * i], ys[i] = xy[2 * i + 1];
  int cnt = pyset_order(xs, ys, n, ox, oy);
  for (int i = 0; i < cnt; i++) out_xy[2 * i] = ox[i], out_xy[2 * i + 1] = oy[i];
  return cnt;
}

/* world.get_all_object_locs(obj) (world.py:278-291) for the objects `match` accepts:
 * get_object_locs(is_held=True) + get_object_locs(is_held=False), each in the order of the
 * name's list, then list(set(...)).  Returns the number of distinct locations, in CPython's
 * set order (element [0] is what the shaping terms use). */
typedef int (*MatchFn)(const Env *, const Obj *, const void *);
static int all_object_locs(const Env *e, MatchFn match, const void *arg, int *xs, int *ys) {
  int lst[MAXOBJ * 2], rx[MAXOBJ], ry[MAXOBJ];
  int n = world_list(e, lst), m = 0;
  for (int held = 1; held >= 0; held--)
    for (int i = 0; i < n; i++) {
      const Obj *o = &e->objs[lst[i]];
      if (o->is_held == held && match(e, o, arg)) {
        rx[m] = o->x;
        ry[m] = o->y;
        m++;
      }
    }
  return pyset_order(rx, ry, m, xs, ys);
}

static int match_goal_fn(const Env *e, const Obj *o, const void *s) { return matches_goal(e, o, (const Sub *)s); }
static int goal_locs(const Env *e, const Sub *s, int *xs, int *ys) {
  return all_object_locs(e, match_goal_fn, s, xs, ys);
}

/* location of "the" fresh X alone, for Chop(X) (start_obj of get_subtask_obj;
 * get_all_object_locs(obj=start_obj)[0], overcooked_environment.py:287) */
static int match_fresh_fn(const Env *e, const Obj *o, const void *food) {
  return o->n == 1 && e->items[o->c[0]].type == *(const int *)food && e->items[o->c[0]].state == 0;
}
static int fresh_loc(const Env *e, int food, int *x, int *y) {
  int xs[MAXOBJ], ys[MAXOBJ];
  int n = all_object_locs(e, match_fresh_fn, &food, xs, ys);
  if (n == 0) return 0;
  *x = xs[0];
  *y = ys[0];
  return 1;
}

/* done (overcooked_environment.py:243-270) */
static int env_done(Env *e) {
  if (e->T && e->t >= e->T) {
    e->successful = 0;
    return 1;
  }
  for (int s = 0; s < e->S; s++) {
    if (e->sub[s].kind != OC_DELIVER) continue;
    int xs[MAXOBJ], ys[MAXOBJ];
    int n = goal_locs(e, &e->sub[s], xs, ys), hit = 0;
    for (int k = 0; k < n; k++)
      if (xs[k] == e->deliv[0] && ys[k] == e->deliv[1]) hit = 1; /* first Delivery tile only (:259) */
    if (!hit) {
      e->successful = 0;
      return 0;
    }
  }
  e->successful = 1;
  return 1;
}

/* subtask_reward + reward (:399-432) */
static int env_reward(Env *e) {
  int reward = 0;
  for (int s = 0; s < e->S; s++) {
    int r = 0;
    int xs[MAXOBJ], ys[MAXOBJ];
    int n = goal_locs(e, &e->sub[s], xs, ys);
    if (e->sub[s].kind == OC_DELIVER) {
      for (int k = 0; k < n; k++)
        if (xs[k] == e->deliv[0] && ys[k] == e->deliv[1]) r = 3;
    } else {
      if (n > e->goalcnt[s]) r = 1;
      e->goalcnt[s] = n;
    }
    reward += r;
    if (r != 0) e->completed[s] = 1;
  }
  return reward;
}

/* calculate_reward_shaping (:272-397).  Every division is int / int in Python, i.e.
 * one correctly rounded fp64 division; sums run left to right in fp64. */
static double reward_shaping(const Env *e, int a) {
  const int MAXP = e->max_path;
  const int agx = e->ax[a], agy = e->ay[a];
  double total = 0.0;
  int total_is_zero = 1; /* Python `total_penalty == 0` */

  /* Chop term (:278-304) */
  int nchop = 0, mind = 0;
  for (int s = 0; s < e->S; s++) {
    if (e->sub[s].kind != OC_CHOP || e->completed[s]) continue;
    int x = 0, y = 0;
    if (!fresh_loc(e, e->sub[s].food, &x, &y)) continue; /* unreachable: IndexError in the reference */
    int d = D(e, agx, agy, x, y);
    if (nchop == 0 || d < mind) mind = d;
    nchop++;
  }
  if (nchop > 0) {
    total += (double)((mind + MAXP) + (nchop - 1) * 2 * MAXP) / (double)MAXP;
    total_is_zero = (total == 0.0);
  }

  /* pair term (:319-363): Plate + recipes[0] ingredient names */
  int lx[OC_MAX_PAIR][MAXOBJ], ly[OC_MAX_PAIR][MAXOBJ], ln[OC_MAX_PAIR];
  for (int p = 0; p < e->npair; p++) ln[p] = 0;
  {
    int lst[MAXOBJ * 2];
    int n = world_list(e, lst);
    for (int i = 0; i < n; i++) {
      const Obj *o = &e->objs[lst[i]];
      for (int c = 0; c < o->n; c++)
        for (int p = 0; p < e->npair; p++)
          if (e->items[o->c[c]].type == e->pair[p]) {
            lx[p][ln[p]] = o->x;
            ly[p][ln[p]] = o->y;
            ln[p]++;
          }
    }
  }
  int npairs = 0, minpair = 0;
  for (int p = 0; p < e->npair; p++)
    for (int q = p + 1; q < e->npair; q++) {
      int val;
      if (ln[p] > 0 && ln[q] > 0) {
        int m = MAXP;
        for (int i = 0; i < ln[p]; i++)
          for (int j = 0; j < ln[q]; j++) {
            int d = D(e, lx[p][i], ly[p][i], lx[q][j], ly[q][j]);
            if (d < m) m = d;
          }
        if (m == 0) continue;
        val = m;
      } else {
        val = MAXP;
      }
      if (npairs == 0 || val < minpair) minpair = val;
      npairs++;
    }
  if (npairs > 0) {
    if (total_is_zero)
      total += (double)(minpair + (npairs - 1) * MAXP) / (double)MAXP;
    else
      total += (double)(npairs * MAXP) / (double)MAXP;
  }

  /* Deliver term (:370-395) */
  for (int s = 0; s < e->S; s++) {
    if (e->sub[s].kind != OC_DELIVER || e->completed[s]) continue;
    int xs[MAXOBJ], ys[MAXOBJ];
    int n = goal_locs(e, &e->sub[s], xs, ys);
    if (n == 0) {
      total += 2.0;
    } else {
      int d = D(e, agx, agy, xs[0], ys[0]) + iabs(agx - xs[0]) + iabs(agy - ys[0]);
      if (d == 0) {
        int best = 0;
        for (int k = 0; k < e->ndeliv; k++) {
          int dx = e->deliv[2 * k], dy = e->deliv[2 * k + 1];
          int dd = D(e, agx, agy, dx, dy) + iabs(agx - dx) + iabs(agy - dy);
          if (k == 0 || dd < best) best = dd;
        }
        total += (double)best / (double)MAXP;
      } else {
        total += (double)d / (double)MAXP + 1.0;
      }
    }
  }
  return total;
}

/* ------------------------------------------------------------------ public API */
#define OC_EXPORT __attribute__((visibility("default")))

OC_EXPORT void *oc_oracle_create(const int32_t *blob, int n_words) {
  if (!blob || n_words < OC_LV_HEADER_WORDS || blob[OC_LV_MAGIC] != OC_LV_MAGIC_VALUE ||
      blob[OC_LV_VERSION] != OC_LV_VERSION_VALUE || blob[OC_LV_TOTAL] != n_words)
    return NULL;
  Env *e = (Env *)calloc(1, sizeof(Env));
  e->blob = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_words);
  memcpy(e->blob, blob, sizeof(int32_t) * (size_t)n_words);
  const int32_t *b = e->blob;
  e->W = b[OC_LV_W]; e->H = b[OC_LV_H]; e->A = b[OC_LV_A]; e->M = b[OC_LV_M];
  e->S = b[OC_LV_S]; e->T = b[OC_LV_T]; e->max_path = b[OC_LV_MAX_PATH];
  e->allergic = b[OC_LV_ALLERGIC]; e->npair = b[OC_LV_NPAIR]; e->ndeliv = b[OC_LV_NDELIV];
  e->play = b[OC_LV_FLAGS] & OC_FLAG_PLAY;
  e->cells = b + b[OC_LV_OFF_CELLS];
  e->dist = b + b[OC_LV_OFF_DIST];
  e->agent0 = b + b[OC_LV_OFF_AGENTS];
  e->item0 = b + b[OC_LV_OFF_ITEMS];
  e->pair = b + b[OC_LV_OFF_PAIR];
  e->deliv = b + b[OC_LV_OFF_DELIV];
  for (int s = 0; s < e->S; s++) {
    const int32_t *p = b + b[OC_LV_OFF_SUBTASKS] + 4 * s;
    e->sub[s].kind = p[0]; e->sub[s].sig = p[1]; e->sub[s].food = p[2]; e->sub[s].ncont = p[3];
  }
  for (int i = 0; i < OC_MAX_ITEMS; i++) e->next_place[i] = -1;
  env_reset(e);
  return e;
}

OC_EXPORT void oc_oracle_destroy(void *h) {
  Env *e = (Env *)h;
  if (!e) return;
  free(e->blob);
  free(e);
}

OC_EXPORT void *oc_oracle_clone(const void *h) {
  const Env *s = (const Env *)h;
  Env *e = (Env *)malloc(sizeof(Env));
  memcpy(e, s, sizeof(Env));
  size_t nw = (size_t)s->blob[OC_LV_TOTAL];
  e->blob = (int32_t *)malloc(sizeof(int32_t) * nw);
  memcpy(e->blob, s->blob, sizeof(int32_t) * nw);
  ptrdiff_t shift = e->blob - s->blob;
  e->cells += shift; e->dist += shift; e->agent0 += shift; e->item0 += shift;
  e->pair += shift; e->deliv += shift;
  return e;
}

OC_EXPORT void oc_oracle_reset(void *h) { env_reset((Env *)h); }

/* Set the item start cells used by every following reset of this env (packed x | y<<4,
 * one per item in world order; NULL restores the blob's defaults). */
OC_EXPORT void oc_oracle_set_placement(void *h, const int32_t *cells) {
  Env *e = (Env *)h;
  for (int i = 0; i < e->M; i++) e->next_place[i] = cells ? cells[i] : -1;
}

/* OvercookedEnvironment.step (overcooked_environment.py:211-241).
 * actions: A codes (0..4).  shaping[2] = agent_0 / agent_1 reward shaping. */
OC_EXPORT void oc_oracle_step(void *h, const int32_t *actions, int32_t *reward, int32_t *done,
                              double *shaping) {
  Env *e = (Env *)h;
  e->t += 1;
  for (int a = 0; a < e->A; a++) {
    int c = actions[a];
    if (c < 0 || c > 4) { /* no such NAV action: flagged, executed as (0, 0) (include/oc_level.h) */
      e->err |= OC_ERR_ACTION;
      c = 4;
    }
    e->adx[a] = NAV_DX[c];
    e->ady[a] = NAV_DY[c];
  }
  check_collisions(e);
  for (int a = 0; a < e->A; a++) interact(e, a); /* execute_navigation (:615-618) */
  int d = env_done(e);
  int r = env_reward(e);
  shaping[0] = reward_shaping(e, 0);
  shaping[1] = reward_shaping(e, 1);
  *reward = r;
  *done = d;
}

/* TEST ONLY: put a freshly reset env into a hand-made state of single-content objects, so that
 * corners random play practically never reaches (e.g. the World.remove alias, world.py:239-247)
 * can be staged.  agents [A][3] = x, y, held item id (-1 = empty hands); items [M][3] = x, y,
 * state_index (a held item sits on its holder's cell whatever x, y say); completed / goalcnt [S]
 * (NULL = zeros): completed_subtasks and goal_objects_count, which must agree with the objects
 * (a chopped food means its Chop subtask was rewarded) or the state is one the reference
 * cannot be in. */
OC_EXPORT void oc_oracle_debug_set(void *h, const int32_t *agents, const int32_t *items, const int32_t *completed,
                                   const int32_t *goalcnt) {
  Env *e = (Env *)h;
  env_reset(e); /* object i = item i; keys by type in world order */
  for (int s = 0; s < e->S; s++) { /* the bookkeeping that goes with the staged objects */
    e->completed[s] = completed ? completed[s] : 0;
    e->goalcnt[s] = goalcnt ? goalcnt[s] : 0;
  }
  for (int i = 0; i < e->M; i++) {
    e->objs[i].x = items[3 * i];
    e->objs[i].y = items[3 * i + 1];
    e->objs[i].is_held = 0;
    e->items[i].state = items[3 * i + 2];
  }
  for (int a = 0; a < e->A; a++) {
    e->ax[a] = agents[3 * a];
    e->ay[a] = agents[3 * a + 1];
    e->ahold[a] = agents[3 * a + 2];
    if (e->ahold[a] >= 0) {
      Obj *o = &e->objs[e->ahold[a]];
      o->is_held = 1;
      o->x = e->ax[a];
      o->y = e->ay[a];
    }
  }
}

OC_EXPORT int oc_oracle_successful(const void *h) { return ((const Env *)h)->successful; }
OC_EXPORT int oc_oracle_error(const void *h) { return ((const Env *)h)->err; }

/* Canonical snapshot, the format of tests/golden base fixtures:
 *   items  [M][5] : x, y, state_index, group (smallest item id in its object), holder agent (-1)
 *   order  [M]    : groups in world.objects iteration order, -1 padded
 *   agents [A][3] : x, y, held group (-1)
 *   misc   [2]    : t, number of objects in the world
 *   completed[S], goalcnt[S] */
OC_EXPORT void oc_oracle_snapshot(const void *h, int32_t *items, int32_t *order, int32_t *agents,
                                  int32_t *misc, int32_t *completed, int32_t *goalcnt) {
  const Env *e = (const Env *)h;
  int lst[MAXOBJ * 2];
  int n = world_list(e, lst);
  int group_of[MAXOBJ];
  for (int i = 0; i < MAXOBJ; i++) group_of[i] = -2;
  for (int i = 0; i < e->M; i++) order[i] = -1;
  for (int i = 0; i < e->M * 5; i++) items[i] = -9;
  for (int i = 0; i < n; i++) {
    const Obj *o = &e->objs[lst[i]];
    int g = o->c[0];
    for (int c = 1; c < o->n; c++)
      if (o->c[c] < g) g = o->c[c];
    group_of[lst[i]] = g;
    if (i < e->M) order[i] = g;
    int holder = -1;
    for (int a = 0; a < e->A; a++)
      if (e->ahold[a] == lst[i]) holder = a;
    for (int c = 0; c < o->n; c++) {
      int32_t *row = items + 5 * o->c[c];
      row[0] = o->x; row[1] = o->y; row[2] = e->items[o->c[c]].state; row[3] = g; row[4] = holder;
    }
  }
  for (int a = 0; a < e->A; a++) {
    agents[3 * a] = e->ax[a];
    agents[3 * a + 1] = e->ay[a];
    agents[3 * a + 2] = e->ahold[a] >= 0 ? group_of[e->ahold[a]] : -1;
  }
  misc[0] = e->t;
  misc[1] = n;
  for (int s = 0; s < e->S; s++) {
    completed[s] = e->completed[s];
    goalcnt[s] = e->goalcnt[s];
  }
}

/* OvercookedMultiEnv.get_observation2 (gym_comm/envs/overcooked_env.py:105-159) for
 * viewer 0 or 1.  out layout (int32, F = 22 + S + 2C entries):
 *   x[4] y[4] state[4] hidden[4] completed[S] agent1_location[2] agent2_location[2]
 *   agent_is_holding[2] agent1_comm[C] agent2_comm[C]
 * comm[k] = index of the one-hot bit of per_agent_communications[k], -1 = all zeros.
 * timestep = t / max_num_timesteps in fp64. */
OC_EXPORT void oc_oracle_obs(const void *h, int viewer, int radius, int viewer_blind, int ego_blind,
                             int C, const int32_t *comm, int32_t *out, double *timestep) {
  const Env *e = (const Env *)h;
  int dx[4] = {0, 0, 0, 0}, dy[4] = {0, 0, 0, 0}, st[4] = {0, 0, 0, 0}, hid[4] = {1, 1, 1, 1};
  int vx = e->ax[viewer], vy = e->ay[viewer];
  if (!viewer_blind) {
    int lst[MAXOBJ * 2];
    int n = world_list(e, lst);
    for (int i = 0; i < n; i++) { /* last writer wins (:121-131) */
      const Obj *o = &e->objs[lst[i]];
      for (int c = 0; c < o->n; c++) {
        const Item *it = &e->items[o->c[c]];
        if (it->type != OC_PLATE) st[it->type] = it->state;
        dx[it->type] = o->x - vx;
        dy[it->type] = o->y - vy;
      }
    }
    for (int k = 0; k < 4; k++) hid[k] = (iabs(dx[k]) + iabs(dy[k]) <= radius) ? 0 : 1;
  }
  int p = 0;
  /* visible_distances: (0,0) when WITHIN the radius, the delta otherwise (:135; sic) */
  for (int k = 0; k < 4; k++) out[p++] = (iabs(dx[k]) + iabs(dy[k]) <= radius) ? 0 : dx[k];
  for (int k = 0; k < 4; k++) out[p++] = (iabs(dx[k]) + iabs(dy[k]) <= radius) ? 0 : dy[k];
  for (int k = 0; k < 4; k++) out[p++] = st[k];
  for (int k = 0; k < 4; k++) out[p++] = hid[k];
  for (int s = 0; s < e->S; s++) out[p++] = e->completed[s];
  out[p++] = viewer_blind ? 0 : e->ax[0];
  out[p++] = viewer_blind ? 0 : e->ay[0];
  out[p++] = viewer_blind ? 0 : e->ax[1];
  out[p++] = viewer_blind ? 0 : e->ay[1];
  out[p++] = ego_blind ? 0 : (e->ahold[viewer] >= 0 ? 1 : 0); /* :154, gated on the EGO's flag */
  out[p++] = 0;
  for (int k = 0; k < 2; k++)
    for (int c = 0; c < C; c++) out[p++] = (comm[k] == c) ? 1 : 0;
  *timestep = (double)e->t / (double)e->T;
}

/* ------------------------------------------------------------------ batch helpers
 * (used by tests to drive many oracle envs with the same [row][n] tensors the HIP
 * library takes, and by bench.py's cpu_baseline leg) */
OC_EXPORT void oc_oracle_batch_step(void **envs, int64_t n0, int64_t n1, int64_t n_stride,
                                    const int32_t *actions /*[A][n_stride]*/, int32_t *reward,
                                    int32_t *done, double *shaping /*[2][n_stride]*/, int auto_reset) {
  for (int64_t i = n0; i < n1; i++) {
    Env *e = (Env *)envs[i];
    int32_t act[OC_MAX_AGENTS];
    for (int a = 0; a < e->A; a++) act[a] = actions[(int64_t)a * n_stride + i];
    double sh[2];
    oc_oracle_step(e, act, &reward[i], &done[i], sh);
    shaping[i] = sh[0];
    shaping[n_stride + i] = sh[1];
    if (auto_reset && done[i]) env_reset(e);
  }
}

/* gym_comm OvercookedMultiEnv.multi_step (overcooked_env.py:207-282) over a range of
 * envs: actions [4][n] = ego move, ego comm, alt move, alt comm.  comm [2][n] is the
 * persistent per_agent_communications state.  obs [2][F][n]. */
OC_EXPORT void oc_oracle_batch_multi_step(void **envs, int64_t n0, int64_t n1, int64_t n_stride,
                                          const int32_t *actions, int32_t *comm, int radius,
                                          int blind_mask, int C, int communication_on, int ego_led,
                                          int ego_agent_idx, int can_move_mask, int32_t *obs,
                                          double *timestep, double *reward, int32_t *done,
                                          int auto_reset) {
  for (int64_t i = n0; i < n1; i++) {
    Env *e = (Env *)envs[i];
    int F = 22 + e->S + 2 * C;
    int ego_mv = actions[0 * n_stride + i], ego_cm = actions[1 * n_stride + i];
    int alt_mv = actions[2 * n_stride + i], alt_cm = actions[3 * n_stride + i];
    /* one_hot[idx] = 1 / NAV_ACTIONS[idx] raise IndexError for an index out of range
     * (:227-248; both move indices are looked up, moved or not): the defined behaviour of
     * include/oc_level.h OC_ERR_ACTION -- flag, send nothing / stand still */
    int ego_talks = communication_on, alt_talks = communication_on && !ego_led;
    int bad = (ego_talks && (ego_cm < 0 || ego_cm >= C)) || (alt_talks && (alt_cm < 0 || alt_cm >= C)) ||
              ego_mv < 0 || ego_mv > 3 || alt_mv < 0 || alt_mv > 3;
    comm[i] = (ego_talks && ego_cm >= 0 && ego_cm < C) ? ego_cm : -1;               /* :227-246 */
    comm[n_stride + i] = (alt_talks && alt_cm >= 0 && alt_cm < C) ? alt_cm : -1;
    int32_t act[OC_MAX_AGENTS] = {4, 4, 4, 4};                         /* :250-262 */
    int ego_slot = ego_agent_idx == 0 ? 0 : 1;
    if ((can_move_mask & 1) && ego_mv >= 0 && ego_mv <= 3) act[ego_slot] = ego_mv;
    if ((can_move_mask & 2) && alt_mv >= 0 && alt_mv <= 3) act[1 - ego_slot] = alt_mv;
    int32_t r, d;
    double sh[2];
    if (bad) e->err |= OC_ERR_ACTION;
    oc_oracle_step(e, act, &r, &d, sh);
    reward[i] = (double)r - sh[0] - sh[1];                             /* :282 */
    done[i] = d;
    if (auto_reset && d) env_reset(e);
    int32_t tmp[22 + OC_MAX_SUBTASKS + 2 * 128]; /* C <= 128 */
    int32_t cm[2] = {comm[i], comm[n_stride + i]};
    for (int v = 0; v < 2; v++) {
      double ts;
      oc_oracle_obs(e, v, radius, (blind_mask >> v) & 1, blind_mask & 1, C, cm, tmp, &ts);
      for (int f = 0; f < F; f++) obs[((int64_t)v * F + f) * n_stride + i] = tmp[f];
      timestep[i] = ts;
    }
  }
}

/* snapshot of envs [0, n): arrays are [n][M][5], [n][M], [n][A][3], [n][2], [n][S], [n][S],
 * plus err[n] */
OC_EXPORT void oc_oracle_batch_snapshot(void **envs, int64_t n, int32_t *items, int32_t *order,
                                        int32_t *agents, int32_t *misc, int32_t *completed,
                                        int32_t *goalcnt, int32_t *err) {
  for (int64_t i = 0; i < n; i++) {
    const Env *e = (const Env *)envs[i];
    oc_oracle_snapshot(e, items + i * e->M * 5, order + i * e->M, agents + i * e->A * 3, misc + i * 2,
                       completed + i * e->S, goalcnt + i * e->S);
    err[i] = e->err;
  }
}

OC_EXPORT void oc_oracle_batch_reset(void **envs, int64_t n, const int32_t *mask) {
  for (int64_t i = 0; i < n; i++)
    if (!mask || mask[i]) env_reset((Env *)envs[i]);
}

/* placement: [M][n] packed start cells, read by every env (like the HIP library's
 * `placement` tensor: the cells an env uses at its next reset / auto-reset) */
OC_EXPORT void oc_oracle_batch_set_placement(void **envs, int64_t n, const int32_t *placement) {
  for (int64_t i = 0; i < n; i++) {
    Env *e = (Env *)envs[i];
    for (int k = 0; k < e->M; k++) e->next_place[k] = placement ? placement[(int64_t)k * n + i] : -1;
  }
}

/* K consecutive wrapper steps over envs [n0, n1) in one call (one call per thread in
 * bench.py's cpu_baseline leg, so Python stays out of the timed loop).  actions is
 * [K][4][n_stride]; only the last step's outputs are kept. */
OC_EXPORT void oc_oracle_batch_multi_rollout(void **envs, int64_t n0, int64_t n1, int64_t n_stride,
                                             int64_t K, const int32_t *actions, int32_t *comm,
                                             int radius, int blind_mask, int C, int communication_on,
                                             int ego_led, int ego_agent_idx, int can_move_mask,
                                             int32_t *obs, double *timestep, double *reward,
                                             int32_t *done, int auto_reset) {
  for (int64_t k = 0; k < K; k++)
    oc_oracle_batch_multi_step(envs, n0, n1, n_stride, actions + k * 4 * n_stride, comm, radius,
                               blind_mask, C, communication_on, ego_led, ego_agent_idx, can_move_mask,
                               obs, timestep, reward, done, auto_reset);
}

OC_EXPORT void oc_oracle_batch_rollout(void **envs, int64_t n0, int64_t n1, int64_t n_stride, int64_t K,
                                       int A, const int32_t *actions, int32_t *reward, int32_t *done,
                                       double *shaping, int auto_reset) {
  for (int64_t k = 0; k < K; k++)
    oc_oracle_batch_step(envs, n0, n1, n_stride, actions + k * A * n_stride, reward, done, shaping,
                         auto_reset);
}

/* OvercookedMultiEnv.get_partial_observability_FOW (gym_comm/envs/overcooked_env.py:161-202):
 * out[(k*W + x)*H + y], k = 0..6: 0 tile type, 1.. agents, 3 + channel contents; cells
 * farther than `radius` (manhattan) from the viewer are -1 in every plane.
 * holding[2] = (agent 0 holds, agent 1 holds). */
OC_EXPORT void oc_oracle_obs_image(const void *h, int viewer, int radius, int8_t *out, int32_t *holding) {
  const Env *e = (const Env *)h;
  const int W = e->W, H = e->H;
  memset(out, 0, (size_t)7 * W * H);
  int lst[MAXOBJ * 2];
  int n = world_list(e, lst);
  for (int i = 0; i < n; i++) { /* :171-178 */
    const Obj *o = &e->objs[lst[i]];
    for (int c = 0; c < o->n; c++) {
      const Item *it = &e->items[o->c[c]];
      out[((it->type + 3) * W + o->x) * H + o->y] = (int8_t)(it->type == OC_PLATE ? 1 : it->state + 1);
    }
  }
  for (int y = 0; y < H; y++) /* :179-180 */
    for (int x = 0; x < W; x++) out[(0 * W + x) * H + y] = (int8_t)cell_at(e, x, y);
  for (int a = 0; a < e->A; a++) out[((a + 1) * W + e->ax[a]) * H + e->ay[a]] = 1; /* :183-185 */
  for (int x = 0; x < W; x++) /* :187-194 */
    for (int y = 0; y < H; y++)
      if (iabs(x - e->ax[viewer]) + iabs(y - e->ay[viewer]) > radius)
        for (int k = 0; k < 7; k++) out[(k * W + x) * H + y] = -1;
  holding[0] = e->ahold[0] >= 0;
  holding[1] = e->ahold[1] >= 0;
}

/* Replay a whole recorded tape in one call (the golden-vector tests: no Python in the loop).
 * reset_before[k] != 0: reset before step k, first installing placements row pl_index[k]
 * ([rows][M] packed cells) when placements != NULL. */
OC_EXPORT void oc_oracle_replay_base(void *h, int64_t K, const int32_t *actions, const int32_t *reset_before,
                                     const int32_t *pl_index, const int32_t *placements, int32_t *items,
                                     int32_t *order, int32_t *agents, int32_t *misc, int32_t *completed,
                                     int32_t *goalcnt, int32_t *reward, int32_t *done, double *shaping,
                                     int32_t *err) {
  Env *e = (Env *)h;
  for (int64_t k = 0; k < K; k++) {
    if (reset_before[k]) {
      if (placements) oc_oracle_set_placement(e, placements + (int64_t)pl_index[k] * e->M);
      env_reset(e);
    }
    oc_oracle_step(e, actions + k * e->A, &reward[k], &done[k], shaping + 2 * k);
    oc_oracle_snapshot(e, items + k * e->M * 5, order + k * e->M, agents + k * e->A * 3, misc + k * 2,
                       completed + k * e->S, goalcnt + k * e->S);
    err[k] = e->err;
  }
}

/* Same for wrapper tapes: actions [K][4] = ego move, ego comm, alt move, alt comm; obs out
 * [K][2][F]; ts [K][2]; comm[2] is the persistent per_agent_communications state. */
OC_EXPORT void oc_oracle_replay_wrapper(void *h, int64_t K, const int32_t *actions, const int32_t *reset_before,
                                        const int32_t *pl_index, const int32_t *placements, int32_t *comm,
                                        int radius, int blind_mask, int C, int communication_on, int ego_led,
                                        int ego_agent_idx, int can_move_mask, int32_t *obs, double *ts,
                                        double *reward, int32_t *done) {
  Env *e = (Env *)h;
  void *envs[1] = {h};
  const int F = 22 + e->S + 2 * C;
  for (int64_t k = 0; k < K; k++) {
    if (reset_before[k]) {
      if (placements) oc_oracle_set_placement(e, placements + (int64_t)pl_index[k] * e->M);
      env_reset(e);
    }
    int32_t o[2 * (22 + OC_MAX_SUBTASKS + 2 * 128)];
    double t1;
    oc_oracle_batch_multi_step(envs, 0, 1, 1, actions + 4 * k, comm, radius, blind_mask, C, communication_on,
                               ego_led, ego_agent_idx, can_move_mask, o, &t1, &reward[k], &done[k], 0);
    memcpy(obs + k * 2 * F, o, sizeof(int32_t) * 2 * (size_t)F);
    ts[2 * k] = ts[2 * k + 1] = t1;
  }
}

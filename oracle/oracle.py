"""ctypes binding of the CPU oracle (oracle/oc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()`` and by
``bench.py``'s ``cpu_baseline`` leg as the checker / reported baseline -- never by the
product package.  The oracle restates the reference's step/reset/obs semantics and is
pinned against golden vectors recorded from the reference itself
(tests/golden/*.npz, tests/test_oracle_golden.py).
"""
import ctypes
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OC_ORACLE_LIB selects another build of the same source, e.g. the ASan/UBSan one
# (`make -C oracle asan`, tests/oracle_asan.sh)
_LIB_PATH = os.environ.get("OC_ORACLE_LIB") or os.path.join(_HERE, "_build", "liboc_oracle.so")
_lib = None


def build(force=False):
    if os.environ.get("OC_ORACLE_LIB"):
        return _LIB_PATH
    src = os.path.join(_HERE, "oc_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "oc_level.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


_I32P = ctypes.POINTER(ctypes.c_int32)
_F64P = ctypes.POINTER(ctypes.c_double)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.oc_oracle_create.restype = ctypes.c_void_p
        L.oc_oracle_create.argtypes = [_I32P, ctypes.c_int]
        L.oc_oracle_clone.restype = ctypes.c_void_p
        L.oc_oracle_clone.argtypes = [ctypes.c_void_p]
        L.oc_oracle_destroy.argtypes = [ctypes.c_void_p]
        L.oc_oracle_reset.argtypes = [ctypes.c_void_p]
        L.oc_oracle_step.argtypes = [ctypes.c_void_p, _I32P, _I32P, _I32P, _F64P]
        L.oc_oracle_successful.argtypes = [ctypes.c_void_p]
        L.oc_oracle_error.argtypes = [ctypes.c_void_p]
        L.oc_oracle_snapshot.argtypes = [ctypes.c_void_p] + [_I32P] * 6
        L.oc_oracle_obs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_int, _I32P, _I32P, _F64P]
        L.oc_oracle_batch_step.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int64,
                                           ctypes.c_int64, ctypes.c_int64, _I32P, _I32P, _I32P,
                                           _F64P, ctypes.c_int]
        L.oc_oracle_batch_multi_step.argtypes = (
            [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
             _I32P, _I32P] + [ctypes.c_int] * 7 + [_I32P, _F64P, _F64P, _I32P, ctypes.c_int])
        L.oc_oracle_replay_base.argtypes = ([ctypes.c_void_p, ctypes.c_int64] + [_I32P] * 12 + [_F64P, _I32P])
        L.oc_oracle_replay_wrapper.argtypes = ([ctypes.c_void_p, ctypes.c_int64] + [_I32P] * 5
                                               + [ctypes.c_int] * 7 + [_I32P, _F64P, _F64P, _I32P])
        L.oc_oracle_obs_image.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(ctypes.c_int8), _I32P]
        L.oc_oracle_set_placement.argtypes = [ctypes.c_void_p, _I32P]
        L.oc_oracle_batch_set_placement.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int64, _I32P]
        L.oc_oracle_batch_multi_rollout.argtypes = (
            [ctypes.POINTER(ctypes.c_void_p)] + [ctypes.c_int64] * 4 + [_I32P, _I32P]
            + [ctypes.c_int] * 7 + [_I32P, _F64P, _F64P, _I32P, ctypes.c_int])
        L.oc_oracle_batch_rollout.argtypes = (
            [ctypes.POINTER(ctypes.c_void_p)] + [ctypes.c_int64] * 4 + [ctypes.c_int, _I32P, _I32P,
                                                                         _I32P, _F64P, ctypes.c_int])
        L.oc_oracle_batch_snapshot.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int64] + [_I32P] * 7
        L.oc_oracle_batch_reset.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int64, _I32P]
        L.oc_oracle_pyset_order.argtypes = [_I32P, ctypes.c_int, _I32P]
        L.oc_oracle_debug_set.argtypes = [ctypes.c_void_p, _I32P, _I32P, _I32P, _I32P]
        _lib = L
    return _lib


def _p32(a):
    return a.ctypes.data_as(_I32P)


def _p64(a):
    return a.ctypes.data_as(_F64P)


class OracleEnv:
    """One environment instance of the oracle."""

    def __init__(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.int32)
        self.blob = blob
        self.A, self.M, self.S = int(blob[4]), int(blob[5]), int(blob[6])
        self._h = lib().oc_oracle_create(_p32(blob), int(blob.size))
        if not self._h:
            raise ValueError("bad level blob")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oc_oracle_destroy(self._h)
            self._h = None

    def reset(self):
        lib().oc_oracle_reset(self._h)

    def replay_base(self, actions, reset_before, pl_index=None, placements=None):
        """Run a recorded base-env tape in one C call; returns per-step arrays."""
        K = len(actions)
        A, M, S = self.A, self.M, self.S
        act = np.ascontiguousarray(actions, dtype=np.int32).reshape(K, A)
        rb = np.ascontiguousarray(reset_before, dtype=np.int32)
        pi = np.ascontiguousarray(pl_index if pl_index is not None else np.zeros(K), dtype=np.int32)
        pl = None if placements is None else np.ascontiguousarray(placements, dtype=np.int32)
        out = {"items": np.zeros((K, M, 5), np.int32), "order": np.zeros((K, M), np.int32),
               "agents": np.zeros((K, A, 3), np.int32), "misc": np.zeros((K, 2), np.int32),
               "completed": np.zeros((K, S), np.int32), "goal_count": np.zeros((K, S), np.int32),
               "reward": np.zeros(K, np.int32), "done": np.zeros(K, np.int32),
               "shaping": np.zeros((K, 2), np.float64), "error": np.zeros(K, np.int32)}
        lib().oc_oracle_replay_base(
            self._h, K, _p32(act), _p32(rb), _p32(pi), _p32(pl) if pl is not None else None,
            _p32(out["items"]), _p32(out["order"]), _p32(out["agents"]), _p32(out["misc"]),
            _p32(out["completed"]), _p32(out["goal_count"]), _p32(out["reward"]), _p32(out["done"]),
            _p64(out["shaping"]), _p32(out["error"]))
        out["t"], out["nobj"] = out["misc"][:, 0], out["misc"][:, 1]
        return out

    def replay_wrapper(self, actions, reset_before, radius, blind_mask, C, communication_on, ego_led,
                       ego_agent_idx, can_move_mask, pl_index=None, placements=None, comm=None):
        K = len(actions)
        F = 22 + self.S + 2 * C
        act = np.ascontiguousarray(actions, dtype=np.int32).reshape(K, 4)
        rb = np.ascontiguousarray(reset_before, dtype=np.int32)
        pi = np.ascontiguousarray(pl_index if pl_index is not None else np.zeros(K), dtype=np.int32)
        pl = None if placements is None else np.ascontiguousarray(placements, dtype=np.int32)
        cm = np.zeros(2, np.int32) if comm is None else np.ascontiguousarray(comm, dtype=np.int32)
        obs = np.zeros((K, 2, F), np.int32)
        ts = np.zeros((K, 2), np.float64)
        rew = np.zeros(K, np.float64)
        done = np.zeros(K, np.int32)
        lib().oc_oracle_replay_wrapper(
            self._h, K, _p32(act), _p32(rb), _p32(pi), _p32(pl) if pl is not None else None, _p32(cm),
            radius, blind_mask, C, int(communication_on), int(ego_led), ego_agent_idx, can_move_mask,
            _p32(obs), _p64(ts), _p64(rew), _p32(done))
        return obs, ts, rew, done

    def obs_image(self, viewer, radius):
        W, H = int(self.blob[2]), int(self.blob[3])
        out = np.zeros((7, W, H), np.int8)
        hold = np.zeros(2, np.int32)
        lib().oc_oracle_obs_image(self._h, viewer, radius,
                                  out.ctypes.data_as(ctypes.POINTER(ctypes.c_int8)), _p32(hold))
        return out, hold

    def set_placement(self, cells):
        """Packed start cells (x | y<<4 per item, world order) for the following resets."""
        if cells is None:
            lib().oc_oracle_set_placement(self._h, None)
        else:
            a = np.ascontiguousarray(cells, dtype=np.int32)
            lib().oc_oracle_set_placement(self._h, _p32(a))

    def debug_set(self, agents, items, completed=None, goal_count=None):
        """TEST ONLY: stage a state of single-content objects (oc_oracle_debug_set): agents
        [A][3] = x, y, held item (-1); items [M][3] = x, y, state_index; completed / goal_count
        [S] (default zeros)."""
        a = np.ascontiguousarray(agents, dtype=np.int32).reshape(self.A, 3)
        it = np.ascontiguousarray(items, dtype=np.int32).reshape(self.M, 3)
        cs = np.ascontiguousarray(completed if completed is not None else np.zeros(self.S), dtype=np.int32)
        gc = np.ascontiguousarray(goal_count if goal_count is not None else np.zeros(self.S), dtype=np.int32)
        lib().oc_oracle_debug_set(self._h, _p32(a), _p32(it), _p32(cs), _p32(gc))

    def step(self, actions):
        act = np.ascontiguousarray(actions, dtype=np.int32)
        r = ctypes.c_int32()
        d = ctypes.c_int32()
        sh = np.zeros(2, dtype=np.float64)
        lib().oc_oracle_step(self._h, _p32(act), ctypes.byref(r), ctypes.byref(d), _p64(sh))
        return r.value, d.value, sh

    @property
    def error(self):
        return lib().oc_oracle_error(self._h)

    @property
    def successful(self):
        return lib().oc_oracle_successful(self._h)

    def snapshot(self):
        items = np.zeros((self.M, 5), np.int32)
        order = np.zeros(self.M, np.int32)
        agents = np.zeros((self.A, 3), np.int32)
        misc = np.zeros(2, np.int32)
        comp = np.zeros(self.S, np.int32)
        gc = np.zeros(self.S, np.int32)
        lib().oc_oracle_snapshot(self._h, _p32(items), _p32(order), _p32(agents), _p32(misc),
                                 _p32(comp), _p32(gc))
        return {"items": items, "order": order, "agents": agents, "t": int(misc[0]),
                "nobj": int(misc[1]), "completed": comp, "goal_count": gc}

    def obs(self, viewer, radius, viewer_blind, ego_blind, C, comm):
        out = np.zeros(22 + self.S + 2 * C, np.int32)
        ts = ctypes.c_double()
        cm = np.ascontiguousarray(comm, dtype=np.int32)
        lib().oc_oracle_obs(self._h, viewer, radius, int(viewer_blind), int(ego_blind), C,
                            _p32(cm), _p32(out), ctypes.byref(ts))
        return out, ts.value


class OracleBatch:
    """N oracle envs driven with the same [row][n] arrays the HIP library takes.
    ``threads`` > 1 splits the env range over Python threads (ctypes drops the GIL)."""

    def __init__(self, blob, n, threads=1):
        self.blob = np.ascontiguousarray(blob, dtype=np.int32)
        self.n = int(n)
        self.A, self.M, self.S = int(blob[4]), int(blob[5]), int(blob[6])
        self.threads = max(1, int(threads))
        L = lib()
        self._handles = (ctypes.c_void_p * self.n)()
        first = L.oc_oracle_create(_p32(self.blob), int(self.blob.size))
        if not first:
            raise ValueError("bad level blob")
        self._handles[0] = first
        for i in range(1, self.n):
            self._handles[i] = L.oc_oracle_clone(first)

    def __del__(self):
        L = lib()
        for i in range(getattr(self, "n", 0)):
            if self._handles[i]:
                L.oc_oracle_destroy(self._handles[i])
                self._handles[i] = None

    def _ranges(self):
        k = min(self.threads, self.n)
        edges = [self.n * i // k for i in range(k + 1)]
        return [(edges[i], edges[i + 1]) for i in range(k)]

    def _run(self, fn):
        rs = self._ranges()
        if len(rs) == 1:
            fn(*rs[0])
            return
        ts = [threading.Thread(target=fn, args=r) for r in rs]
        for t in ts:
            t.start()
        for t in ts:
            t.join()

    def reset(self, mask=None):
        m = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.int32)
            m = _p32(mask)
        lib().oc_oracle_batch_reset(self._handles, self.n, m)

    def set_placement(self, placement):
        """[M][n] packed start cells used at each env's next reset / auto-reset."""
        if placement is None:
            lib().oc_oracle_batch_set_placement(self._handles, self.n, None)
        else:
            a = np.ascontiguousarray(placement, dtype=np.int32)
            assert a.shape == (self.M, self.n)
            lib().oc_oracle_batch_set_placement(self._handles, self.n, _p32(a))

    def snapshot_all(self):
        n, A, M, S = self.n, self.A, self.M, self.S
        items = np.zeros((n, M, 5), np.int32)
        order = np.zeros((n, M), np.int32)
        agents = np.zeros((n, A, 3), np.int32)
        misc = np.zeros((n, 2), np.int32)
        comp = np.zeros((n, S), np.int32)
        gc = np.zeros((n, S), np.int32)
        err = np.zeros(n, np.int32)
        lib().oc_oracle_batch_snapshot(self._handles, n, _p32(items), _p32(order), _p32(agents),
                                       _p32(misc), _p32(comp), _p32(gc), _p32(err))
        return {"items": items, "order": order, "agents": agents, "t": misc[:, 0].copy(),
                "nobj": misc[:, 1].copy(), "completed": comp, "goal_count": gc, "error": err}

    def step(self, actions, auto_reset=False):
        actions = np.ascontiguousarray(actions, dtype=np.int32)
        assert actions.shape == (self.A, self.n)
        reward = np.zeros(self.n, np.int32)
        done = np.zeros(self.n, np.int32)
        shaping = np.zeros((2, self.n), np.float64)
        L = lib()
        self._run(lambda a, b: L.oc_oracle_batch_step(
            self._handles, a, b, self.n, _p32(actions), _p32(reward), _p32(done), _p64(shaping),
            int(auto_reset)))
        return reward, done, shaping

    def multi_step(self, actions, comm, radius, blind_mask, C, communication_on=True,
                   ego_led=False, ego_agent_idx=0, can_move_mask=3, auto_reset=False):
        actions = np.ascontiguousarray(actions, dtype=np.int32)
        assert actions.shape == (4, self.n) and comm.shape == (2, self.n)
        F = 22 + self.S + 2 * C
        obs = np.zeros((2, F, self.n), np.int32)
        ts = np.zeros(self.n, np.float64)
        reward = np.zeros(self.n, np.float64)
        done = np.zeros(self.n, np.int32)
        L = lib()
        self._run(lambda a, b: L.oc_oracle_batch_multi_step(
            self._handles, a, b, self.n, _p32(actions), _p32(comm), radius, blind_mask, C,
            int(communication_on), int(ego_led), ego_agent_idx, can_move_mask, _p32(obs),
            _p64(ts), _p64(reward), _p32(done), int(auto_reset)))
        return obs, ts, reward, done

    def multi_rollout(self, actions, comm, radius, blind_mask, C, communication_on=True,
                      ego_led=False, ego_agent_idx=0, can_move_mask=3, auto_reset=True):
        """K wrapper steps per call, actions [K][4][n]; each thread makes ONE C call."""
        actions = np.ascontiguousarray(actions, dtype=np.int32)
        K = actions.shape[0]
        assert actions.shape == (K, 4, self.n)
        F = 22 + self.S + 2 * C
        obs = np.zeros((2, F, self.n), np.int32)
        ts = np.zeros(self.n, np.float64)
        reward = np.zeros(self.n, np.float64)
        done = np.zeros(self.n, np.int32)
        L = lib()
        self._run(lambda a, b: L.oc_oracle_batch_multi_rollout(
            self._handles, a, b, self.n, K, _p32(actions), _p32(comm), radius, blind_mask, C,
            int(communication_on), int(ego_led), ego_agent_idx, can_move_mask, _p32(obs),
            _p64(ts), _p64(reward), _p32(done), int(auto_reset)))
        return obs, ts, reward, done

    def rollout(self, actions, auto_reset=True):
        """K base-env steps per call, actions [K][A][n]."""
        actions = np.ascontiguousarray(actions, dtype=np.int32)
        K = actions.shape[0]
        assert actions.shape == (K, self.A, self.n)
        reward = np.zeros(self.n, np.int32)
        done = np.zeros(self.n, np.int32)
        shaping = np.zeros((2, self.n), np.float64)
        L = lib()
        self._run(lambda a, b: L.oc_oracle_batch_rollout(
            self._handles, a, b, self.n, K, self.A, _p32(actions), _p32(reward), _p32(done),
            _p64(shaping), int(auto_reset)))
        return reward, done, shaping

    def snapshot(self, i):
        e = OracleEnv.__new__(OracleEnv)
        e.A, e.M, e.S = self.A, self.M, self.S
        e._h = self._handles[i]
        try:
            return e.snapshot()
        finally:
            e._h = None

    def obs(self, i, viewer, radius, viewer_blind, ego_blind, C, comm):
        e = OracleEnv.__new__(OracleEnv)
        e.A, e.M, e.S = self.A, self.M, self.S
        e._h = self._handles[i]
        try:
            return e.obs(viewer, radius, viewer_blind, ego_blind, C, comm)
        finally:
            e._h = None


_TILE_GLYPH = [" ", "-", "/", "*"]          # Floor, Counter, Cutboard, Delivery (utils/core.py:18-26)
_TYPE_LETTER = ["t", "l", "o", "p"]


def render_ascii(blob, snap):
    """``str(OvercookedEnvironment)`` from an oracle snapshot -- the checker's own rendering,
    derived from the reference's display code, not from the product's:
    ``World.update_display`` (gym_cooking/utils/world.py:38-48: every object of
    ``world.objects`` in iteration order writes ``str(obj)`` at its location, the Tomato-named
    ones once more), then the agents (overcooked_environment.py:442-446), joined as in
    ``__str__`` (:62-65).  ``str(Object)`` = its contents sorted by name, ``-``-joined;
    a Food prints ``<state_index + 1><first letter>``, a Plate ``p`` (utils/core.py:149-377).
    Grid squares and movable objects share ``world.objects``; a grid square's key is created
    before any object can sit on it is drawn, and agents are drawn last, so drawing tiles first
    gives the same picture."""
    blob = np.asarray(blob)
    W, H, M = int(blob[2]), int(blob[3]), int(blob[5])
    cells = blob[int(blob[16]):int(blob[16]) + W * H].reshape(H, W)
    types = [int(blob[int(blob[19]) + 3 * i]) for i in range(M)]
    rep = [[_TILE_GLYPH[int(cells[y][x])] for x in range(W)] for y in range(H)]
    items, order, agents = snap["items"], snap["order"], snap["agents"]

    def obj_str(g):
        members = sorted((i for i in range(M) if items[i][3] == g),
                         key=lambda i: ["Tomato", "Lettuce", "Onion", "Plate"][types[i]])
        return "-".join("p" if types[i] == 3 else "%d%s" % (items[i][2] + 1, _TYPE_LETTER[types[i]])
                        for i in members)

    groups = [int(g) for g in order if g >= 0]
    for g in groups:
        rep[int(items[g][1])][int(items[g][0])] = obj_str(g)
    for g in groups:                       # `for obj in self.objects["Tomato"]` (world.py:46-47)
        if [types[i] for i in range(M) if items[i][3] == g] == [0]:
            rep[int(items[g][1])][int(items[g][0])] = obj_str(g)
    for a in range(len(agents)):
        rep[int(agents[a][1])][int(agents[a][0])] = str(a)
    return "\n".join("".join(c + " " for c in row) for row in rep)


def pyset_order(locs):
    """The oracle's restatement of ``list(set(locs))`` for (x, y) locations, as [(x, y), ...]."""
    a = np.ascontiguousarray(locs, dtype=np.int32).reshape(-1, 2)
    out = np.zeros((max(len(a), 1), 2), np.int32)
    n = lib().oc_oracle_pyset_order(_p32(a), len(a), _p32(out))
    return [tuple(int(v) for v in out[i]) for i in range(n)]

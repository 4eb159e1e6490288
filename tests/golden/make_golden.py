#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Build-container only (needs /root/reference).  Run as

    cd /root/repo && PYTHONHASHSEED=0 python tests/golden/make_golden.py

PYTHONHASHSEED=0 matters: the reference's subtask order comes from iterating a
Python ``set`` of Action objects (recipe_planner/stripsworld.py:72-77,
envs/overcooked_environment.py:457), so it changes with the hash seed.  Every
fixture records the order it was generated with.

Fixtures (all ``.npz``, compressed, a few hundred KB in total):

  base_<level>_a<A>.npz     gym_cooking.envs.OvercookedEnvironment.step/reset
                            (overcooked_environment.py:180-241): per step the
                            action codes fed in and the full post-step state,
                            reward, done, completed_subtasks, goal_objects_count
                            and the raw fp64 bits of both shaping terms.
  wrap_<name>.npz           gym_comm OvercookedMultiEnv.multi_step/multi_reset
                            (gym_comm/envs/overcooked_env.py:207-297): per step
                            the 11 observation fields of both viewers, the raw
                            fp64 bits of the shaped reward, done.

Action codes used in the tapes: 0=(0,1) 1=(0,-1) 2=(-1,0) 3=(1,0) 4=(0,0)
(0..3 are World.NAV_ACTIONS in order, utils/world.py:16).
"""
import json
import os
import random
import sys
from collections import deque

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as H  # noqa: E402

NAV = [(0, 1), (0, -1), (-1, 0), (1, 0), (0, 0)]
LETTER = {"D": 0, "U": 1, "L": 2, "R": 3, "N": 4}


def codes(s):
    return [LETTER[c] for c in s]


# ---- known-answer scripts ----------
KAT1_TOMATO = list(zip(codes("DDDRR" + "N" * 18), codes("RULLLLLDDDDRRRRDLLLLUUL")))
KAT2_SALAD = [(LETTER[a], LETTER[b]) for a, b in [
    "NR", "NU", "NL", "NL", "RR", "LR", "LL", "UL", "RD", "RD", "LD", "LD",
    "UR", "DR", "DL", "DL", "DN", "RN", "RN", "LN", "UN", "UN", "LN"]]


# ---- a purposeful random policy, so tapes reach chops/merges/deliveries ----
class Purposeful:
    """Each agent repeatedly picks a random non-floor target cell next to a
    reachable floor cell, walks to it along a BFS path over floor cells (other
    agents ignored, so collisions happen), presses into it, and re-targets.
    With probability eps an action is replaced by a uniform 5-way draw."""

    def __init__(self, cells, n_agents, rng, eps=0.1):
        self.cells = cells
        self.h = len(cells)
        self.w = len(cells[0])
        self.rng = rng
        self.eps = eps
        self.plan = [deque() for _ in range(n_agents)]
        self.targets = []
        for y in range(self.h):
            for x in range(self.w):
                if cells[y][x] != 0 and any(self._floor(x + dx, y + dy) for dx, dy in NAV[:4]):
                    self.targets.append((x, y))

    def _floor(self, x, y):
        return 0 <= x < self.w and 0 <= y < self.h and self.cells[y][x] == 0

    def _path_to(self, src, target, interesting):
        # BFS over floor from src to any floor cell adjacent to target
        prev = {src: None}
        q = deque([src])
        goal = None
        while q:
            c = q.popleft()
            for k, (dx, dy) in enumerate(NAV[:4]):
                if (c[0] + dx, c[1] + dy) == target:
                    goal = (c, k)
                    q.clear()
                    break
            else:
                for k, (dx, dy) in enumerate(NAV[:4]):
                    n = (c[0] + dx, c[1] + dy)
                    if self._floor(*n) and n not in prev:
                        prev[n] = (c, k)
                        q.append(n)
        if goal is None:
            return None
        c, last = goal
        acts = [last]
        while prev[c] is not None:
            c, k = prev[c]
            acts.append(k)
        acts.reverse()
        return acts

    def act(self, env, ai):
        if self.rng.random() < self.eps:
            return self.rng.randrange(5)
        if not self.plan[ai]:
            a = env.sim_agents[ai]
            # bias towards cells holding items / cutboards / delivery
            hot = [tuple(o.location) for o in H.world_objects(env) if not o.is_held]
            hot += [(x, y) for (x, y) in self.targets if self.cells[y][x] in (2, 3)]
            pool = hot if (hot and self.rng.random() < 0.7) else self.targets
            for _ in range(8):
                t = pool[self.rng.randrange(len(pool))]
                p = self._path_to(tuple(a.location), t, hot)
                if p is not None:
                    self.plan[ai] = deque(p)
                    break
            if not self.plan[ai]:
                return self.rng.randrange(5)
        return self.plan[ai].popleft()


def run_base(level, A, T, tapes, ego_config=None, partner_config=None, play=False):
    """tapes: list of (kind, spec).  Returns dict of arrays + static json."""
    arg = H.make_arglist(level, A, T, ego_config=ego_config, partner_config=partner_config, play=play)
    env = H.base_env(arg)
    static = H.static_tables(env)
    static.update(level=level, num_agents=A, max_num_timesteps=T, play=bool(play),
                  ego_config=arg.ego_config, partner_config=arg.partner_config,
                  hashseed=os.environ.get("PYTHONHASHSEED", "unset"))
    S = len(env.all_subtasks)
    rec = {k: [] for k in ("actions", "items", "order", "agents", "nobj", "t", "reward",
                           "done", "completed", "goal_count", "shaping", "reset_before",
                           "tape_id", "pl_index")}
    names = [a.name for a in env.sim_agents]

    placements = []

    def fresh():
        with H.quiet():
            env.reset()
        # item start cells of this episode, world order (random-* levels scatter them,
        # overcooked_environment.py:157-173)
        placements.append([[o.location[0], o.location[1]] for o in H.world_objects(env)])
        return H.base_items(env)

    for tape_id, (kind, spec) in enumerate(tapes):
        items = fresh()
        need_reset_flag = 1
        if kind == "script":
            steps = len(spec)
            rng = None
        else:
            steps, seed = spec
            rng = random.Random(seed)
        pol = Purposeful(static["cells"], A, rng) if kind == "purpose" else None
        for k in range(steps):
            if kind == "script":
                acts = list(spec[k])
                acts += [4] * (A - len(acts))
            elif kind == "rand5":
                acts = [rng.randrange(5) for _ in range(A)]
            elif kind == "rand4":
                acts = [rng.randrange(4) for _ in range(A)]
            elif kind == "purpose":
                acts = [pol.act(env, ai) for ai in range(A)]
            else:
                raise ValueError(kind)
            with H.quiet():
                r, d, info = env.step({n: NAV[a] for n, a in zip(names, acts)})
            rows, order, agents, nobj = H.snapshot(env, items)
            rec["actions"].append(acts)
            rec["items"].append(rows)
            rec["order"].append(order)
            rec["agents"].append(agents)
            rec["nobj"].append(nobj)
            rec["t"].append(env.t)
            rec["reward"].append(int(r))
            rec["done"].append(1 if d else 0)
            rec["completed"].append(list(env.completed_subtasks))
            rec["goal_count"].append(list(env.goal_objects_count))
            rec["shaping"].append([H.f64_bits(info["agent_0_reward_shaping"]),
                                   H.f64_bits(info["agent_1_reward_shaping"])])
            rec["reset_before"].append(need_reset_flag)
            rec["pl_index"].append(len(placements) - 1)
            rec["tape_id"].append(tape_id)
            need_reset_flag = 0
            if d:
                items = fresh()
                need_reset_flag = 1
                if pol is not None:
                    pol.plan = [deque() for _ in range(A)]
    out = {
        "static_json": np.array(json.dumps(static)),
        "actions": np.array(rec["actions"], dtype=np.int8),
        "items": np.array(rec["items"], dtype=np.int8),
        "order": np.array(rec["order"], dtype=np.int8),
        "agents": np.array(rec["agents"], dtype=np.int8),
        "nobj": np.array(rec["nobj"], dtype=np.int8),
        "t": np.array(rec["t"], dtype=np.int32),
        "reward": np.array(rec["reward"], dtype=np.int32),
        "done": np.array(rec["done"], dtype=np.int8),
        "completed": np.array(rec["completed"], dtype=np.int8).reshape(-1, S),
        "goal_count": np.array(rec["goal_count"], dtype=np.int8).reshape(-1, S),
        "shaping_bits": np.array(rec["shaping"], dtype=np.uint64),
        "reset_before": np.array(rec["reset_before"], dtype=np.int8),
        "tape_id": np.array(rec["tape_id"], dtype=np.int16),
        # one row per env.reset() call, in call order (= every step with reset_before == 1)
        "placements": np.array(placements, dtype=np.int8),
        "pl_index": np.array(rec["pl_index"], dtype=np.int16),   # row of `placements` in force at step k
    }
    return out, int(np.sum(out["reward"])), int(np.sum(out["done"]))


OBS_KEYS = ["timestep", "object_encodings_x", "object_encodings_y", "state_encodings",
            "is_hidden", "completed_subtasks", "agent1_location", "agent2_location",
            "agent_is_holding", "agent1_comm", "agent2_comm"]


def flat_obs(o):
    """11 fields -> (timestep f64 bits, int64 vector of everything else, dtype tags)."""
    ts = H.f64_bits(float(o["timestep"][0]))
    vec = []
    tags = []
    for k in OBS_KEYS[1:]:
        a = np.asarray(o[k])
        tags.append(str(a.dtype))
        # comm vectors are float64 one-hots; every other field is integral
        vec.extend(int(v) for v in a.reshape(-1))
        assert all(float(v) == int(v) for v in a.reshape(-1)), (k, a)
    return ts, vec, tags


def run_wrapper(name, level, T, steps, seed, kind="purpose", ego_agent_idx=0, **cfg):
    arg = H.make_arglist(level, 2, T, **cfg)
    env = H.wrapper_env(arg, ego_agent_idx=ego_agent_idx)
    base = env.base_env
    static = H.static_tables(base)
    C = arg.num_communication
    static.update(level=level, num_agents=2, max_num_timesteps=T, play=bool(cfg.get("play", False)),
                  ego_config=arg.ego_config, partner_config=arg.partner_config,
                  num_communication=C, communication_on=arg.communication_on,
                  ego_led=arg.ego_led, fow_radius=arg.fow_radius,
                  ego_agent_idx=ego_agent_idx,
                  hashseed=os.environ.get("PYTHONHASHSEED", "unset"))
    rng = random.Random(seed)
    pol = Purposeful(static["cells"], 2, rng, eps=0.15) if kind == "purpose" else None
    rec = {k: [] for k in ("actions", "ts_bits", "obs", "rew_bits", "done", "reset_before", "pl_index")}
    placements = []
    # the constructor already did multi_reset(); record the obs of a fresh multi_reset
    with H.quiet():
        o0, o1 = env.multi_reset()
    placements.append([[o.location[0], o.location[1]] for o in H.world_objects(base)])
    t0, v0, tags = flat_obs(o0)
    t1, v1, _ = flat_obs(o1)
    reset_obs = {"ts_bits": [t0, t1], "obs": [v0, v1]}
    need_reset_flag = 1
    for k in range(steps):
        if pol is not None:
            mv = [pol.act(base, 0), pol.act(base, 1)]
            mv = [m if m < 4 else rng.randrange(4) for m in mv]   # wrapper has no no-op index
        else:
            mv = [rng.randrange(4), rng.randrange(4)]
        cm = [rng.randrange(C), rng.randrange(C)]
        # multi_step(ego_action, alt_action): the ego drives sim agent `ego_agent_idx`
        # (overcooked_env.py:253-262).  Tapes are stored as (ego, alt).
        if ego_agent_idx == 0:
            ego, alt = (mv[0], cm[0]), (mv[1], cm[1])
        else:
            ego, alt = (mv[1], cm[0]), (mv[0], cm[1])
        with H.quiet():
            (o0, o1), (r0, r1), d, _ = env.multi_step(ego, alt)
        assert H.f64_bits(r0) == H.f64_bits(r1)
        t0, v0, _ = flat_obs(o0)
        t1, v1, _ = flat_obs(o1)
        rec["actions"].append([ego[0], ego[1], alt[0], alt[1]])
        rec["ts_bits"].append([t0, t1])
        rec["obs"].append([v0, v1])
        rec["rew_bits"].append(H.f64_bits(r0))
        rec["done"].append(1 if d else 0)
        rec["reset_before"].append(need_reset_flag)
        rec["pl_index"].append(len(placements) - 1)
        need_reset_flag = 0
        if d:
            with H.quiet():
                env.multi_reset()
            placements.append([[o.location[0], o.location[1]] for o in H.world_objects(base)])
            need_reset_flag = 1
            if pol is not None:
                pol.plan = [deque(), deque()]
    static["obs_dtypes"] = dict(zip(OBS_KEYS[1:], tags))
    out = {
        "static_json": np.array(json.dumps(static)),
        "actions": np.array(rec["actions"], dtype=np.int8),
        "ts_bits": np.array(rec["ts_bits"], dtype=np.uint64),
        "obs": np.array(rec["obs"], dtype=np.int16),
        "rew_bits": np.array(rec["rew_bits"], dtype=np.uint64),
        "done": np.array(rec["done"], dtype=np.int8),
        "reset_before": np.array(rec["reset_before"], dtype=np.int8),
        "reset_ts_bits": np.array(reset_obs["ts_bits"], dtype=np.uint64),
        "reset_obs": np.array(reset_obs["obs"], dtype=np.int16),
        "placements": np.array(placements, dtype=np.int8),
        "pl_index": np.array(rec["pl_index"], dtype=np.int16),
    }
    return out, int(np.sum(out["done"]))


def run_ascii(level, A, T, script):
    """str(env) (overcooked_environment.py:62-65) and holdings after every scripted step."""
    env = H.base_env(H.make_arglist(level, A, T))
    names = [a.name for a in env.sim_agents]
    out = {"level": level, "num_agents": A, "T": T, "script": [list(s) for s in script],
           "subtasks": [[st.name, list(st.args)] for st in env.all_subtasks],
           "reset_str": str(env), "steps": []}
    for acts in script:
        acts = list(acts) + [4] * (A - len(acts))
        with H.quiet():
            r, d, info = env.step({n: NAV[a] for n, a in zip(names, acts)})
        out["steps"].append({"str": str(env), "holding": [a.get_holding() for a in env.sim_agents],
                             "locations": [list(a.location) for a in env.sim_agents],
                             "termination_info": env.termination_info, "successful": env.successful,
                             "reward": int(r), "done": bool(d)})
    return out


def run_fow(level, T, steps, seed, radius):
    """OvercookedMultiEnv.get_partial_observability_FOW (overcooked_env.py:161-202; not called
    by the reference's live code) for both viewers along a goal-directed tape."""
    arg = H.make_arglist(level, 2, T)
    env = H.wrapper_env(arg)
    base = env.base_env
    static = H.static_tables(base)
    static.update(level=level, num_agents=2, max_num_timesteps=T, radius=radius,
                  ego_config=arg.ego_config, partner_config=arg.partner_config,
                  hashseed=os.environ.get("PYTHONHASHSEED", "unset"))
    rng = random.Random(seed)
    pol = Purposeful(static["cells"], 2, rng, eps=0.1)
    names = [a.name for a in base.sim_agents]
    rec = {k: [] for k in ("actions", "maps", "holding", "completed", "reset_before")}
    with H.quiet():
        base.reset()
    flag = 1

    def grab():
        out = []
        for v in range(2):
            with H.quiet():
                o = env.get_partial_observability_FOW(v, radius=radius)
            out.append(o)
        rec["maps"].append([np.asarray(o["blockworld_map"]) for o in out])
        assert (out[0]["agent_is_holding"] == out[1]["agent_is_holding"]).all()
        rec["holding"].append([int(x) for x in out[0]["agent_is_holding"]])
        rec["completed"].append([int(x) for x in out[0]["completed_subtasks"]])

    for k in range(steps):
        acts = [pol.act(base, 0), pol.act(base, 1)]
        with H.quiet():
            r, d, info = base.step({n: NAV[a] for n, a in zip(names, acts)})
        rec["actions"].append(acts)
        rec["reset_before"].append(flag)
        flag = 0
        grab()
        if d:
            with H.quiet():
                base.reset()
            pol.plan = [deque(), deque()]
            flag = 1
    out = {"static_json": np.array(json.dumps(static)),
           "actions": np.array(rec["actions"], dtype=np.int8),
           "maps": np.array(rec["maps"], dtype=np.int8),          # [K][2][7][X][Y]
           "holding": np.array(rec["holding"], dtype=np.int8),
           "completed": np.array(rec["completed"], dtype=np.int8),
           "reset_before": np.array(rec["reset_before"], dtype=np.int8)}
    return out


CUSTOM_LEVELS = {
    # our own maps (not reference files): they pin what no shipped level exercises --
    # OnionSalad / the Onion channel (29 subtasks, 5 items), two recipes that share
    # subtasks, and two Delivery tiles (only the first one pays, shaping takes the nearer)
    "custom-onion_salad": "-o---t-\n/     l\n/     -\n*     -\n-     -\n-     p\n-----p-\n\nOnionSalad\n\n2 1\n4 1\n4 4\n2 4",
    "custom-two_recipes": "-----t-\n/     l\n/     -\n*     -\n-     -\n-     p\n-----p-\n\nSalad\nSimpleTomato\n\n2 1\n4 1\n4 4\n2 4",
    "custom-two_deliveries": "--*-t-\n/    l\n*    p\n-    p\n------\n\nSimpleTomato\n\n1 1\n3 2\n4 3",
}

DUP_LEVELS = {
    # maps that REPEAT a content type (no shipped level does): the reference's engine handles
    # multisets -- mergeable() has no duplicate check (utils/core.py:240-257), world.objects is a
    # dict of lists (utils/world.py:236-247), goal counts go above 1
    # (overcooked_environment.py:408-415) and the `[0]` of a location *set* picks the goal object
    # the shaping walks to (:287,:374)
    "custom-two_tomatoes": "-t---t-\n/     l\n/     -\n*     -\n-     -\n-     p\n-----p-\n\nSimpleTomato\n\n2 1\n4 1\n4 4\n2 4",
    "custom-two_lettuces_salad": "-l---t-\n/     l\n/     -\n*     -\n-     -\n-     p\n-----p-\n\nSalad\n\n2 1\n4 1\n4 4\n2 4",
    "custom-two_tomatoes_small": "-t/t-\n-   p\n*   p\n-----\n\nSimpleTomato\n\n1 1\n3 2\n2 1",
    # THREE of a type (round 3): the count == 3 paths -- the 2-bit goal counts at their
    # maximum, three candidates in the set-order lookup, five items
    "custom-three_tomatoes": "-t-t-t-\n/     -\n/     -\n*     -\n-     -\n-     p\n-----p-\n\nSimpleTomato\n\n2 1\n4 1\n4 4\n2 4",
}


def main_play(summary):
    """arglist.play = True: the "playable" branches of interact() (utils/interact.py:44-47,52,
    66-67) -- a merge lands on the counter, a fresh food is put down on a Cutboard and chopped by
    the next empty-handed press."""
    for level, A, T, tapes in [("open-divider_tomato", 2, 150, [("purpose", (6000, 90))]),
                               ("full-divider_salad", 2, 200, [("purpose", (8000, 91))]),
                               ("partial-divider_tl", 3, 150, [("purpose", (5000, 92))])]:
        out, sr, nd = run_base(level, A, T, tapes, play=True)
        fn = "pbase_%s_a%d.npz" % (level, A)
        np.savez_compressed(os.path.join(HERE, fn), **out)
        summary[fn] = {"steps": int(len(out["t"])), "sum_reward": sr, "episodes": nd}
        print(fn, summary[fn], flush=True)
    for name, level, T, steps, seed, kw in [("play_tomato_r2", "open-divider_tomato", 150, 5000, 150, {"play": True}),
                                            ("play_salad_c3", "open-divider_salad", 200, 6000, 151,
                                             {"play": True, "num_communication": 3, "fow_radius": 1})]:
        out, nd = run_wrapper(name, level, T, steps, seed, **kw)
        fn = "pwrap_%s.npz" % name
        np.savez_compressed(os.path.join(HERE, fn), **out)
        summary[fn] = {"steps": int(len(out["done"])), "episodes": nd}
        print(fn, summary[fn], flush=True)


def main_dup(summary, only3=False):
    H.use_custom_levels(DUP_LEVELS)
    jobs = [("custom-three_tomatoes", 2, 200, [("purpose", (9000, 86))]),
            ("custom-three_tomatoes", 3, 150, [("purpose", (5000, 87))])]
    jobs += [] if only3 else [("custom-two_tomatoes", 2, 200, [("purpose", (7000, 80))]),
            ("custom-two_tomatoes", 3, 150, [("purpose", (4000, 81))]),
            ("custom-two_lettuces_salad", 2, 250, [("purpose", (9000, 82))]),
            ("custom-two_tomatoes_small", 2, 120, [("rand5", (1500, 83)), ("purpose", (6000, 84))]),
            ("custom-two_tomatoes_small", 3, 120, [("purpose", (4000, 85))])]
    for level, A, T, tapes in jobs:
        out, sr, nd = run_base(level, A, T, tapes)
        st = json.loads(str(out["static_json"]))
        st["level_text"] = DUP_LEVELS[level]
        out["static_json"] = np.array(json.dumps(st))
        fn = "cbase_dup_%s_a%d.npz" % (level.replace("custom-", ""), A)
        np.savez_compressed(os.path.join(HERE, fn), **out)
        summary[fn] = {"steps": int(len(out["t"])), "sum_reward": sr, "episodes": nd,
                       "steps_with_goal_count_above_1": int((out["goal_count"] > 1).any(axis=1).sum()),
                       "max_objects": int(out["nobj"].max()), "min_objects": int(out["nobj"].min())}
        print(fn, summary[fn], "S=%d" % len(st["subtasks"]), flush=True)
    wrap_jobs = [("dup_three_tomatoes_r2", "custom-three_tomatoes", 200, 7000, 143, {})]
    for name, level, T, steps, seed, kw in wrap_jobs + ([] if only3 else [
            ("dup_two_tomatoes_r2", "custom-two_tomatoes", 200, 6000, 140, {}),
            ("dup_two_lettuces_salad_c3", "custom-two_lettuces_salad", 250, 8000, 141, {"num_communication": 3, "fow_radius": 1}),
            ("dup_two_tomatoes_small_r1", "custom-two_tomatoes_small", 120, 5000, 142, {"fow_radius": 1})]):
        out, nd = run_wrapper(name, level, T, steps, seed, **kw)
        st = json.loads(str(out["static_json"]))
        st["level_text"] = DUP_LEVELS[level]
        out["static_json"] = np.array(json.dumps(st))
        fn = "cwrap_%s.npz" % name
        np.savez_compressed(os.path.join(HERE, fn), **out)
        summary[fn] = {"steps": int(len(out["done"])), "episodes": nd}
        print(fn, summary[fn], flush=True)


def main_custom(summary):
    H.use_custom_levels(CUSTOM_LEVELS)
    jobs = [("custom-onion_salad", 2, 200, [("purpose", (6000, 70))]),
            ("custom-onion_salad", 3, 150, [("purpose", (3000, 71))]),
            ("custom-two_recipes", 2, 150, [("purpose", (4000, 72))]),
            ("custom-two_deliveries", 2, 80, [("rand5", (500, 73)), ("purpose", (3000, 74))]),
            ("custom-two_deliveries", 3, 80, [("purpose", (2000, 75))])]
    for level, A, T, tapes in jobs:
        out, sr, nd = run_base(level, A, T, tapes)
        st = json.loads(str(out["static_json"]))
        st["level_text"] = CUSTOM_LEVELS[level]
        out["static_json"] = np.array(json.dumps(st))
        fn = "cbase_%s_a%d.npz" % (level, A)
        np.savez_compressed(os.path.join(HERE, fn), **out)
        summary[fn] = {"steps": int(len(out["t"])), "sum_reward": sr, "episodes": nd}
        print(fn, summary[fn], "S=%d" % len(st["subtasks"]), flush=True)
    for name, level, T, steps, seed, kw in [("conion_r2", "custom-onion_salad", 200, 4000, 130, {"num_communication": 3}),
                                            ("ctwodeliv_r1", "custom-two_deliveries", 80, 2000, 131, {"fow_radius": 1})]:
        out, nd = run_wrapper(name, level, T, steps, seed, **kw)
        st = json.loads(str(out["static_json"]))
        st["level_text"] = CUSTOM_LEVELS[level]
        out["static_json"] = np.array(json.dumps(st))
        fn = "cwrap_%s.npz" % name
        np.savez_compressed(os.path.join(HERE, fn), **out)
        summary[fn] = {"steps": int(len(out["done"])), "episodes": nd}
        print(fn, summary[fn], flush=True)


def main_random(summary):
    """random-* levels: items scattered on random Counters at every
    reset with Python's global `random` (seeded here so the run is repeatable)."""
    base_jobs = [
        ("random-open-divider_salad_small", 2, 80, [("rand5", (600, 50)), ("purpose", (2500, 60))]),
        ("random-open-divider_salad_small_wide", 2, 100, [("purpose", (2500, 61))]),
        ("random-salad-superwide", 2, 100, [("rand5", (400, 52)), ("purpose", (2500, 62))]),
        ("random-open-divider_tomato", 3, 80, [("purpose", (2500, 63))]),
        ("random-full-divider_salad", 2, 80, [("purpose", (1500, 64))]),
        ("random-open-divider_salad_small_wide_big", 2, 100, [("purpose", (2000, 65))]),
    ]
    for level, A, T, tapes in base_jobs:
        random.seed(1000 + A)
        out, sr, nd = run_base(level, A, T, tapes)
        fn = "rbase_%s_a%d.npz" % (level, A)
        np.savez_compressed(os.path.join(HERE, fn), **out)
        summary[fn] = {"steps": int(len(out["t"])), "sum_reward": sr, "episodes": nd}
        print(fn, summary[fn], flush=True)
    wrap_jobs = [
        ("rsmall_r2", "random-open-divider_salad_small", 80, 2500, 120, {}),
        ("rsuperwide_c5", "random-salad-superwide", 100, 2000, 121, {"num_communication": 5}),
        ("rwide_r1", "random-open-divider_salad_small_wide", 100, 2000, 122, {"fow_radius": 1}),
    ]
    for name, level, T, steps, seed, kw in wrap_jobs:
        random.seed(seed)
        out, nd = run_wrapper(name, level, T, steps, seed, **kw)
        fn = "rwrap_%s.npz" % name
        np.savez_compressed(os.path.join(HERE, fn), **out)
        summary[fn] = {"steps": int(len(out["done"])), "episodes": nd}
        print(fn, summary[fn], flush=True)


def main():
    assert os.environ.get("PYTHONHASHSEED") == "0", "run with PYTHONHASHSEED=0"
    summary = {}
    if "--fow-only" in sys.argv:
        for name, level, radius, seed in (("tomato_r2", "open-divider_tomato", 2, 300),
                                          ("salad_r3", "full-divider_salad", 3, 301)):
            out = run_fow(level, 100, 1500, seed, radius)
            np.savez_compressed(os.path.join(HERE, "fow_%s.npz" % name), **out)
            print("fow_%s.npz" % name, out["maps"].shape, int(out["completed"].sum()))
        return
    if "--play-only" in sys.argv:
        with open(os.path.join(HERE, "SUMMARY.json")) as f:
            summary = json.load(f)
        main_play(summary)
        with open(os.path.join(HERE, "SUMMARY.json"), "w") as f:
            json.dump(summary, f, indent=1, sort_keys=True)
        return
    if "--dup-only" in sys.argv or "--dup3-only" in sys.argv:
        with open(os.path.join(HERE, "SUMMARY.json")) as f:
            summary = json.load(f)
        main_dup(summary, only3="--dup3-only" in sys.argv)
        with open(os.path.join(HERE, "SUMMARY.json"), "w") as f:
            json.dump(summary, f, indent=1, sort_keys=True)
        return
    if "--custom-only" in sys.argv:
        with open(os.path.join(HERE, "SUMMARY.json")) as f:
            summary = json.load(f)
        main_custom(summary)
        with open(os.path.join(HERE, "SUMMARY.json"), "w") as f:
            json.dump(summary, f, indent=1, sort_keys=True)
        return
    if "--random-only" in sys.argv:
        with open(os.path.join(HERE, "SUMMARY.json")) as f:
            summary = json.load(f)
        main_random(summary)
        with open(os.path.join(HERE, "SUMMARY.json"), "w") as f:
            json.dump(summary, f, indent=1, sort_keys=True)
        return
    asc = [run_ascii("open-divider_tomato", 2, 100, KAT1_TOMATO),
           run_ascii("full-divider_salad", 2, 100, KAT2_SALAD)]
    with open(os.path.join(HERE, "ascii_kat.json"), "w") as f:
        json.dump(asc, f, indent=0)
    print("ascii_kat.json", len(asc))
    if "--ascii-only" in sys.argv:
        return

    base_jobs = [
        # level, A, T, tapes
        ("open-divider_tomato", 2, 100,
         [("script", KAT1_TOMATO), ("rand5", (1500, 0)), ("rand4", (500, 10)), ("purpose", (2500, 20))]),
        ("full-divider_salad", 2, 100,
         [("script", KAT2_SALAD), ("rand5", (1500, 1)), ("purpose", (3000, 21))]),
        ("partial-divider_tl", 3, 100,
         [("rand5", (1500, 2)), ("rand4", (500, 12)), ("purpose", (3000, 22))]),
        ("open-divider_salad", 2, 150, [("rand5", (800, 3)), ("purpose", (2500, 23))]),
        ("open-divider_tl", 2, 150, [("rand5", (800, 4)), ("purpose", (2500, 24))]),
        ("open-divider_tl", 3, 100, [("rand5", (800, 6)), ("purpose", (2500, 26))]),
        ("partial-divider_salad", 3, 100, [("rand5", (800, 5)), ("purpose", (2500, 25))]),
        ("partial-divider_tomato", 4, 60, [("rand5", (600, 7)), ("purpose", (1500, 27))]),
        ("full-divider_tl", 4, 80, [("purpose", (1500, 28))]),
        ("full-divider_tomato", 2, 50, [("purpose", (1000, 29))]),
    ]
    for level, A, T, tapes in base_jobs:
        out, sr, nd = run_base(level, A, T, tapes)
        fn = "base_%s_a%d.npz" % (level, A)
        np.savez_compressed(os.path.join(HERE, fn), **out)
        summary[fn] = {"steps": int(len(out["t"])), "sum_reward": sr, "episodes": nd}
        print(fn, summary[fn], flush=True)

    # base env with an ALLERGIC ego / partner (agent.py:296-298)
    out, sr, nd = run_base("open-divider_tomato", 2, 80, [("purpose", (1200, 40))],
                           ego_config={"ALLERGIC": True})
    np.savez_compressed(os.path.join(HERE, "base_allergic_ego.npz"), **out)
    summary["base_allergic_ego.npz"] = {"steps": int(len(out["t"])), "sum_reward": sr, "episodes": nd}
    out, sr, nd = run_base("partial-divider_tl", 3, 80, [("purpose", (1200, 41))],
                           partner_config={"ALLERGIC": True})
    np.savez_compressed(os.path.join(HERE, "base_allergic_partner.npz"), **out)
    summary["base_allergic_partner.npz"] = {"steps": int(len(out["t"])), "sum_reward": sr, "episodes": nd}

    wrap_jobs = [
        # name, level, T, steps, seed, kwargs
        ("tomato_r2", "open-divider_tomato", 100, 2500, 100, {}),
        ("tomato_r0", "open-divider_tomato", 60, 800, 101, {"fow_radius": 0}),
        ("tomato_r1", "open-divider_tomato", 60, 800, 102, {"fow_radius": 1}),
        ("tomato_r1000", "open-divider_tomato", 60, 800, 103, {"fow_radius": 1000}),
        ("salad_r2", "full-divider_salad", 100, 2500, 104, {}),
        ("salad_open_c5", "open-divider_salad", 120, 2500, 105, {"num_communication": 5}),
        ("tl_open_r3", "open-divider_tl", 120, 2500, 106, {"fow_radius": 3}),
        ("tomato_commoff", "open-divider_tomato", 60, 600, 107, {"communication_on": False}),
        ("tomato_egoled", "open-divider_tomato", 60, 600, 108, {"ego_led": True}),
        ("tomato_blind_ego", "open-divider_tomato", 60, 800, 109, {"ego_config": {"BLIND": True}}),
        ("tomato_blind_partner", "open-divider_tomato", 60, 800, 110, {"partner_config": {"BLIND": True}}),
        ("tomato_allergic_ego", "open-divider_tomato", 60, 800, 111, {"ego_config": {"ALLERGIC": True}}),
        ("tomato_allergic_partner", "open-divider_tomato", 60, 800, 112, {"partner_config": {"ALLERGIC": True}}),
        ("tomato_pinned_ego", "open-divider_tomato", 60, 600, 113, {"ego_config": {"CAN_MOVE": False}}),
        ("tomato_pinned_partner", "open-divider_tomato", 60, 600, 114, {"partner_config": {"CAN_MOVE": False}}),
        ("salad_partial_rand", "partial-divider_salad", 80, 1000, 115, {"kind": "rand"}),
        ("tomato_egoidx1", "open-divider_tomato", 60, 800, 116, {"ego_agent_idx": 1}),
        ("tl_full_blind_allergic", "full-divider_tl", 60, 800, 117,
         {"ego_config": {"BLIND": True}, "partner_config": {"ALLERGIC": True}, "ego_agent_idx": 1}),
    ]
    for name, level, T, steps, seed, kw in wrap_jobs:
        kw = dict(kw)
        kind = kw.pop("kind", "purpose")
        eidx = kw.pop("ego_agent_idx", 0)
        out, nd = run_wrapper(name, level, T, steps, seed, kind=kind, ego_agent_idx=eidx, **kw)
        fn = "wrap_%s.npz" % name
        np.savez_compressed(os.path.join(HERE, fn), **out)
        summary[fn] = {"steps": int(len(out["done"])), "episodes": nd}
        print(fn, summary[fn], flush=True)

    main_random(summary)
    main_custom(summary)          # last: it changes the working directory
    with open(os.path.join(HERE, "SUMMARY.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Randomized parity soak of the FUSED wrapper kernel (oc_multi_step) on a GPU box: every
built-in level with 2 agents, a different wrapper configuration per level (fog radius,
number of comm channels, communication_on / ego_led, BLIND / ALLERGIC / CAN_MOVE, ego_agent_idx,
observation element type), thousands of envs with per-env action streams, auto-reset on; HIP vs
the CPU oracle on all 11 observation fields of both viewers, timestep and shaped-reward bits,
done flags every step, and the full state at the end.  Not part of the test-suite (minutes);
run it when the kernels change:  python tests/soak_wrapper.py [steps] [envs] [generic|spec] [config seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from gym_comm_amd import compiler, levels
from gym_comm_amd.batched import BatchedOvercooked
from hip_util import assert_snapshots_equal
from oracle import oracle

bits = lambda a: np.ascontiguousarray(a).view(np.uint64)


def run(name, cfg, n, steps, seed, spec):
    rng = np.random.default_rng(seed)
    lv = compiler.compile_level(name, 2, cfg["T"], ego_allergic=cfg["ego"]["ALLERGIC"],
                                partner_allergic=cfg["partner"]["ALLERGIC"])
    if cfg["shuffle"]:      # a random subtask order, as another PYTHONHASHSEED would give the reference
        perm = [int(v) for v in rng.permutation(lv.num_subtasks)]
        lv = compiler.compile_level(name, 2, cfg["T"], ego_allergic=cfg["ego"]["ALLERGIC"],
                                    partner_allergic=cfg["partner"]["ALLERGIC"], subtask_order=perm)
    C = cfg["C"]
    kw = {}
    ora = oracle.OracleBatch(lv.blob, n, threads=16)
    if lv.random_placement:
        place = np.zeros((lv.num_items, n), np.int32)
        for i in range(n):
            pick = rng.choice(len(lv.counters), size=len(lv.scatter_items), replace=False)
            for k, item in enumerate(lv.scatter_items):
                x, y = lv.counters[pick[k]]
                place[item, i] = x | (y << 4)
        ora.set_placement(place)
        ora.reset()
        kw["placement_mode"] = "host"
    env = BatchedOvercooked(lv, num_envs=n, auto_reset=True, specialize_level=spec,
                            ego_config=cfg["ego"], partner_config=cfg["partner"], num_communication=C,
                            communication_on=cfg["comm_on"], ego_led=cfg["ego_led"],
                            fow_radius=cfg["radius"], ego_agent_idx=cfg["ego_idx"],
                            obs_dtype=cfg["odt"], episode_stats=cfg["src"] != 0,
                            waves_per_64=cfg.get("waves", 0), **kw)
    if lv.random_placement:
        env.set_placement(torch.from_numpy(place).cuda())
        env.reset()
    blind = (1 if cfg["ego"]["BLIND"] else 0) | (2 if cfg["partner"]["BLIND"] else 0)
    can = (1 if cfg["ego"]["CAN_MOVE"] else 0) | (2 if cfg["partner"]["CAN_MOVE"] else 0)
    comm = np.zeros((2, n), np.int32)
    rsum = 0.0
    dones = 0
    for k in range(steps):
        a = np.stack([rng.integers(0, 4, n), rng.integers(0, C, n), rng.integers(0, 4, n),
                      rng.integers(0, C, n)]).astype(np.int32)
        # keep a direction for a while so that agents get somewhere
        if k % 3:
            a[0], a[2] = prev[0], prev[2]
        prev = a
        if cfg["src"] == 0:         # the four action rows
            o, t, r, d = env.multi_step(torch.from_numpy(a).cuda())
        else:                       # [n][2] pairs, int32 or int64 (oc_step_opts)
            dt = torch.int32 if cfg["src"] == 1 else torch.int64
            ego = torch.from_numpy(np.ascontiguousarray(a[0:2].T)).cuda().to(dt)
            alt = torch.from_numpy(np.ascontiguousarray(a[2:4].T)).cuda().to(dt)
            o, t, r, d = env.multi_step(None, ego_pairs=ego, alt_pairs=alt)
        oo, to, ro, do = ora.multi_step(a, comm, cfg["radius"], blind, C, communication_on=cfg["comm_on"],
                                        ego_led=cfg["ego_led"], ego_agent_idx=cfg["ego_idx"],
                                        can_move_mask=can, auto_reset=True)
        ctx = "%s step %d" % (name, k)
        assert np.array_equal(o.cpu().numpy(), oo), ctx + " obs"
        assert np.array_equal(d.cpu().numpy(), do), ctx + " done"
        assert np.array_equal(bits(r.cpu().numpy()), bits(ro)), ctx + " reward bits"
        assert np.array_equal(bits(t.cpu().numpy()), bits(to)), ctx + " timestep bits"
        rsum += float(ro.sum())
        dones += int(do.sum())
    hs, os_ = env.snapshot(), ora.snapshot_all()
    assert_snapshots_equal(hs, os_, name + " final state", where=os_["error"] == 0)
    return rsum, dones


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    use_spec = len(sys.argv) > 3 and sys.argv[3] == "spec"
    t0 = time.time()
    total = 0
    crng = np.random.default_rng(int(sys.argv[4]) if len(sys.argv) > 4 else 2026)
    for name in sorted(levels.BUILTIN):
        if name == "random-open-divider_salad_small_cramped":
            continue        # agent 0 starts boxed in at (0,0): every move is out of bounds
        flag = lambda p: bool(crng.random() < p)
        cfg = {"T": int(crng.choice([40, 100, 250])), "C": int(crng.choice([2, 2, 3, 5, 10])),
               "radius": int(crng.choice([0, 1, 2, 2, 3, 1000])), "comm_on": flag(0.8), "ego_led": flag(0.3),
               "ego_idx": int(crng.choice([0, 0, 1])), "shuffle": flag(0.6), "src": int(crng.integers(0, 3)),
               "odt": [torch.int32, torch.int32, torch.int8, torch.float32][int(crng.integers(0, 4))],
               "ego": {"ALLERGIC": flag(0.15), "BLIND": flag(0.15), "CAN_MOVE": not flag(0.1)},
               "partner": {"ALLERGIC": flag(0.15), "BLIND": flag(0.15), "CAN_MOVE": not flag(0.1)}}
        cfg["waves"] = int(crng.choice([1, 4]))   # one wave per 64 envs / the split launch (same results)
        rsum, dones = run(name, cfg, n, steps, 500 + total % 97, use_spec)
        total += n * steps
        short = {k: (str(v).replace("torch.", "") if k == "odt" else v) for k, v in cfg.items()}
        print("%-46s ok  shaped_reward_sum=%-12.1f episodes=%-6d %s (%.0fs)"
              % (name, rsum, dones, short, time.time() - t0), flush=True)
    print("wrapper soak ok (%s kernels): %d env-steps compared bit-exactly in %.0f s"
          % ("per-level specialised" if use_spec else "generic", total, time.time() - t0))


if __name__ == "__main__":
    main()

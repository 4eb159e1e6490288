#!/usr/bin/env python3
"""Long randomized parity soak on a GPU box: every built-in fixed level x 2..4 agents and
every random-* level, thousands of envs with per-env action streams, HIP vs the CPU oracle
with a full state compare every step.  Not part of the test-suite (minutes, not seconds);
run it when the kernels change:  python tests/soak.py [steps] [envs] [generic|spec]
("spec" compiles a per-level specialised library for every configuration on first use)."""
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
from hip_util import assert_snapshots_equal, bits, scripted_then_random
from oracle import oracle


def run(level, A, T, n, steps, seed, spec):
    lv = compiler.compile_level(level, A, T)
    rng = np.random.default_rng(seed)
    acts = scripted_then_random(rng, level if isinstance(level, str) else level.name, steps, A, n)
    ora = oracle.OracleBatch(lv.blob, n, threads=16)
    kw = {}
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
    env = BatchedOvercooked(lv, num_envs=n, auto_reset=True, specialize_level=spec, **kw)
    if lv.random_placement:
        env.set_placement(torch.from_numpy(place).cuda())
        env.reset()
    a_d = torch.from_numpy(acts).cuda()
    rsum = flagged = 0
    was = np.zeros(n, bool)
    lost = np.zeros(n, bool)
    for k in range(steps):
        r, d, sh = env.step(a_d[k])
        ro, do, sho = ora.step(acts[k], auto_reset=True)
        hs, os_ = env.snapshot(), ora.snapshot_all()
        # Both sides must RAISE a flag on the same env at the same step.  An env that is flagged (a
        # state where the reference itself raises or corrupts its store: from there on neither side
        # imitates it, and the two need not agree) is left out until its episode ends -- INCLUDING the
        # step that ends it: the auto-reset clears the flag, but that step's reward / shaping were
        # computed on the corrupt state.  If the two sides end that episode at DIFFERENT steps (seen
        # once: a dup level, 3 agents, step 1 164 of 1 500 envs) the env's two trajectories are out of
        # step for good: it is dropped from the rest of the run and counted.
        differ = (os_["error"] != hs["error"]) & ~lost
        assert not (differ & ~was).any(), (level, A, k, "error flags differ on an env that was clean")
        lost |= differ
        clean = (os_["error"] == 0) & (hs["error"] == 0) & ~was & ~lost
        was = (os_["error"] != 0) | (hs["error"] != 0)
        flagged += int((~clean).sum())
        ctx = "%s a%d step %d" % (level if isinstance(level, str) else level.name, A, k)
        assert np.array_equal(r.cpu().numpy()[clean], ro[clean]), ctx
        assert np.array_equal(d.cpu().numpy()[clean], do[clean]), ctx
        assert np.array_equal(bits(sh.cpu().numpy())[:, clean], bits(sho)[:, clean]), ctx
        assert_snapshots_equal(hs, os_, ctx, where=clean)
        rsum += int(ro.sum())
    if lost.any():
        print("  (%d env(s) of %s a%d dropped after their flagged episode ended at different steps on the two sides)"
              % (int(lost.sum()), level if isinstance(level, str) else level.name, A), flush=True)
    return rsum, flagged


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    use_spec = len(sys.argv) > 3 and sys.argv[3] == "spec"
    t0 = time.time()
    total = 0
    for name in sorted(levels.BUILTIN):
        spec = levels.load_level(name)
        for A in range(2, min(4, len(spec.agent_starts)) + 1):
            if name == "random-open-divider_salad_small_cramped":
                continue        # agent 0 starts boxed in at (0,0): every move is out of bounds
            rsum, flagged = run(name, A, 120, n, steps, 1000 + A, spec=use_spec)
            total += n * steps
            print("%-46s A=%d  ok  reward_sum=%-6d flagged_env_steps=%-5d  (%.0fs)"
                  % (name, A, rsum, flagged, time.time() - t0), flush=True)
    # maps that repeat a content type (the library's dup kernels): the level texts of the
    # tests/golden/c*_dup_* fixtures, 2..3 agents
    import glob
    import json
    seen = set()
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "cbase_dup_*.npz"))):
        st = json.loads(str(np.load(f)["static_json"]))
        if st["level"] in seen:
            continue
        seen.add(st["level"])
        spec_lv = levels.parse_level_text(st["level"], st["level_text"])
        for A in range(2, min(3, len(spec_lv.agent_starts)) + 1):
            rsum, flagged = run(spec_lv, A, 120, n, steps, 2000 + A, spec=use_spec)
            total += n * steps
            print("%-46s A=%d  ok  reward_sum=%-6d flagged_env_steps=%-5d  (%.0fs)"
                  % (st["level"] + " (dup)", A, rsum, flagged, time.time() - t0), flush=True)
    # the out-of-bounds level: both sides must raise OC_ERR_OOB on the same envs
    rsum, flagged = run("random-open-divider_salad_small_cramped", 2, 120, n, min(steps, 200), 7, spec=use_spec)
    print("random-open-divider_salad_small_cramped        A=2  ok  flagged_env_steps=%d" % flagged)
    print("soak ok (%s kernels): %d env-steps compared bit-exactly in %.0f s"
          % ("per-level specialised" if use_spec else "generic", total, time.time() - t0))


if __name__ == "__main__":
    main()

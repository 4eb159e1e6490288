#!/usr/bin/env python3
"""Re-run ONE configuration of tests/soak.py (a dup level of the golden fixtures) and, at the first
step whose error flags differ between HIP and the oracle, print the env: its state before the step
on both sides, the actions, the flags.   python tests/soak_repro.py <level> <A> <steps> <n> <seed> [spec]"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from gym_comm_amd import compiler, levels
from gym_comm_amd.batched import BatchedOvercooked
from hip_util import scripted_then_random
from oracle import oracle


def main():
    name, A, steps, n, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    use_spec = len(sys.argv) > 6 and sys.argv[6] == "spec"
    level = name
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "cbase_dup_*.npz"))):
        st = json.loads(str(np.load(f)["static_json"]))
        if st["level"] == name:
            level = levels.parse_level_text(st["level"], st["level_text"])
    lv = compiler.compile_level(level, A, 120)
    rng = np.random.default_rng(seed)
    acts = scripted_then_random(rng, name, steps, A, n)
    ora = oracle.OracleBatch(lv.blob, n, threads=16)
    env = BatchedOvercooked(lv, num_envs=n, auto_reset=True, specialize_level=use_spec)
    a_d = torch.from_numpy(acts).cuda()
    prev_h, prev_o = env.snapshot(), ora.snapshot_all()
    for k in range(steps):
        env.step(a_d[k])
        ora.step(acts[k], auto_reset=True)
        hs, os_ = env.snapshot(), ora.snapshot_all()
        bad = np.nonzero(os_["error"] != hs["error"])[0]
        if len(bad):
            e = int(bad[0])
            np.set_printoptions(linewidth=200)
            print("step %d: %d env(s) differ; env %d: oracle error %d, hip error %d; actions %s"
                  % (k, len(bad), e, os_["error"][e], hs["error"][e], acts[k][:, e]))
            for tag, before, after in (("hip", prev_h, hs), ("oracle", prev_o, os_)):
                for key in sorted(before):
                    v0, v1 = np.asarray(before[key]), np.asarray(after[key])
                    if v0.ndim >= 1 and v0.shape[-1] == n:
                        print("  %-7s %-12s before %s  after %s" % (tag, key, v0[..., e].tolist(), v1[..., e].tolist()))
            print("  history of this env's actions, last 12 steps:", acts[max(0, k - 11):k + 1, :, e].tolist())
            return 1
        prev_h, prev_o = hs, os_
    print("no difference in %d steps" % steps)
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Diagnostic: where one wave of k_multi_step spends its cycles (s_memtime stamps).
Builds a -DOC_STAMPS specialised library (never shipped, never timed), runs a few hundred
steps and prints the median cycle count of every phase.  GPU box only.
    python tools/stamp_phases.py [n] [waves per 64 envs: 0 = the library's choice, 1, 4] [pairs | stats | rng]
(pairs: actions as [n][2] pairs; stats: rows + in-kernel episode statistics; rng: ego pairs + the
in-kernel random partner + statistics -- what OvercookedVecEnv.step_tensors launches)"""
import os
import sys

os.environ["OC_HIP_EXTRA_FLAGS"] = (os.environ.get("OC_HIP_EXTRA_FLAGS", "") + " -DOC_STAMPS").strip()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gym_comm_amd.batched import BatchedOvercooked

NAMES = ["start->state+actions in regs", "collisions+interact (+ position lookups issued)",
         "done/reward (+ Deliver lookups issued)", "state stores, distances consumed, quotients formed",
         "obs x2 (stores issued)", "fp64 sums", "shaped reward stored", "metrics"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    env = BatchedOvercooked("open-divider_tomato", num_agents=2, num_envs=n, max_num_timesteps=500,
                            num_communication=2, auto_reset=True, specialize_level=True,
                            waves_per_64=int(sys.argv[2]) if len(sys.argv) > 2 else 0,
                            episode_stats=len(sys.argv) > 3 and sys.argv[3] in ("stats", "rng"))   # rows + statistics: the options variant
    sp = env.launch_waves(general=len(sys.argv) > 3)    # waves per 64 envs (4 = split launch): the kernel writes one record per wave
    waves = (n + 63) // 64 * sp
    dbg = torch.zeros((waves, 16), dtype=torch.int64, device="cuda")
    env.reward = dbg.view(torch.int32)          # the stamps build writes its stamps through `sparse`
    gen = torch.Generator(device="cuda").manual_seed(1)
    hi = torch.tensor([4, 2, 4, 2], device="cuda").view(4, 1)
    rows = []
    pairs = len(sys.argv) > 3 and sys.argv[3] == "pairs"     # the options variant: actions as [n][2] pairs
    rng_mode = len(sys.argv) > 3 and sys.argv[3] == "rng"
    rng = torch.randint(0, 2 ** 31 - 1, (n,), generator=gen, device="cuda", dtype=torch.int32)
    played = torch.zeros((2, n), dtype=torch.int32, device="cuda")
    for k in range(300):
        a = (torch.rand((4, n), generator=gen, device="cuda") * hi).to(torch.int32)
        if rng_mode:
            env.multi_step(a, ego_pairs=a[0:2].T.contiguous(), alt_rng=rng, alt_played=played)
        elif pairs:
            env.multi_step(None, ego_pairs=a[0:2].T.contiguous(), alt_pairs=a[2:4].T.contiguous())
        else:
            env.multi_step(a)
        if k >= 100:
            rows.append(dbg.cpu().numpy().copy())
    t = np.stack(rows).astype(np.int64)          # [steps][waves][16]
    print("n = %d, %d waves (%d per 64 envs); median shader cycles per phase (one wave):" % (n, waves, sp))
    for role in range(sp):
        tr = t[:, role::sp, :9]
        taken = [k for k in range(9) if (tr[:, :, k] != 0).all()]    # a wave only stamps the phases of its duties
        if sp > 1:
            print(" wave %d of the workgroup (stamps %s):" % (role, taken))
        tot = np.median(tr[:, :, taken[-1]] - tr[:, :, taken[0]])
        for a, b in zip(taken[:-1], taken[1:]):
            v = np.median(tr[:, :, b] - tr[:, :, a])
            name = NAMES[a] if b == a + 1 else "stamp %d -> %d" % (a, b)
            print("  %-52s %8.0f  (%4.1f %%)" % (name, v, 100 * v / tot))
        print("  %-52s %8.0f" % ("total (first stamp -> last stamp)", tot))
    # (per workgroup: the stamps of different XCDs do not share an origin)
    grp = t.reshape(t.shape[0], -1, sp, 16)[:, :, :, :9]
    first = np.where(grp == 0, np.iinfo(np.int64).max, grp).min(axis=(2, 3))
    span = np.median(grp.max(axis=(2, 3)) - first)
    print("  first stamp -> last stamp of a workgroup: %.0f cycles" % span)


if __name__ == "__main__":
    main()

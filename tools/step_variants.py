#!/usr/bin/env python3
"""Per-launch time of oc_multi_step by action source and options (include/oc_hip.h, oc_step_opts):
action rows; int32 / int64 [n][2] pairs for both players; pairs + in-kernel episode statistics;
ego pairs + in-kernel random partner (+ statistics): what OvercookedVecEnv launches.  hipGraph
replays of 64 launches, HIP events.  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_comm_amd.batched import BatchedOvercooked
from aux_kernel_rates import timed


def main():
    for n in (4096, 131072):
        gen = torch.Generator(device="cuda").manual_seed(1)
        rows = torch.randint(0, 2, (4, n), generator=gen, device="cuda", dtype=torch.int32)
        p32 = [torch.randint(0, 2, (n, 2), generator=gen, device="cuda", dtype=torch.int32) for _ in range(2)]
        p64 = [p.long() for p in p32]
        rng = torch.randint(0, 2 ** 31 - 1, (n,), generator=gen, device="cuda", dtype=torch.int32)
        played = torch.zeros((2, n), dtype=torch.int32, device="cuda")
        out = ["open-divider_tomato x2, n = %d:" % n]
        for stats in (False, True):
            env = BatchedOvercooked("open-divider_tomato", num_agents=2, num_envs=n, max_num_timesteps=500,
                                    num_communication=2, communication_on=True, fow_radius=2,
                                    episode_stats=stats)
            tag = " + episode statistics" if stats else ""
            if not stats:
                out.append("rows %.2f us" % timed(lambda: env.multi_step(rows)))
            out.append("int32 pairs%s %.2f us" % (tag, timed(lambda: env.multi_step(None, ego_pairs=p32[0], alt_pairs=p32[1]))))
            out.append("int64 pairs%s %.2f us" % (tag, timed(lambda: env.multi_step(None, ego_pairs=p64[0], alt_pairs=p64[1]))))
            out.append("ego pairs + in-kernel partner%s %.2f us"
                       % (tag, timed(lambda: env.multi_step(rows, ego_pairs=p32[0], alt_rng=rng, alt_played=played))))
        print("  ".join(out), flush=True)


if __name__ == "__main__":
    main()

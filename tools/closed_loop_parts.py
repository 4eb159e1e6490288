#!/usr/bin/env python3
"""Closed loop taken apart: per-launch time of the fused policy kernel alone (two players), of the
fused env step alone (int32 pairs of both players + episode statistics), and of the two alternating
-- is a closed-loop step the sum of its parts?  hipGraph replays of 64 launches, HIP events.
GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_comm_amd.batched import BatchedOvercooked
from gym_comm_amd.vec_env import FusedMLPPartner, MLPPolicy
from aux_kernel_rates import timed


def main():
    for odt in (torch.int32, torch.int8):
        for n in (4096, 32768, 131072):
            env = BatchedOvercooked("open-divider_tomato", num_agents=2, num_envs=n, max_num_timesteps=500,
                                    num_communication=2, communication_on=True, fow_radius=2,
                                    episode_stats=True, obs_dtype=odt)
            env.observe()
            pl = [FusedMLPPartner(MLPPolicy(env.S, 2, seed=s).cuda(), sample=True, seed=s) for s in (1, 2)]
            rows = [env.obs[0], env.obs[1]]
            policy = lambda: FusedMLPPartner.launch(pl, rows, env.timestep)
            policy()
            step = lambda: env.multi_step(None, ego_pairs=pl[0].pairs, alt_pairs=pl[1].pairs)

            def both():
                policy()
                step()
            tp, ts, tb = timed(policy), timed(step), timed(both, per=32)
            print("n = %6d, %s rows: policy x2 %.2f us  step %.2f us  sum %.2f us  alternating %.2f us per (policy + step)"
                  % (n, str(odt).replace("torch.", ""), tp, ts, tp + ts, tb), flush=True)


if __name__ == "__main__":
    main()

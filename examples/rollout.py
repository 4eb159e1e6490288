#!/usr/bin/env python3
"""Random-policy rollout on 1..8 MI355X: one process per GPU, independent env shards, one
RCCL all-gather of the 64-byte metrics vector at the end.

    python examples/rollout.py --envs 131072 --steps 2000
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        examples/rollout.py --envs 1048576 --steps 2000      # --envs is the WHOLE job
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_comm_amd import dist as ocdist
from gym_comm_amd.arglist import load_env_args
from gym_comm_amd.vec_env import OvercookedVecEnv


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--json-path", default=None, help="a run config with the reference's schema")
    p.add_argument("--level", default="open-divider_tomato")
    p.add_argument("--envs", type=int, default=131072, help="envs in the whole job")
    p.add_argument("--steps", type=int, default=1000)
    a = p.parse_args()
    rank, local_rank, world = ocdist.env_rank_world()
    torch.cuda.set_device(local_rank)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    cfg = load_env_args(a.json_path) if a.json_path else load_env_args(
        {"level": a.level, "num_agents": 2, "max_num_timesteps": 500, "communication_on": True,
         "num_communication": 2})
    start, count = ocdist.shard(a.envs, rank, world)
    venv = OvercookedVecEnv(cfg, count, device="cuda:%d" % local_rank, seed=ocdist.rank_seed(0, rank),
                            track_episode_stats=False)
    venv.reset_tensors()
    gen = torch.Generator(device="cuda").manual_seed(ocdist.rank_seed(1, rank))
    C = venv._b.C
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ego = torch.stack([torch.randint(0, 4, (count,), generator=gen, device="cuda"),
                           torch.randint(0, C, (count,), generator=gen, device="cuda")], dim=1)
        venv.step_tensors(ego)
    torch.cuda.synchronize()
    g = ocdist.gather_rollout_metrics(venv._b.metrics_vector(), time.perf_counter() - t0)
    if rank == 0:
        tot = g["total"]
        print("ranks %d  env-steps %d  episodes %d  successes %d  reward_sum %d  %.3g env-steps/s"
              % (world, tot["env_steps"], tot["episodes"], tot["successes"], tot["reward_sum"],
                 ocdist.whole_job_rate(tot["env_steps"], g["elapsed_s"])))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

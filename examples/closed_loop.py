#!/usr/bin/env python3
"""Policies in the loop on one MI355X: an ego and a partner torch module act on the observation
rows the step kernel leaves in HBM; ego policy -> partner policy -> fused step (+ in-kernel episode
statistics) run as one hipGraph per 8 steps; env 0's episodes are recorded as ASCII frames.

    python examples/closed_loop.py --envs 4096 --steps 512
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_comm_amd.arglist import load_env_args
from gym_comm_amd.vec_env import MLPPolicy, OvercookedVecEnv, TorchPolicyPartner


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--level", default="open-divider_salad")
    p.add_argument("--envs", type=int, default=4096)
    p.add_argument("--steps", type=int, default=512)
    a = p.parse_args()
    C = 3
    cfg = load_env_args({"level": a.level, "num_agents": 2, "max_num_timesteps": 100,
                         "communication_on": True, "num_communication": C, "fow_radius": 2})
    partner = TorchPolicyPartner(MLPPolicy(9, C, hidden=64, seed=1).cuda(), sample=True, seed=7)
    venv = OvercookedVecEnv(cfg, a.envs, partner=partner, obs_dtype=torch.float32)   # float32 rows: no cast before the GEMM
    S = venv._b.S
    partner.policy = MLPPolicy(S, C, hidden=64, seed=1).cuda()
    ego = TorchPolicyPartner(MLPPolicy(S, C, hidden=64, seed=2).cuda(), sample=True)
    venv.reset_tensors()
    loop = venv.closed_loop(ego, steps=8)           # 8 x (ego forward, partner forward, fused step) per replay
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps // 8):
        obs, rew, done = loop.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m = venv.metrics()
    print("%d envs x %d steps: %.1f us/step, %.3g env-steps/s; episodes %d, successes %d, reward_sum %d"
          % (a.envs, a.steps, dt / a.steps * 1e6, a.envs * a.steps / dt, m["episodes"], m["successes"], m["reward_sum"]))
    print("env 0 after the rollout (t = %d):" % venv.get_attr("t", 0)[0])
    print(venv.env_method("__str__", indices=[0])[0])


if __name__ == "__main__":
    main()

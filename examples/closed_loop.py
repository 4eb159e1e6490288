#!/usr/bin/env python3
"""Policies in the loop on one MI355X: an ego and a partner MLP act on the observation rows the
step kernel leaves in HBM; both policies (ONE launch of the MFMA policy kernel, include/oc_policy.h)
-> fused step (+ in-kernel episode statistics) run as one hipGraph per 8 steps; env 0 is printed as
ASCII at the end.  --policy torch runs the same modules through torch instead (~25 launches per step).

    python examples/closed_loop.py --envs 4096 --steps 512
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_comm_amd.arglist import load_env_args
from gym_comm_amd.vec_env import FusedMLPPartner, MLPPolicy, OvercookedVecEnv, TorchPolicyPartner


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--level", default="open-divider_salad")
    p.add_argument("--envs", type=int, default=4096)
    p.add_argument("--steps", type=int, default=512)
    p.add_argument("--policy", default="fused", choices=["fused", "torch"])
    a = p.parse_args()
    C = 3
    cfg = load_env_args({"level": a.level, "num_agents": 2, "max_num_timesteps": 100,
                         "communication_on": True, "num_communication": C, "fow_radius": 2})
    fused = a.policy == "fused"
    # torch modules want float32 rows (no cast before the GEMM); the fused kernel reads int32 rows as they lie
    venv = OvercookedVecEnv(cfg, a.envs, obs_dtype=torch.int32 if fused else torch.float32)
    S = venv._b.S
    seat = (lambda pol, sd: FusedMLPPartner(pol, sample=True, seed=sd)) if fused else \
           (lambda pol, sd: TorchPolicyPartner(pol, sample=True, seed=sd))
    venv.partner = seat(MLPPolicy(S, C, hidden=64, seed=1).cuda(), 7)
    ego = seat(MLPPolicy(S, C, hidden=64, seed=2).cuda(), 8)
    venv.reset_tensors()
    loop = venv.closed_loop(ego, steps=8)           # 8 x (both policies, fused step) per replay
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

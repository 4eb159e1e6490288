#!/usr/bin/env python3
"""Where does a step of the tensor boundary go?  Runs one variant (argv[1]) of
OvercookedVecEnv.step_tensors / ClosedLoop for a fixed number of steps, prints the host wall time
per step; meant to be run under `rocprofv3 --kernel-trace --stats` for the kernel side.
  variants: eager | graph | loop1 | loop16      second argument: number of envs"""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_comm_amd.vec_env import OvercookedVecEnv, RandomPartner


def main():
    variant, n = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    arg = SimpleNamespace(level="open-divider_tomato", num_agents=2, max_num_timesteps=500,
                          ego_config={}, partner_config={}, num_communication=2,
                          communication_on=True, ego_led=False, fow_radius=2)
    venv = OvercookedVecEnv(arg, n, seed=1, use_graph=(variant == "graph"))
    venv.reset_tensors()
    ego = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    if variant in ("eager", "graph"):
        fn = lambda: venv.step_tensors(ego)
        per = 1
    else:
        per = int(variant[4:])
        loop = venv.closed_loop(RandomPartner(2, seed=9), graph=True, steps=per)
        fn = loop.step
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps // per):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s n=%d: %.1f us/step host enqueue, %.1f us/step to completion"
          % (variant, n, (t1 - t0) / (steps // per * per) * 1e6, (t2 - t0) / (steps // per * per) * 1e6), flush=True)


if __name__ == "__main__":
    main()

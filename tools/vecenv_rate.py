#!/usr/bin/env python3
"""Throughput of the SB3-shaped boundary: tensor API (device-resident policy) and numpy
API (host buffers in and out: the PCIe-inclusive rate quoted in DESIGN.md)."""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_comm_amd.vec_env import OvercookedVecEnv


def main():
    for n in (4096, 131072):
        arg = SimpleNamespace(level="open-divider_tomato", num_agents=2, max_num_timesteps=500,
                              ego_config={}, partner_config={}, num_communication=2,
                              communication_on=True, ego_led=False, fow_radius=2)
        venv = OvercookedVecEnv(arg, n, seed=1)
        venv.reset_tensors()
        gen = torch.Generator(device="cuda").manual_seed(0)
        acts = [torch.stack([torch.randint(0, 4, (n,), generator=gen, device="cuda"),
                             torch.randint(0, 2, (n,), generator=gen, device="cuda")], dim=1)
                for _ in range(64)]
        for k in range(50):
            venv.step_tensors(acts[k % 64])
        torch.cuda.synchronize()
        steps = 1000 if n <= 4096 else 300
        t0 = time.perf_counter()
        for k in range(steps):
            venv.step_tensors(acts[k % 64])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("n=%d tensor API (random partner on device): %.1f us/step, %.3g env-steps/s"
              % (n, dt / steps * 1e6, n * steps / dt))
        acts_np = [a.cpu().numpy() for a in acts]
        venv.reset()
        steps = 200 if n <= 4096 else 30
        t0 = time.perf_counter()
        for k in range(steps):
            venv.step(acts_np[k % 64])
        dt = time.perf_counter() - t0
        print("n=%d numpy API (host actions in, 11 host obs arrays out, PCIe-inclusive): %.1f us/step, %.3g env-steps/s"
              % (n, dt / steps * 1e6, n * steps / dt))


if __name__ == "__main__":
    main()

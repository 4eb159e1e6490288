#!/usr/bin/env python3
"""Throughput of the SB3-shaped boundary: tensor API (device-resident policy; eager launches
and the one-hipGraph-replay variant), the ego-policy-in-the-graph closed loop, and the numpy
API (host buffers in and out: the PCIe-inclusive rate quoted in DESIGN.md)."""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_comm_amd.vec_env import OvercookedVecEnv, RandomPartner


def timed(fn, steps):
    # >= 150 ms of continuous work first: the first tens of ms after host-side idling can run 10x
    # slow on this box (tools/slow_replay_probe.py)
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < 0.15:
        fn(k)
        k += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        fn(k)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    for n in (4096, 131072):
        arg = SimpleNamespace(level="open-divider_tomato", num_agents=2, max_num_timesteps=500,
                              ego_config={}, partner_config={}, num_communication=2,
                              communication_on=True, ego_led=False, fow_radius=2)
        gen = torch.Generator(device="cuda").manual_seed(0)
        acts = [torch.stack([torch.randint(0, 4, (n,), generator=gen, device="cuda"),   # int64, as a policy gives them
                             torch.randint(0, 2, (n,), generator=gen, device="cuda")], dim=1)
                for _ in range(64)]
        for use_graph in (False, True):
            venv = OvercookedVecEnv(arg, n, seed=1, use_graph=use_graph)
            venv.reset_tensors()
            timed(lambda k: venv.step_tensors(acts[k % 64]), 50)
            steps = 2000 if n <= 4096 else 500
            dt = timed(lambda k: venv.step_tensors(acts[k % 64]), steps)
            print("n=%d step_tensors(ego_actions), random partner on device, %s: %.1f us/step, %.3g env-steps/s"
                  % (n, "partner + fused step as ONE hipGraph replay" if use_graph else "eager launches",
                     dt * 1e6, n / dt), flush=True)
        # both players inside the graph (random ego too): no host-side copy at all
        for k_steps in (1, 16):
            venv = OvercookedVecEnv(arg, n, seed=1)
            venv.reset_tensors()
            loop = venv.closed_loop(RandomPartner(2, seed=9), graph=True, steps=k_steps)
            timed(lambda k: loop.step(), 20)
            reps = (2000 if n <= 4096 else 500) // k_steps
            dt = timed(lambda k: loop.step(), reps) / k_steps
            print("n=%d ClosedLoop (ego + partner + fused step, %d step(s) per hipGraph replay): %.1f us/step, %.3g env-steps/s"
                  % (n, k_steps, dt * 1e6, n / dt), flush=True)
        acts_np = [a.cpu().numpy() for a in acts]
        for reuse in (False, True):
            venv = OvercookedVecEnv(arg, n, seed=1, reuse_host_buffers=reuse)
            venv.reset()
            for k in range(10):
                venv.step(acts_np[k % 64])
            steps = 400 if n <= 4096 else 40
            t0 = time.perf_counter()
            for k in range(steps):
                venv.step(acts_np[k % 64])
            dt = (time.perf_counter() - t0) / steps
            print("n=%d numpy API (host actions in, 11 host obs arrays + infos out, PCIe-inclusive; %s): %.1f us/step, %.3g env-steps/s"
                  % (n, "views of two alternating pinned buffers" if reuse else "fresh arrays every step",
                     dt * 1e6, n / dt), flush=True)


if __name__ == "__main__":
    main()

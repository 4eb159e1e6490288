#!/usr/bin/env python3
"""Fourth look at the slow stretch: is the one-time ~85 ms gap (slow_stretch_probe2.py, row (b))
BEFORE the device starts on the submitted replays, or AFTER the last one (a late host wake-up)?
Repeats probe 2's scenario -- churn of eager batches, a fresh VecEnv, capture, 30 replays of a
16-step graph -- eight times and, per round, reports the host clock at which the FIRST event and
the LAST event completed, relative to the start of submission.  GPU box only."""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_comm_amd.vec_env import OvercookedVecEnv, RandomPartner

N = 131072
ARG = SimpleNamespace(level="open-divider_tomato", num_agents=2, max_num_timesteps=500, ego_config={},
                      partner_config={}, num_communication=2, communication_on=True, ego_led=False, fow_radius=2)


def churn(use_graph):
    for _ in range(2):
        venv = OvercookedVecEnv(ARG, N, seed=1, use_graph=use_graph)
        venv.reset_tensors()
        a = torch.zeros((N, 2), dtype=torch.int64, device="cuda")
        for _ in range(20):
            venv.step_tensors(a)
        torch.cuda.synchronize()
        del venv


def loop16(tag, reps=30):
    venv = OvercookedVecEnv(ARG, N, seed=1)
    venv.reset_tensors()
    loop = venv.closed_loop(RandomPartner(2, seed=9), graph=True, steps=16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for k in range(reps):
        loop.step()
    e1.record()
    t_sub = time.perf_counter() - t0
    e0.synchronize()
    t_first = time.perf_counter() - t0
    e1.synchronize()
    t_last = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_sync = time.perf_counter() - t0
    print("%-44s submitted in %.2f ms | first event done at %7.2f ms | last at %7.2f ms (gpu span %.2f ms) | device sync returned at %7.2f ms"
          % (tag, t_sub * 1e3, t_first * 1e3, t_last * 1e3, e0.elapsed_time(e1), t_sync * 1e3), flush=True)


def main():
    loop16("first thing in the process")
    for r in range(8):
        churn(r % 2 == 1)
        loop16("round %d after churn(%s)" % (r, "graph" if r % 2 else "eager"))


if __name__ == "__main__":
    main()

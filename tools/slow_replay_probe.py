#!/usr/bin/env python3
"""Probe for the slow ClosedLoop graph replays seen inside tools/vecenv_rate.py at 131 072 envs
(75-130 us per step instead of 11): the same loop (a) first thing in a process, (b) after two other
131 072-env batches were built, stepped and dropped, (c) the same with gc + empty_cache in
between, (d) built while the dropped batches' hipGraphs are still alive.  GPU box only."""
import gc
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


def loop16(tag, keep=None):
    venv = OvercookedVecEnv(ARG, N, seed=1)
    venv.reset_tensors()
    loop = venv.closed_loop(RandomPartner(2, seed=9), graph=True, steps=16)
    for _ in range(5):
        loop.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        loop.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (30 * 16)
    # the same again behind 150 ms of continuous work (is a slow stretch a clock ramp after idling?)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        loop.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        loop.step()
    torch.cuda.synchronize()
    dw = (time.perf_counter() - t0) / (30 * 16)
    # the same env, the same launches, issued eagerly (no graph): is it the memory or the graph instance?
    t0 = time.perf_counter()
    for _ in range(200):
        loop.enqueue()
    torch.cuda.synchronize()
    de = (time.perf_counter() - t0) / 200
    # ... and a second graph captured on the same env
    loop2 = venv.closed_loop(RandomPartner(2, seed=9), graph=True, steps=16)
    for _ in range(3):
        loop2.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        loop2.step()
    torch.cuda.synchronize()
    d2 = (time.perf_counter() - t0) / (30 * 16)
    print("%-70s graph %.1f us/step (%.1f behind 150 ms of work), eager %.1f, a second graph on the same env %.1f"
          % (tag, dt * 1e6, dw * 1e6, de * 1e6, d2 * 1e6), flush=True)
    if keep is not None:
        keep.append((venv, loop))


def churn(use_graph):
    for _ in range(2):
        venv = OvercookedVecEnv(ARG, N, seed=1, use_graph=use_graph)
        venv.reset_tensors()
        a = torch.zeros((N, 2), dtype=torch.int64, device="cuda")
        for _ in range(20):
            venv.step_tensors(a)
        torch.cuda.synchronize()
        del venv


def main():
    loop16("(a) first thing in the process")
    churn(False)
    loop16("(b) after two eager batches were built, stepped and dropped")
    churn(True)
    loop16("(b') after two graph-capturing batches were built, stepped and dropped")
    gc.collect()
    torch.cuda.empty_cache()
    loop16("(c) after gc.collect() + torch.cuda.empty_cache()")
    keep = []
    loop16("(d) a loop kept alive ...", keep)
    loop16("(d) ... and another one beside it", keep)
    del keep
    churn(True)
    gc.collect()
    loop16("(e) after two more graph-capturing batches and gc.collect() only")
    churn(True)
    torch.cuda.empty_cache()
    loop16("(f) after two more and torch.cuda.empty_cache() only")
    churn(True)
    torch.cuda.synchronize()
    time.sleep(0.5)
    loop16("(g) after two more, a device sync and 0.5 s")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Second look at round 2's slow stretch (profiles/r02_v6_slow_replay_probe.txt: graph replays at
75-160 us per step instead of 11 right after graph-capturing batches had been built and dropped).
tools/slow_stretch_probe.py found nothing slow INSIDE the launches (normal shader clock, normal
kernel-active spans and boundaries on a 480-launch graph after every kind of churn).  Round 2's
loop was different: 30 replays of a 16-step graph, timed on the HOST.  This probe repeats exactly
that (ClosedLoop over OvercookedVecEnv with the in-kernel RandomPartner at 131 072 envs) and times
every replay three ways: the host time spent inside `graph.replay()`, the GPU time between two
events around it, and the wall time to the sync -- so a stall of the launch CALL (runtime-side
work: freeing a dropped graph's resources) can be told from slow execution on the device.
GPU box only."""
import gc
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
    t0 = time.perf_counter()
    loop = venv.closed_loop(RandomPartner(2, seed=9), graph=True, steps=16)
    t_capture = time.perf_counter() - t0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    host = []
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    ev[0].record()
    for k in range(reps):
        t0 = time.perf_counter()
        loop.step()
        host.append(time.perf_counter() - t0)
        ev[k + 1].record()
    t_issue = time.perf_counter() - t_all
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_all
    gpu = [ev[k].elapsed_time(ev[k + 1]) * 1e3 / 16 for k in range(reps)]      # us per step
    host_us = [h * 1e6 for h in host]
    print("%-64s capture %.1f ms | wall %.1f us/step | issue (host) %.1f ms of %.1f ms | per replay: host call us %s ... | gpu us/step %s ..."
          % (tag, t_capture * 1e3, wall / (reps * 16) * 1e6, t_issue * 1e3, wall * 1e3,
             " ".join("%.0f" % h for h in host_us[:8]), " ".join("%.1f" % g for g in gpu[:8])), flush=True)
    print("%-64s   host call: median %.0f max %.0f us; gpu: median %.1f max %.1f us/step" % (
        "", np.median(host_us), max(host_us), np.median(gpu), max(gpu)), flush=True)


def main():
    loop16("(a) first thing in the process")
    churn(False)
    loop16("(b) after two eager batches were built, stepped and dropped")
    churn(True)
    loop16("(b') after two graph-capturing batches built, stepped, dropped")
    churn(True)
    loop16("(b'') the same again")
    churn(True)
    gc.collect()
    loop16("(e) after two more and gc.collect() only")
    churn(True)
    torch.cuda.empty_cache()
    loop16("(f) after two more and torch.cuda.empty_cache() only")
    churn(True)
    torch.cuda.synchronize()
    time.sleep(0.5)
    loop16("(g) after two more, a device sync and 0.5 s")


if __name__ == "__main__":
    main()

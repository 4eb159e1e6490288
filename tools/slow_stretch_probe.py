#!/usr/bin/env python3
"""What is the ~10x slow stretch that follows building / graph-capturing / dropping batches?

Round 2 saw graph replays (and eager launches) run 5-30x slow for tens of milliseconds after a
process had built, captured and dropped other batches (profiles/r02_v6_slow_replay_probe.txt) and
called it "a clock / power ramp after idling" -- although the probe's own rows (a) "first thing in
the process" and (g) "a device sync and 0.5 s" were fast.  This probe looks INSIDE the slow
launches: a 480-launch hipGraph of the fused step on the TIMELINE build (every wave stamps the
chip-wide 100 MHz clock and the shader clock; include/oc_hip.h: oc_timeline_begin) is replayed
  (1) warm, as the baseline;
  (2) right after `churn`: two graph-capturing 131 072-env batches built, stepped and dropped;
  (3) after churn + gc.collect() + torch.cuda.synchronize()  (explicit teardown, then a sync);
  (4) after churn + torch.cuda.empty_cache();
  (5) after 0.5 s of plain host sleep with nothing torn down (idling alone).
For every replay it prints, over the 480 launches in order: period / kernel-active / boundary
(us), the shader clock the waves ran at (s_memtime cycles per realtime tick), and the same for
the first 32 launches alone -- a clock ramp shows as a LOW shader clock with normal cycles per
wave; work queued on the device by a teardown shows as a normal clock with long boundaries (or
long active spans with a normal clock if the memory system is busy).
GPU box only."""
import ctypes
import gc
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gym_comm_amd.batched import BatchedOvercooked
from gym_comm_amd.vec_env import OvercookedVecEnv

N = int(os.environ.get("PROBE_N", "131072"))
STEPS = 480
ARG = SimpleNamespace(level="open-divider_tomato", num_agents=2, max_num_timesteps=500, ego_config={},
                      partner_config={}, num_communication=2, communication_on=True, ego_led=False, fow_radius=2)
BIG = torch.iinfo(torch.int64).max


def churn(use_graph=True, n=131072):
    for _ in range(2):
        venv = OvercookedVecEnv(ARG, n, seed=1, use_graph=use_graph)
        venv.reset_tensors()
        a = torch.zeros((n, 2), dtype=torch.int64, device="cuda")
        for _ in range(20):
            venv.step_tensors(a)
        torch.cuda.synchronize()
        del venv


class Probe:
    def __init__(self):
        self.env = BatchedOvercooked("open-divider_tomato", num_envs=N, max_num_timesteps=500, device="cuda:0",
                                     specialize_level="timeline")
        self.stride = 4 * ((N + 63) // 64)
        self.rec = torch.zeros((STEPS, self.stride, 4), dtype=torch.int32, device="cuda")
        acts = torch.randint(0, 2, (16, 4, N), dtype=torch.int32, device="cuda")
        acts[:, 0] = torch.randint(0, 4, (16, N), dtype=torch.int32, device="cuda")
        acts[:, 2] = torch.randint(0, 4, (16, N), dtype=torch.int32, device="cuda")
        self.acts = [acts[k].contiguous() for k in range(16)]
        self.stream = torch.cuda.Stream()
        with torch.cuda.stream(self.stream):
            for k in range(8):
                self.env.multi_step(self.acts[k])
            self.stream.synchronize()
            assert self.env._L.oc_timeline_begin(ctypes.c_void_p(self.rec.data_ptr()), STEPS, self.stride) == 0
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                for k in range(STEPS):
                    self.env.multi_step(self.acts[k % 16])
            self.env._L.oc_timeline_begin(None, 0, 0)
            for _ in range(20):
                self.graph.replay()
            self.stream.synchronize()

    def replay(self, tag):
        with torch.cuda.stream(self.stream):
            self.rec.fill_(-1)
            self.stream.synchronize()
            t0 = time.perf_counter()
            self.graph.replay()
            self.stream.synchronize()
            wall = time.perf_counter() - t0
            import bench
            wrote, st, issue, _, cycles = bench.timeline_reduce(self.rec)
            start = torch.where(wrote, st, torch.full_like(st, BIG)).min(dim=1).values
            end = torch.where(wrote, issue, torch.zeros_like(st)).max(dim=1).values
            span = ((issue - st) * wrote).sum(dim=1).double()
            cyc = (cycles * wrote).sum(dim=1).double()
            s, e = start.cpu().numpy(), end.cpu().numpy()
            mhz = (100.0 * cyc / span.clamp(min=1)).cpu().numpy()
            cpw = (cyc / wrote.sum(dim=1).clamp(min=1)).cpu().numpy()          # shader cycles per wave
        act, gap, per = (e - s) * 0.01, (s[1:] - e[:-1]) * 0.01, (s[1:] - s[:-1]) * 0.01
        f = lambda x, sl: float(np.mean(x[sl]))
        head, tail = slice(0, 32), slice(STEPS // 2, None)
        print("%-58s wall %7.1f us/step | launches 0-31: period %7.2f active %6.2f boundary %7.2f clock %4.0f MHz cycles/wave %6.0f"
              " | launches %d-: period %6.2f active %6.2f boundary %5.2f clock %4.0f MHz cycles/wave %6.0f"
              % (tag, wall / STEPS * 1e6, f(per, head), f(act, head), f(gap, head), f(mhz, head), f(cpw, head),
                 STEPS // 2, f(per, tail), f(act, tail), f(gap, tail), f(mhz, tail), f(cpw, tail)), flush=True)
        return wall / STEPS * 1e6


def main():
    p = Probe()
    for k in range(3):
        p.replay("(1) warm baseline #%d" % k)
    churn(True)
    for k in range(3):
        p.replay("(2) right after churn (2 graph batches built+dropped) #%d" % k)
    churn(True)
    gc.collect()
    torch.cuda.synchronize()
    for k in range(2):
        p.replay("(3) churn + gc.collect() + device sync #%d" % k)
    churn(True)
    torch.cuda.empty_cache()
    for k in range(2):
        p.replay("(4) churn + empty_cache() #%d" % k)
    time.sleep(0.5)
    for k in range(2):
        p.replay("(5) 0.5 s of host sleep, nothing torn down #%d" % k)
    churn(False)
    for k in range(2):
        p.replay("(6) right after churn of EAGER batches #%d" % k)
    # the eager path right after a churn, timed on the host (row (f) of round 2's probe: 336 us)
    churn(True)
    torch.cuda.empty_cache()
    with torch.cuda.stream(p.stream):
        t0 = time.perf_counter()
        for k in range(200):
            p.env.multi_step(p.acts[k % 16])
        p.stream.synchronize()
        print("(7) 200 eager launches right after churn + empty_cache(): %.1f us/step" % ((time.perf_counter() - t0) / 200 * 1e6))
        t0 = time.perf_counter()
        for k in range(200):
            p.env.multi_step(p.acts[k % 16])
        p.stream.synchronize()
        print("(7) ... and 200 more: %.1f us/step" % ((time.perf_counter() - t0) / 200 * 1e6))


if __name__ == "__main__":
    main()

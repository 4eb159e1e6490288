#!/usr/bin/env python3
"""Third look at the slow stretch: WHAT makes the next submission start tens of ms late?

slow_stretch_probe.py: nothing is slow inside the launches (shader clock, kernel-active spans and
boundaries normal after every kind of churn).  slow_stretch_probe2.py: the replays of a "slow" run
execute at the normal rate and the host calls return at once, yet the region's wall clock is ~85 ms
longer -- a ONE-TIME delay before the device starts on newly submitted work, which divided over a
few hundred steps reads as "10x slow".  This probe times exactly that start latency -- host clock
from submission of (event, one 16-step graph replay, event) to the FIRST event's completion, and to
the last -- right after each of a list of host-side actions, to find which one arms the delay:
allocating / freeing device memory through torch's caching allocator and past it (empty_cache ->
hipFree), pinned and pageable host buffers and copies from them, building and dropping a level
(hipMalloc + hipMemcpy + hipFree in oc_level_create/destroy), capturing and dropping a graph,
building a whole batch.  GPU box only."""
import gc
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gym_comm_amd.batched import BatchedOvercooked

N = 131072


class Replayer:
    def __init__(self):
        self.env = BatchedOvercooked("open-divider_tomato", num_envs=N, max_num_timesteps=500, device="cuda:0")
        acts = torch.randint(0, 2, (16, 4, N), dtype=torch.int32, device="cuda")
        self.acts = [acts[k].contiguous() for k in range(16)]
        for k in range(8):
            self.env.multi_step(self.acts[k])
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            for k in range(16):
                self.env.multi_step(self.acts[k])
        for _ in range(50):
            self.graph.replay()
        torch.cuda.synchronize()
        self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def measure(self, tag):
        """(start latency, total) in ms of one replay submitted NOW, then the same again at once."""
        out = []
        for _ in range(3):
            t0 = time.perf_counter()
            self.e0.record()
            self.graph.replay()
            self.e1.record()
            t_sub = time.perf_counter() - t0
            self.e0.synchronize()
            t_first = time.perf_counter() - t0
            self.e1.synchronize()
            t_all = time.perf_counter() - t0
            out.append((t_sub * 1e3, t_first * 1e3, t_all * 1e3, self.e0.elapsed_time(self.e1)))
        print("%-72s submit %.2f ms | first event done after %7.2f ms | all done after %7.2f ms (gpu %.2f ms) | again: %.2f / %.2f, %.2f / %.2f"
              % (tag, out[0][0], out[0][1], out[0][2], out[0][3], out[1][1], out[1][2], out[2][1], out[2][2]), flush=True)


def main():
    r = Replayer()
    mb = 1 << 20
    r.measure("baseline (warm, nothing done)")
    time.sleep(0.3)
    r.measure("after 0.3 s of host sleep")

    t = torch.empty(256 * mb, dtype=torch.uint8, device="cuda"); t.zero_(); torch.cuda.synchronize(); del t
    r.measure("after torch alloc + fill + free of 256 MB (cached by the allocator)")
    torch.cuda.empty_cache()
    r.measure("after torch.cuda.empty_cache() (hipFree of the cached 256 MB)")
    t = torch.empty(256 * mb, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    r.measure("after a fresh 256 MB hipMalloc through torch (not touched)")
    t.zero_(); torch.cuda.synchronize()
    r.measure("... after its first touch (fill kernel)")
    del t; torch.cuda.empty_cache()
    r.measure("... after freeing it to the driver")

    h = torch.empty(64 * mb, dtype=torch.uint8, pin_memory=True)
    r.measure("after allocating 64 MB of pinned host memory")
    d = torch.empty(64 * mb, dtype=torch.uint8, device="cuda"); d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    r.measure("after a 64 MB pinned H2D copy")
    del h; gc.collect()
    r.measure("after freeing the pinned buffer")
    a = np.ones(64 * mb, np.uint8); d.copy_(torch.from_numpy(a)); torch.cuda.synchronize()
    r.measure("after a 64 MB PAGEABLE H2D copy (numpy source)")
    del a; gc.collect()
    r.measure("after freeing that numpy array")
    x = d.cpu(); del x; gc.collect()
    r.measure("after a 64 MB D2H copy into pageable memory, freed")
    del d; torch.cuda.empty_cache()
    r.measure("after freeing the 64 MB device buffer to the driver")

    from gym_comm_amd import compiler
    lv = compiler.compile_level("open-divider_tomato", 2, 500)
    e = BatchedOvercooked(lv, num_envs=64, device="cuda:0"); torch.cuda.synchronize()
    r.measure("after building a 64-env batch (oc_level_create: hipMalloc + hipMemcpy)")
    del e; gc.collect()
    r.measure("after dropping it (oc_level_destroy: hipFree)")

    e = BatchedOvercooked(lv, num_envs=N, device="cuda:0"); torch.cuda.synchronize()
    r.measure("after building a 131072-env batch")
    g = torch.cuda.CUDAGraph()
    a4 = r.acts[0]
    e.multi_step(a4); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(16):
            e.multi_step(a4)
    r.measure("after capturing a 16-step graph on it (not replayed)")
    g.replay(); torch.cuda.synchronize()
    r.measure("after replaying that graph once")
    del g; gc.collect()
    r.measure("after dropping the graph")
    del e; gc.collect()
    r.measure("after dropping the batch")
    torch.cuda.empty_cache()
    r.measure("after empty_cache()")

    from types import SimpleNamespace
    from gym_comm_amd.vec_env import OvercookedVecEnv
    arg = SimpleNamespace(level="open-divider_tomato", num_agents=2, max_num_timesteps=500, ego_config={},
                          partner_config={}, num_communication=2, communication_on=True, ego_led=False, fow_radius=2)
    for rep in range(2):
        v = OvercookedVecEnv(arg, N, seed=1, use_graph=False); v.reset_tensors()
        a2 = torch.zeros((N, 2), dtype=torch.int64, device="cuda")
        for _ in range(20):
            v.step_tensors(a2)
        torch.cuda.synchronize()
        r.measure("after building + stepping an eager 131072-env VecEnv (#%d)" % rep)
        del v, a2
        r.measure("after dropping it (#%d)" % rep)


if __name__ == "__main__":
    main()

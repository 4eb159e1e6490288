#!/usr/bin/env python3
"""Per-launch time of the kernels around the fused step: oc_step + oc_obs (the unfused pair),
oc_reset (masked), oc_obs_image, oc_random_actions -- hipGraph replays of 64 launches, HIP events.
GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gym_comm_amd.batched import BatchedOvercooked
from gym_comm_amd.vec_env import RandomPartner


def timed(fn, reps=30, per=64):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(per):
                fn()
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * per)


def main():
    for level, A in (("open-divider_tomato", 2), ("partial-divider_tl", 3)):
        for n in (4096, 131072):
            env = BatchedOvercooked(level, num_agents=A, num_envs=n, max_num_timesteps=500,
                                    num_communication=2, communication_on=True, fow_radius=2)
            gen = torch.Generator(device="cuda").manual_seed(1)
            acts = torch.randint(0, 4, (A, n), generator=gen, device="cuda", dtype=torch.int32)
            mask = (torch.rand(n, generator=gen, device="cuda") < 0.01).to(torch.int32)
            out = ["%s x%d, n = %d:" % (level, A, n), "oc_step %.2f us" % timed(lambda: env.step(acts))]
            if A == 2:
                a4 = torch.randint(0, 2, (4, n), generator=gen, device="cuda", dtype=torch.int32)
                out.append("oc_obs %.2f us" % timed(env.observe))
                out.append("oc_multi_step %.2f us" % timed(lambda: env.multi_step(a4)))
                out.append("oc_obs_image %.2f us" % timed(env.observe_image and (lambda: env.observe_image(packed=True))))
                rp = RandomPartner(2, seed=3)
                rows = torch.zeros((2, n), dtype=torch.int32, device="cuda")
                rp.act_into(None, rows[0], rows[1])
                out.append("oc_random_actions %.2f us" % timed(lambda: rp.act_into(None, rows[0], rows[1])))
            out.append("oc_reset(mask 1%%) %.2f us" % timed(lambda: env.reset(mask)))
            print("  ".join(out), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Soak of the one-launch closed loop (oc_step_opts.policy: the step kernel evaluates both MLP
policies behind the step) against the two-launch path (policy kernel, then step) on a GPU box:
several levels, comm channel counts and observation element types, thousands of envs, many
episodes with auto-reset; state, observations, rewards, done flags, episode statistics, the comm
rows and the pairs about to be executed compared after EVERY step (torch.equal on the device).
Not part of the test-suite; run it when the fused kernels change:
    python tests/soak_closed_loop.py [steps] [envs]"""
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from gym_comm_amd.vec_env import FusedMLPPartner, MLPPolicy, OvercookedVecEnv

CASES = [("open-divider_tomato", 2, torch.int32, 60), ("full-divider_salad", 4, torch.int8, 80),
         ("open-divider_tl", 3, torch.float32, 50), ("partial-divider_salad", 2, torch.int32, 40),
         ("random-open-divider_salad_small", 3, torch.int32, 45), ("full-divider_tomato", 1, torch.int8, 30)]


def make(level, C, odt, T, n, one_launch, seed):
    arg = SimpleNamespace(level=level, num_agents=2, max_num_timesteps=T, ego_config={}, partner_config={},
                          num_communication=C, communication_on=True, ego_led=False, fow_radius=2)
    venv = OvercookedVecEnv(arg, n, seed=seed, obs_dtype=odt)
    S = venv._b.S
    ego = FusedMLPPartner(MLPPolicy(S, C, seed=seed + 1).cuda(), sample=True, seed=seed + 2)
    venv.partner = alt = FusedMLPPartner(MLPPolicy(S, C, seed=seed + 3).cuda(), sample=True, seed=seed + 4)
    venv.reset_tensors()
    return venv, ego, alt, venv.closed_loop(ego, graph=False, one_launch=one_launch)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    total, t0 = 0, time.time()
    for k, (level, C, odt, T) in enumerate(CASES):
        two = make(level, C, odt, T, n, False, 100 + k)
        one = make(level, C, odt, T, n, True, 100 + k)
        assert one[3].one_launch and not two[3].one_launch
        for s in range(steps):
            two[3].step()
            nxt = (one[1].pairs.clone(), one[2].pairs.clone())     # what the one-launch loop is about to execute
            one[3].step()
            assert torch.equal(nxt[0], two[1].pairs) and torch.equal(nxt[1], two[2].pairs), (level, s)
            for name in ("state", "obs", "shaped_reward", "done", "ep_return", "ep_length", "comm", "timestep"):
                assert torch.equal(getattr(two[0]._b, name), getattr(one[0]._b, name)), (level, s, name)
        m = one[0]._b.read_metrics()
        assert m == two[0]._b.read_metrics() and m["env_steps"] == steps * n
        total += steps * n
        print("%-36s C=%d %-8s ok  episodes=%-7d successes=%-5d reward_sum=%-7d (%.0fs)"
              % (level, C, str(odt).replace("torch.", ""), m["episodes"], m["successes"], m["reward_sum"],
                 time.time() - t0), flush=True)
    print("closed-loop soak ok: one launch == policy kernel + step on %d env-steps (compared after every step) in %.0f s"
          % (total, time.time() - t0))


if __name__ == "__main__":
    main()

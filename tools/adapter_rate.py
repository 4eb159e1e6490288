#!/usr/bin/env python3
"""Rate of the single-env drop-in (BASELINE configs[0]): OvercookedMultiEnv.multi_step with
host actions in and host observation dicts out, one launch per step.  GPU box only."""
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gym_comm_amd.envs import OvercookedMultiEnv


def main():
    arg = SimpleNamespace(level="open-divider_tomato", num_agents=2, max_num_timesteps=500,
                          max_num_subtasks=14, ego_config={}, partner_config={}, num_communication=2,
                          communication_on=True, ego_led=False, fow_radius=2)
    env = OvercookedMultiEnv(arg)
    rng = np.random.default_rng(0)
    acts = rng.integers(0, [4, 2, 4, 2], size=(4096, 4))
    for look in (False, True):
        env.multi_reset()
        for k in range(200):
            env.multi_step(acts[k, :2], acts[k, 2:])
        steps, t0 = 3000, time.perf_counter()
        for k in range(steps):
            obs, rew, done, info = env.multi_step(acts[k % 4096, :2], acts[k % 4096, 2:])
            if look:
                _ = env.base_env.t, env.base_env.sim_agents[0].location    # forces the state mirror
            if done:
                env.multi_reset()
        dt = time.perf_counter() - t0
        print("single-env adapter, %s: %.1f us/step, %.3g env-steps/s"
              % ("reading base_env.t / sim_agents every step" if look else "observations, reward, done only",
                 dt / steps * 1e6, steps / dt))


if __name__ == "__main__":
    main()

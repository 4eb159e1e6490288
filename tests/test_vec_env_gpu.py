"""-m gpu: the SB3-VecEnv-shaped boundary over the batched env (gym_comm_amd.vec_env)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

KEYS = ["object_encodings_x", "object_encodings_y", "state_encodings", "is_hidden",
        "completed_subtasks", "agent1_location", "agent2_location", "agent_is_holding",
        "agent1_comm", "agent2_comm"]


def _arglist(st):
    return SimpleNamespace(level=st["level"], num_agents=2, max_num_timesteps=st["max_num_timesteps"],
                           max_num_subtasks=14, ego_config=st["ego_config"],
                           partner_config=st["partner_config"],
                           num_communication=st["num_communication"],
                           communication_on=st["communication_on"], ego_led=st["ego_led"],
                           fow_radius=st["fow_radius"])


class TapePartner:
    """Batched partner replaying the fixture's partner actions; checks what it is shown."""

    def __init__(self, tape, expect_obs, n):
        self.tape, self.k, self.n, self.expect = tape, 0, n, expect_obs
        self.updates = 0

    def __call__(self, obs):
        flat = torch.cat([obs[k].to(torch.int64) for k in KEYS], dim=1).cpu().numpy()
        assert (flat == self.expect[self.k][None, :]).all(), self.k     # partner sees viewer-1 obs
        a = torch.tensor(self.tape[self.k], dtype=torch.int32).repeat(self.n, 1)
        self.k += 1
        return a

    def update(self, rewards, dones):
        assert rewards.shape == (self.n,) and dones.shape == (self.n,)
        self.updates += 1


@pytest.mark.parametrize("terminal_obs", [False, True])
def test_vec_env_replays_wrapper_fixture(terminal_obs):
    from gym_comm_amd.vec_env import SPACE_DTYPE, OvercookedVecEnv
    z, st = load_golden(os.path.join(GOLDEN, "wrap_tomato_r2.npz"))
    n, K = 48, 420
    # what the partner is shown before step k: viewer-1 obs after step k-1 (or the reset obs);
    # after a done the reference calls multi_reset(), whose viewer-1 obs equals reset_obs[1]
    # except for the comm vectors, which persist (overcooked_env.py:284-297)
    S, C = len(st["subtasks"]), st["num_communication"]
    expect = []
    prev = z["reset_obs"][1].astype(np.int64)
    for k in range(K):
        expect.append(prev.copy())
        nxt = z["obs"][k][1].astype(np.int64)
        if z["done"][k]:
            r = z["reset_obs"][1].astype(np.int64).copy()
            r[-2 * C:] = nxt[-2 * C:]
            nxt = r
        prev = nxt
    partner = TapePartner([(int(a[2]), int(a[3])) for a in z["actions"]], expect, n)
    venv = OvercookedVecEnv(_arglist(st), n, partner=partner, subtask_order=st["subtasks"],
                            terminal_obs=terminal_obs)
    assert venv.num_envs == n
    obs = venv.reset()
    flat0 = np.concatenate([obs[k].astype(np.int64) for k in KEYS], axis=1)
    assert (flat0 == z["reset_obs"][0][None, :]).all()
    for k, v in obs.items():
        assert v.dtype == SPACE_DTYPE[k] and v.shape[0] == n
    dones_seen = 0
    for k in range(K):
        a = z["actions"][k]
        obs, rew, done, infos = venv.step(np.tile(np.array([[a[0], a[1]]]), (n, 1)))
        exp_r = np.float32(z["rew_bits"][k:k + 1].view(np.float64)[0])
        assert (rew == exp_r).all() and rew.dtype == np.float32
        assert (done == bool(z["done"][k])).all() and done.dtype == bool
        flat = np.concatenate([obs[key].astype(np.int64) for key in KEYS], axis=1)
        if z["done"][k]:
            dones_seen += 1
            # auto-reset: the returned obs is the new episode's first obs (comm persists)
            r = z["reset_obs"][0].astype(np.int64).copy()
            r[-2 * C:] = z["obs"][k][0][-2 * C:]
            assert (flat == r[None, :]).all()
            assert all("episode" in i for i in infos)
            assert infos[0]["episode"]["l"] <= st["max_num_timesteps"]
            if terminal_obs:
                t = infos[0]["terminal_observation"]
                tf = np.concatenate([t[key].astype(np.int64).reshape(-1) for key in KEYS])
                assert (tf == z["obs"][k][0]).all()
                assert np.float32(t["timestep"][0]) == np.float32(z["ts_bits"][k:k + 1, 0].view(np.float64)[0])
            else:
                assert "terminal_observation" not in infos[0]
            assert np.float32(obs["timestep"][0, 0]) == 0.0
        else:
            assert (flat == z["obs"][k][0][None, :]).all()
            assert all(i == {} for i in infos)
    assert dones_seen >= 3 and partner.updates == K
    m = venv.metrics()
    assert m["env_steps"] == n * K and m["episodes"] == n * dones_seen


def test_vec_env_random_partner_and_tensor_api():
    from gym_comm_amd.vec_env import OvercookedVecEnv
    arg = SimpleNamespace(level="random-open-divider_salad_small", num_agents=2, max_num_timesteps=30,
                          ego_config={}, partner_config={}, num_communication=3, communication_on=True,
                          ego_led=False, fow_radius=2)
    n = 1000
    venv = OvercookedVecEnv(arg, n, seed=3)
    obs = venv.reset_tensors()
    assert obs["object_encodings_x"].shape == (n, 4) and obs["agent1_comm"].shape == (n, 3)
    gen = torch.Generator(device="cuda").manual_seed(0)
    total_done = 0
    for _ in range(100):
        a = torch.stack([torch.randint(0, 4, (n,), generator=gen, device="cuda"),
                         torch.randint(0, 3, (n,), generator=gen, device="cuda")], dim=1)
        obs, rew, done = venv.step_tensors(a)
        assert rew.dtype == torch.float64 and rew.shape == (n,) and bool((rew <= 3 * 9).all())
        total_done += int(done.sum().item())
    assert total_done == n * 3                       # T = 30 -> every env finishes 3 episodes
    assert venv.metrics()["episodes"] == total_done

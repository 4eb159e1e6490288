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
    ep_r, ep_l = 0.0, 0            # what Monitor would report: fp64 running sum, step count
    for k in range(K):
        a = z["actions"][k]
        obs, rew, done, infos = venv.step(np.tile(np.array([[a[0], a[1]]]), (n, 1)))
        ep_r += float(z["rew_bits"][k:k + 1].view(np.float64)[0])
        ep_l += 1
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
            # the kernel keeps the episode statistics: exactly the left-to-right fp64 sum
            assert all(i["episode"] == {"r": ep_r, "l": ep_l} for i in infos), (infos[0], ep_r, ep_l)
            ep_r, ep_l = 0.0, 0
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


def _tomato_args(T=40, C=2, level="open-divider_tomato"):
    return SimpleNamespace(level=level, num_agents=2, max_num_timesteps=T, ego_config={},
                           partner_config={}, num_communication=C, communication_on=True,
                           ego_led=False, fow_radius=2)


def test_step_tensors_graph_replay_equals_eager_launches():
    """use_graph=True (partner draw + fused step replayed as one hipGraph) walks the same
    trajectory as the eager launches: observations, rewards, done flags, episode statistics."""
    from gym_comm_amd.vec_env import OvercookedVecEnv
    n, K = 777, 130
    a = OvercookedVecEnv(_tomato_args(), n, seed=3, use_graph=False)
    b = OvercookedVecEnv(_tomato_args(), n, seed=3, use_graph=True)
    oa, ob = a.reset_tensors(), b.reset_tensors()
    gen = torch.Generator(device="cuda").manual_seed(1)
    finished = 0
    for k in range(K):
        ego = torch.stack([torch.randint(0, 4, (n,), generator=gen, device="cuda"),
                           torch.randint(0, 2, (n,), generator=gen, device="cuda")], dim=1)
        oa, ra, da = a.step_tensors(ego)
        ob, rb, db = b.step_tensors(ego)
        assert torch.equal(da, db) and torch.equal(ra, rb), k
        for key in oa:
            assert torch.equal(oa[key], ob[key]), (k, key)
        assert torch.equal(a.episode_returns, b.episode_returns)
        assert torch.equal(a.episode_lengths, b.episode_lengths)
        finished += int(da.sum().item())
        # Monitor semantics: a finished env's row holds the episode total until the next step
        ln = a.episode_lengths.cpu().numpy()
        assert (ln[da.cpu().numpy() != 0] <= 40).all() and (ln >= 1).all()
    assert finished >= 3 * n and a.metrics() == b.metrics()


@pytest.mark.parametrize("sample", [False, True])
def test_closed_loop_with_torch_policies(sample, oracle_lib):
    """TorchPolicyPartner in both seats (an MLP on the viewer's observation rows), the whole
    step -- ego policy, partner policy, fused kernel, episode statistics -- as one hipGraph.
    The env side is checked against the oracle: it is fed the actions the policies wrote."""
    from gym_comm_amd.vec_env import MLPPolicy, OvercookedVecEnv, TorchPolicyPartner
    n, K, C, T = 640, 90, 3, 30
    args = _tomato_args(T=T, C=C, level="full-divider_salad")
    partner = TorchPolicyPartner(MLPPolicy(9, C, hidden=32, seed=5).cuda(), sample=sample, seed=77)
    venv = OvercookedVecEnv(args, n, partner=partner, seed=1, obs_dtype=torch.float32)
    ego = TorchPolicyPartner(MLPPolicy(9, C, hidden=32, seed=6).cuda(), sample=sample)
    venv.reset_tensors()
    loop = venv.closed_loop(ego, graph=True)
    ora = oracle_lib.OracleBatch(venv._b.level.blob, n, threads=4)
    comm = np.zeros((2, n), np.int32)
    seen = set()
    ep_ret, prev_done = np.zeros(n), np.zeros(n, np.int32)
    for k in range(K):
        obs, rew, done = loop.step()
        acts = venv._act.cpu().numpy()               # what the two policies chose this step
        assert acts[[0, 2]].min() >= 0 and acts[[0, 2]].max() <= 3 and acts[[1, 3]].max() < C
        seen |= set(np.unique(acts[0]).tolist())
        oo, to, ro, do = ora.multi_step(acts, comm, 2, 0, C, auto_reset=True)
        assert np.array_equal(done.cpu().numpy(), do), k
        assert np.array_equal(rew.cpu().numpy().view(np.uint64), ro.view(np.uint64)), k
        assert np.array_equal(venv._b.obs.cpu().numpy(), oo.astype(np.float32)), k
        assert obs["agent1_comm"].shape == (n, C) and obs.rows.shape == (22 + 9 + 2 * C, n)
        # running return: an env that finished at the previous step starts over
        ep_ret = np.where(prev_done != 0, ro, ep_ret + ro)
        assert np.array_equal(venv.episode_returns.cpu().numpy().view(np.uint64), ep_ret.view(np.uint64)), k
        prev_done = do.copy()
    assert len(seen) > 1 or not sample                # sampled actions vary (an argmax policy may be constant)
    assert venv.metrics()["env_steps"] == n * K and venv.metrics()["episodes"] >= n * (K // T)


def test_per_env_views_of_a_batch_match_the_oracle_render(oracle_lib):
    """VecEnv.get_attr('base_env', indices) / env_method on a 4096-env batch: str(), t, holdings,
    completed_subtasks of envs 0, 63, 64, 4095 against the oracle's own per-env rendering
    (oracle.render_ascii) after every step -- what EpisodeRecorder / ParallelEpisodeRecorder read
    (episode_recorder.py:15-46,48-85)."""
    from gym_comm_amd.vec_env import BatchEpisodeRecorder, OvercookedVecEnv
    from hip_util import scripted_then_random
    n, K, C, T = 4096, 70, 2, 35
    level = "open-divider_salad"
    venv = OvercookedVecEnv(_tomato_args(T=T, C=C, level=level), n, seed=2)
    blob = venv._b.level.blob
    ora = oracle_lib.OracleBatch(blob, n, threads=8)
    comm = np.zeros((2, n), np.int32)
    rng = np.random.default_rng(8)
    mv = scripted_then_random(rng, level, K, 2, n, nact=4)
    cm = rng.integers(0, C, (K, 2, n)).astype(np.int32)
    idx = [0, 63, 64, 4095]
    venv.reset_tensors()
    views = venv.get_attr("base_env", idx)
    assert [v._index for v in views] == idx
    worlds = venv.get_attr("world", idx)              # persistent objects, refreshed in place
    for v, i in zip(views, idx):
        assert str(v) == oracle_lib.render_ascii(blob, ora.snapshot(i))
    with pytest.raises(RuntimeError):
        views[0].step({"agent-0": (0, 1), "agent-1": (0, 1)})
    with pytest.raises(IndexError):
        venv.get_attr("t", [n])

    class Tape:                                       # the partner's recorded actions
        k = 0

        def act_into(self, obs, move_row, comm_row):
            move_row.copy_(torch.from_numpy(mv[self.k, 1]).cuda())
            comm_row.copy_(torch.from_numpy(cm[self.k, 1]).cuda())
            self.k += 1
    venv.partner = Tape()
    for k in range(K):
        ego = torch.from_numpy(np.stack([mv[k, 0], cm[k, 0]], axis=1)).cuda()
        venv.step_tensors(ego)
        a = np.stack([mv[k, 0], cm[k, 0], mv[k, 1], cm[k, 1]]).astype(np.int32)
        ora.multi_step(a, comm, 2, 0, C, auto_reset=True)
        strs = venv.env_method("__str__", indices=idx)
        ts = venv.get_attr("t", idx)
        comp = venv.get_attr("completed_subtasks", idx)
        for j, i in enumerate(idx):
            snap = ora.snapshot(i)
            assert strs[j] == oracle_lib.render_ascii(blob, snap), (k, i)
            assert ts[j] == snap["t"] and comp[j] == snap["completed"].tolist(), (k, i)
            assert venv.get_attr("world", i)[0] is worlds[j]
            held = [ag.holding is not None for ag in venv.get_attr("sim_agents", i)[0]]
            assert held == [snap["agents"][a][2] >= 0 for a in range(2)], (k, i)
    frames = venv.env_method("render_frame", indices=[63])
    assert frames[0].ndim == 3 and frames[0].dtype == np.uint8
    assert venv.get_attr("num_envs", [0, 1]) == [n, n]          # a VecEnv attribute: replicated

    # the ParallelEpisodeRecorder-shaped hook: frames of the recorded envs, one list per episode
    rec = BatchEpisodeRecorder(venv, record_interval=1, indices=(0, 4095), ascii_only=True)
    rec.reset()
    venv.partner = Tape()
    for k in range(K):
        rec.step(np.stack([mv[k, 0], cm[k, 0]], axis=1))
    assert len(rec.episodes) >= 2 * (K // T)
    for i, ep, frames in rec.episodes:
        assert i in (0, 4095) and 1 <= len(frames) <= T and all(isinstance(f, str) for f in frames)


@pytest.mark.gpu
def test_numpy_api_arrays_with_and_without_reused_host_buffers():
    """The numpy boundary through oc_pack_host (include/oc_hostio.h): the arrays have the declared
    dtypes and equal the device tensors; by default they are fresh every step; with
    reuse_host_buffers=True they are views of two alternating pinned buffers (valid until the
    step after next) with the same contents."""
    import numpy as np
    import torch
    from types import SimpleNamespace
    from gym_comm_amd.vec_env import OvercookedVecEnv, SPACE_DTYPE
    arg = SimpleNamespace(level="open-divider_salad", num_agents=2, max_num_timesteps=30, ego_config={},
                          partner_config={}, num_communication=3, communication_on=True, ego_led=False,
                          fow_radius=1)
    n = 777
    va = OvercookedVecEnv(arg, n, seed=3)
    vb = OvercookedVecEnv(arg, n, seed=3, reuse_host_buffers=True)
    oa, ob = va.reset(), vb.reset()
    rng = np.random.default_rng(0)
    kept = []
    for k in range(70):
        acts = np.stack([rng.integers(0, 4, n), rng.integers(0, 3, n)], axis=1)
        oa, ra, da, ia = va.step(acts)
        ob, rb, db, ib = vb.step(acts)
        dev = va._obs_tensors(0)
        for key in oa:
            assert oa[key].dtype == np.dtype(SPACE_DTYPE[key]) == ob[key].dtype, key
            want = dev[key].cpu().numpy() if key != "timestep" else va._b.timestep.cpu().numpy().reshape(-1, 1)
            assert np.array_equal(oa[key], want.astype(SPACE_DTYPE[key])), (k, key)
            assert np.array_equal(oa[key], ob[key]), (k, key)
        assert ra.dtype == np.float32 and np.array_equal(ra, va._b.shaped_reward.cpu().numpy().astype(np.float32))
        assert np.array_equal(ra, rb) and np.array_equal(da, db) and da.dtype == bool
        assert np.array_equal(da, va._b.done.cpu().numpy() != 0)
        for i in np.nonzero(da)[0][:5]:
            assert ia[i]["episode"] == ib[i]["episode"] and ia[i]["episode"]["l"] > 0
        kept.append((oa["object_encodings_x"], oa["object_encodings_x"].copy()))
    for view, snapshot in kept:            # fresh arrays stay what they were
        assert np.array_equal(view, snapshot)

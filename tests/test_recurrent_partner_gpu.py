"""-m gpu: the stateful partner protocol (gym_comm_amd.vec_env.RecurrentPolicyPartner + RolloutSink).

The reference seats ``OnPolicyAgent(RecurrentPPO('MultiInputPolicy', ...))`` beside the env
(trainer.py:92-112; pantheonrl/common/agents.py:112-214): per-env LSTM state, ``episode_starts``,
value / log-prob recorded at ``get_action``, the reward added at ``update``.  Checked here:
  * the batched partner inside ``OvercookedVecEnv`` (eager launches AND the captured hipGraph, with
    the sink recording inside the graph) walks the same trajectory, bit for bit, as a hand-written
    eager loop that keeps the LSTM state, the episode_start mask, the sampling noise and the
    rollout buffer itself -- 64 envs x 200 steps with T = 25, so every env resets several times;
  * the env side against the oracle, fed the actions that were recorded;
  * per env: a float64 copy of the policy run on ONE env at a time (batch of 1, its own state)
    reproduces the greedy actions and the values of the batch.
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

S, C, T = 3, 2, 25
F = 22 + S + 2 * C
HID = 32


class LSTMPolicy(torch.nn.Module):
    """obs rows (+ timestep) -> LSTMCell -> move / comm logits and a value; feature-major rows in,
    [n, k] logits out.  Honours episode_start itself too (sb3_contrib resets the states of starting
    envs inside the policy, recurrent/policies.py `_process_sequence`)."""

    def __init__(self, seed=0, dtype=torch.float32):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.cell = torch.nn.LSTMCell(F + 1, HID)
        self.move = torch.nn.Linear(HID, 4)
        self.comm = torch.nn.Linear(HID, C)
        self.val = torch.nn.Linear(HID, 1)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 0.8)
        self.to(dtype)
        self.dtype = dtype

    def forward(self, obs, state, episode_start):
        x = torch.cat([obs.rows.to(self.dtype).T, obs.timestep.to(self.dtype).unsqueeze(1)], dim=1)
        keep = (1.0 - episode_start.to(self.dtype)).unsqueeze(1)
        h, c = self.cell(x, (state[0] * keep, state[1] * keep))
        return self.move(h), self.comm(h), (h, c), self.val(h)


def _args():
    return SimpleNamespace(level="open-divider_tomato", num_agents=2, max_num_timesteps=T, ego_config={},
                           partner_config={}, num_communication=C, communication_on=True, ego_led=False,
                           fow_radius=2)


def _state(n, dtype=torch.float32):
    return (torch.zeros(n, HID, dtype=dtype, device="cuda"), torch.zeros(n, HID, dtype=dtype, device="cuda"))


@pytest.mark.parametrize("use_graph", [False, True], ids=["eager", "graph"])
def test_recurrent_partner_equals_hand_written_loop_and_oracle(use_graph, oracle_lib):
    from gym_comm_amd.vec_env import OvercookedVecEnv, RecurrentPolicyPartner, RolloutSink
    n, K = 64, 200
    pol = LSTMPolicy(seed=3).cuda()
    sink = RolloutSink(K, n, F, obs_dtype=torch.float32)
    partner = RecurrentPolicyPartner(pol, _state(n), sample=True, sink=sink)
    assert partner.graph_safe
    venv = OvercookedVecEnv(_args(), n, partner=partner, seed=5, obs_dtype=torch.float32, use_graph=use_graph)
    ref = OvercookedVecEnv(_args(), n, seed=5, obs_dtype=torch.float32)        # stepped by the hand-written loop
    gen = torch.Generator(device="cuda").manual_seed(11)
    ego = torch.stack([torch.randint(0, 4, (K, n), generator=gen, device="cuda"),
                       torch.randint(0, C, (K, n), generator=gen, device="cuda")], dim=2).to(torch.int32)

    # ---- the hand-written loop: state, masks, sampling and buffer kept by hand (agents.py:112-214) ----
    torch.cuda.manual_seed(1234)
    ref.reset_tensors()
    h, c = _state(n)
    es = torch.ones(n, device="cuda")
    buf = {k: [] for k in ("obs", "act", "logp", "val", "es", "rew", "done")}
    with torch.no_grad():
        for k in range(K):
            ov = ref._obs_tensors(1)                                   # the partner sees viewer 1
            start = es.unsqueeze(1) > 0                                # starting envs: initial (zero) state
            h, c = torch.where(start, torch.zeros_like(h), h), torch.where(start, torch.zeros_like(c), c)
            mv, cm, (h, c), val = pol(ov, (h, c), es)
            a_mv = (mv - torch.empty_like(mv).exponential_().log_()).argmax(dim=1)
            a_cm = (cm - torch.empty_like(cm).exponential_().log_()).argmax(dim=1)
            logp = (torch.log_softmax(mv, 1).gather(1, a_mv[:, None]) + torch.log_softmax(cm, 1).gather(1, a_cm[:, None]))[:, 0]
            buf["obs"].append(ov.rows.clone()); buf["es"].append(es.clone())
            buf["act"].append(torch.stack([a_mv, a_cm]).to(torch.int32)); buf["logp"].append(logp); buf["val"].append(val[:, 0])
            acts = torch.stack([ego[k, :, 0], ego[k, :, 1], a_mv.to(torch.int32), a_cm.to(torch.int32)]).contiguous()
            _, _, rew, done = ref._b.multi_step(acts)
            buf["rew"].append(rew.clone()); buf["done"].append(done.clone())
            es = done.float()

    # ---- the partner protocol ----
    torch.cuda.manual_seed(1234)
    venv.reset_tensors()
    assert float(partner.episode_start.min()) == 1.0
    ora = oracle_lib.OracleBatch(venv._b.level.blob, n, threads=4)
    comm = np.zeros((2, n), np.int32)
    resets = 0
    for k in range(K):
        _, rew, done = venv.step_tensors(ego[k].contiguous())
        assert torch.equal(rew.view(torch.int64), buf["rew"][k].view(torch.int64)), k
        assert torch.equal(done, buf["done"][k]), k
        resets += int(done.sum().item())
    assert resets >= 4 * n                                              # every env started over several times
    assert sink.steps() == K and sink.full()
    stack = lambda key: torch.stack(buf[key])
    assert torch.equal(sink.obs, stack("obs"))
    assert torch.equal(sink.actions, stack("act"))
    assert torch.equal(sink.log_probs, stack("logp")) and torch.equal(sink.values, stack("val"))
    assert torch.equal(sink.episode_starts, stack("es")) and torch.equal(sink.dones, stack("done"))
    assert torch.equal(sink.rewards.view(torch.int64), stack("rew").view(torch.int64))
    assert torch.equal(partner.state[0], h) and torch.equal(partner.state[1], c)
    assert torch.equal(venv._b.state, ref._b.state) and torch.equal(venv._b.obs, ref._b.obs)
    # ---- env side against the oracle, fed the recorded actions ----
    act = sink.actions.cpu().numpy()
    eg = ego.cpu().numpy()
    for k in range(K):
        a4 = np.stack([eg[k, :, 0], eg[k, :, 1], act[k, 0], act[k, 1]]).astype(np.int32)
        oo, to, ro, do = ora.multi_step(a4, comm, 2, 0, C, auto_reset=True)
        assert np.array_equal(do, sink.dones[k].cpu().numpy()), k
        assert np.array_equal(ro.view(np.uint64), sink.rewards[k].cpu().numpy().view(np.uint64)), k
        if k + 1 < K:      # what the partner was shown before step k + 1 = viewer 1 after step k
            assert np.array_equal(oo[1].astype(np.float32), sink.obs[k + 1].cpu().numpy()), k


def test_recurrent_partner_matches_per_env_float64_loops():
    """Greedy actions and values of the batch against the same policy in float64 run on ONE env at a
    time -- a batch of 1 with its own state, episode_start and (cloned) env -- for 6 envs x 60 steps."""
    from gym_comm_amd.batched import BatchedOvercooked
    from gym_comm_amd.vec_env import ObsView, OvercookedVecEnv, RecurrentPolicyPartner
    n, K, picks = 64, 60, [0, 7, 19, 33, 62, 63]
    pol64 = LSTMPolicy(seed=9, dtype=torch.float64).cuda()
    partner = RecurrentPolicyPartner(pol64, _state(n, torch.float64), sample=False)
    venv = OvercookedVecEnv(_args(), n, partner=partner, seed=2, obs_dtype=torch.float32)
    gen = torch.Generator(device="cuda").manual_seed(4)
    ego = torch.stack([torch.randint(0, 4, (K, n), generator=gen, device="cuda"),
                       torch.randint(0, C, (K, n), generator=gen, device="cuda")], dim=2).to(torch.int32)
    venv.reset_tensors()
    played, values = [], []
    for k in range(K):
        venv.step_tensors(ego[k].contiguous())
        played.append(venv._act[2:4].clone())
        values.append(partner.value.clone())
    played, values = torch.stack(played), torch.stack(values)
    for i in picks:
        one = BatchedOvercooked("open-divider_tomato", num_envs=1, max_num_timesteps=T, num_communication=C,
                                fow_radius=2, obs_dtype=torch.float32, episode_stats=True)
        one.reset()
        one.observe()
        h, c = _state(1, torch.float64)
        es = torch.ones(1, device="cuda")
        with torch.no_grad():
            for k in range(K):
                ov = ObsView()
                ov.rows, ov.timestep = one.obs[1], one.timestep
                mv, cm, (h, c), val = pol64(ov, (h, c), es)
                a = torch.stack([mv.argmax(1), cm.argmax(1)]).to(torch.int32)
                assert torch.equal(a[:, 0], played[k, :, i]), (i, k)
                assert abs(float(val[0, 0]) - float(values[k, i])) < 1e-5, (i, k)
                acts = torch.stack([ego[k, i:i + 1, 0], ego[k, i:i + 1, 1], a[0], a[1]]).contiguous()
                _, _, _, done = one.multi_step(acts)
                es = done.float()

"""Vectorised trainer boundary: N Overcooked envs behind the
Stable-Baselines3 ``VecEnv`` API, ego-perspective, with the partner policy evaluated on
the whole batch.

The reference trains on ONE env: ``RecurrentPPO`` wraps ``OvercookedMultiEnv`` in
``Monitor + DummyVecEnv(1)`` and pantheonrl asks the partner agent for one action per
step (pantheonrl/common/multiagentenv.py:149-215, trainer.py:87-121).  Here the same
ego/partner protocol runs on tensors:

    obs_ego = venv.reset()
    obs_ego, rewards, dones, infos = venv.step(ego_actions)      # ego_actions [n, 2] = (move, comm)

Call order of one step, as ``MultiAgentEnv.step`` (multiagentenv.py:149-215):
``partner.act_into(obs_partner)`` on the observations the partner saw after the previous step
(``_get_actions``: ``agent.get_action(ob)``, SimultaneousEnv semantics :395-404); the fused
``oc_multi_step`` kernel once (``n_step``); ``partner.update(rewards, dones)`` if the partner
has that method (``_update_players`` -> ``OnPolicyAgent.update``, agents.py:168-185).  Finished
envs are auto-reset inside the kernel, so -- as with any SB3 VecEnv -- the observation returned
for a done env is the first observation of its next episode and ``rewards``/``dones`` belong to
the finished step.  ``infos[i]["terminal_observation"]`` is only filled with
``terminal_obs=True`` (three launches per step instead of one).  Episode return / length
(what SB3's ``Monitor`` adds: ``infos[i]["episode"]``) are kept by the kernel itself.

Partners
  ``RandomPartner``        uniform random, drawn inside the step kernel (no launch of its own);
  ``TorchPolicyPartner``   any ``torch.nn.Module`` mapping the observation dict to (move logits,
                           comm logits) -- the batched stand-in for pantheonrl's
                           ``OnPolicyAgent.get_action`` / ``update`` (agents.py:112-194);
  ``RecurrentPolicyPartner``  a STATEFUL torch policy (per-env recurrent state, ``episode_start``
                           masks, log-probs / values, a graph-capturable ``RolloutSink``) -- what
                           the reference seats there: ``OnPolicyAgent(RecurrentPPO(...))``;
  ``FusedMLPPartner``      an ``MLPPolicy`` (64 tanh units) as ONE launch of the hand-written MFMA
                           policy kernel (include/oc_policy.h), ego and partner in the same launch;
  any callable ``partner(obs_dict) -> [n, 2]`` still works (slow path).

``ClosedLoop`` (``venv.closed_loop(ego)``) captures ego policy -> partner policy -> fused step
(+ in-kernel episode statistics) in one hipGraph, so a rollout costs one graph replay per step
(or per k steps) instead of a dozen host-side launches.

Subclasses ``stable_baselines3.common.vec_env.VecEnv`` when SB3 is importable; otherwise a
structural stand-in with the same methods.  ``step_tensors`` is the zero-copy variant for
policies that live on the GPU; the numpy API packs a step's arrays with one launch of
``oc_pack_host`` (include/oc_hostio.h) and one PCIe copy.
"""
import contextlib
import ctypes
import os
import weakref

import numpy as np
import torch

from . import _lib
from .batched import BatchedOvercooked, OBS_KEYS
from .envs import OvercookedEnvironment, _arg, make_spaces

try:                                                    # pragma: no cover - SB3 absent in CI image
    from stable_baselines3.common.vec_env import VecEnv as _VecEnvBase
except Exception:
    class _VecEnvBase:                                  # structural stand-in
        def __init__(self, num_envs, observation_space, action_space):
            self.num_envs = num_envs
            self.observation_space = observation_space
            self.action_space = action_space

        def step(self, actions):
            self.step_async(actions)
            return self.step_wait()


# dtype each key has after SB3 copies it into a buffer of the declared space
# (gym_comm/envs/overcooked_env.py:41-85: Box float32 / Box int64 / MultiBinary int8)
SPACE_DTYPE = {
    "timestep": np.float32, "object_encodings_x": np.int64, "object_encodings_y": np.int64,
    "state_encodings": np.int8, "is_hidden": np.int8, "completed_subtasks": np.int8,
    "agent1_location": np.float32, "agent2_location": np.float32, "agent_is_holding": np.int8,
    "agent1_comm": np.int8, "agent2_comm": np.int8,
}


class ObsView(dict):
    """One viewer's observation: the 11 keys of get_observation2 as [n, k] tensor views, plus
    the storage they are views of -- ``rows`` ([F][n], the kernel's env-major rows; a policy
    that consumes them feature-major needs no gather, transpose or concat) and ``timestep``
    (f64 [n]).  The same objects every step; the next step overwrites their contents."""
    rows = None
    timestep = None


class RandomPartner:
    """Uniform random partner (move 0..3, comm 0..C-1) from a PCG32 stream per env.  In the
    partner seat of an ``OvercookedVecEnv`` it costs NO launch: the fused step kernel draws the
    actions itself (``oc_step_opts.alt_rng``, include/oc_hip.h) and reports them in action rows
    2, 3.  Anywhere else (``act_into`` / call) it is one launch of ``oc_random_actions``."""
    graph_safe = True
    in_kernel = True            # OvercookedVecEnv hands `rng_state(n)` to the step kernel

    def __init__(self, num_comm, seed=0, device="cuda"):
        self.C = int(num_comm)
        self.seed = int(seed)
        self.device = torch.device(device)
        self._L = _lib.load()
        self._rng = None

    def rng_state(self, n):
        """int32 [n] tensor holding the n PCG32 states (uint32 bit patterns)."""
        return self._state(n)

    def _state(self, n):
        if self._rng is None or self._rng.numel() != n:
            g = torch.Generator(device="cpu").manual_seed(self.seed)
            self._rng = torch.randint(0, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int64
                                      ).to(torch.int32).to(self.device)
        return self._rng

    def act_into(self, obs, move_row, comm_row):
        n = move_row.numel()
        dev = move_row.device.index
        if torch.cuda.current_device() != dev:
            with torch.cuda.device(dev):
                return self.act_into(obs, move_row, comm_row)
        rc = self._L.oc_random_actions(self._state(n).data_ptr(), move_row.data_ptr(), comm_row.data_ptr(),
                                       self.C, n, torch._C._cuda_getCurrentRawStream(dev))
        if rc:
            _lib.check(rc, "oc_random_actions", self._L)

    def get_state(self, n):
        """A copy of the generator state for n envs (allocated on first use)."""
        return self._state(n).clone()

    def set_state(self, st):
        self._state(st.numel()).copy_(st)

    def __call__(self, obs):
        n = next(iter(obs.values())).shape[0]
        out = torch.empty((2, n), dtype=torch.int32, device=self.device)
        self.act_into(obs, out[0], out[1])
        return out.T


class MLPPolicy(torch.nn.Module):
    """A small two-layer policy over get_observation2's 22 + S + 2C features (+ timestep):
    returns (move logits [4][n], comm logits [C][n]).  Feature-major on purpose: the kernel
    leaves a viewer's observation as [F][n] rows in HBM (float32 with ``obs_dtype=float32``), so
    the first layer is one GEMM on those rows as they lie -- no gather, no transpose, no concat.
    An observation dict without ``rows`` (any other env) is concatenated the slow way."""
    feature_major = True

    def __init__(self, num_subtasks, num_comm, hidden=64, seed=0):
        super().__init__()
        F = 22 + int(num_subtasks) + 2 * int(num_comm)
        g = torch.Generator().manual_seed(int(seed))
        init = lambda *shape: torch.nn.Parameter(
            (torch.rand(shape, generator=g) * 2 - 1) / float(np.sqrt(shape[-1])))
        self.w1, self.b1, self.wt = init(hidden, F), init(hidden, 1), init(hidden, 1)
        self.w2, self.b2 = init(4 + int(num_comm), hidden), init(4 + int(num_comm), 1)
        self.C = int(num_comm)

    def forward(self, obs):
        rows = getattr(obs, "rows", None)
        if rows is None:
            rows = torch.cat([obs[k].reshape(obs[k].shape[0], -1) for k in OBS_KEYS], dim=1).T
            ts = obs["timestep"].reshape(1, -1)
        else:
            ts = obs.timestep.unsqueeze(0)
        if rows.dtype != torch.float32:
            rows = rows.to(torch.float32)
        h = torch.addmm(self.b1, self.w1, rows)                 # [H][n]
        h = torch.addcmul(h, self.wt, ts.to(torch.float32))
        out = torch.addmm(self.b2, self.w2, torch.tanh_(h))     # [4 + C][n]
        return out[:4], out[4:]


class TorchPolicyPartner:
    """A torch module in the partner (or ego) seat.  ``policy(obs) -> (move_logits,
    comm_logits)``, each [n, k] -- or [k, n] when ``policy.feature_major`` is true.  Actions are
    the argmax (``sample=False``) or a categorical sample (Gumbel-max on the default CUDA
    generator, so the draw is hipGraph-capturable), written straight into the kernel's action
    rows.  ``update(rewards, dones)`` -- pantheonrl's ``Agent.update`` -- is forwarded to
    ``on_update`` (e.g. a rollout buffer's add), called after every step."""

    def __init__(self, policy, sample=True, seed=None, device="cuda", on_update=None):
        self.policy = policy
        self.sample = bool(sample)
        self.device = torch.device(device)
        self.on_update = on_update
        # stateless between steps and fixed-shape => one capture serves every later step
        self.graph_safe = on_update is None
        if seed is not None:
            with torch.cuda.device(self.device):
                torch.cuda.manual_seed(int(seed))

    def _pick(self, logits, dim):
        if self.sample:     # argmax(logits - log E), E ~ Exp(1)  ==  a categorical sample
            logits = logits - torch.empty_like(logits).exponential_().log_()
        return logits.argmax(dim=dim)

    @torch.no_grad()
    def act_into(self, obs, move_row, comm_row):
        mv, cm = self.policy(obs)
        dim = 0 if getattr(self.policy, "feature_major", False) else 1
        move_row.copy_(self._pick(mv, dim))
        comm_row.copy_(self._pick(cm, dim))

    def __call__(self, obs):
        n = next(iter(obs.values())).shape[0]
        out = torch.empty((2, n), dtype=torch.int32, device=self.device)
        self.act_into(obs, out[0], out[1])
        return out.T

    def update(self, rewards, dones):
        if self.on_update is not None:
            self.on_update(rewards, dones)


class RolloutSink:
    """What pantheonrl's ``OnPolicyAgent`` keeps per step in its rollout buffer -- ``buf.add(obs,
    action, [0], episode_start, value, log_prob)`` in ``get_action`` and ``buf.rewards[pos - 1] +=
    reward`` in ``update`` (pantheonrl/common/agents.py:112-214) -- for a whole batch, in
    PREALLOCATED ``[n_steps][...][n]`` device tensors whose write position is itself a device scalar.
    Every write is an ``index_copy_`` / ``index_add_`` on that scalar, so recording neither
    synchronises nor changes shape: it can sit inside a captured hipGraph (a learner's hook that runs
    host code per step cannot).  ``full()`` / ``steps()`` read the counter (one sync, when asked);
    ``reset()`` starts the next rollout.  Past ``n_steps`` the position wraps (a ring)."""

    def __init__(self, n_steps, n, rows, device="cuda", obs_dtype=torch.int32):
        dev = torch.device(device)
        T = self.n_steps = int(n_steps)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self.obs = z((T, int(rows), n), obs_dtype)       # the viewer's [F][n] rows as they lie
        self.timestep = z((T, n), torch.float64)
        self.actions = z((T, 2, n), torch.int32)         # (move, comm)
        self.log_probs = z((T, n), torch.float32)
        self.values = z((T, n), torch.float32)
        self.rewards = z((T, n), torch.float64)
        self.episode_starts = z((T, n), torch.float32)
        self.dones = z((T, n), torch.int32)
        self.pos = z((1,), torch.int64)                  # next slot
        self.last = z((1,), torch.int64)                 # slot of the most recent add()
        self.count = z((1,), torch.int64)                # adds since reset()

    def add(self, rows, timestep, move, comm, log_prob, value, episode_start):
        i = self.pos
        self.obs.index_copy_(0, i, rows.unsqueeze(0))
        self.timestep.index_copy_(0, i, timestep.unsqueeze(0))
        self.actions.index_copy_(0, i, torch.stack([move, comm]).to(torch.int32).unsqueeze(0))
        self.log_probs.index_copy_(0, i, log_prob.to(torch.float32).unsqueeze(0))
        if value is not None:
            self.values.index_copy_(0, i, value.reshape(1, -1).to(torch.float32))
        self.episode_starts.index_copy_(0, i, episode_start.to(torch.float32).unsqueeze(0))
        self.rewards.index_fill_(0, i, 0.0)              # buf.add(..., [0], ...): update() adds
        self.last.copy_(i)
        self.pos.add_(1).remainder_(self.n_steps)
        self.count.add_(1)

    def add_reward(self, rewards, dones):
        """``update(reward, done)`` of the step the most recent ``add`` belongs to."""
        self.rewards.index_add_(0, self.last, rewards.to(torch.float64).unsqueeze(0))
        self.dones.index_copy_(0, self.last, dones.to(torch.int32).unsqueeze(0))

    def steps(self):
        return int(self.count.item())

    def full(self):
        return self.steps() >= self.n_steps

    def reset(self):
        self.pos.zero_()
        self.last.zero_()
        self.count.zero_()

    def get_state(self):
        return self.pos.clone(), self.last.clone(), self.count.clone()

    def set_state(self, st):
        self.pos.copy_(st[0])
        self.last.copy_(st[1])
        self.count.copy_(st[2])


class RecurrentPolicyPartner:
    """A STATEFUL torch policy in the partner (or ego) seat: the batched form of what the
    reference seats there -- ``OnPolicyAgent(RecurrentPPO('MultiInputPolicy', ...))``
    (trainer.py:92-112; pantheonrl/common/agents.py:112-214; sb3_contrib's recurrent policies
    carry LSTM states per env and reset them where ``episode_starts`` is set).

        policy(obs, state, episode_start) -> (move_logits, comm_logits, new_state[, value])

    ``obs``: the viewer's ``ObsView`` ([n, k] keys, ``rows`` [F][n], ``timestep``); ``state``: a
    tuple of ``[n, ...]`` tensors OWNED BY THE PARTNER (updated in place, so a captured graph keeps
    their addresses); ``episode_start``: float32 [n], 1 where the env's previous step returned done
    (the step kernel's ``done`` row, handed over by ``update``) and for every env on the first step
    after a reset -- SB3's ``_last_episode_starts``.  With ``mask_state`` (default) the partner
    itself puts the initial state back into the rows of starting envs before the call, so a policy
    that ignores ``episode_start`` is still correct.  Logits are ``[n, k]``, or ``[k, n]`` when
    ``policy.feature_major``; ``value`` is optional ([n] or [n, 1]).

    Actions: argmax, or a categorical sample (Gumbel-max on the default CUDA generator:
    hipGraph-capturable), written into the kernel's action rows; their log-probability under the
    policy is computed either way.  ``sink`` (a ``RolloutSink``) records (obs, action, log_prob,
    value, episode_start) at ``act_into`` and (reward, done) at ``update`` WITHOUT leaving the
    device or the graph; ``on_update(rewards, dones)`` is the host-side learner hook (it switches
    graph capture off, as for ``TorchPolicyPartner``)."""

    def __init__(self, policy, state, sample=True, seed=None, device="cuda", sink=None, on_update=None,
                 mask_state=True):
        self.policy = policy
        self.sample = bool(sample)
        self.device = torch.device(device)
        self.sink = sink
        self.on_update = on_update
        self.mask_state = bool(mask_state)
        self.graph_safe = on_update is None
        state = tuple(state) if isinstance(state, (tuple, list)) else (state,)
        self.initial = tuple(s.detach().clone().to(self.device) for s in state)
        self.state = tuple(s.clone() for s in self.initial)
        n = self.state[0].shape[0]
        self.episode_start = torch.ones(n, dtype=torch.float32, device=self.device)
        self.log_prob = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.value = torch.zeros(n, dtype=torch.float32, device=self.device)
        if seed is not None:
            with torch.cuda.device(self.device):
                torch.cuda.manual_seed(int(seed))

    def reset(self):
        """A fresh rollout: every env starts an episode (called by ``reset_tensors``)."""
        self.episode_start.fill_(1.0)
        for s, s0 in zip(self.state, self.initial):
            s.copy_(s0)

    def _pick(self, logits, dim):
        if self.sample:
            logits = logits - torch.empty_like(logits).exponential_().log_()
        return logits.argmax(dim=dim)

    @torch.no_grad()
    def act_into(self, obs, move_row, comm_row):
        es = self.episode_start
        if self.mask_state:
            for s, s0 in zip(self.state, self.initial):
                s.copy_(torch.where(es.view((-1,) + (1,) * (s.dim() - 1)) > 0, s0, s))
        out = self.policy(obs, self.state, es)
        mv, cm, new_state = out[0], out[1], out[2]
        value = out[3] if len(out) > 3 else None
        dim = 0 if getattr(self.policy, "feature_major", False) else 1
        a_mv, a_cm = self._pick(mv, dim), self._pick(cm, dim)
        lp = (torch.log_softmax(mv.float(), dim=dim).gather(dim, a_mv.unsqueeze(dim)).squeeze(dim)
              + torch.log_softmax(cm.float(), dim=dim).gather(dim, a_cm.unsqueeze(dim)).squeeze(dim))
        move_row.copy_(a_mv)
        comm_row.copy_(a_cm)
        self.log_prob.copy_(lp)
        if value is not None:
            self.value.copy_(value.reshape(-1))
        new_state = tuple(new_state) if isinstance(new_state, (tuple, list)) else (new_state,)
        for s, ns in zip(self.state, new_state):
            s.copy_(ns)
        if self.sink is not None:
            rows = getattr(obs, "rows", None)
            if rows is None:
                rows = torch.cat([obs[k].reshape(obs[k].shape[0], -1) for k in OBS_KEYS], dim=1).T
            ts = obs.timestep if getattr(obs, "timestep", None) is not None else obs["timestep"].reshape(-1)
            self.sink.add(rows, ts, move_row, comm_row, lp, value, es)

    def __call__(self, obs):
        n = self.episode_start.numel()
        out = torch.empty((2, n), dtype=torch.int32, device=self.device)
        self.act_into(obs, out[0], out[1])
        return out.T

    def update(self, rewards, dones):
        """pantheonrl's ``Agent.update(reward, done)`` for the batch: ``dones`` become the next
        step's ``episode_start`` (agents.py:207-208), the reward joins the recorded transition."""
        self.episode_start.copy_(dones)
        if self.sink is not None:
            self.sink.add_reward(rewards, dones)
        if self.on_update is not None:
            self.on_update(rewards, dones)

    def get_state(self, n=None):
        return (tuple(s.clone() for s in self.state), self.episode_start.clone(), self.log_prob.clone(),
                self.value.clone(), None if self.sink is None else self.sink.get_state())

    def set_state(self, st):
        for s, v in zip(self.state, st[0]):
            s.copy_(v)
        self.episode_start.copy_(st[1])
        self.log_prob.copy_(st[2])
        self.value.copy_(st[3])
        if self.sink is not None:
            self.sink.set_state(st[4])


class FusedMLPPartner:
    """An ``MLPPolicy`` in the partner (or ego) seat as ONE launch of the hand-written policy
    kernel (include/oc_policy.h, csrc/oc_policy.hip): both products on the matrix cores
    (v_mfma_f32_32x32x16_f16, one wave = 32 envs, the hidden layer never leaves the accumulator
    registers), tanh, a categorical sample from two PCG32 streams per env, and the result written as
    the int32 [n][2] (move, comm) pairs the step kernel consumes as they lie.  What
    ``TorchPolicyPartner(MLPPolicy(...))`` does in ~12 torch launches.  Two of them (ego + partner)
    share one launch (``FusedMLPPartner.launch``).  Weights are packed once (``refresh()`` after an
    optimiser step); fp16 operands, fp32 accumulation: logits within 2e-2 of the fp32 module's."""
    graph_safe = True
    _seats = 0          # partners built so far: the default stream seed differs from seat to seat

    def __init__(self, policy, sample=True, seed=None, device="cuda", keep_logits=False):
        # seed=None: a different default per partner built (two seats left at their defaults -- an
        # ego and a partner -- must not share per-env random streams: their samples would use the
        # same uniform draw every step); pass a seed for a reproducible stream
        if seed is None:
            seed = 0x5EED + 1000003 * FusedMLPPartner._seats
        FusedMLPPartner._seats += 1
        if not isinstance(policy, MLPPolicy):
            raise TypeError("FusedMLPPartner runs gym_comm_amd.vec_env.MLPPolicy; wrap any other module "
                            "in TorchPolicyPartner")
        if policy.w1.shape[0] != 64:
            raise ValueError("the fused kernel has 64 hidden units (got %d)" % policy.w1.shape[0])
        if not 1 <= policy.C <= 16:
            raise ValueError("the fused kernel samples at most 16 comm channels (got %d)" % policy.C)
        self.policy, self.sample, self.seed = policy, bool(sample), int(seed)
        self.device = torch.device(device)
        self.keep_logits = bool(keep_logits)
        self._L = _lib.load_policy()
        self.F = int(policy.w1.shape[1])
        self.C = int(policy.C)
        self._w = None
        self._rng = self.pairs = self.logits = None
        self.refresh()

    def refresh(self):
        """(Re)pack the module's current weights into MFMA fragment order and upload them."""
        L, pol = self._L, self.policy
        f32 = lambda t: np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))
        fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
        ks = L.oc_policy_ksteps(self.F)
        w1, wt, b1, w2, b2 = (f32(t) for t in (pol.w1, pol.wt, pol.b1, pol.w2, pol.b2))
        o1 = np.zeros((2, ks, 64, 8), np.uint16)
        o2 = np.zeros((4, 64, 8), np.uint16)
        ob = np.zeros((64, 16), np.float32)
        for rc, what in ((L.oc_policy_pack_w1(fp(w1), fp(wt.reshape(-1)), fp(b1.reshape(-1)), self.F,
                                              o1.ctypes.data_as(ctypes.c_void_p)), "oc_policy_pack_w1"),
                         (L.oc_policy_pack_w2(fp(w2), self.C, o2.ctypes.data_as(ctypes.c_void_p)), "oc_policy_pack_w2"),
                         (L.oc_policy_pack_b2(fp(b2.reshape(-1)), fp(w2), self.C, fp(ob)), "oc_policy_pack_b2")):
            if rc:
                raise _lib.OcError("%s failed: %s" % (what, L.oc_policy_last_error().decode()))
        dev = self.device
        new = tuple(torch.from_numpy(a.view(np.int16) if a.dtype == np.uint16 else a).to(dev) for a in (o1, o2, ob))
        if self._w is None:
            self._w = new
        else:                       # in place: a captured graph keeps the addresses
            for old, cur in zip(self._w, new):
                old.copy_(cur)

    def _buffers(self, n):
        if self.pairs is None or self.pairs.shape[0] != n:
            g = torch.Generator(device="cpu").manual_seed(self.seed)
            self._rng = torch.randint(0, 2 ** 31 - 1, (2, n), generator=g, dtype=torch.int64
                                      ).to(torch.int32).to(self.device)
            self.pairs = torch.zeros((n, 2), dtype=torch.int32, device=self.device)
            self.logits = (torch.zeros((4 + self.C, n), dtype=torch.float32, device=self.device)
                           if self.keep_logits else None)

    def _player(self, rows):
        n = rows.shape[1]
        if rows.shape[0] != self.F:
            raise ValueError("the policy was built for %d observation rows, the env has %d" % (self.F, rows.shape[0]))
        self._buffers(n)
        return _lib.PolicyPlayer(rows.data_ptr(), self._w[0].data_ptr(), self._w[1].data_ptr(),
                                 self._w[2].data_ptr(), self._rng.data_ptr() if self.sample else None,
                                 self.pairs.data_ptr(), self.logits.data_ptr() if self.logits is not None else None)

    @staticmethod
    def launch(players, rows, timestep):
        """One launch for one or two FusedMLPPartner, each on its own observation rows ([F][n]
        views of the env's obs tensor); their ``pairs`` tensors hold the result."""
        first = players[0]
        if any(pl.F != first.F or pl.C != first.C for pl in players):
            raise ValueError("players of one launch share F and C")
        arr = (_lib.PolicyPlayer * len(players))(*[pl._player(r) for pl, r in zip(players, rows)])
        ot = {torch.int32: 0, torch.int8: 1, torch.float32: 2}[rows[0].dtype]
        dev = rows[0].device.index
        n = rows[0].shape[1]
        with (contextlib.nullcontext() if torch.cuda.current_device() == dev else torch.cuda.device(dev)):
            rc = first._L.oc_policy_mlp(arr, len(players), timestep.data_ptr(), first.F, first.C, ot, n,
                                        torch._C._cuda_getCurrentRawStream(dev))
        if rc:
            raise _lib.OcError("oc_policy_mlp failed (%d): %s" % (rc, first._L.oc_policy_last_error().decode()))

    _PCG_MULT, _PCG_INC = 747796405, 2891336453          # csrc/oc_policy_device.h: pcg32
    _PCG_MULT_INV = pow(747796405, -1, 1 << 32)

    def rewind_rng(self):
        """Step both PCG32 streams of every env back by ONE draw (the generator's state update is
        an invertible affine map mod 2^32): the next evaluation repeats the last one's draw.  How a
        one-launch closed loop re-primes after a reset without leaving the sequence the two-launch
        form walks (``ClosedLoop.prime``)."""
        if self._rng is None or not self.sample:
            return
        s = self._rng.to(torch.int64) & 0xFFFFFFFF
        s = ((s - self._PCG_INC) * self._PCG_MULT_INV) & 0xFFFFFFFF
        self._rng.copy_(torch.where(s >= (1 << 31), s - (1 << 32), s).to(torch.int32))

    def step_policy(self, n):
        """(w1, w2, b2, rng) device addresses for oc_step_opts.policy (the step kernel evaluates
        this policy itself; its result lands in ``self.pairs``)."""
        self._buffers(n)
        return (self._w[0].data_ptr(), self._w[1].data_ptr(), self._w[2].data_ptr(),
                self._rng.data_ptr() if self.sample else None)

    def pairs_for(self, obs):
        """obs: an ``ObsView`` of the env (``rows`` [F][n], ``timestep``).  Returns int32 [n][2]."""
        FusedMLPPartner.launch([self], [obs.rows], obs.timestep)
        return self.pairs

    def act_into(self, obs, move_row, comm_row):       # the generic partner protocol (two more copies)
        pr = self.pairs_for(obs)
        move_row.copy_(pr[:, 0])
        comm_row.copy_(pr[:, 1])

    def __call__(self, obs):
        return self.pairs_for(obs)

    def get_state(self, n):
        """The player's mutable state: its random streams and its pairs (in a one-launch closed
        loop the pairs are the NEXT step's actions, i.e. state)."""
        self._buffers(n)
        return self._rng.clone(), self.pairs.clone()

    def set_state(self, st):
        self._rng.copy_(st[0])
        self.pairs.copy_(st[1])


class ClosedLoop:
    """ego policy -> partner policy -> fused step, ``steps`` times, as ONE hipGraph.

    ``ego``: something with ``act_into(obs, move_row, comm_row)`` (``TorchPolicyPartner``,
    ``RandomPartner``), or None -- then the caller writes the ego's (move, comm) into
    ``venv.ego_action_rows`` before every ``step()``.  ``enqueue()`` issues the launches of one
    step eagerly (what gets captured; also usable inside a caller's own capture)."""

    def __init__(self, venv, ego=None, graph=True, steps=1, one_launch=None):
        self.venv, self.ego, self.steps = venv, ego, int(steps)
        self.graph = None
        # ego and partner both FusedMLPPartner: let the step kernel evaluate them (one launch per
        # step instead of two) where the library can -- a specialised one, at most 4 comm channels --
        # and where it pays: in a split launch each of the four waves takes one (viewer, half)
        # pass (7.7 -> 6.7 us per step at 4 096 envs); a lone wave per 64 envs would run all four
        # passes back to back (9.6 -> 13.7 us at 32 768 envs), so larger batches keep two launches
        pt = venv.partner
        can = (isinstance(ego, FusedMLPPartner) and isinstance(pt, FusedMLPPartner) and pt.F == ego.F
               and pt.C == ego.C and ego.C <= 4 and venv._b.kernel_flavour == "spec"
               and venv._b.standard_wrapper_config)
        if one_launch and not can:
            raise ValueError("one_launch needs two FusedMLPPartner of one shape, C <= 4, a specialised library "
                             "and the wrapper's standard configuration")
        self.one_launch = (can and venv._b.launch_waves(general=True) == 4) if one_launch is None else bool(one_launch)
        self._primed = False
        self.prime()
        venv._loops.add(self)       # reset_tensors() re-primes every live loop (weak references)
        if graph:
            for pl in (ego, venv.partner):
                if pl is not None and not getattr(pl, "graph_safe", False):
                    raise ValueError("%r is not marked graph_safe (stateless, fixed-shape)" % (pl,))
            self.graph = venv._capture(self.enqueue, players=[ego], repeat=self.steps)

    def prime(self):
        """One-launch closed loop only: ``ego.pairs`` / ``partner.pairs`` hold the NEXT step's actions,
        computed by the previous step's launch from the observations it wrote.  After anything that
        rewrites the observations behind the loop's back -- ``reset_tensors()`` above all, which
        calls this for every live loop -- they must be recomputed from the CURRENT observations (one
        eager launch of the policy kernel, outside any graph), or the first step after the reset
        would play actions sampled for the pre-reset state.  Order for a loop built
        before the first reset: ``closed_loop(...)``, ``reset_tensors()`` (primes), ``step()``..."""
        if not self.one_launch:
            return
        v = self.venv
        if self._primed:
            # the pending pairs being replaced consumed one draw of each stream: take it back, so the
            # re-evaluation on the new observations uses the draw the two-launch form would use next
            self.ego.rewind_rng()
            v.partner.rewind_rng()
        o0, o1 = v._obs_tensors(0), v._obs_tensors(1)
        FusedMLPPartner.launch([self.ego, v.partner], [o0.rows, o1.rows], o0.timestep)
        self._primed = True

    def enqueue(self):
        v = self.venv
        ego, pt = self.ego, v.partner
        if isinstance(ego, FusedMLPPartner):
            if isinstance(pt, FusedMLPPartner) and pt.F == ego.F and pt.C == ego.C:
                o0, o1 = v._obs_tensors(0), v._obs_tensors(1)
                if self.one_launch:
                    # the step kernel evaluates both policies itself, behind the step, on the
                    # observations it has just written, and leaves the NEXT step's pairs in
                    # ego.pairs / pt.pairs (oc_step_opts.policy): ONE launch per step.  (The pairs
                    # of the very first step came from one launch of the policy kernel: __init__.)
                    n = v.num_envs
                    v._b.multi_step(v._act, ego_pairs=ego.pairs, alt_pairs=pt.pairs,
                                    policy=(ego.step_policy(n), pt.step_policy(n)))
                else:
                    # both policies in ONE launch, their pairs consumed by the step as they lie
                    FusedMLPPartner.launch([ego, pt], [o0.rows, o1.rows], o0.timestep)
                    v._b.multi_step(v._act, ego_pairs=ego.pairs, alt_pairs=pt.pairs)
            else:
                v._partner_and_step(ego.pairs_for(v._obs_tensors(0)))
        else:
            if ego is not None:
                ego.act_into(v._obs_tensors(0), v._act[0], v._act[1])
            v._partner_and_step(None)
        # MultiAgentEnv._update_players (multiagentenv.py:186-195): every seated agent hears the
        # step's reward and done -- inside the captured graph when the player's update() is device work
        for pl in (ego, pt):
            if pl is not None and hasattr(pl, "update"):
                pl.update(v._b.shaped_reward, v._b.done)
        v._version += 1

    def step(self):
        """Returns (ego obs, shaped reward f64 [n], done int32 [n]) -- views of the tensors the
        next step overwrites."""
        v = self.venv
        if self.graph is not None:
            self.graph.replay()
            v._version += self.steps
        else:
            for _ in range(self.steps):
                self.enqueue()
        return v._obs_tensors(0), v._b.shaped_reward, v._b.done


class OvercookedVecEnv(_VecEnvBase):
    def __init__(self, arglist, num_envs, partner=None, device="cuda", terminal_obs=False,
                 ego_agent_idx=0, subtask_order=None, level_dir=None, seed=0,
                 track_episode_stats=True, use_graph=False, reuse_host_buffers=False, **batched_kw):
        if _arg(arglist, "num_agents") != 2:
            raise ValueError("the gym_comm wrapper drives exactly 2 agents")
        self.arglist = arglist
        self.terminal_obs = bool(terminal_obs)
        self.track_episode_stats = bool(track_episode_stats)
        self._b = BatchedOvercooked(
            _arg(arglist, "level"), num_agents=2, num_envs=num_envs,
            max_num_timesteps=_arg(arglist, "max_num_timesteps", 100),
            max_num_subtasks=_arg(arglist, "max_num_subtasks", 14),
            ego_config=dict(_arg(arglist, "ego_config", {}) or {}),
            partner_config=dict(_arg(arglist, "partner_config", {}) or {}),
            num_communication=_arg(arglist, "num_communication", 10),
            communication_on=_arg(arglist, "communication_on", False),
            ego_led=_arg(arglist, "ego_led", False), fow_radius=_arg(arglist, "fow_radius", 2),
            ego_agent_idx=ego_agent_idx, device=device, subtask_order=subtask_order,
            level_dir=level_dir, auto_reset=True, seed=seed,
            episode_stats=self.track_episode_stats, play=bool(_arg(arglist, "play", False)), **batched_kw)
        lv = self._b.level
        obs_space, act_space = make_spaces(lv.width, lv.height, lv.num_subtasks, self._b.C)
        super().__init__(num_envs, obs_space, act_space)
        self._fast = self._graph = None
        self._loops = weakref.WeakSet()                  # live ClosedLoop objects (re-primed by reset_tensors)
        self.partner = partner if partner is not None else RandomPartner(self._b.C, seed, self._b.device)
        self._act = torch.zeros((4, num_envs), dtype=torch.int32, device=self._b.device)
        self._pending = None
        # the kernel overwrites the same observation rows every step, so the 11-key views of
        # both viewers are built once (22 slice + transpose objects cost ~60 us per step otherwise)
        self._views = [None, None]
        self._infos = [{} for _ in range(num_envs)]      # reused; only finished envs are touched
        self._dirty = []
        self._host_plan_cache = None
        self._last_terminal = None
        self._version = 0                                # bumped by every step / reset
        self._env_views = {}                             # env index -> (OvercookedEnvironment view, version)
        self._env_attrs = {}                             # (env index, name) -> value set by set_attr(indices=...)
        self._use_graph = bool(use_graph) and not self.terminal_obs
        self._ego_pairs = None
        self._act_pinned = None
        self._host_stats = None
        # numpy API: False = every step returns fresh arrays (a host copy out of the pinned
        # buffer); True = the arrays are views of two alternating pinned buffers, valid until
        # the step after next (what a rollout collector that copies them on arrival needs)
        self._reuse_host = bool(reuse_host_buffers)
        self._host_flip = 0

    @property
    def partner(self):
        return self._partner

    @partner.setter
    def partner(self, pt):          # a new partner invalidates the prepared launch plan / captured graph
        self._partner = pt
        self._fast = self._graph = None

    # the running episode statistics live in the batch (kept by the kernel)
    @property
    def episode_returns(self):
        return self._b.ep_return

    @property
    def episode_lengths(self):
        return self._b.ep_length

    @property
    def ego_action_rows(self):
        """int32 [2][n]: the ego's (move, comm) rows of the kernel's action tensor."""
        return self._act[0:2]

    # -- tensors ---------------------------------------------------------------------
    def _obs_tensors(self, viewer):
        """The 11 keys as [n, k] device tensors (views of the kernel's [k][n] rows; the same
        tensor objects every step -- the next step overwrites their contents)."""
        if self._views[viewer] is None:
            d = self._b.obs_dict(viewer)
            v = ObsView((k, t.T) for k, t in d.items())
            v.rows, v.timestep = self._b.obs[viewer], self._b.timestep
            self._views[viewer] = v
        return self._views[viewer]

    def reset_tensors(self):
        b = self._b
        b.reset()
        b.done.zero_()
        if self.track_episode_stats:
            b.ep_return.zero_()
            b.ep_length.zero_()
        b.observe()
        self._version += 1
        # players with per-episode state start over (a recurrent partner's states and episode_start),
        # and every live one-launch closed loop recomputes its pending actions from the new observations
        seen = set()
        for pl in [self.partner] + [lp.ego for lp in self._loops]:
            if pl is not None and id(pl) not in seen and hasattr(pl, "reset"):
                seen.add(id(pl))
                pl.reset()
        for lp in self._loops:
            lp.prime()
        return self._obs_tensors(0)

    def closed_loop(self, ego=None, graph=True, steps=1, one_launch=None):
        """ego policy -> partner policy -> fused step as one hipGraph (see ``ClosedLoop``)."""
        return ClosedLoop(self, ego, graph=graph, steps=steps, one_launch=one_launch)

    def _fast_plan(self):
        """Addresses for the single-launch path of step_tensors, or False when it does not apply
        (a partner with launches or an update() of its own, terminal_obs, a captured graph)."""
        pt = self.partner
        if self._use_graph or self.terminal_obs or not getattr(pt, "in_kernel", False) or hasattr(pt, "update"):
            return False
        rng = pt.rng_state(self.num_envs).data_ptr()
        return {"act": self._act.data_ptr(), "played": self._act[2:4].data_ptr(),
                "rng": rng, "obs": self._obs_tensors(0),
                "pair_shape": torch.Size((self.num_envs, 2)), "dev": self._b._dev_index,
                # the prepared call (include/oc_hip.h: oc_multi_step_prepare): everything but the ego's pairs fixed
                "launch": self._b.prepare_multi_step(self._act.data_ptr(), alt_rng_ptr=rng,
                                                     alt_played_ptr=self._act[2:4].data_ptr(), auto_reset=True)}

    def _capture(self, fn, players=(), repeat=1):
        """`repeat` calls of fn() as one hipGraph.  One eager warm-up call first (module loads,
        rocBLAS workspaces) that must not count: the env's tensors, the action rows, the metrics
        and every random stream (the env's, the players', torch's) are put back afterwards."""
        b = self._b
        if not getattr(self.partner, "graph_safe", False):
            raise ValueError("%r is not marked graph_safe (stateless, fixed-shape)" % (self.partner,))
        players = [pl for pl in list(players) + [self.partner] if pl is not None]
        keep, act = b._arena.clone(), self._act.clone()
        rng = None if b.rng is None else b.rng.clone()
        met = None if b.metrics is None else b.metrics.clone()
        torch_rng = torch.cuda.get_rng_state(b.device)
        saved = [pl.get_state(self.num_envs) if hasattr(pl, "get_state") else None for pl in players]
        ver = self._version
        fn()
        torch.cuda.current_stream(b.device).synchronize()
        b._arena.copy_(keep)
        self._act.copy_(act)
        if rng is not None:
            b.rng.copy_(rng)
        if met is not None:
            b.metrics.copy_(met)
        torch.cuda.set_rng_state(torch_rng, b.device)
        for pl, st in zip(players, saved):
            if hasattr(pl, "set_state"):
                pl.set_state(st)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(repeat):
                fn()
        self._version = ver
        return graph

    def _partner_and_step(self, ego_pairs, auto_reset=None):
        """The partner's move and the fused step.  ego_pairs: the ego's int32 [n][2] (move, comm)
        pairs, or None = rows 0, 1 of ``self._act``.  Launches: none for an in-kernel partner
        (``RandomPartner``), the partner's own otherwise; then ONE for the step."""
        b, pt = self._b, self.partner
        if getattr(pt, "in_kernel", False):
            b.multi_step(self._act, auto_reset=auto_reset, ego_pairs=ego_pairs,
                         alt_rng=pt.rng_state(self.num_envs), alt_played=self._act[2:4])
            return
        partner_obs = self._obs_tensors(1)
        alt_pairs = None
        if isinstance(pt, FusedMLPPartner):       # one launch; its pairs are consumed as they lie
            alt_pairs = pt.pairs_for(partner_obs)
        elif hasattr(pt, "act_into"):     # rows of the action tensor: ego move, ego comm, alt move, alt comm
            pt.act_into(partner_obs, self._act[2], self._act[3])
        else:
            pa = torch.as_tensor(pt(partner_obs), device=b.device)
            if pa.dtype == torch.int32 and pa.is_contiguous():
                alt_pairs = pa                               # consumed as it lies
            else:
                self._act[2:4].copy_(pa.T)
        b.multi_step(self._act, auto_reset=auto_reset, ego_pairs=ego_pairs, alt_pairs=alt_pairs)

    def step_tensors(self, ego_actions=None):
        """ego_actions: int tensor [n, 2] on the device (None: already written into
        ``ego_action_rows``).  A contiguous int32 or int64 tensor is handed to the kernel as it
        lies (no copy, no extra launch).  Returns (ego obs dict of [n, k] tensors, shaped reward f64 [n],
        done int32 [n]) -- views that the next step overwrites."""
        b = self._b
        f = self._fast
        if f is None:
            f = self._fast = self._fast_plan()
        if f:   # one launch, no partner launch, nothing to capture: the per-call cost is one ctypes call
            ego_ptr = None
            if ego_actions is not None:
                ea = ego_actions
                if not (isinstance(ea, torch.Tensor) and (ea.dtype is torch.int32 or ea.dtype is torch.int64)
                        and ea.is_cuda and ea.is_contiguous() and ea.shape == f["pair_shape"]
                        and ea.get_device() == f["dev"]):
                    if self._ego_pairs is None:
                        self._ego_pairs = torch.zeros(f["pair_shape"], dtype=torch.int32, device=b.device)
                    self._ego_pairs.copy_(torch.as_tensor(ego_actions, device=b.device))
                    ea = self._ego_pairs
                ego_ptr = ea.data_ptr()
                i64 = ea.dtype is torch.int64
            f["launch"](ego_ptr, ego_ptr is not None and i64)
            self._version += 1
            self._last_terminal = None
            return f["obs"], b.shaped_reward, b.done
        ego_pairs = None
        if ego_actions is not None:
            ea = torch.as_tensor(ego_actions, device=b.device)
            if ea.dtype == torch.int32 and ea.is_contiguous() and not self._use_graph:
                ego_pairs = ea                 # (the slow path keeps to int32: alt_pairs may join it)
            else:                       # other dtypes / layouts, or the captured graph's fixed input
                if self._ego_pairs is None:
                    self._ego_pairs = torch.zeros((self.num_envs, 2), dtype=torch.int32, device=b.device)
                self._ego_pairs.copy_(ea)
                ego_pairs = self._ego_pairs
        term = None
        if self._use_graph:
            if self._ego_pairs is None:
                self._ego_pairs = torch.zeros((self.num_envs, 2), dtype=torch.int32, device=b.device)
            if ego_pairs is None:       # ego rows written by the caller: bring them into the graph's input
                self._ego_pairs.copy_(self._act[0:2].T)
            if self._graph is None:     # partner -> fused step (-> the partner's update), captured once
                def one_step():
                    self._partner_and_step(self._ego_pairs)
                    if hasattr(self.partner, "update"):
                        self.partner.update(b.shaped_reward, b.done)
                self._graph = self._capture(one_step)
            self._graph.replay()
            self._version += 1
            self._last_terminal = None
            return self._obs_tensors(0), b.shaped_reward, b.done
        else:
            if self.terminal_obs:
                self._partner_and_step(ego_pairs, auto_reset=False)
                term = {k: v.clone() for k, v in self._obs_tensors(0).items()}
                b.reset(b.done)                                   # mask = done flags
                b.observe()
            else:
                self._partner_and_step(ego_pairs)
            self._version += 1
        rew, done = b.shaped_reward, b.done
        self._last_terminal = term
        if hasattr(self.partner, "update"):
            self.partner.update(rew, done)
        return self._obs_tensors(0), rew, done

    # -- SB3 VecEnv API (numpy) --------------------------------------------------------
    @staticmethod
    def _to_numpy(obs):
        return {k: v.cpu().numpy().astype(SPACE_DTYPE[k]) for k, v in obs.items()}

    def _host_plan(self):
        """The pack plan of the numpy boundary (include/oc_hostio.h): which observation row goes
        to which column of which dtype block (int64 / float32 / int8 as the declared spaces,
        overcooked_env.py:41-85), the device and pinned host buffers, and the numpy views' offsets."""
        b, n = self._b, self.num_envs
        L = _lib.load_hostio()
        block_of = {np.dtype(np.int64): 0, np.dtype(np.float32): 1, np.dtype(np.int8): 2}
        width, plan, cols = [0, 0, 0], np.zeros(b.F, np.int32), {}
        for k, (lo, hi) in b._layout.items():
            blk = block_of[np.dtype(SPACE_DTYPE[k])]
            cols[k] = (blk, width[blk], width[blk] + hi - lo)
            for r in range(lo, hi):
                plan[r] = (blk << 16) | width[blk]
                width[blk] += 1
        stats = b.ep_return is not None
        total = int(L.oc_pack_host_bytes(width[0], width[1], width[2], 1, int(stats), 1, int(stats), n))
        off, o = {}, 0
        for name, nbytes in (("b64", n * width[0] * 8), ("ret", n * 8 if stats else 0), ("b32", n * width[1] * 4),
                             ("ts", n * 4), ("rew", n * 4), ("done", n * 4), ("len", n * 4 if stats else 0),
                             ("b8", n * width[2])):
            off[name] = (o, o + nbytes)
            o += nbytes
        assert o == total
        return {"L": L, "plan": torch.from_numpy(plan).to(b.device), "width": width, "cols": cols, "stats": stats,
                "dev": torch.empty(max(total, 1), dtype=torch.uint8, device=b.device),
                "pinned": [torch.empty(max(total, 1), dtype=torch.uint8, pin_memory=True)
                           for _ in range(2 if self._reuse_host else 1)], "off": off,
                "ot": {torch.int32: 0, torch.int8: 1, torch.float32: 2}[b.obs.dtype]}

    def _host_step(self, rew=None, done=None):
        """Everything the numpy API returns for one step: ONE launch (``oc_pack_host``,
        include/oc_hostio.h: the ego viewer's [F][n] rows gathered by target dtype, transposed
        and converted; float32 timestep and reward, done flags, episode return / length) into one
        device buffer, ONE device->host copy into pinned memory, and a host copy out of it (the
        caller may keep the arrays); the per-key arrays are column views of that copy."""
        b, n = self._b, self.num_envs
        hp = self._host_plan_cache
        if hp is None:
            hp = self._host_plan_cache = self._host_plan()
        dp = lambda t: None if t is None else t.data_ptr()
        with b._on_device():
            rc = hp["L"].oc_pack_host(b.obs[0].data_ptr(), hp["ot"], b.F, hp["plan"].data_ptr(), hp["width"][0],
                                      hp["width"][1], hp["width"][2], b.timestep.data_ptr(),
                                      b.shaped_reward.data_ptr(), dp(b.ep_return), b.done.data_ptr(),
                                      dp(b.ep_length), hp["dev"].data_ptr(), n, b._raw_stream())
        if rc:
            raise _lib.OcError("oc_pack_host failed (%d): %s" % (rc, hp["L"].oc_hostio_last_error().decode()))
        self._host_flip ^= 1
        pinned = hp["pinned"][self._host_flip if self._reuse_host else 0]
        pinned.copy_(hp["dev"], non_blocking=True)
        torch.cuda.current_stream(b.device).synchronize()
        host = pinned.numpy() if self._reuse_host else pinned.numpy().copy()
        view = lambda name, dt, shape: host[hp["off"][name][0]:hp["off"][name][1]].view(dt).reshape(shape)
        blocks = (view("b64", np.int64, (n, hp["width"][0])), view("b32", np.float32, (n, hp["width"][1])),
                  view("b8", np.int8, (n, hp["width"][2])))
        obs = {k: blocks[blk][:, lo:hi] for k, (blk, lo, hi) in hp["cols"].items()}
        obs["timestep"] = view("ts", np.float32, (n, 1))
        self._host_stats = ((view("ret", np.float64, (n,)), view("len", np.int32, (n,))) if hp["stats"] else None)
        return (obs, view("rew", np.float32, (n,)) if rew is not None else None,
                view("done", np.int32, (n,)) if done is not None else None)

    def _ego_obs_numpy(self):
        return self._host_step()[0]

    def reset(self):
        self.reset_tensors()
        return self._ego_obs_numpy()

    def step_async(self, actions):
        self._pending = np.asarray(actions)

    def step_wait(self):
        # host actions -> pinned staging -> the device pairs the kernel consumes as they lie
        if self._act_pinned is None:
            self._act_pinned = torch.empty((self.num_envs, 2), dtype=torch.int32, pin_memory=True)
            if self._ego_pairs is None:
                self._ego_pairs = torch.zeros((self.num_envs, 2), dtype=torch.int32, device=self._b.device)
        np.copyto(self._act_pinned.numpy(), self._pending.reshape(self.num_envs, 2), casting="unsafe")
        self._ego_pairs.copy_(self._act_pinned, non_blocking=True)
        _, rew, done = self.step_tensors(self._ego_pairs)
        obs_np, rew_np, done_i32 = self._host_step(rew, done)
        done_np = done_i32.astype(bool)
        infos = self._infos                       # the same list every step (as DummyVecEnv's buf_infos)
        for i in self._dirty:
            infos[i] = {}
        idx = np.nonzero(done_np)[0]
        self._dirty = idx.tolist()
        if len(idx):
            term = self._to_numpy(self._last_terminal) if self._last_terminal is not None else None
            if self.track_episode_stats:
                # after a step that returned done the kernel's rows hold the finished episode's
                # totals; they crossed PCIe with the observation (oc_pack_host)
                ret, ln = self._host_stats
            for i in idx:
                if self.track_episode_stats:
                    infos[i]["episode"] = {"r": float(ret[i]), "l": int(ln[i])}     # Monitor-style
                if term is not None:
                    infos[i]["terminal_observation"] = {k: v[i] for k, v in term.items()}
        return obs_np, rew_np, done_np, list(infos)   # shallow copy: holders keep their dicts

    def close(self):
        pass

    def seed(self, seed=None):
        return [seed] * self.num_envs

    # -- per-env access (VecEnv.get_attr / env_method) -------------------
    # ParallelEpisodeRecorder reads env.get_attr('t' | 'world' | 'sim_agents' | 'arglist', i)[0]
    # and keeps the world / sim_agents objects between steps (episode_recorder.py:48-85), so the
    # view of env i is ONE persistent object whose mirror is refreshed from the device -- a
    # column of A+M+2 state words -- when it is read after the batch has moved on.
    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        idx = [int(i) for i in np.atleast_1d(indices)]
        for i in idx:
            if not 0 <= i < self.num_envs:
                raise IndexError("env index %d outside 0..%d" % (i, self.num_envs - 1))
        return idx

    def base_env(self, i):
        """The ``OvercookedEnvironment``-shaped view of env i (read-only; ``t``, ``world``,
        ``sim_agents``, ``completed_subtasks``, ``rep``, ``str()``, ``display()``,
        ``render_frame()``)."""
        i = self._indices([i])[0]
        ent = self._env_views.get(i)
        if ent is None:
            ent = [OvercookedEnvironment(self.arglist, _batch=self._b, _index=i, _view=True), -1]
            self._env_views[i] = ent
        if ent[1] != self._version:
            ent[0].mark_dirty()
            ent[1] = self._version
        return ent[0]

    _PER_ENV_ATTRS = ("base_env", "t", "world", "sim_agents", "completed_subtasks", "goal_objects_count",
                      "rep", "all_subtasks", "recipes")

    def get_attr(self, attr_name, indices=None):
        idx = self._indices(indices)
        if attr_name == "base_env":
            return [self.base_env(i) for i in idx]
        if attr_name in self._PER_ENV_ATTRS:
            return [getattr(self.base_env(i), attr_name) for i in idx]
        # an attribute set for some envs only (set_attr with indices) lives on those envs' views
        return [self._env_attrs[(i, attr_name)] if (i, attr_name) in self._env_attrs else getattr(self, attr_name)
                for i in idx]

    def set_attr(self, attr_name, value, indices=None):
        """SB3's ``VecEnv.set_attr``: the attribute of the envs in ``indices`` (None = all).  The
        device state of an env is not settable this way (the per-env views are read-only mirrors):
        state-derived names raise.  ``indices=None`` sets a batch-level attribute of the VecEnv;
        a subset is remembered per env and returned by ``get_attr`` for those envs."""
        if attr_name in self._PER_ENV_ATTRS:
            raise AttributeError("%r mirrors device state and is read-only" % attr_name)
        if indices is None:
            setattr(self, attr_name, value)
            for key in [k for k in self._env_attrs if k[1] == attr_name]:
                del self._env_attrs[key]
            return
        for i in self._indices(indices):
            self._env_attrs[(i, attr_name)] = value

    _PER_ENV_METHODS = ("render_frame", "render_rgb", "display", "get_agent_names", "__str__")

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        idx = self._indices(indices)
        if method_name in self._PER_ENV_METHODS:
            return [getattr(self.base_env(i), method_name)(*method_args, **method_kwargs) for i in idx]
        return [getattr(self, method_name)(*method_args, **method_kwargs)] * len(idx)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def metrics(self):
        return self._b.read_metrics()


class BatchEpisodeRecorder:
    """``ParallelEpisodeRecorder`` (episode_recorder.py:48-85) for the batched env: wraps an
    ``OvercookedVecEnv``, counts episodes per env and, for the envs in ``indices`` whose episode
    number is a multiple of ``record_interval``, collects one frame per step --
    ``render_frame()`` of that env's view, i.e. the composition of the reference's pygame
    renderer from device state (the ASCII ``str(env)`` without sprites).  A finished episode's
    frames are handed to ``on_episode(env_index, episode, frames)`` (default: kept in
    ``self.episodes``; ``save_dir`` writes them as .npz).  wandb / imageio / pygame are not
    required."""

    def __init__(self, venv, record_interval=-1, indices=(0,), sprite_dir=None, ascii_only=False,
                 on_episode=None, save_dir=None):
        self.env = venv
        self.num_envs = venv.num_envs
        self.record_interval = int(record_interval)
        self.indices = [int(i) for i in indices]
        self.episode_counts = {i: 0 for i in self.indices}
        self.sprite_dir, self.ascii_only = sprite_dir, bool(ascii_only)
        self.on_episode, self.save_dir = on_episode, save_dir
        self._frames = {i: [] for i in self.indices}
        self.episodes = []

    def __getattr__(self, name):                   # gym.Wrapper-style pass-through
        return getattr(self.env, name)

    def _recording(self, i):
        return self.record_interval > 0 and self.episode_counts[i] % self.record_interval == 0

    def _frame(self, i):
        v = self.env.base_env(i)
        return str(v) if self.ascii_only else v.render_frame(self.sprite_dir)

    def reset(self):
        out = self.env.reset()
        for i in self.indices:
            self.episode_counts[i] += 1
            self._frames[i] = [self._frame(i)] if self._recording(i) else []
        return out

    def step(self, actions):
        obs, rew, dones, infos = self.env.step(actions)
        for i in self.indices:
            if dones[i]:
                # the env was auto-reset inside the kernel: its view shows the first frame of the
                # next episode, which starts that episode's recording
                if self._frames[i]:
                    ep = (i, self.episode_counts[i], self._frames[i])
                    if self.on_episode is not None:
                        self.on_episode(*ep)
                    else:
                        self.episodes.append(ep)
                    if self.save_dir and not self.ascii_only:
                        os.makedirs(self.save_dir, exist_ok=True)
                        np.savez_compressed(os.path.join(self.save_dir, "env%d_ep%d.npz" % (i, ep[1])),
                                            frames=np.stack(ep[2]))
                self.episode_counts[i] += 1
                self._frames[i] = []
            if self._recording(i):
                self._frames[i].append(self._frame(i))
        return obs, rew, dones, infos

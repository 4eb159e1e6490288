"""Vectorised trainer boundary: N Overcooked envs behind the
Stable-Baselines3 ``VecEnv`` API, ego-perspective, with the partner policy evaluated on
the whole batch.

The reference trains on ONE env: ``RecurrentPPO`` wraps ``OvercookedMultiEnv`` in
``Monitor + DummyVecEnv(1)`` and pantheonrl asks the partner agent for one action per
step (pantheonrl/common/multiagentenv.py:149-215, trainer.py:87-121).  Here the same
ego/partner protocol runs on tensors:

    obs_ego = venv.reset()
    obs_ego, rewards, dones, infos = venv.step(ego_actions)      # ego_actions [n, 2] = (move, comm)

Every ``step`` asks ``partner(obs_partner)`` for the partner's [n, 2] actions on the
observations it saw after the previous step (SimultaneousEnv semantics,
multiagentenv.py:395-404), calls the fused ``oc_multi_step`` kernel once, and reports
``partner.update(rewards, dones)`` if the partner has that method.  Finished envs are
auto-reset inside the kernel, so -- as with any SB3 VecEnv -- the observation returned for
a done env is the first observation of its next episode and ``rewards``/``dones`` belong to
the finished step.  ``infos[i]["terminal_observation"]`` is only filled with
``terminal_obs=True`` (three launches per step instead of one).

Subclasses ``stable_baselines3.common.vec_env.VecEnv`` when SB3 is importable; otherwise a
structural stand-in with the same methods.  ``step_tensors`` is the zero-copy variant for
policies that live on the GPU.
"""
import numpy as np
import torch

from .batched import BatchedOvercooked
from .envs import _arg, make_spaces

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


class RandomPartner:
    """Uniform random partner (move 0..3, comm 0..C-1), generated on the device."""

    def __init__(self, num_comm, seed=0, device="cuda"):
        self.C = num_comm
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self.device = device

    def __call__(self, obs):
        n = next(iter(obs.values())).shape[0]
        mv = torch.randint(0, 4, (n,), generator=self.gen, device=self.device, dtype=torch.int32)
        cm = torch.randint(0, self.C, (n,), generator=self.gen, device=self.device, dtype=torch.int32)
        return torch.stack([mv, cm], dim=1)

    def act_into(self, obs, move_row, comm_row):
        """Same draw, written straight into the kernel's action rows (int32 [n] each): two
        launches instead of four (no stack, no transpose-copy)."""
        move_row.random_(0, 4, generator=self.gen)
        comm_row.random_(0, self.C, generator=self.gen)


class OvercookedVecEnv(_VecEnvBase):
    def __init__(self, arglist, num_envs, partner=None, device="cuda", terminal_obs=False,
                 ego_agent_idx=0, subtask_order=None, level_dir=None, seed=0,
                 track_episode_stats=True, **batched_kw):
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
            level_dir=level_dir, auto_reset=True, seed=seed, **batched_kw)
        lv = self._b.level
        obs_space, act_space = make_spaces(lv.width, lv.height, lv.num_subtasks, self._b.C)
        super().__init__(num_envs, obs_space, act_space)
        self.partner = partner if partner is not None else RandomPartner(self._b.C, seed, self._b.device)
        self._act = torch.zeros((4, num_envs), dtype=torch.int32, device=self._b.device)
        self._pending = None
        self._partner_obs = None
        # the kernel overwrites the same observation rows every step, so the 11-key views of
        # both viewers are built once (22 slice + transpose objects cost ~60 us per step otherwise)
        self._views = [None, None]
        self._infos = [{} for _ in range(num_envs)]      # reused; only finished envs are touched
        self._dirty = []
        self._dtype_groups = None
        self.episode_returns = torch.zeros(num_envs, dtype=torch.float64, device=self._b.device)
        self.episode_lengths = torch.zeros(num_envs, dtype=torch.int64, device=self._b.device)

    # -- tensors ---------------------------------------------------------------------
    def _obs_tensors(self, viewer):
        """The 11 keys as [n, k] device tensors (views of the kernel's [k][n] rows; the same
        tensor objects every step -- the next step overwrites their contents)."""
        if self._views[viewer] is None:
            d = self._b.obs_dict(viewer)
            self._views[viewer] = {k: v.T for k, v in d.items()}
        return self._views[viewer]

    def reset_tensors(self):
        self._b.reset()
        self._b.observe()
        self.episode_returns.zero_()
        self.episode_lengths.zero_()
        self._partner_obs = self._obs_tensors(1)
        return self._obs_tensors(0)

    def step_tensors(self, ego_actions):
        """ego_actions: int tensor [n, 2] on the device.  Returns (ego obs dict of [n, k]
        tensors, shaped reward f64 [n], done int32 [n]) -- views that the next step
        overwrites."""
        b = self._b
        # rows of the action tensor: ego move, ego comm, alt move, alt comm
        if hasattr(self.partner, "act_into"):
            self.partner.act_into(self._partner_obs, self._act[2], self._act[3])
        else:
            pa = torch.as_tensor(self.partner(self._partner_obs), device=b.device)
            self._act[2:4].copy_(pa.T)
        self._act[0:2].copy_(torch.as_tensor(ego_actions, device=b.device).T)
        term = None
        if self.terminal_obs:
            b.multi_step(self._act, auto_reset=False)
            term = {k: v.clone() for k, v in self._obs_tensors(0).items()}
            b.reset(b.done)                                   # mask = done flags
            b.observe()
        else:
            b.multi_step(self._act)
        rew, done = b.shaped_reward, b.done
        if self.track_episode_stats:
            self.episode_returns += rew
            self.episode_lengths += 1
        self._last_terminal = term
        if hasattr(self.partner, "update"):
            self.partner.update(rew, done)
        self._partner_obs = self._obs_tensors(1)
        return self._obs_tensors(0), rew, done

    # -- SB3 VecEnv API (numpy) --------------------------------------------------------
    @staticmethod
    def _to_numpy(obs):
        return {k: v.cpu().numpy().astype(SPACE_DTYPE[k]) for k, v in obs.items()}

    def _host_step(self, rew=None, done=None):
        """Everything the numpy API returns for one step, in ONE device->host copy.  The ego
        viewer's [F][n] rows are gathered by target dtype (int64 / float32 / int8 as the
        declared spaces, overcooked_env.py:41-85), transposed and cast on the GPU; those
        blocks, the float32 timestep, the float32 reward and the done flags are concatenated
        as bytes (widest elements first, so every block stays aligned) and cross PCIe together;
        the per-key arrays are column views of the host copy (no host-side casts)."""
        b = self._b
        if self._dtype_groups is None:
            groups = {}
            for k, (lo, hi) in b._layout.items():
                groups.setdefault(SPACE_DTYPE[k], []).append((k, lo, hi))
            self._dtype_groups = []
            for dt in sorted(groups, key=lambda d: -np.dtype(d).itemsize):
                keys = groups[dt]
                rows = torch.tensor([r for _, lo, hi in keys for r in range(lo, hi)], device=b.device)
                cols, c = {}, 0
                for k, lo, hi in keys:
                    cols[k] = (c, c + hi - lo)
                    c += hi - lo
                self._dtype_groups.append((np.dtype(dt), getattr(torch, np.dtype(dt).name), rows, cols, c))
        n = self.num_envs
        parts, plan = [], []            # device byte blocks; (name, numpy dtype, shape, cols)
        for ndt, tdt, rows, cols, width in self._dtype_groups:
            parts.append(b.obs[0].index_select(0, rows).T.to(tdt).contiguous())
            plan.append(("obs", ndt, (n, width), cols))
        f32 = [("timestep", b.timestep)] + ([("rew", rew)] if rew is not None else [])
        for name, t in f32:
            parts.append(t.to(torch.float32))
            plan.append((name, np.dtype(np.float32), (n,), None))
        if done is not None:
            parts.append(done)
            plan.append(("done", np.dtype(np.int32), (n,), None))
        # order by element size, widest first (stable): offsets stay multiples of the element size
        order = sorted(range(len(parts)), key=lambda i: -plan[i][1].itemsize)
        # small batches are latency-bound (one copy instead of six: 268 -> 235 us at n = 4096);
        # large ones are bandwidth-bound and the byte-wise concatenation only adds a pass
        # (1.6 -> 2.4 ms at n = 131072), so there every block crosses on its own
        single = n <= 16384
        if single:
            host = torch.cat([parts[i].reshape(-1).view(torch.uint8) for i in order]).cpu().numpy()
        out, obs, off = {}, {}, 0
        for i in order:
            name, ndt, shape, cols = plan[i]
            nb = int(np.prod(shape)) * ndt.itemsize
            if single:
                arr = host[off:off + nb].view(ndt).reshape(shape)
            else:
                arr = parts[i].cpu().numpy().reshape(shape)
            off += nb
            if name == "obs":
                for k, (lo, hi) in cols.items():
                    obs[k] = arr[:, lo:hi]
            elif name == "timestep":
                obs["timestep"] = arr.reshape(-1, 1)
            else:
                out[name] = arr
        return obs, out.get("rew"), out.get("done")

    def _ego_obs_numpy(self):
        return self._host_step()[0]

    def reset(self):
        self.reset_tensors()
        return self._ego_obs_numpy()

    def step_async(self, actions):
        self._pending = np.asarray(actions)

    def step_wait(self):
        _, rew, done = self.step_tensors(torch.from_numpy(self._pending.astype(np.int32)))
        obs_np, rew_np, done_i32 = self._host_step(rew, done)
        done_np = done_i32.astype(bool)
        infos = self._infos                       # the same list every step (as DummyVecEnv's buf_infos)
        for i in self._dirty:
            infos[i] = {}
        idx = np.nonzero(done_np)[0]
        self._dirty = idx.tolist()
        if len(idx):
            ret = self.episode_returns.cpu().numpy()
            ln = self.episode_lengths.cpu().numpy()
            term = self._to_numpy(self._last_terminal) if self._last_terminal is not None else None
            for i in idx:
                infos[i]["episode"] = {"r": float(ret[i]), "l": int(ln[i])}     # Monitor-style
                if term is not None:
                    infos[i]["terminal_observation"] = {k: v[i] for k, v in term.items()}
            m = done.bool()
            self.episode_returns[m] = 0
            self.episode_lengths[m] = 0
        return obs_np, rew_np, done_np, list(infos)   # shallow copy: holders keep their dicts

    def close(self):
        pass

    def seed(self, seed=None):
        return [seed] * self.num_envs

    def get_attr(self, attr_name, indices=None):
        n = self.num_envs if indices is None else len(np.atleast_1d(indices))
        return [getattr(self, attr_name)] * n

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        n = self.num_envs if indices is None else len(np.atleast_1d(indices))
        return [getattr(self, method_name)(*method_args, **method_kwargs)] * n

    def env_is_wrapped(self, wrapper_class, indices=None):
        n = self.num_envs if indices is None else len(np.atleast_1d(indices))
        return [False] * n

    def metrics(self):
        return self._b.read_metrics()

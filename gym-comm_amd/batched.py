"""BatchedOvercooked: N independent Overcooked envs of one level on one MI355X.

Host side of the hot path: owns the PyTorch-ROCm tensors (env-major int32 SoA) and calls
the hand-written HIP kernels of liboc_hip.so through the C ABI (include/oc_hip.h) with
raw device pointers on torch's current stream.  No compute happens in Python or torch.

Mirrors, batched:
  gym_cooking.envs.OvercookedEnvironment.step/reset    (overcooked_environment.py:180-241)
  gym_comm.envs.OvercookedMultiEnv.multi_step/multi_reset/get_observation2
                                                       (overcooked_env.py:105-297)
"""
import contextlib
import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib, compiler, specialize
from .state import unpack_state

OBS_KEYS = ["object_encodings_x", "object_encodings_y", "state_encodings", "is_hidden",
            "completed_subtasks", "agent1_location", "agent2_location", "agent_is_holding",
            "agent1_comm", "agent2_comm"]


def obs_layout(S, C):
    """Row ranges of each observation key inside the [F] axis (F = 22 + S + 2C)."""
    sizes = [4, 4, 4, 4, S, 2, 2, 2, C, C]
    out, r = {}, 0
    for k, s in zip(OBS_KEYS, sizes):
        out[k] = (r, r + s)
        r += s
    return out


class BatchedOvercooked:
    def __init__(self, level, num_agents=2, num_envs=4096, max_num_timesteps=100,
                 ego_config=None, partner_config=None, num_communication=2,
                 communication_on=True, ego_led=False, fow_radius=2, ego_agent_idx=0,
                 device="cuda", subtask_order=None, placements=None, level_dir=None,
                 max_num_subtasks=14, auto_reset=True, track_metrics=True, specialize_level="auto",
                 seed=0, placement_mode="rng", obs_dtype=torch.int32, episode_stats=False, play=False,
                 waves_per_64=0):
        cfg = {"ALLERGIC": False, "BLIND": False, "CAN_MOVE": True}   # missing CAN_MOVE = True
        self.ego_config = dict(cfg, **(ego_config or {}))
        self.partner_config = dict(cfg, **(partner_config or {}))
        if isinstance(level, compiler.CompiledLevel):
            self.level = level
        else:
            self.level = compiler.compile_level(
                level, num_agents, max_num_timesteps, max_num_subtasks,
                ego_allergic=self.ego_config["ALLERGIC"],
                partner_allergic=self.partner_config["ALLERGIC"],
                subtask_order=subtask_order, placements=placements, level_dir=level_dir, play=play)
        lv = self.level
        if not lv.hip_supported:
            raise ValueError("level %r: more than three items of one type, or more than 16 distinct "
                             "merged object names (limits of the packed item words)" % lv.name)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.OcError("BatchedOvercooked needs a ROCm device (got %s); there is no CPU path"
                               % self.device)
        self.n = int(num_envs)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", self._dev_index)
        self.A, self.M, self.S = lv.num_agents, lv.num_items, lv.num_subtasks
        self.C = int(num_communication)
        self.auto_reset = bool(auto_reset)
        # launch hint of the fused step (include/oc_hip.h, oc_step_opts.waves_per_64): 0 = the
        # library decides, 1 = one wave per 64 envs, 4 = split launch; results are identical
        if waves_per_64 not in (0, 1, 2, 4):
            raise ValueError("waves_per_64 must be 0 (auto), 1, 2 or 4")
        self.waves_per_64 = int(waves_per_64)
        blob = np.ascontiguousarray(lv.blob, dtype=np.int32)
        # per-level specialised kernels when available (specialize.py), else the generic library
        self.kernel_flavour, self._L = specialize.load_for(blob, specialize_level)
        with torch.cuda.device(self.device):
            h = ctypes.c_void_p()
            _lib.check(self._L.oc_level_create(blob.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                               int(blob.size), ctypes.byref(h)), "oc_level_create", self._L)
        self._h = h
        # where the state keeps the bits of subtask s (the library's canonical subtask order)
        self.subtask_slot, self._goal_index, self._dup = _lib.subtask_info(blob, self._L)
        self.W_state = self._L.oc_state_words(h)
        self.F = self._L.oc_obs_rows(h, self.C)
        n = self.n
        if obs_dtype not in (torch.int32, torch.int8, torch.float32):
            raise ValueError("obs_dtype must be torch.int32, torch.int8 or torch.float32")
        # Every tensor a step reads or writes is a view of ONE device allocation (each view
        # 256-byte aligned): a host consumer fetches the whole step -- state, rewards, done,
        # observations -- with a single device->host copy (`fetch()`), instead of one per tensor.
        spec = [("state", (self.W_state, n), torch.int32), ("reward", (n,), torch.int32),
                ("done", (n,), torch.int32), ("shaping", (2, n), torch.float64),
                ("comm", (2, n), torch.int32),                  # one-hot(0) (overcooked_env.py:89-91)
                ("obs", (2, self.F, n), obs_dtype), ("timestep", (n,), torch.float64),
                ("shaped_reward", (n,), torch.float64)]
        if episode_stats:           # kept by the fused kernel (include/oc_hip.h: ep_return / ep_length)
            spec += [("ep_return", (n,), torch.float64), ("ep_length", (n,), torch.int32)]
        else:
            self.ep_return = self.ep_length = None
        self._arena_layout, off = {}, 0
        for name, shape, dt in spec:
            nb = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
            self._arena_layout[name] = (off, nb, shape, dt)
            off += (nb + 255) // 256 * 256
        self._arena = torch.zeros(max(off, 256), dtype=torch.uint8, device=self.device)
        for name, (o, nb, shape, dt) in self._arena_layout.items():
            setattr(self, name, self._arena[o:o + nb].view(dt).view(shape))
        self.metrics = (torch.zeros((self._L.oc_metrics_slots(n), 8), dtype=torch.int64,
                                    device=self.device) if track_metrics else None)
        self._obs_cfg = _lib.ObsCfg(int(fow_radius),
                                    (1 if self.ego_config["BLIND"] else 0) |
                                    (2 if self.partner_config["BLIND"] else 0), self.C,
                                    {torch.int32: 0, torch.int8: 1, torch.float32: 2}[obs_dtype])
        self._wrap_cfg = _lib.WrapCfg(self._obs_cfg, int(bool(communication_on)), int(bool(ego_led)),
                                      int(ego_agent_idx),
                                      (1 if self.ego_config["CAN_MOVE"] else 0) |
                                      (2 if self.partner_config["CAN_MOVE"] else 0))
        self._layout = obs_layout(self.S, self.C)
        self._ms_args = None
        self._policy_arr = None
        self._arena_pinned = None
        self._fetch_plan = None
        # random-* levels: item start cells differ per env and per episode
        self.placement = None
        self.rng = None
        if lv.random_placement:
            if placement_mode == "rng":
                g = torch.Generator(device="cpu").manual_seed(int(seed))
                self.rng = torch.randint(0, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int64
                                         ).to(torch.int32).to(self.device)
            elif placement_mode == "host":
                nominal = torch.tensor(lv.pack_placement([(x, y) for _, x, y in lv.items]),
                                       dtype=torch.int32)
                self.placement = nominal.view(-1, 1).repeat(1, n).contiguous().to(self.device)
            else:
                raise ValueError("placement_mode must be 'rng' or 'host'")
        self.reset()

    def __del__(self):
        for c in getattr(self, "_calls", []):
            self._L.oc_call_destroy(c)
        self._calls = []
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._L.oc_level_destroy(h)
            self._h = None

    @property
    def launch_waves_per_64(self):
        """Waves per 64 envs the PLAIN fused step is launched with (1, 2 or 4 = split launches)."""
        return self.launch_waves(general=False)

    @property
    def standard_wrapper_config(self):
        """The wrapper configuration of the reference's own run files (communication on, not ego-led,
        both players CAN_MOVE, ego = sim agent 0, nobody BLIND, play off): what the plain step and
        the options-on-the-standard-configuration variant of the fused kernel fold at compile time
        (csrc/oc_kernels.hip: XO = 0 / 1); anything else runs the general variant."""
        c = self._wrap_cfg
        return bool(c.communication_on and not c.ego_led and c.can_move_mask == 3 and c.ego_agent_idx == 0
                    and c.obs.blind_mask == 0 and not self.level.play)

    def launch_waves(self, general=False):
        """Waves per 64 envs the library launches for this batch (include/oc_hip.h:
        oc_multi_step_waves): the plain step, or -- ``general`` -- the general variant (pairs /
        in-kernel partner / episode statistics / policies / a non-standard wrapper configuration:
        what ``OvercookedVecEnv`` launches), which splits four ways or not at all."""
        return int(self._L.oc_multi_step_waves(self.n, self.waves_per_64, 1 if general else 0))

    # -- helpers ---------------------------------------------------------------
    def _on_device(self):
        """Kernels must be launched with this env's device current (one process per GPU is
        the normal case and costs nothing here)."""
        if torch.cuda.current_device() == self._dev_index:
            return contextlib.nullcontext()
        return torch.cuda.device(self.device)

    def _stream(self):
        return ctypes.c_void_p(self._raw_stream())

    def _raw_stream(self):
        """hipStream_t of torch's current stream on this env's device, as an int."""
        try:
            return torch._C._cuda_getCurrentRawStream(self._dev_index)      # ~0.3 us
        except AttributeError:                                             # older/newer torch
            return torch.cuda.current_stream(self.device).cuda_stream

    @staticmethod
    def _p(t: Optional[torch.Tensor]):
        return ctypes.c_void_p(0 if t is None else t.data_ptr())

    def _check_tensor(self, t, shape, dtype, name):
        if not isinstance(t, torch.Tensor):
            raise TypeError("%s must be a torch tensor" % name)
        if t.device != self.state.device or t.dtype != dtype or tuple(t.shape) != tuple(shape) \
                or not t.is_contiguous():
            raise ValueError("%s must be a contiguous %s tensor of shape %s on %s (got %s %s %s)"
                             % (name, dtype, tuple(shape), self.state.device, t.dtype,
                                tuple(t.shape), t.device))

    # -- API -------------------------------------------------------------------
    def reset(self, mask: Optional[torch.Tensor] = None):
        """Reset all envs, or those with mask[n] != 0 (int32 [n])."""
        if mask is not None:
            self._check_tensor(mask, (self.n,), torch.int32, "mask")
        with self._on_device():
            _lib.check(self._L.oc_reset(self._h, self._p(self.state), self._p(mask),
                                        self._p(self.placement), self._p(self.rng), self.n,
                                        self._stream()), "oc_reset", self._L)

    def set_placement(self, placement: torch.Tensor):
        """placement_mode='host': the start cells (int32 [M][n], x | y<<4, world order) every
        env uses at its next reset / auto-reset."""
        if self.placement is None:
            raise ValueError("this env was not created with placement_mode='host' on a random-* level")
        self._check_tensor(placement, (self.M, self.n), torch.int32, "placement")
        self.placement.copy_(placement)

    def step(self, actions: torch.Tensor, auto_reset: Optional[bool] = None):
        """Base-env step.  actions: int32 [A][n] action codes 0..4 (4 = stay).
        Returns (reward int32[n], done int32[n], shaping f64[2][n]) -- views of
        pre-allocated tensors, overwritten by the next call."""
        self._check_tensor(actions, (self.A, self.n), torch.int32, "actions")
        ar = self.auto_reset if auto_reset is None else auto_reset
        with self._on_device():
            _lib.check(self._L.oc_step(self._h, self._p(self.state), self._p(actions),
                                       self._p(self.reward), self._p(self.done), self._p(self.shaping),
                                       int(ar), self._p(self.metrics), self._p(self.placement),
                                       self._p(self.rng), self.n, self._stream()), "oc_step", self._L)
        return self.reward, self.done, self.shaping

    def observe(self):
        """Both viewers' observations of the current state.  Returns (obs int32 [2][F][n],
        timestep f64 [n])."""
        with self._on_device():
            _lib.check(self._L.oc_obs(self._h, self._p(self.state), self._p(self.comm),
                                      ctypes.byref(self._obs_cfg), self._p(self.obs),
                                      self._p(self.timestep), self.n, self._stream()), "oc_obs", self._L)
        return self.obs, self.timestep

    def multi_step(self, actions: Optional[torch.Tensor] = None, auto_reset: Optional[bool] = None,
                   ego_pairs: Optional[torch.Tensor] = None, alt_pairs: Optional[torch.Tensor] = None,
                   alt_rng: Optional[torch.Tensor] = None, alt_played: Optional[torch.Tensor] = None,
                   policy=None):
        """gym_comm wrapper step in one launch.  actions: int32 [4][n] = ego move (0..3),
        ego comm, alt move, alt comm.  Per player the (move, comm) may instead come as an
        int32 or int64 [n][2] tensor of pairs (`ego_pairs` / `alt_pairs`: a policy's [n, 2] output as it
        lies), and the partner may be drawn by the kernel itself, uniformly, from a per-env
        PCG32 stream (`alt_rng`, int32/uint32 [n]; `alt_played` int32 [2][n] receives the draw)
        -- include/oc_hip.h, oc_step_opts.  `policy`: a pair of device-address tuples
        (w1, w2, b2, rng or None) -- the two players' packed MLPs (include/oc_policy.h): the kernel
        evaluates them behind the step and overwrites `ego_pairs` / `alt_pairs` (int32) with the
        NEXT step's actions: the closed loop in one launch.
        Returns (obs, timestep, shaped_reward f64[n], done)."""
        n = self.n
        if actions is not None:
            self._check_tensor(actions, (4, n), torch.int32, "actions")
        elif ego_pairs is None or (alt_pairs is None and alt_rng is None):
            raise ValueError("multi_step needs `actions`, or `ego_pairs` and one of `alt_pairs` / `alt_rng`")
        pdt = None
        for name, t in (("ego_pairs", ego_pairs), ("alt_pairs", alt_pairs)):
            if t is not None:
                if t.dtype not in (torch.int32, torch.int64) or (pdt is not None and t.dtype != pdt):
                    raise ValueError("ego_pairs / alt_pairs must be int32 or int64, both the same")
                pdt = t.dtype
                self._check_tensor(t, (n, 2), pdt, name)
        if alt_rng is not None:
            self._check_tensor(alt_rng, (n,), torch.int32, "alt_rng")
        if alt_played is not None:
            self._check_tensor(alt_played, (2, n), torch.int32, "alt_played")
        ptr = lambda t: None if t is None else t.data_ptr()
        if policy is not None:
            if ego_pairs is None or alt_pairs is None or pdt is not torch.int32 or alt_rng is not None:
                raise ValueError("policy= needs int32 ego_pairs and alt_pairs (it overwrites them) and no alt_rng")
            if self._policy_arr is None:
                self._policy_arr = (_lib.StepPolicy * 2)()
            for k, (w1, w2, b2, rng) in enumerate(policy):
                self._policy_arr[k] = _lib.StepPolicy(w1, w2, b2, rng)
        self.multi_step_raw(ptr(actions) or 0, ptr(ego_pairs), ptr(alt_pairs), ptr(alt_rng), ptr(alt_played),
                            int(self.auto_reset if auto_reset is None else auto_reset), pdt is torch.int64,
                            policy=self._policy_arr if policy is not None else None)
        return self.obs, self.timestep, self.shaped_reward, self.done

    def multi_step_raw(self, actions_ptr, ego_pairs_ptr, alt_pairs_ptr, alt_rng_ptr, alt_played_ptr, auto_reset,
                       pairs_int64=False, policy=None):
        """multi_step on raw device addresses (int, None = absent), nothing checked: the per-call
        cost is one ctypes call.  For callers that validated their tensors once (vec_env)."""
        a = self._ms_args
        if a is None:       # every pointer but the action sources is fixed for the life of the env
            dp = lambda t: 0 if t is None else t.data_ptr()
            self._ms_opts = _lib.StepOpts(dp(self.ep_return) or None, dp(self.ep_length) or None,
                                          None, None, None, None, 0, self.waves_per_64)
            a = self._ms_args = [self._h, dp(self.state), dp(self.comm), 0, ctypes.byref(self._wrap_cfg),
                                 dp(self.obs), dp(self.timestep), dp(self.shaped_reward), dp(self.done),
                                 dp(self.reward), 0, dp(self.metrics), dp(self.placement), dp(self.rng),
                                 ctypes.byref(self._ms_opts), self.n, 0]
        o = self._ms_opts
        o.ego_pairs, o.alt_pairs, o.alt_rng, o.alt_played = ego_pairs_ptr, alt_pairs_ptr, alt_rng_ptr, alt_played_ptr
        o.pairs_int64 = 1 if pairs_int64 else 0
        o.policy = policy if policy is not None else None    # (_lib.StepPolicy * 2) or NULL
        a[3] = actions_ptr
        a[10] = auto_reset
        a[16] = self._raw_stream()
        if torch.cuda.current_device() == self._dev_index:
            rc = self._L.oc_multi_step(*a)
        else:
            with torch.cuda.device(self.device):
                rc = self._L.oc_multi_step(*a)
        if rc:
            _lib.check(rc, "oc_multi_step", self._L)

    def prepare_multi_step(self, actions_ptr, alt_rng_ptr=None, alt_played_ptr=None, alt_pairs_ptr=None,
                           auto_reset=True):
        """A PREPARED fused step (include/oc_hip.h: oc_multi_step_prepare): every argument but the ego's
        pair tensor fixed once; returns ``launch(ego_pairs_ptr_or_None, pairs_int64)`` whose per-call
        cost is a 4-argument foreign call (the 17-argument one costs a Python caller ~2 us more)."""
        dp = lambda t: 0 if t is None else t.data_ptr()
        opts = _lib.StepOpts(dp(self.ep_return) or None, dp(self.ep_length) or None, None, alt_pairs_ptr,
                             alt_rng_ptr, alt_played_ptr, 0, self.waves_per_64)
        h = ctypes.c_void_p()
        with self._on_device():
            _lib.check(self._L.oc_multi_step_prepare(
                self._h, dp(self.state), dp(self.comm), actions_ptr or 0, ctypes.byref(self._wrap_cfg), dp(self.obs),
                dp(self.timestep), dp(self.shaped_reward), dp(self.done), dp(self.reward), int(auto_reset),
                dp(self.metrics), dp(self.placement), dp(self.rng), ctypes.byref(opts), self.n, ctypes.byref(h)),
                "oc_multi_step_prepare", self._L)
        self._calls = getattr(self, "_calls", [])
        self._calls.append(h)
        L, dev, call = self._L, self._dev_index, h
        raw_stream = torch._C._cuda_getCurrentRawStream

        def launch(ego_pairs_ptr=None, pairs_int64=False):
            if torch.cuda.current_device() == dev:
                rc = L.oc_call_launch(call, ego_pairs_ptr, 1 if pairs_int64 else 0, raw_stream(dev))
            else:
                with torch.cuda.device(dev):
                    rc = L.oc_call_launch(call, ego_pairs_ptr, 1 if pairs_int64 else 0, raw_stream(dev))
            if rc:
                _lib.check(rc, "oc_call_launch", L)
        return launch

    def observe_image(self, radius: Optional[int] = None, packed: bool = False):
        """Image-style fog-of-war observation of both viewers
        (get_partial_observability_FOW, overcooked_env.py:161-202).  Returns
        (maps int8 [2][7][W][H][n], holding int8 [2][n]).  The kernel writes four consecutive
        cells of a plane of an env per dword ([2][7*ceil(WH/4)][n] int32, include/oc_hip.h); the
        [2][7][W][H][n] result is one re-layout copy of that -- `packed=True` returns the
        kernel's tensor itself."""
        lv = self.level
        if getattr(self, "_image", None) is None:
            self._image_words = self._L.oc_image_words(self._h)
            self._image = torch.zeros((2, self._image_words, self.n), dtype=torch.int32, device=self.device)
            self._holding = torch.zeros((2, self.n), dtype=torch.int8, device=self.device)
        r = self._obs_cfg.fow_radius if radius is None else int(radius)
        with self._on_device():
            _lib.check(self._L.oc_obs_image(self._h, self._p(self.state), r, self._p(self._image),
                                            self._p(self._holding), self.n, self._stream()),
                       "oc_obs_image", self._L)
        if packed:
            return self._image, self._holding
        cells = lv.width * lv.height
        q = self._image_words // 7
        img = self._image.view(torch.int8).view(2, 7, q, self.n, 4).permute(0, 1, 2, 4, 3).reshape(2, 7, 4 * q, self.n)
        return img[:, :, :cells].reshape(2, 7, lv.width, lv.height, self.n), self._holding

    def completed_subtasks(self):
        """completed_subtasks of every env as int32 [S][n] (from the packed state)."""
        word = self.state[self.A + self.M]
        if getattr(self, "_slot_t", None) is None:
            self._slot_t = torch.tensor(self.subtask_slot, device=self.device, dtype=torch.int32).view(-1, 1)
        return (word.view(1, -1) >> self._slot_t) & 1

    def obs_dict(self, viewer: int):
        """The 11 observation keys of get_observation2 as tensor views: key -> [k][n]
        (``.T`` gives the [n][k] batch a policy takes); 'timestep' is f64 [1][n]."""
        d = {"timestep": self.timestep.unsqueeze(0)}
        for k, (a, b) in self._layout.items():
            d[k] = self.obs[viewer, a:b]
        return d

    def snapshot(self):
        """Named fields of every env's state (host numpy), see state.unpack_state."""
        return unpack_state(self.state.cpu().numpy(), self.A, self.M, self.S, **self.unpack_kw())

    def unpack_kw(self):
        """Extra arguments state.unpack_state needs for this level: the bit of every subtask in
        the state's subtask words and, in dup mode, where its goal count lives."""
        kw = {"slot": self.subtask_slot}
        if self._dup:
            kw.update(goal_index=self._goal_index,
                      deliver=[s.kind == compiler.KIND_DELIVER for s in self.level.subtasks])
        return kw

    def fetch(self):
        """ONE device->host copy of everything a step produced: returns {name: numpy view}
        for state, reward, done, shaping, comm, obs, timestep, shaped_reward (host copies
        with the device tensors' shapes and dtypes).  Synchronises with the stream."""
        # through a pinned staging buffer (an asynchronous copy + one stream sync: about half the
        # latency of a pageable .cpu() for the single-env adapter's 1 KB arena), then a host copy
        # so that the caller owns what it gets
        if self._arena_pinned is None:
            self._arena_pinned = torch.empty(self._arena.numel(), dtype=torch.uint8, pin_memory=True)
        self._arena_pinned.copy_(self._arena, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        host = self._arena_pinned.numpy().copy()
        if self._fetch_plan is None:
            self._fetch_plan = [(name, o, o + nb, np.dtype(str(dt).replace("torch.", "")), shape)
                                for name, (o, nb, shape, dt) in self._arena_layout.items()]
        return {name: host[lo:hi].view(ndt).reshape(shape) for name, lo, hi, ndt, shape in self._fetch_plan}

    def metrics_vector(self):
        """int64[8] totals (sum over the per-wave slots), on the device; None when the env was
        built with track_metrics=False."""
        if self.metrics is None:
            return None
        return self.metrics.sum(dim=0)

    def read_metrics(self):
        if self.metrics is None:
            return None
        m = self.metrics_vector().cpu().tolist()
        return {"env_steps": m[0], "episodes": m[1], "successes": m[2], "reward_sum": m[3],
                "completed_subtasks_sum": m[4], "errors": m[5]}

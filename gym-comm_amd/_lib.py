"""ctypes loader for liboc_hip.so (include/oc_hip.h) and its per-level specialisations.

There is no CPU fallback: if the library cannot be loaded, or a call fails, this
raises.  (The CPU oracle under oracle/ is test infrastructure and is never used here.)
"""
import ctypes
import os

from . import build as _build

_I32P = ctypes.POINTER(ctypes.c_int32)
ABI_VERSION = 6          # include/oc_hip.h: OC_ABI_VERSION

SYMBOLS = ["oc_abi_version", "oc_last_error", "oc_level_create", "oc_level_destroy",
           "oc_level_spec_source", "oc_is_specialized", "oc_level_subtask_info",
           "oc_metrics_slots", "oc_state_words", "oc_obs_rows", "oc_reset", "oc_step", "oc_obs",
           "oc_obs_image", "oc_image_words", "oc_multi_step", "oc_multi_step_waves", "oc_random_actions",
           "oc_timeline_begin", "oc_multi_step_prepare", "oc_call_launch", "oc_call_destroy"]


class ObsCfg(ctypes.Structure):
    _fields_ = [("fow_radius", ctypes.c_int32), ("blind_mask", ctypes.c_int32),
                ("num_comm", ctypes.c_int32), ("obs_int8", ctypes.c_int32)]


class WrapCfg(ctypes.Structure):
    _fields_ = [("obs", ObsCfg), ("communication_on", ctypes.c_int32),
                ("ego_led", ctypes.c_int32), ("ego_agent_idx", ctypes.c_int32),
                ("can_move_mask", ctypes.c_int32)]


class StepPolicy(ctypes.Structure):
    """oc_step_policy (include/oc_hip.h): one player's packed MLP for oc_step_opts.policy."""
    _fields_ = [("w1", ctypes.c_void_p), ("w2", ctypes.c_void_p), ("b2", ctypes.c_void_p), ("rng", ctypes.c_void_p)]


class StepOpts(ctypes.Structure):
    """oc_step_opts (include/oc_hip.h): optional device pointers of oc_multi_step."""
    _fields_ = [("ep_return", ctypes.c_void_p), ("ep_length", ctypes.c_void_p),
                ("ego_pairs", ctypes.c_void_p), ("alt_pairs", ctypes.c_void_p),
                ("alt_rng", ctypes.c_void_p), ("alt_played", ctypes.c_void_p),
                ("pairs_int64", ctypes.c_int32), ("waves_per_64", ctypes.c_int32),
                ("policy", ctypes.POINTER(StepPolicy))]


POLICY_ABI_VERSION = 1   # include/oc_policy.h: OC_POLICY_ABI_VERSION
POLICY_SYMBOLS = ["oc_policy_abi_version", "oc_policy_last_error", "oc_policy_ksteps", "oc_policy_pack_w1",
                  "oc_policy_pack_w2", "oc_policy_pack_b2", "oc_policy_mlp"]


class PolicyPlayer(ctypes.Structure):
    """oc_policy_player (include/oc_policy.h)."""
    _fields_ = [("obs", ctypes.c_void_p), ("w1", ctypes.c_void_p), ("w2", ctypes.c_void_p),
                ("b2", ctypes.c_void_p), ("rng", ctypes.c_void_p), ("pairs", ctypes.c_void_p),
                ("logits", ctypes.c_void_p)]


class OcError(RuntimeError):
    pass


_libs = {}
_hip_preloaded = False


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).
    Our library must bind to THAT runtime instance -- device pointers and streams come
    from torch -- so make sure it is the one already loaded when liboc_hip.so resolves
    its libamdhip64.so.7 dependency."""
    global _hip_preloaded
    if _hip_preloaded:
        return
    import torch
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    _hip_preloaded = True


def _declare(L):
    vp = ctypes.c_void_p
    L.oc_abi_version.restype = ctypes.c_int
    L.oc_is_specialized.restype = ctypes.c_int
    L.oc_last_error.restype = ctypes.c_char_p
    L.oc_level_create.argtypes = [_I32P, ctypes.c_int32, ctypes.POINTER(vp)]
    L.oc_level_destroy.argtypes = [vp]
    L.oc_level_spec_source.argtypes = [_I32P, ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p, ctypes.c_int32]
    L.oc_level_subtask_info.argtypes = [_I32P, ctypes.c_int32, _I32P, _I32P, _I32P]
    L.oc_metrics_slots.argtypes = [ctypes.c_int64]
    L.oc_metrics_slots.restype = ctypes.c_int64
    L.oc_state_words.argtypes = [vp]
    L.oc_state_words.restype = ctypes.c_int32
    L.oc_obs_rows.argtypes = [vp, ctypes.c_int32]
    L.oc_obs_rows.restype = ctypes.c_int32
    L.oc_image_words.argtypes = [vp]
    L.oc_image_words.restype = ctypes.c_int32
    L.oc_reset.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int64, vp]
    L.oc_step.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_int32, vp, vp, vp, ctypes.c_int64, vp]
    L.oc_obs.argtypes = [vp, vp, vp, ctypes.POINTER(ObsCfg), vp, vp, ctypes.c_int64, vp]
    L.oc_obs_image.argtypes = [vp, vp, ctypes.c_int32, vp, vp, ctypes.c_int64, vp]
    L.oc_multi_step.argtypes = [vp, vp, vp, vp, ctypes.POINTER(WrapCfg), vp, vp, vp, vp, vp,
                                ctypes.c_int32, vp, vp, vp, ctypes.POINTER(StepOpts), ctypes.c_int64, vp]
    L.oc_random_actions.argtypes = [vp, vp, vp, ctypes.c_int32, ctypes.c_int64, vp]
    L.oc_multi_step_waves.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
    L.oc_multi_step_waves.restype = ctypes.c_int32
    L.oc_timeline_begin.argtypes = [vp, ctypes.c_int64, ctypes.c_int64]
    L.oc_multi_step_prepare.argtypes = [vp, vp, vp, vp, ctypes.POINTER(WrapCfg), vp, vp, vp, vp, vp,
                                        ctypes.c_int32, vp, vp, vp, ctypes.POINTER(StepOpts), ctypes.c_int64,
                                        ctypes.POINTER(vp)]
    L.oc_call_launch.argtypes = [vp, vp, ctypes.c_int32, vp]
    L.oc_call_destroy.argtypes = [vp]
    for f in ("oc_level_create", "oc_level_destroy", "oc_level_spec_source", "oc_level_subtask_info",
              "oc_reset", "oc_step",
              "oc_obs", "oc_obs_image", "oc_multi_step", "oc_random_actions", "oc_timeline_begin",
              "oc_multi_step_prepare", "oc_call_launch", "oc_call_destroy"):
        getattr(L, f).restype = ctypes.c_int
    if L.oc_abi_version() != ABI_VERSION:
        raise OcError("liboc_hip.so ABI version mismatch")
    return L


def load(path=None):
    """Load (once per path) and type a library.  Default: the generic liboc_hip.so."""
    path = os.path.abspath(path or os.environ.get("OC_HIP_LIB") or _build.LIB)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise OcError(
            "HIP extension %s not built: run `python -c 'import __graft_entry__ as g; g.build()'`"
            " (there is no CPU fallback)" % path)
    _preload_torch_hip_runtime()
    L = _declare(ctypes.CDLL(path))
    L._oc_path = path
    _libs[path] = L
    return L


def check(rc, what, lib=None):
    if rc != 0:
        msg = (lib or load()).oc_last_error()
        raise OcError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def subtask_info(blob, lib=None):
    """(slot, goal_index, dup) of a level blob (include/oc_hip.h: oc_level_subtask_info): where the
    state tensor keeps the bits of the blob's subtask s.  Host only."""
    import numpy as np
    L = lib or load()
    blob = np.ascontiguousarray(blob, dtype=np.int32)
    S = int(blob[6])
    slot, gi = np.zeros(S, np.int32), np.zeros(S, np.int32)
    dup = ctypes.c_int32()
    check(L.oc_level_subtask_info(blob.ctypes.data_as(_I32P), int(blob.size), slot.ctypes.data_as(_I32P),
                                  gi.ctypes.data_as(_I32P), ctypes.byref(dup)), "oc_level_subtask_info", L)
    return slot.tolist(), gi.tolist(), bool(dup.value)


def load_policy(path=None):
    """Load (once) and type liboc_policy.so (include/oc_policy.h): the fused MLP policy kernel."""
    path = os.path.abspath(path or os.environ.get("OC_POLICY_LIB") or _build.POLICY_LIB)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise OcError(
            "HIP extension %s not built: run `python -c 'import __graft_entry__ as g; g.build()'`"
            " (there is no CPU fallback)" % path)
    _preload_torch_hip_runtime()
    L = ctypes.CDLL(path)
    vp, fp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)
    L.oc_policy_abi_version.restype = ctypes.c_int
    L.oc_policy_last_error.restype = ctypes.c_char_p
    L.oc_policy_ksteps.argtypes = [ctypes.c_int32]
    L.oc_policy_ksteps.restype = ctypes.c_int32
    L.oc_policy_pack_w1.argtypes = [fp, fp, fp, ctypes.c_int32, vp]
    L.oc_policy_pack_w2.argtypes = [fp, ctypes.c_int32, vp]
    L.oc_policy_pack_b2.argtypes = [fp, fp, ctypes.c_int32, fp]
    L.oc_policy_mlp.argtypes = [ctypes.POINTER(PolicyPlayer), ctypes.c_int32, vp, ctypes.c_int32, ctypes.c_int32,
                                ctypes.c_int32, ctypes.c_int64, vp]
    for f in ("oc_policy_pack_w1", "oc_policy_pack_w2", "oc_policy_pack_b2", "oc_policy_mlp"):
        getattr(L, f).restype = ctypes.c_int
    if L.oc_policy_abi_version() != POLICY_ABI_VERSION:
        raise OcError("liboc_policy.so ABI version mismatch")
    L._oc_path = path
    _libs[path] = L
    return L


HOSTIO_ABI_VERSION = 1   # include/oc_hostio.h: OC_HOSTIO_ABI_VERSION
HOSTIO_SYMBOLS = ["oc_hostio_abi_version", "oc_hostio_last_error", "oc_pack_host_bytes", "oc_pack_host"]


def load_hostio(path=None):
    """Load (once) and type liboc_hostio.so (include/oc_hostio.h): the numpy boundary's pack kernel."""
    path = os.path.abspath(path or os.environ.get("OC_HOSTIO_LIB") or _build.HOSTIO_LIB)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise OcError(
            "HIP extension %s not built: run `python -c 'import __graft_entry__ as g; g.build()'`"
            " (there is no CPU fallback)" % path)
    _preload_torch_hip_runtime()
    L = ctypes.CDLL(path)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.oc_hostio_abi_version.restype = ctypes.c_int
    L.oc_hostio_last_error.restype = ctypes.c_char_p
    L.oc_pack_host_bytes.argtypes = [i32, i32, i32, i32, i32, i32, i32, i64]
    L.oc_pack_host_bytes.restype = i64
    L.oc_pack_host.argtypes = [vp, i32, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, i64, vp]
    L.oc_pack_host.restype = ctypes.c_int
    if L.oc_hostio_abi_version() != HOSTIO_ABI_VERSION:
        raise OcError("liboc_hostio.so ABI version mismatch")
    L._oc_path = path
    _libs[path] = L
    return L

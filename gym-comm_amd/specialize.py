"""Per-level kernel specialisation.

The generic liboc_hip.so reads a level's static tables (map bit-planes, goal sets,
item types, shaping lookup programs) from kernel arguments at run time.  For a level
that is stepped millions of times it pays to fold them into the code instead: the same
source (csrc/oc_kernels.hip) compiled with ``-DOC_SPECIALIZED`` and a generated header
that holds the level as a ``constexpr LevelHdr`` gives straight-line kernels with no
scalar loads, no uniform branches and fully unrolled loops.  The result is an ordinary
shared library exporting the same C ABI (include/oc_hip.h) that refuses any other level.

Libraries are cached in csrc/_spec/ under a hash of (generated header, kernel source,
C headers), so they are built once -- by ``__graft_entry__.build()`` for the BASELINE
levels, or on first use wherever hipcc is available -- and travel with the repo snapshot.
max_num_timesteps and the ALLERGIC flags stay run-time arguments and do not select a
specialisation; the subtask order does (it fixes bit positions).
"""
import ctypes
import hashlib
import os
import subprocess
import threading

import numpy as np

from . import _lib
from . import build as _build

SPEC_DIR = os.path.join(_build.CSRC, "_spec")
_I32P = ctypes.POINTER(ctypes.c_int32)


_PROFILER_VARS = ("ROCP_", "ROCPROF", "ROCTRACER", "HSA_TOOLS_LIB", "ROCPROFILER")


def profiler_attached() -> bool:
    """True when this process runs under rocprofv3 / rocprof (their tool library is preloaded
    or injected through HSA_TOOLS_LIB)."""
    pre = os.environ.get("LD_PRELOAD", "")
    if "rocprof" in pre or "roctracer" in pre:
        return True
    return any(k.startswith(_PROFILER_VARS) for k in os.environ)


def clean_env():
    """Environment for the hipcc child chain: nothing of a profiler's injection survives."""
    env = {k: v for k, v in os.environ.items() if not k.startswith(_PROFILER_VARS)}
    env.pop("LD_PRELOAD", None)
    return env


def spec_header_text(blob) -> str:
    """The generated header for a level blob (host only, no GPU needed)."""
    L = _lib.load()
    blob = np.ascontiguousarray(blob, dtype=np.int32)
    buf = ctypes.create_string_buffer(16384)
    n = L.oc_level_spec_source(blob.ctypes.data_as(_I32P), int(blob.size), buf, len(buf))
    if n < 0:
        _lib.check(n, "oc_level_spec_source", L)
    return buf.value.decode()


def _source_digest():
    h = hashlib.sha1()
    inc = os.path.join(_build.CSRC, "..", "..", "include")
    files = [os.path.join(_build.CSRC, s) for s in _build.SOURCES]
    files += sorted(os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(_build.FLAGS).encode())
    return h


def spec_key(blob) -> str:
    h = _source_digest()
    h.update(spec_header_text(blob).encode())
    return h.hexdigest()[:16]


def spec_lib_path(blob) -> str:
    return os.path.join(SPEC_DIR, "liboc_spec_%s.so" % spec_key(blob))


def ensure(blob, verbose=False):
    """Return the path of the specialised library for this level, building it if needed.
    Returns None when it is not cached and hipcc is unavailable."""
    text = spec_header_text(blob)
    h = _source_digest()
    h.update(text.encode())
    key = h.hexdigest()[:16]
    path = os.path.join(SPEC_DIR, "liboc_spec_%s.so" % key)
    if os.path.exists(path):
        return path
    if profiler_attached():
        # hipcc execs clang and lld; under rocprofv3 this process has already initialised the
        # GPU (the profiler's preloaded library does) and those exec hops would inherit its
        # LD_PRELOAD -- the pattern that takes this pool's machines down.  Never JIT here: the
        # caller falls back to the generic library (or fails, with mode=True).
        return None
    try:
        hipcc = _build.hipcc_path()
    except RuntimeError:
        return None
    os.makedirs(SPEC_DIR, exist_ok=True)
    # several ranks may reach this point for the same level at once (one process per GPU):
    # every writer uses its own temporary names and publishes with an atomic rename
    hdr = os.path.join(SPEC_DIR, "spec_%s.h" % key)
    uniq = "%d.%d" % (os.getpid(), threading.get_ident())   # ranks AND threads may build the same level at once
    tmp_hdr = "%s.%s.tmp" % (hdr, uniq)
    with open(tmp_hdr, "w") as f:
        f.write(text)
    os.replace(tmp_hdr, hdr)
    tmp_lib = "%s.%s.tmp" % (path, uniq)
    cmd = [hipcc, "--offload-arch=" + _build.ARCH] + _build.FLAGS
    cmd += ["-DOC_SPECIALIZED", '-DOC_SPEC_FILE="%s"' % hdr]
    cmd += [os.path.join(_build.CSRC, s) for s in _build.SOURCES] + ["-o", tmp_lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    try:
        subprocess.check_call(cmd, env=clean_env())
        os.replace(tmp_lib, path)
    finally:
        if os.path.exists(tmp_lib):
            os.remove(tmp_lib)
    return path


def load_for(blob, mode="auto", verbose=False):
    """The library to use for a level: ('spec' | 'generic', typed CDLL).
    mode: True (must specialise), False (generic), 'auto' (specialise when possible)."""
    if os.environ.get("OC_SPECIALIZE") == "0" and mode == "auto":
        mode = False
    if mode is False:
        return "generic", _lib.load()
    path = ensure(blob, verbose=verbose)
    if path is None:
        if mode is True:
            raise _lib.OcError("no cached specialisation for this level and hipcc is unavailable")
        return "generic", _lib.load()
    return "spec", _lib.load(path)

"""Kernel specialisation.

The generic liboc_hip.so reads a level's static tables (map bit-planes, goal sets,
item types, shaping lookup programs) from kernel arguments at run time.  For a level
that is stepped millions of times it pays to fold them into the code instead: the same
source (csrc/oc_kernels.hip) compiled with ``-DOC_SPECIALIZED`` and a generated header
that holds the level as a ``constexpr LevelHdr`` gives straight-line kernels with no
uniform branches and fully unrolled loops.  The result is an ordinary shared library
exporting the same C ABI (include/oc_hip.h).  Two flavours:

  "level" library      (``-DOC_SPEC_GEOMETRY``) folds everything, the map included: the
                       fastest code (2.82 us per step, tomato-2 x 4096 envs), for ONE map;
  "structure" library  folds what the recipes, the item multiset, the agent count and the
                       border kind fix; the map (size, tiles, positions) stays a run-time
                       argument (~4 % slower).  It serves EVERY map of that structure -- e.g. a
                       user-made map with the Salad recipe on a box that has no hipcc, which
                       the generic library would step ~1.8x slower.
(Round-3 figures; the ratios level : structure : generic were 3.07 : 3.18 :
5.55 us in round 2.)

``load_for`` picks, in this order: the cached level library; the cached structure library;
a level library compiled now (hipcc present, no profiler attached); the generic library.
Libraries are cached in csrc/_spec/ under a hash of (generated header, kernel source,
C headers, flags) -- built by ``__graft_entry__.build()`` (structure libraries for every shipped
level, level libraries for the BASELINE configs and the tested levels) and travelling with the repo
snapshot.  Measurement variants of a level library (``variant="timeline"``: -DOC_TIMELINE, bench.py
--decompose) are never picked unless asked for.  max_num_timesteps, the ALLERGIC flags
and the subtask order stay run-time arguments and select nothing.
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


def spec_header_text(blob, geometry=True) -> str:
    """The generated header for a level blob (host only, no GPU needed): the whole level
    (`geometry`, a "level" library) or its structure alone."""
    L = _lib.load()
    blob = np.ascontiguousarray(blob, dtype=np.int32)
    buf = ctypes.create_string_buffer(16384)
    n = L.oc_level_spec_source(blob.ctypes.data_as(_I32P), int(blob.size), 1 if geometry else 0, buf, len(buf))
    if n < 0:
        _lib.check(n, "oc_level_spec_source", L)
    return buf.value.decode()


def _source_digest():
    h = hashlib.sha1()
    inc = os.path.join(_build.CSRC, "..", "..", "include")
    files = [os.path.join(_build.CSRC, s) for s in list(_build.SOURCES) + list(_build.LOCAL_HEADERS)]
    files += [os.path.join(inc, f) for f in _build.HEADERS]      # what the source includes
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(_build.FLAGS).encode())
    return h


# build variants of a specialised library beside the product one (measurement only)
VARIANT_FLAGS = {"": [], "timeline": ["-DOC_TIMELINE=1"], "timeline-drain": ["-DOC_TIMELINE=2"]}    # include/oc_hip.h: oc_timeline_begin


def spec_key(blob, geometry=True, variant="") -> str:
    h = _source_digest()
    h.update(b"level" if geometry else b"structure")
    h.update(spec_header_text(blob, geometry).encode())
    if variant:
        h.update(" ".join(VARIANT_FLAGS[variant]).encode())
    return h.hexdigest()[:16]


def spec_lib_path(blob, geometry=True, variant="") -> str:
    return os.path.join(SPEC_DIR, "liboc_spec_%s.so" % spec_key(blob, geometry, variant))


def ensure(blob, verbose=False, geometry=True, compile=True, variant=""):
    """Return the path of the specialised library of this level (`geometry`: its level library,
    else its structure library), building it if needed.  Returns None when it is not cached and
    cannot / may not be compiled (`compile=False`, no hipcc, a profiler attached).
    `variant="timeline"`: the diagnostic flavour whose waves stamp the constant-rate clock
    (bench.py --decompose); never picked by `load_for` unless asked for."""
    text = spec_header_text(blob, geometry)
    key = spec_key(blob, geometry, variant)
    path = os.path.join(SPEC_DIR, "liboc_spec_%s.so" % key)
    if os.path.exists(path):
        return path
    if not compile:
        return None
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
    cmd += ["-DOC_SPECIALIZED", '-DOC_SPEC_FILE="%s"' % hdr] + (["-DOC_SPEC_GEOMETRY"] if geometry else [])
    cmd += VARIANT_FLAGS[variant]
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
    mode: True (must specialise), False (generic), 'auto' (specialise when possible),
    'structure' (the structure library: tests, measurements), 'timeline' (the level library's
    -DOC_TIMELINE flavour: measurements).  Order for True / 'auto': cached
    level library, cached structure library, a level library compiled now, generic."""
    if os.environ.get("OC_SPECIALIZE") == "0" and mode == "auto":
        mode = False
    if os.environ.get("OC_SPECIALIZE") == "structure" and mode in ("auto", True):
        mode = "structure"          # (measurements / a test-suite pass on the structure libraries)
    if mode is False:
        return "generic", _lib.load()
    if mode in ("timeline", "timeline-drain"):   # measurement flavours of the level library (bench.py --decompose)
        path = ensure(blob, verbose=verbose, geometry=True, variant=mode)
        if path is None:
            raise _lib.OcError("no cached timeline library for this level and hipcc is unavailable")
        return "spec", _lib.load(path)
    if mode == "structure":
        path = ensure(blob, verbose=verbose, geometry=False)
        if path is None:
            raise _lib.OcError("no cached structure library for this level and hipcc is unavailable")
        return "spec", _lib.load(path)
    path = (ensure(blob, geometry=True, compile=False) or ensure(blob, geometry=False, compile=False)
            or ensure(blob, verbose=verbose, geometry=True))
    if path is None:
        if mode is True:
            raise _lib.OcError("no cached specialisation for this level and hipcc is unavailable")
        return "generic", _lib.load()
    return "spec", _lib.load(path)

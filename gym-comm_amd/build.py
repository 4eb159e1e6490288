"""Build liboc_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU; the built library travels to the GPU box with the
repo snapshot.  ``-ffp-contract=off``: reward shaping must reproduce CPython's fp64
arithmetic bit for bit, so no fused multiply-adds may be formed.
"""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["oc_kernels.hip"]
HEADERS = ["oc_hip.h", "oc_level.h"]          # include/: what SOURCES include
LOCAL_HEADERS = ["oc_policy_device.h"]        # csrc/: the policy's device code, shared with POLICY_SOURCES
LIB = os.path.join(CSRC, "liboc_hip.so")
# the policy library (include/oc_policy.h): its own translation unit and shared object, so that
# the stepper's specialised builds neither contain nor depend on it
POLICY_SOURCES = ["oc_policy.hip"]
POLICY_HEADERS = ["oc_policy.h"]
POLICY_LIB = os.path.join(CSRC, "liboc_policy.so")
POLICY_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# the host-I/O library (include/oc_hostio.h): the numpy boundary's pack-for-PCIe kernel
HOSTIO_SOURCES = ["oc_hostio.hip"]
HOSTIO_HEADERS = ["oc_hostio.h"]
HOSTIO_LIB = os.path.join(CSRC, "liboc_hostio.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
         # leading scalar kernel arguments (k_multi_step) arrive in SGPRs at wave launch
         "-mllvm", "-amdgpu-kernarg-preload-count=9"]
# experiment switches (e.g. OC_HIP_EXTRA_FLAGS=-DOC_TABLES_IN_LDS); they enter the
# specialisation cache key, so variants never collide
FLAGS += [f for f in os.environ.get("OC_HIP_EXTRA_FLAGS", "").split() if f]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=)")


def needs_build(lib=LIB, sources=SOURCES, headers=HEADERS, local_headers=LOCAL_HEADERS):
    if not os.path.exists(lib):
        return True
    deps = [os.path.join(CSRC, s) for s in list(sources) + list(local_headers)]
    inc = os.path.join(CSRC, "..", "..", "include")
    deps += [os.path.join(inc, f) for f in headers]
    return os.path.getmtime(lib) < max(os.path.getmtime(d) for d in deps)


def _compile(lib, sources, flags, verbose):
    cmd = [hipcc_path(), "--offload-arch=" + ARCH] + list(flags)
    tmp = "%s.%d.tmp" % (lib, os.getpid())        # concurrent builders never share a temporary
    cmd += [os.path.join(CSRC, s) for s in sources] + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    env = {k: v for k, v in os.environ.items()
           if k != "LD_PRELOAD" and not k.startswith(("ROCP_", "ROCPROF", "ROCTRACER", "HSA_TOOLS_LIB"))}
    try:
        subprocess.check_call(cmd, env=env)
        os.replace(tmp, lib)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return lib


def build(force=False, verbose=False, extra_flags=()):
    """Compile the stepper into csrc/liboc_hip.so.  Returns the library path."""
    if not force and not needs_build():
        return LIB
    return _compile(LIB, SOURCES, FLAGS + list(extra_flags), verbose)


def build_policy(force=False, verbose=False):
    """Compile the MLP policy kernel (include/oc_policy.h) into csrc/liboc_policy.so."""
    if not force and not needs_build(POLICY_LIB, POLICY_SOURCES, POLICY_HEADERS, LOCAL_HEADERS):
        return POLICY_LIB
    return _compile(POLICY_LIB, POLICY_SOURCES, POLICY_FLAGS, verbose)


def build_hostio(force=False, verbose=False):
    """Compile the numpy boundary's pack kernel (include/oc_hostio.h) into csrc/liboc_hostio.so."""
    if not force and not needs_build(HOSTIO_LIB, HOSTIO_SOURCES, HOSTIO_HEADERS, ()):
        return HOSTIO_LIB
    return _compile(HOSTIO_LIB, HOSTIO_SOURCES, POLICY_FLAGS, verbose)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_policy(force=True, verbose=True))
    print(build_hostio(force=True, verbose=True))

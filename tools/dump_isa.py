#!/usr/bin/env python3
"""Device assembly of a level's specialised kernels + static instruction statistics per kernel.

    python tools/dump_isa.py open-divider_tomato 2 [--out /tmp/k.s] [--variant timeline]
                             [--structure] [--match k_multi_step] [--flags "-DX ..."]

Compiles csrc/oc_kernels.hip exactly as specialize.ensure() would (-S --cuda-device-only instead
of -shared) and prints, for every kernel whose demangled name matches: VGPRs / SGPRs / scratch and
the static count of VALU / SALU / VMEM / SMEM / LDS / waitcnt / branch instructions.  Static counts of a
split kernel cover all arms; per-arm figures come from the `; %bb` ranges (use --arms)."""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def classify(op):
    if op.startswith(("v_", "ds_bpermute")):
        return "VALU"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
        return "SMEM"
    if op.startswith("s_waitcnt"):
        return "WAIT"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setpc", "s_swappc")):
        return "CTRL"
    if op.startswith("s_nop"):
        return "NOP"
    if op.startswith("s_"):
        return "SALU"
    return "OTHER"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("level")
    ap.add_argument("agents", type=int)
    ap.add_argument("--out", default="/tmp/oc_isa.s")
    ap.add_argument("--variant", default="")
    ap.add_argument("--structure", action="store_true")
    ap.add_argument("--generic", action="store_true")
    ap.add_argument("--match", default="k_multi_step|k_step")
    ap.add_argument("--flags", default="")
    a = ap.parse_args()
    from gym_comm_amd import build, compiler, specialize
    lv = compiler.compile_level(a.level, a.agents, 500)
    hdr = a.out + ".spec.h"
    with open(hdr, "w") as f:
        f.write(specialize.spec_header_text(lv.blob, not a.structure))
    cmd = [build.hipcc_path(), "--offload-arch=" + build.ARCH]
    cmd += [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    if not a.generic:
        cmd += ["-DOC_SPECIALIZED", '-DOC_SPEC_FILE="%s"' % hdr] + ([] if a.structure else ["-DOC_SPEC_GEOMETRY"])
    cmd += specialize.VARIANT_FLAGS[a.variant] + a.flags.split()
    cmd += ["-S", "--cuda-device-only", "-o", a.out, os.path.join(build.CSRC, "oc_kernels.hip")]
    subprocess.run(cmd, check=True)
    text = open(a.out).read()
    # kernels: "<mangled>:" ... ".end_amdhsa_kernel" / s_endpgm region; split on .globl
    parts = re.split(r"\n\t\.globl\t(\S+)", text)
    rx = re.compile(a.match)
    for k in range(1, len(parts), 2):
        name, body = parts[k], parts[k + 1]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if not rx.search(dem):
            continue
        code = body.split(".section\t.rodata")[0]
        counts = {}
        for ln in code.splitlines():
            m = re.match(r"\t([a-z_0-9]+)", ln)
            if m and not ln.startswith("\t."):
                c = classify(m.group(1))
                counts[c] = counts.get(c, 0) + 1
        meta = {key: re.search(r"; %s: (\d+)" % key, body) for key in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy")}
        meta = {k2: (int(v.group(1)) if v else -1) for k2, v in meta.items()}
        short = re.sub(r"^void \(anonymous namespace\)::", "", dem)
        short = re.sub(r"\(.*$", "", short)
        print("%-64s vgpr %3d sgpr %3d scratch %d | %s | total %d" % (
            short, meta["NumVgprs"], meta["NumSgprs"], meta["ScratchSize"],
            " ".join("%s %d" % (c, counts.get(c, 0)) for c in ("VALU", "SALU", "VMEM", "SMEM", "LDS", "WAIT", "CTRL", "NOP")),
            sum(counts.values())))
    print("assembly:", a.out)


if __name__ == "__main__":
    main()

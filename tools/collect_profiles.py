#!/usr/bin/env python3
"""After tools/profile_round.sh <tag> (GPU box) has been merged back into gpurun_out/:
condense it into committed summaries under profiles/ and refresh profiles/traffic.json.

    python tools/collect_profiles.py <tag> <round-prefix, e.g. r01_v6>
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = {"tomato_n4096": ("open-divider_tomato", 2, 4096, "k_multi_step"),
       "salad_n32768": ("full-divider_salad", 2, 32768, "k_multi_step"),
       "tl3_n65536": ("partial-divider_tl", 3, 65536, "k_step"),
       "tomato_n131072": ("open-divider_tomato", 2, 131072, "k_multi_step"),
       "eager_tomato_n4096": ("open-divider_tomato", 2, 4096, "k_multi_step")}


def main():
    tag, pre = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    summ = os.path.join(ROOT, "tools", "summarize_profile.py")
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    # the kernel source these counters belong to: bench.py drops an entry whose digest is not the
    # current source's (a stored constant must not outlive the kernel it was measured on)
    from gym_comm_amd import specialize
    digest = specialize._source_digest().hexdigest()[:16]
    for name, (level, A, n, kern) in CFG.items():
        sdir = os.path.join(src, "stats_" + name)
        if os.path.isdir(sdir):
            subprocess.check_call([sys.executable, summ, "stats", sdir,
                                   os.path.join(ROOT, "profiles", "%s_kernel_stats_%s.md" % (pre, name))])
            bj = os.path.join(src, "stats_%s.json" % name)
            if os.path.exists(bj) and os.path.getsize(bj):
                with open(bj) as f, open(os.path.join(ROOT, "profiles", "%s_bench_under_rocprof_%s.json" % (pre, name)), "w") as o:
                    o.write(f.read())
        dirs = [os.path.join(src, "pmc_%s_%s" % (k, name)) for k in "abs"]
        if all(os.path.isdir(d) for d in dirs):
            out = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (pre, name))
            subprocess.check_call([sys.executable, summ, "pmc", kern, out] + dirs)
            res = json.load(open(out))
            h = res.get("_hbm_bytes_per_launch")
            if h:
                traffic["%s_a%d_n%d" % (level, A, n)] = {
                    "hbm_bytes_per_launch": h["total"], "fetch_corrected": h["fetch_corrected"],
                    "write": h["write"], "source": "profiles/" + os.path.basename(out), "kernel": kern,
                    "kernel_source_digest": digest}
    json.dump(traffic, open(tpath, "w"), indent=1)
    print("updated", tpath)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into small, committed
summaries under profiles/.

    python tools/summarize_profile.py stats  <rocprof_dir> <out.md>      kernel-trace --stats
    python tools/summarize_profile.py pmc    <name> <out.json> <dir>...  --pmc passes (averaged per launch)
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name, n=90):
    return name if len(name) <= n else name[:n - 3] + "..."


def stats(d, out):
    f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tr = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    extra = {}
    if tr:
        for r in csv.DictReader(open(tr[0])):
            extra.setdefault(r["Kernel_Name"], r)
    with open(out, "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats summary\n\nsource: `%s`\n\n" % d)
        o.write("| kernel | calls | total ns | avg ns | % | min ns | max ns | VGPR | SGPR | WG | grid |\n")
        o.write("|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows:
            e = extra.get(r["Name"], {})
            o.write("| `%s` | %s | %s | %.1f | %s | %s | %s | %s | %s | %s | %s |\n" % (
                short(r["Name"]), r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]),
                r["Percentage"], r["MinNs"], r["MaxNs"], e.get("VGPR_Count", ""),
                e.get("SGPR_Count", ""), e.get("Workgroup_Size_X", ""), e.get("Grid_Size_X", "")))
    print("wrote", out)


def pmc(name, out, dirs):
    res = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if name in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                res[k] = {"avg_per_launch": sum(v) / len(v), "launches": len(v)}
    if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
        # MI355X_MICROARCH.md (HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
        # reports 1/2 of the bytes of a coalesced streaming read -> doubled here
        fb = res["FETCH_SIZE"]["avg_per_launch"] * 1024 * 2
        wb = res["WRITE_SIZE"]["avg_per_launch"] * 1024
        res["_hbm_bytes_per_launch"] = {"fetch_corrected": fb, "write": wb, "total": fb + wb}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("wrote", out)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4:])

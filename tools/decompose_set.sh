#!/bin/bash
# GPU box: bench.py --decompose for BASELINE configs[1..4] (kernel-active vs launch boundary from the
# in-graph constant-rate clock stamps of the -DOC_TIMELINE build; no profiler).   tools/decompose_set.sh <tag>
# Output: gpurun_out/decomp_<tag>/*.json; copy into profiles/<prefix>_decompose_*.json.
TAG="$1"; OUT=gpurun_out/decomp_$TAG; mkdir -p $OUT
cd "$(dirname "$0")/.."
d() { name=$1; shift; python bench.py "$@" --steps 20 --warmup 5 --decompose --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.log || echo "FAILED $name";
  python - "$OUT/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
dd = r["decompose"]
print("%-16s step %.3f us = active %.3f + boundary %.3f | timeline build %.3f: active %.3f + boundary %.3f = %.3f (closure %.3f) | drain %s | wave life %.3f spread %.3f clock %.0f MHz | frac %.4f frac_active %.4f"
      % (sys.argv[2], d["ms_per_step"] * 1e3, r["kernel_active_us_product"], r["boundary_us"], dd["timeline_build_event_us_per_step"], r["kernel_active_us"],
         r["boundary_us"], dd["period_us"], r["decompose_closure"], dd["store_drain_us"], dd["wave_lifetime_us"],
         dd["wave_start_spread_us"], dd["shader_clock_mhz"], r["frac"], r["frac_kernel_active"]))
print("                 waves of a workgroup, start..end us after the launch's first wave: " + "  ".join("%.2f..%.2f" % (w["start_us"], w["end_us"]) for w in dd["by_wave_in_workgroup"]))
PY
}
d tomato_n4096
d salad_n32768 --level full-divider_salad --envs 32768
d tl3_n65536 --level partial-divider_tl --agents 3 --envs 65536
d tomato_n131072 --envs 131072

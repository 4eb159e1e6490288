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
print("%-28s step %.3f us | timeline build %.3f us: active %.3f + boundary %.3f = %.3f (closure %.3f) | issue span %.3f | frac %.4f frac_active %.4f"
      % (sys.argv[2], d["ms_per_step"] * 1e3, r["decompose"]["timeline_build_event_us_per_step"], r["kernel_active_us"],
         r["boundary_us"], r["decompose"]["period_us"], r["decompose_closure"], r["decompose"]["issue_span_us"], r["frac"], r["frac_kernel_active"]))
PY
}
d tomato_n4096
d salad_n32768 --level full-divider_salad --envs 32768
d tl3_n65536 --level partial-divider_tl --agents 3 --envs 65536
d tomato_n131072 --envs 131072

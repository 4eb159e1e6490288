#!/bin/bash
# GPU box: the set of bench.py lines profiles/ keeps per kernel revision.   tools/bench_set.sh <tag>
# Output: gpurun_out/bench_<tag>/*.json (one JSON line each); copy into profiles/<prefix>_bench_*.json.
TAG="$1"; OUT=gpurun_out/bench_$TAG; mkdir -p $OUT
cd "$(dirname "$0")/.."
b() { name=$1; shift; python bench.py "$@" > $OUT/$name.json 2> $OUT/$name.log || echo "FAILED $name"; echo "$name $(python -c "import json,sys; d=json.loads(open('$OUT/$name.json').read()); print('%.3f us/step' % (d['ms_per_step']*1e3), '%.3g' % d['value'], d['unit'], 'frac', d.get('roofline',{}).get('frac'))" 2>/dev/null)"; }
b open-divider_tomato_4096                                   # the default: BASELINE configs[1], with cpu_baseline
b tomato_4096_steps20 --steps 20 --warmup 5 --no-cpu-baseline    # the driver's form
b tomato_4096_steps20000 --steps 20000 --no-cpu-baseline
b tomato_4096_one_wave --waves-per-64 1 --no-cpu-baseline        # the launch the split replaced
b tomato_4096_decompose --steps 20 --warmup 5 --decompose --no-cpu-baseline
b full-divider_salad_32768 --level full-divider_salad --envs 32768 --no-cpu-baseline
b partial-divider_tl_65536 --level partial-divider_tl --agents 3 --envs 65536 --no-cpu-baseline
b open-divider_tomato_131072 --envs 131072 --no-cpu-baseline
for n in 8192 16384 32768 65536 262144; do b $n --envs $n --no-cpu-baseline; done
b T100 --T 100 --no-cpu-baseline
b int8_131072 --envs 131072 --obs-dtype int8 --no-cpu-baseline
b int8_4096 --obs-dtype int8 --no-cpu-baseline
b rsmall --level random-open-divider_salad_small --no-cpu-baseline
OC_SPECIALIZE=structure b structure_4096 --no-cpu-baseline
OC_SPECIALIZE=0 b generic_4096 --no-cpu-baseline
b closed_loop_4096 --mode closed-loop --no-cpu-baseline
b closed_loop_torch_4096 --mode closed-loop --policy torch --no-cpu-baseline
b closed_loop_131072 --mode closed-loop --envs 131072 --no-cpu-baseline
b closed_loop_torch_131072 --mode closed-loop --policy torch --envs 131072 --no-cpu-baseline
b gloo2_same_gpu --gpus 2 --same-gpu --backend gloo --no-cpu-baseline

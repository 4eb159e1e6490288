#!/bin/bash
# GPU box: a few bench lines for a kernel A/B (steps 20 form, no cpu baseline).  tools/ab_set.sh <tag>
TAG="$1"; OUT=gpurun_out/ab_$TAG; mkdir -p $OUT
cd "$(dirname "$0")/.."
b() { name=$1; shift; python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline > $OUT/$name.json 2> $OUT/$name.log || echo "FAILED $name"; echo "$name $(python -c "import json,sys; d=json.loads(open('$OUT/$name.json').read()); print('%.3f us/step' % (d['ms_per_step']*1e3), '%.3g' % d['value'], 'frac %.4f' % d.get('roofline',{}).get('frac'))" 2>/dev/null)"; }
b tomato_4096
b tomato_4096_w1 --waves-per-64 1
b tomato_16384 --envs 16384
b tomato_32768 --envs 32768
b tomato_32768_w4 --envs 32768 --waves-per-64 4
b tomato_40960_w4 --envs 40960 --waves-per-64 4
b tomato_40960_w1 --envs 40960 --waves-per-64 1
b tomato_49152_w4 --envs 49152 --waves-per-64 4
b tomato_49152_w1 --envs 49152 --waves-per-64 1
b tomato_65536 --envs 65536
b tomato_65536_w4 --envs 65536 --waves-per-64 4
b tomato_65536_w2 --envs 65536 --waves-per-64 2
b tomato_131072 --envs 131072
b salad_32768 --level full-divider_salad --envs 32768
b salad_32768_w4 --level full-divider_salad --envs 32768 --waves-per-64 4
b salad_32768_w1 --level full-divider_salad --envs 32768 --waves-per-64 1
b tl3_65536 --level partial-divider_tl --agents 3 --envs 65536
OC_LAUNCH=step_split=2 b tl3_65536_split2 --level partial-divider_tl --agents 3 --envs 65536
b tl3_16384 --level partial-divider_tl --agents 3 --envs 16384
OC_LAUNCH=step_split=1 b tl3_16384_split1 --level partial-divider_tl --agents 3 --envs 16384
b tl3_24576 --level partial-divider_tl --agents 3 --envs 24576
OC_LAUNCH=step_split=2 b tl3_24576_split2 --level partial-divider_tl --agents 3 --envs 24576
b tl3_32768 --level partial-divider_tl --agents 3 --envs 32768
OC_LAUNCH=step_split=2 b tl3_32768_split2 --level partial-divider_tl --agents 3 --envs 32768

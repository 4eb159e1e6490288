#!/bin/bash
# A/B of kernel build variants on the GPU box: tools/ab_bench.sh "<extra hipcc flags>" <tag>
# (the specialised level library is keyed on the flags, so a variant is compiled on first use)
set -e
FLAGS="$1"; TAG="$2"
for cfg in "open-divider_tomato 2 4096" "full-divider_salad 2 32768" "partial-divider_tl 3 65536" "open-divider_tomato 2 131072"; do
  set -- $cfg
  for od in int32 int8; do
    if [ "$2" = 3 ] && [ "$od" = int8 ]; then continue; fi
    OC_HIP_EXTRA_FLAGS="$FLAGS" python bench.py --level $1 --agents $2 --envs $3 --obs-dtype $od --no-cpu-baseline --steps 10240 \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$TAG', '$1', $2, $3, '$od', '%.2f us  %.3e env-steps/s  frac %.3f' % (d['ms_per_step']*1e3, d['value'], d['roofline']['frac']))"
  done
done

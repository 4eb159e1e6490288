#!/bin/bash
# launch-geometry sweeps on the GPU box (tuning aid, not product): store policy, block size.  Every line is one bench.py run of tomato-2 (10 240 steps, hipGraph).
one() {  # level agents envs obs  (env overrides come from the caller)
  python bench.py --level $1 --agents $2 --envs $3 --obs-dtype $4 --no-cpu-baseline --steps 10240 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('WT=${OC_LAUNCH:-auto} ', '$1', $2, $3, '$4', '%.2f us  %.3e env-steps/s  frac %.3f' % (d['ms_per_step']*1e3, d['value'], d['roofline']['frac']))"
}
for n in 4096 8192 16384 32768; do
  for wt in 0 1; do OC_LAUNCH=wt=$wt one open-divider_tomato 2 $n int32; done
done
one full-divider_salad 2 32768 int32
for n in 65536 131072 262144 524288; do
  for b in 64 128 256; do OC_LAUNCH=block=$b one open-divider_tomato 2 $n int32; done
done
for b in 64 128; do OC_LAUNCH=block=$b one partial-divider_tl 3 65536 int32; done

#!/bin/bash
# GPU box: write-back (OC_LAUNCH=wt=0) against write-through (=1) stores at small batch sizes
# -- what the launcher's store policy (write_through() in csrc/oc_kernels.hip) rests on.
for n in 64 512 2048 4096; do for wt in 0 1; do OC_LAUNCH=wt=$wt python bench.py --no-cpu-baseline --envs $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('WT', $wt, 'n', $n, 'tomato', '%.3f us' % (d['ms_per_step']*1e3))"; done; done
for wt in 0 1; do OC_LAUNCH=wt=$wt python bench.py --no-cpu-baseline --envs 4096 --level full-divider_salad 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('WT', $wt, 'salad 4096', '%.3f us' % (d['ms_per_step']*1e3))"; OC_LAUNCH=wt=$wt python bench.py --no-cpu-baseline --envs 4096 --level partial-divider_tl --agents 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('WT', $wt, 'tl3 4096', '%.3f us' % (d['ms_per_step']*1e3))"; done

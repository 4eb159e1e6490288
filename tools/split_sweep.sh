#!/bin/bash
# GPU box: the fused step as ONE wave per 64 envs (--waves-per-64 1) against the split launch
# (4 waves per 64 envs, see multi_step_body in csrc/oc_kernels.hip) over the batch size --
# what the launcher's split policy (split_for()) rests on.
LEVEL=${1:-open-divider_tomato}
for n in 64 1024 4096 8192 16384 32768 65536 131072; do for sp in 1 4; do python bench.py --no-cpu-baseline --waves-per-64 $sp --envs $n --level $LEVEL 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('split', $sp, 'n', $n, '$LEVEL', '%.3f us' % (d['ms_per_step']*1e3), d['config']['kernel_flavour'], d['config']['waves_per_64_envs'])"; done; done

#!/bin/bash
# GPU box: the base step (oc_step, 3 agents) as one wave per 64 envs against the two-wave split
# launch (state wave + shaping wave) over the batch size -- what step_split_for() rests on.
for n in 1024 4096 8192 16384 32768 65536; do for sp in 1 2; do OC_LAUNCH=step_split=$sp python bench.py --no-cpu-baseline --envs $n --level partial-divider_tl --agents 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('step split', $sp, 'n', $n, 'tl-3', '%.3f us' % (d['ms_per_step']*1e3), d['config']['kernel_flavour'])"; done; done

#!/bin/bash
# GPU box: rocprofv3 evidence for one kernel revision.   tools/profile_round.sh <tag>
#   kernel-trace --stats of the bench command (per BASELINE config) and three separate --pmc
#   passes (FETCH_SIZE | WRITE_SIZE | SQ instruction counters), as MI355X_MICROARCH.md's
#   HBM / PMC-slot sections prescribe (FETCH_SIZE and WRITE_SIZE do not fit one pass;
#   counters never share a run with a trace).  Output: gpurun_out/prof_<tag>/...
set -e
TAG="$1"; OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
# every library the runs below load is built NOW, outside the profiler: under rocprofv3 the
# package refuses to JIT-compile a missing specialisation (gym-comm_amd/specialize.py)
python3 -c 'import __graft_entry__ as g; g.build()' > $OUT/build.log 2>&1
cfgs=("tomato_n4096 --level open-divider_tomato --agents 2 --envs 4096"
      "salad_n32768 --level full-divider_salad --agents 2 --envs 32768"
      "tl3_n65536 --level partial-divider_tl --agents 3 --envs 65536"
      "tomato_n131072 --level open-divider_tomato --agents 2 --envs 131072")
for c in "${cfgs[@]}"; do
  set -- $c; name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$name -o s -- python3 bench.py "$@" --steps 8192 --warmup 512 --reps 3 --no-cpu-baseline > $OUT/stats_$name.json 2> $OUT/stats_$name.log
  echo "stats $name done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_a_$name -o a -- python3 bench.py "$@" --steps 200 --warmup 20 --reps 1 --no-graph --no-cpu-baseline > /dev/null 2> $OUT/pmc_a_$name.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_b_$name -o b -- python3 bench.py "$@" --steps 200 --warmup 20 --reps 1 --no-graph --no-cpu-baseline > /dev/null 2> $OUT/pmc_b_$name.log
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_s_$name -o s -- python3 bench.py "$@" --steps 200 --warmup 20 --reps 1 --no-graph --no-cpu-baseline > /dev/null 2> $OUT/pmc_s_$name.log
  echo "pmc $name done"
done
# the 4096-env kernel once more with eager launches (no graph): back-to-back graph nodes
# under the profiler's per-dispatch signals read ~1 us longer than the kernel itself
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_eager_tomato_n4096 -o s -- python3 bench.py --no-graph --steps 4096 --warmup 256 --reps 3 --no-cpu-baseline > $OUT/stats_eager_tomato_n4096.json 2> $OUT/stats_eager_tomato_n4096.log
# keep only small files (the merge back is capped at 64 MiB for the WHOLE gpurun_out/): the
# per-dispatch trace is cut to its head (the summariser only reads one row per kernel from it)
find $OUT -name "*.db" -delete 2>/dev/null || true
for f in $(find $OUT -name "*kernel_trace.csv"); do head -n 300 "$f" > "$f.head" && mv "$f.head" "$f"; done
du -sh $OUT

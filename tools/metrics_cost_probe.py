#!/usr/bin/env python3
"""Does the per-wave metrics accumulation (six no-return L2 atomics per wave, the only stores of the
step that are not write-through) cost anything at the launch boundary?  The bench's own workload and
graphs, once as shipped and once with the metrics pointer withheld (oc_multi_step / oc_step accept
NULL: no counters).  GPU box only; prints us per step for both, alternating, three rounds."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench


def timed(blocks, stream, steps, reps=9):
    out = []
    with torch.cuda.stream(stream):
        blocks.run(steps)
        stream.synchronize()
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            blocks.run(steps)
            e1.record(stream)
            e1.synchronize()
            out.append(e0.elapsed_time(e1) * 1e3 / steps)
    return sorted(out)[len(out) // 2]


def main():
    argv = sys.argv[1:]
    args = bench.parse(argv)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream()
    variants = {}
    for name in ("metrics", "no-metrics"):
        env, step_fn, _ = bench.open_loop_workload(args, dev, 1)
        if name == "no-metrics":
            saved = env.metrics
            env.metrics = None            # the launchers pass NULL
            env._ms_args = None
        with torch.cuda.stream(stream):
            for k in range(8):
                step_fn(k)
            stream.synchronize()
            blocks = bench.StepBlocks(step_fn, stream, 240, True)
            blocks.prepare([2400])
        variants[name] = (env, blocks)
    for r in range(3):
        print("  ".join("%s %.3f us/step" % (k, timed(v[1], stream, 2400)) for k, v in variants.items()), flush=True)


if __name__ == "__main__":
    main()

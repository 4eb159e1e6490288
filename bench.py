#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched Overcooked stepper (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: the fused ``oc_multi_step`` launch
(action decode -> collisions -> interact -> done/reward -> fp64 shaping -> auto-reset ->
both observations) for 2-agent levels, ``oc_step`` for 3+ agents (the reference has no
observation encoder for them).  Inputs (state, a 256-step window of pre-generated
actions) are resident in HBM before the timed region.  Envs are independent, so N GPUs =
N shards of the same per-GPU batch with no data-path collective ("weak" scaling); RCCL
only all-gathers the 64-byte metrics vector at the end of the rollout.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
WINDOW = 256                # pre-generated action steps, cycled


def algorithmic_bytes_per_env_step(A, M, S, C, with_obs):
    """4 B x (2*W_state + A + W_out + W_obs), int32 SoA accounting with
    W_state = 3A + 5M + 1 + 2S, W_out = 6, W_obs = 2*(23 + S + 2C)."""
    w_state = 3 * A + 5 * M + 1 + 2 * S
    w_obs = 2 * (23 + S + 2 * C) if with_obs else 0
    return 4 * (2 * w_state + A + 6 + w_obs)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20000)
    p.add_argument("--warmup", type=int, default=512)
    p.add_argument("--level", default="open-divider_tomato")
    p.add_argument("--agents", type=int, default=2)
    p.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    p.add_argument("--T", type=int, default=500, help="max_num_timesteps (README.md:49)")
    p.add_argument("--comm", type=int, default=2)
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    p.add_argument("--graph-steps", type=int, default=WINDOW)
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse N>1 on one GPU)")
    p.add_argument("--same-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    p.add_argument("--obs-dtype", default="int32", choices=["int32", "int8", "float32"],
                   help="observation rows: int32 (SURVEY 8(d) accounting), int8 (4x fewer bytes) or float32")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    return p.parse_args()


def host_threads():
    """Threads for the CPU baseline: the cores this process may run on, capped at the
    GPU box's per-GPU CPU share (16)."""
    if os.environ.get("OC_CPU_THREADS"):
        return max(1, int(os.environ["OC_CPU_THREADS"]))
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))


def cpu_baseline(level_blob, A, C, wrapper, seconds):
    """The CPU oracle (oracle/oc_oracle.c, the bit-exact restatement of the reference's
    step()+obs) timed on this box's host cores on a bounded sample of the same workload.
    Reported baseline only -- never part of the product path."""
    from oracle import oracle
    cores = host_threads()
    n = 4096
    chunk = 64                                   # steps per C call (one call per thread)
    rng = np.random.default_rng(1234)
    ora = oracle.OracleBatch(level_blob, n, threads=cores)
    if wrapper:
        acts = np.stack([rng.integers(0, 4, (chunk, n)), rng.integers(0, C, (chunk, n)),
                         rng.integers(0, 4, (chunk, n)), rng.integers(0, C, (chunk, n))],
                        axis=1).astype(np.int32)
        comm = np.zeros((2, n), np.int32)
    else:
        acts = rng.integers(0, 4, (chunk, A, n)).astype(np.int32)
    steps = 0
    t0 = time.perf_counter()
    while True:
        if wrapper:
            ora.multi_rollout(acts, comm, 2, 0, C, auto_reset=True)
        else:
            ora.rollout(acts, auto_reset=True)
        steps += chunk
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    # the same loop on ONE host thread (SURVEY 8(d) asks for both), a quarter of the time budget
    ora1 = oracle.OracleBatch(level_blob, n, threads=1)
    steps1 = 0
    t1 = time.perf_counter()
    while True:
        if wrapper:
            ora1.multi_rollout(acts, comm, 2, 0, C, auto_reset=True)
        else:
            ora1.rollout(acts, auto_reset=True)
        steps1 += chunk
        el1 = time.perf_counter() - t1
        if el1 >= seconds / 4:
            break
    return {"value": n * steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "value_1thread": n * steps1 / el1,
            "sample": "%d envs x %d steps of the same workload (%.1f s wall), C oracle "
                      "(oracle/oc_oracle.c), %d threads; 1 thread: %d steps in %.1f s"
                      % (n, steps, el, cores, steps1, el1)}


def main():
    args = parse()
    # native libraries (RCCL's version banner, gloo's rank chatter) print to fd 1; keep
    # stdout clean for the ONE JSON line by pointing fd 1 at stderr until we print it
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(args.backend)

    from gym_comm_amd.batched import BatchedOvercooked
    wrapper = args.agents == 2
    n = args.envs
    env = BatchedOvercooked(args.level, num_agents=args.agents, num_envs=n,
                            max_num_timesteps=args.T, num_communication=args.comm,
                            communication_on=True, fow_radius=2, device=dev, auto_reset=True,
                            obs_dtype=getattr(torch, args.obs_dtype))
    lv = env.level
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    if wrapper:
        hi = torch.tensor([4, args.comm, 4, args.comm], device=dev).view(1, 4, 1)
        acts = (torch.rand((WINDOW, 4, n), generator=gen, device=dev) * hi).to(torch.int32).contiguous()
        step_fn = env.multi_step
    else:
        acts = torch.randint(0, 4, (WINDOW, args.agents, n), generator=gen, device=dev,
                             dtype=torch.int32)
        step_fn = env.step
    acts = [acts[k].contiguous() for k in range(WINDOW)]

    stream = torch.cuda.Stream(device=dev)
    use_graph = not args.no_graph
    G = max(1, min(args.graph_steps, WINDOW))
    graph = None
    with torch.cuda.stream(stream):
        for k in range(max(1, min(args.warmup, 64))):    # touch everything before capture (module load,
            step_fn(acts[k % WINDOW])                    # first-launch work): at least once, also for --warmup 0
        stream.synchronize()
        tails = {}                                       # remainder length -> its own graph

        def capture(length):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for k in range(length):
                    step_fn(acts[k % WINDOW])
            return g

        if use_graph:
            graph = capture(G)
            for cnt in (args.warmup, args.steps):        # a short --steps still runs as ONE graph launch
                if cnt % G:
                    tails.setdefault(cnt % G, capture(cnt % G))

        def run(steps):
            k = 0
            if graph is not None:
                while steps - k >= G:
                    graph.replay()
                    k += G
                if steps - k in tails:
                    tails[steps - k].replay()
                    k = steps
            while k < steps:
                step_fn(acts[k % WINDOW])
                k += 1

        run(args.warmup)
        stream.synchronize()
        if env.metrics is not None:
            env.metrics.zero_()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        run(args.steps)
        ev1.record(stream)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        ev_ms = ev0.elapsed_time(ev1)
        rollout_metrics = env.metrics_vector()          # before the probe launches below

        # kernel duration for the roofline: HIP events on the launch stream around
        # individual launches (no graph), averaged
        probe = 200
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(probe)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(probe)]
        for k in range(probe):
            e0[k].record(stream)
            step_fn(acts[k % WINDOW])
            e1[k].record(stream)
        stream.synchronize()
        per_launch_ms = sorted(a.elapsed_time(b) for a, b in zip(e0, e1))
        launch_ms_bracketed = float(np.mean(per_launch_ms[probe // 10: probe - probe // 10]))

    # end-of-rollout metrics: the only collective on the path (RCCL all-gather, 64 B/rank)
    from gym_comm_amd import dist as ocdist
    if dist is not None and args.backend != "nccl":
        rollout_metrics = rollout_metrics.cpu()          # gloo rehearsal: gather on host tensors
    g = ocdist.gather_rollout_metrics(rollout_metrics, elapsed)
    elapsed = g["elapsed_s"]
    m = [g["total"][k] for k in ocdist.METRIC_NAMES]

    if rank == 0:
        total_env_steps = n * world * args.steps
        value = total_env_steps / elapsed
        bytes_per = algorithmic_bytes_per_env_step(lv.num_agents, lv.num_items, lv.num_subtasks,
                                                   args.comm, wrapper)
        launch_s = (ev_ms / 1e3) / args.steps          # avg launch duration over the timed region
        achieved = bytes_per * n / launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                key = "%s_a%d_n%d" % (args.level, args.agents, n)
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec (whole node)", "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s, %d agents, %d parallel envs per GPU, T=%d, C=%d, %s"
                                   % (args.level, args.agents, n, args.T, args.comm,
                                      "fused multi_step (step+obs)" if wrapper else "step only"),
                       "level": args.level, "num_agents": args.agents, "envs_per_gpu": n,
                       "max_num_timesteps": args.T, "launch": "hipgraph" if graph is not None else "eager",
                       "obs_dtype": args.obs_dtype,
                       "parallelism": "env-sharded x%d" % world},
            "agent_steps_per_sec": value * lv.num_agents,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         # the same with the bytes the PMC counters saw instead of SURVEY's int32-per-
                         # field accounting (the packed state moves fewer): what the memory system did
                         "traffic_gbps": (traffic / launch_s / 1e9) if traffic else None,
                         "traffic_frac": (traffic / launch_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "kernel": "k_multi_step" if wrapper else "k_step",
                         "algorithmic_bytes_per_env_step": bytes_per,
                         "avg_launch_us_timed_region": launch_s * 1e6,
                         "avg_launch_us_event_bracketed": launch_ms_bracketed * 1e3},
            "rollout_metrics": {"env_steps": m[0], "episodes": m[1], "successes": m[2],
                                "reward_sum": m[3], "completed_subtasks_sum": m[4], "errors": m[5]},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(lv.blob, lv.num_agents, args.comm, wrapper,
                                               args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

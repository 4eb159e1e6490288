#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched Overcooked stepper (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: the fused ``oc_multi_step`` launch
(action decode -> collisions -> interact -> done/reward -> fp64 shaping -> auto-reset ->
both observations) for 2-agent levels, ``oc_step`` for 3+ agents (the reference has no
observation encoder for them).  Inputs (state, a 256-step window of pre-generated
actions) are resident in HBM before the timed region.  Envs are independent, so N GPUs =
N shards of the same per-GPU batch with no data-path collective ("weak" scaling); RCCL
only all-gathers the 64-byte metrics vector at the end of the rollout.

``--gpus N`` (N > 1) started WITHOUT torchrun (no RANK in the environment) starts the N
ranks itself -- fresh child processes through ``torch.distributed.run``, before this
process has touched a GPU -- and relays rank 0's line; it never reports a 1-GPU run for an
N-GPU request.

Timing: blocks of EXACTLY ``--steps`` (K) steps are run back to back -- ``reps`` of them,
enough for >= 50 ms of GPU time (``--reps`` overrides) -- inside ONE region bracketed by
barrier + synchronize on both sides.  Blocks are hipGraph replays; a K shorter than the
256-step graph window shares one replay with the next blocks (``blocks_per_replay``), so the
~10 us between two graph launches is not charged to 20 steps.  A HIP event on the launch
stream separates the replays; ``ms_per_step`` is the MEDIAN block time / K (MAX over ranks):
a short ``--steps`` gives the steady-state number, not one launch + sync latency.  The wall
clock over the whole region is reported beside it (``ms_per_step_wall``; for N > 1 it is the
value).  The host POLLS the last event instead of blocking on it: a blocked wait on this box
sometimes returns 50-80 ms after the device has finished (tools/slow_stretch_probe4.py) -- what
round 2 took for a "slow stretch" of the GPU and hid behind 150 ms of untimed replays.  A short
untimed settle (``--settle-ms``, 30 ms: the first replays of a freshly instantiated graph) remains.

``--decompose`` replays the same graph on the TIMELINE build of the level's library (every wave
stamps the chip-wide 100 MHz clock; include/oc_hip.h: oc_timeline_begin) and adds
``roofline.kernel_active_us`` / ``boundary_us`` / ``frac_kernel_active`` beside the unchanged
headline: the step split into the span with waves on the chip and the launch boundary, no profiler.

``--mode closed-loop`` measures the same kernel with policies in the loop (an ego and a
partner MLP on the observations, actions sampled on the device, episode statistics; one
hipGraph per step block) through ``OvercookedVecEnv``; ``--policy fused`` (default) evaluates both
MLPs in one launch of the hand-written MFMA kernel (include/oc_policy.h), ``--policy torch`` as
torch modules; it is reported with its own metric name, never instead of the headline.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
WINDOW = 256                # pre-generated action steps, cycled
MIN_TIMED_MS = 50.0         # GPU time the timed region must cover (reps is derived from it)
SETTLE_MS = 30.0         # untimed replays before calibration and timing (--settle-ms; see main)
ELEM = {"int32": 4, "int8": 1, "float32": 4}


def survey_bytes_per_env_step(A, M, S, C, with_obs, obs_elem=4):
    """4 B x (2*W_state + A + W_out) + elem x W_obs, int32-per-FIELD accounting
    with W_state = 3A + 5M + 1 + 2S, W_out = 6, W_obs = 2*(23 + S + 2C).  The packed layout
    moves fewer bytes than this (see layout_bytes_per_env_step), so a rate formed with it is a
    labelled secondary figure, not the roofline fraction."""
    w_state = 3 * A + 5 * M + 1 + 2 * S
    w_obs = 2 * (23 + S + 2 * C) if with_obs else 0
    return 4 * (2 * w_state + A + 6) + obs_elem * w_obs


def layout_bytes_per_env_step(A, M, S, C, wrapper, obs_elem=4, metrics=True):
    """Bytes one env-step really moves with this repo's layout (include/oc_hip.h):
    read  : A+M+2 packed state words, the action rows;
    write : the state words, and for the fused wrapper step the 2 comm words, 2 viewers x
            (22+S+2C) observation rows, timestep f64, shaped reward f64, done, sparse reward;
            for the base step reward, done, 2 x f64 shaping;
    plus six 8-byte no-return atomics per 64 envs for the metrics.
    Returns (read, written)."""
    state = 4 * (A + M + 2)
    if wrapper:
        read = state + 4 * 4
        written = state + 2 * 4 + 2 * (22 + S + 2 * C) * obs_elem + 8 + 8 + 4 + 4
    else:
        read = state + 4 * A
        written = state + 4 + 4 + 2 * 8
    if metrics:
        written += 6 * 8 / 64.0
    return read, written


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2048)
    p.add_argument("--warmup", type=int, default=256)
    p.add_argument("--reps", type=int, default=0,
                   help="timed repetitions of the --steps block (0 = enough for %.0f ms of GPU time)" % MIN_TIMED_MS)
    p.add_argument("--mode", default="open-loop", choices=["open-loop", "closed-loop"])
    p.add_argument("--level", default="open-divider_tomato")
    p.add_argument("--agents", type=int, default=2)
    p.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    p.add_argument("--T", type=int, default=500, help="max_num_timesteps (README.md:49)")
    p.add_argument("--comm", type=int, default=2)
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    p.add_argument("--graph-steps", type=int, default=WINDOW)
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse N>1 on one GPU)")
    p.add_argument("--same-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    p.add_argument("--obs-dtype", default="int32", choices=sorted(ELEM),
                   help="observation rows: int32 (SURVEY 8(d) accounting), int8 (4x fewer bytes) or float32")
    p.add_argument("--hidden", type=int, default=64, help="closed-loop: width of the two policy MLPs")
    p.add_argument("--policy", default="fused", choices=["fused", "torch"],
                   help="closed-loop: the two MLP policies as ONE launch of the hand-written MFMA kernel "
                        "(include/oc_policy.h) or as torch modules (~12 launches each)")
    p.add_argument("--closed-loop-launches", default="auto", choices=["auto", "1", "2"],
                   help="closed-loop, --policy fused: launches per step -- 1 = the step kernel evaluates both "
                        "policies itself (oc_step_opts.policy), 2 = policy kernel + step; auto = 1 where possible")
    p.add_argument("--waves-per-64", type=int, default=0, choices=[0, 1, 2, 4],
                   help="launch hint of the fused step: 0 = the library decides (four waves per 64 envs up to 24576 envs, two up to 32768), "
                        "1 = one wave per 64 envs, 2 / 4 = split launches (include/oc_hip.h, oc_step_opts)")
    p.add_argument("--decompose", action="store_true",
                   help="also replay the same graph on the TIMELINE build of the level's library (every wave "
                        "stamps the chip-wide 100 MHz clock; include/oc_hip.h: oc_timeline_begin) and report "
                        "roofline.kernel_active_us / boundary_us beside the unchanged headline")
    p.add_argument("--settle-ms", type=float, default=SETTLE_MS,
                   help="untimed graph replays before the calibration and the timed region")
    p.add_argument("--seed", type=int, default=1234)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    return p.parse_args(argv)


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as fresh children (this process has not
    touched a GPU), one per LOCAL_RANK, and let rank 0's JSON line through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "2")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    if proc.returncode != 0 or not lines:
        sys.stderr.write("bench.py: the %d-rank run failed (exit %d)\n" % (args.gpus, proc.returncode))
        return proc.returncode or 1
    out = json.loads(lines[-1])
    if out.get("n_gpus") != args.gpus:
        sys.stderr.write("bench.py: asked for %d ranks, the run reports %r\n" % (args.gpus, out.get("n_gpus")))
        return 1
    print(lines[-1], flush=True)
    return 0


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


def open_loop_workload(args, dev, seed, actions=None, specialize_level="auto"):
    """The open-loop workload exactly as the timed region runs it: the batch (the library's own
    launch choice unless --waves-per-64 says otherwise), a WINDOW-step window of pre-generated
    actions resident in HBM, and ``step_fn(k)`` = one launch of the hot path on window step
    ``k % WINDOW``.  `actions` (int32 [WINDOW][4 or A][n], on `dev`) replaces the seeded random
    window: how tests/test_timed_path_gpu.py replays THIS path against the oracle."""
    from gym_comm_amd.batched import BatchedOvercooked
    n = args.envs
    wrapper = args.agents == 2
    env = BatchedOvercooked(args.level, num_agents=args.agents, num_envs=n,
                            max_num_timesteps=args.T, num_communication=args.comm,
                            communication_on=True, fow_radius=2, device=dev, auto_reset=True,
                            obs_dtype=getattr(torch, args.obs_dtype), seed=seed,
                            waves_per_64=args.waves_per_64, specialize_level=specialize_level)
    rows = 4 if wrapper else args.agents
    if actions is None:
        gen = torch.Generator(device=dev).manual_seed(seed)
        if wrapper:
            hi = torch.tensor([4, args.comm, 4, args.comm], device=dev).view(1, 4, 1)
            actions = (torch.rand((WINDOW, 4, n), generator=gen, device=dev) * hi).to(torch.int32).contiguous()
        else:
            actions = torch.randint(0, 4, (WINDOW, args.agents, n), generator=gen, device=dev,
                                    dtype=torch.int32)
    elif tuple(actions.shape) != (WINDOW, rows, n) or actions.dtype != torch.int32:
        raise ValueError("actions must be int32 [%d][%d][%d]" % (WINDOW, rows, n))
    fn = env.multi_step if wrapper else env.step
    acts = [actions[k].contiguous() for k in range(WINDOW)]
    torch.cuda.synchronize(dev)          # reset + window are in HBM before any other stream steps
    return env, (lambda k: fn(acts[k % WINDOW])), acts


def reference_python_rate(level, agents, wrapper):
    """The reference's OWN Python rate for this workload: a recorded figure (the reference cannot
    travel to the GPU box), written by tests/golden/time_reference.py in the build container into
    profiles/reference_cpu_rate.json -- read like profiles/traffic.json, labelled as stored."""
    path = os.path.join(ROOT, "profiles", "reference_cpu_rate.json")
    try:
        with open(path) as f:
            rj = json.load(f)
        ent = rj["configs"]["%s_a%d_%s" % (level, agents, "wrapper" if wrapper else "base")]
        best = max(ent["runs"], key=lambda r: r["procs"])
        one = min(ent["runs"], key=lambda r: r["procs"])
        return {"value": best["env_steps_per_s"], "unit": "env-steps/s", "cores": best["procs"],
                "value_1core": one["env_steps_per_s"] / one["procs"],
                "hardware": "%s (build container, %d cores)" % (rj["hardware"], rj["cores_available"]),
                "call": ent["call"],
                "source": "profiles/reference_cpu_rate.json (recorded %s by %s; not measured in this run)"
                          % (rj.get("recorded"), rj.get("script"))}
    except (OSError, KeyError, ValueError):
        return None


def timeline_reduce(rec):
    """Per-launch reduction of a timeline record tensor (include/oc_hip.h: oc_timeline_begin), int32
    [launches][stride][4] = {start lo, start hi, issue-span | drain-span << 16, shader cycles} per
    wave, 0xFF-filled where no wave wrote.  Returns int64 tensors: wrote [L][stride] (bool), start
    [L][stride], and per wave issue-end, drain-end (absolute ticks) and shader cycles."""
    r = rec.to(torch.int64) & 0xFFFFFFFF
    start = r[:, :, 0] | (r[:, :, 1] << 32)
    wrote = ~((r[:, :, 0] == 0xFFFFFFFF) & (r[:, :, 1] == 0xFFFFFFFF))
    issue = start + (r[:, :, 2] & 0xFFFF)
    drain = start + (r[:, :, 2] >> 16)
    return wrote, start, issue, drain, r[:, :, 3]


def _timeline_run(args, dev, seed, stream, steps, variant, min_ms):
    """One timeline flavour of the level's library: capture `steps` launches with a record each,
    replay for >= min_ms of GPU time; returns per-launch arrays (10 ns ticks) and per-wave means."""
    import ctypes
    env, step_fn, _ = open_loop_workload(args, dev, seed, specialize_level=variant)
    L = env._L
    n = args.envs
    BIG = torch.iinfo(torch.int64).max
    stride = 4 * ((n + 63) // 64)                     # an upper bound of the waves of one launch
    rec = torch.zeros((steps, stride, 4), dtype=torch.int32, device=dev)
    use_drain = variant == "timeline-drain"
    out = {k: [] for k in ("start", "end", "issue", "life", "spread", "mhz", "ev_us")}
    with torch.cuda.stream(stream):
        for k in range(8):
            step_fn(k)
        stream.synchronize()
        rc = L.oc_timeline_begin(ctypes.c_void_p(rec.data_ptr()), steps, stride)
        if rc:
            raise SystemExit("oc_timeline_begin failed: %s" % L.oc_last_error().decode())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for k in range(steps):
                step_fn(k)
        L.oc_timeline_begin(None, 0, 0)
        t_end = time.perf_counter() + SETTLE_MS / 1e3
        while time.perf_counter() < t_end:
            g.replay()
            stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_end = time.perf_counter() + 10.0
        gpu_ms = 0.0
        while gpu_ms < min_ms and time.perf_counter() < t_end:
            rec.fill_(-1)
            stream.synchronize()
            e0.record(stream)
            g.replay()
            e1.record(stream)
            stream.synchronize()
            wrote, st, issue, drain, cycles = timeline_reduce(rec)
            end_w = drain if use_drain else issue
            start = torch.where(wrote, st, torch.full_like(st, BIG)).min(dim=1).values   # per launch
            end = torch.where(wrote, end_w, torch.zeros_like(st)).max(dim=1).values
            out["start"].append(start.cpu().numpy())
            out["end"].append(end.cpu().numpy())
            out["issue"].append(torch.where(wrote, issue, torch.zeros_like(st)).max(dim=1).values.cpu().numpy())
            span = ((end_w - st) * wrote).sum().double()
            out["life"].append(float(span.item()) / max(1, int(wrote.sum().item())))
            out["mhz"].append(100.0 * float((cycles * wrote).sum().item()) / max(1.0, float(span.item())))
            out["spread"].append(float((torch.where(wrote, st, start.unsqueeze(1)).max(dim=1).values - start)
                                       .double().mean().item()))
            ms = e0.elapsed_time(e1)
            gpu_ms += ms
            out["ev_us"].append(ms / steps * 1e3)
        out["waves"] = int(wrote[0].sum().item())
        # by position of the wave in its workgroup (the last replay): mean start / end offset from the
        # launch's first wave start -- which waves start late, which end last
        wpw = max(1, out["waves"] // max(1, (n + 63) // 64))
        out["by_wave_in_workgroup"] = []
        for j in range(wpw):
            sl = slice(j, out["waves"], wpw)
            out["by_wave_in_workgroup"].append({
                "start_us": float((st[:, sl] - start.unsqueeze(1)).double().mean().item()) * 0.01,
                "end_us": float((end_w[:, sl] - start.unsqueeze(1)).double().mean().item()) * 0.01})
    return out


def decompose_step(args, dev, seed, stream, steps, min_ms=60.0):
    """Split a step of the chained-launch graph into KERNEL-ACTIVE time and LAUNCH BOUNDARY without a
    profiler: the same workload on the TIMELINE build of the level's library (-DOC_TIMELINE=1: the
    product source + two reads of the chip-wide constant-rate clock and ONE 16-byte write-through
    store by one lane per wave), one `steps`-launch hipGraph whose launches each have their own record
    [waves][4 x 32 bits]; reduced over the waves.  Per launch k:
        active_k   = max issue-end_k - min start_k        waves of launch k on the chip
        boundary_k = min start_{k+1} - max issue-end_k     store drain, end-of-kernel cache work, the
                                                           command processor, the next dispatch
        period_k   = active_k + boundary_k                 the step, as the chip saw it
    A second pass on the -DOC_TIMELINE=2 flavour (every wave also WAITS for its stores and stamps
    that) says how much of the boundary is store drain (`store_drain_us`; that flavour is slower:
    its waves outlive their stores).  The clock ticks every 10 ns, so single values are quantised;
    means over >= 10^4 launches are not.  Times in us; the timeline build's own event-timed step is
    reported too, so its perturbation is visible (`decompose_closure` in the bench line = period /
    the product build's ms_per_step)."""
    tick_us = 0.01                                     # s_memrealtime: 100 MHz (MI355X_MICROARCH.md)
    r = _timeline_run(args, dev, seed, stream, steps, "timeline", min_ms)
    cat = lambda xs: np.concatenate(xs).astype(np.float64) * tick_us
    s_, e_ = np.stack(r["start"]), np.stack(r["end"])
    act = ((e_ - s_).astype(np.float64) * tick_us).ravel()
    gap = ((s_[:, 1:] - e_[:, :-1]).astype(np.float64) * tick_us).ravel()
    per = ((s_[:, 1:] - s_[:, :-1]).astype(np.float64) * tick_us).ravel()
    stat = lambda x: {"mean": float(x.mean()), "median": float(np.median(x)), "p95": float(np.percentile(x, 95))}
    out = {"kernel_active_us": float(act.mean()), "boundary_us": float(gap.mean()), "period_us": float(per.mean()),
           "kernel_active": stat(act), "boundary": stat(gap), "period": stat(per),
           "wave_lifetime_us": float(np.mean(r["life"])) * tick_us,
           "wave_start_spread_us": float(np.mean(r["spread"])) * tick_us,
           "shader_clock_mhz": float(np.mean(r["mhz"])),   # s_memtime cycles per 10 ns realtime tick x 100
           "launches_sampled": int(act.size), "waves_per_launch": r["waves"],
           "by_wave_in_workgroup": r["by_wave_in_workgroup"],
           "timeline_build_event_us_per_step": float(np.median(r["ev_us"])),
           "method": "in-graph s_memrealtime stamps (100 MHz, chip-wide) on the -DOC_TIMELINE build of the "
                     "level library: per launch min wave start / max wave end; no profiler attached"}
    try:
        d = _timeline_run(args, dev, seed, stream, steps, "timeline-drain", min_ms / 2)
        ds, de, di = np.stack(d["start"]), np.stack(d["end"]), np.stack(d["issue"])
        out["store_drain_us"] = float(((de - di).astype(np.float64) * tick_us).mean())
        out["drain_build"] = {"kernel_active_to_drain_us": float(((de - ds) * tick_us).mean()),
                              "boundary_after_drain_us": float(((ds[:, 1:] - de[:, :-1]) * tick_us).mean()),
                              "event_us_per_step": float(np.median(d["ev_us"]))}
    except Exception as exc:                           # the drain flavour is optional evidence
        out["store_drain_us"] = None
        out["drain_build"] = {"unavailable": str(exc)[:200]}
    return out


class StepBlocks:
    """K steps as hipGraph replays (K-step graphs of up to `G` steps + one tail graph), or as
    eager launches."""

    def __init__(self, step_fn, stream, G, use_graph):
        self.step_fn, self.stream, self.G, self.use_graph = step_fn, stream, G, use_graph
        self.graphs = {}

    def _graph(self, length):
        if length not in self.graphs:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream):
                for k in range(length):
                    self.step_fn(k)
            self.graphs[length] = g
        return self.graphs[length]

    def prepare(self, counts):
        if not self.use_graph:
            return
        for cnt in counts:
            if cnt >= self.G:
                self._graph(self.G)
            if cnt % self.G:
                self._graph(cnt % self.G)

    def run(self, steps):
        if self.use_graph:
            k = 0
            while steps - k >= self.G:
                self.graphs[self.G].replay()
                k += self.G
            if steps - k:
                self.graphs[steps - k].replay()
            return
        for k in range(steps):
            self.step_fn(k)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))           # before anything here touches a GPU
    # native libraries (RCCL's version banner, gloo's rank chatter) print to fd 1; keep
    # stdout clean for the ONE JSON line by pointing fd 1 at stderr until we print it
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
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

    from gym_comm_amd import dist as ocdist
    wrapper = args.agents == 2
    closed = args.mode == "closed-loop"
    if closed and not wrapper:
        raise SystemExit("--mode closed-loop drives the 2-agent gym_comm wrapper")
    n = args.envs
    K = max(1, args.steps)
    seed = ocdist.rank_seed(args.seed, rank)          # independent shard: actions AND placements
    stream = torch.cuda.Stream(device=dev)
    use_graph = not args.no_graph
    G = max(1, min(args.graph_steps, WINDOW))

    if closed:
        from gym_comm_amd.vec_env import FusedMLPPartner, OvercookedVecEnv, MLPPolicy, TorchPolicyPartner
        arglist = dict(level=args.level, num_agents=2, max_num_timesteps=args.T,
                       num_communication=args.comm, communication_on=True, fow_radius=2)
        fused = args.policy == "fused"
        if fused and (args.hidden != 64 or args.comm > 16):
            raise SystemExit("--policy fused: 64 hidden units, at most 16 comm channels (use --policy torch)")
        # fused: the hand-written MFMA policy kernel reads the int32 rows as they lie; torch: the
        # module's first GEMM wants float32 rows
        venv = OvercookedVecEnv(arglist, n, device=dev, seed=seed,
                                obs_dtype=getattr(torch, args.obs_dtype) if fused else torch.float32)
        env = venv._b
        seat = (lambda pol, sd: FusedMLPPartner(pol, sample=True, seed=sd, device=dev)) if fused else \
               (lambda pol, sd: TorchPolicyPartner(pol, sample=True, seed=sd, device=dev))
        venv.partner = seat(MLPPolicy(env.S, args.comm, hidden=args.hidden, seed=seed + 1).to(dev), seed + 2)
        ego = seat(MLPPolicy(env.S, args.comm, hidden=args.hidden, seed=seed + 3).to(dev), seed + 4)
        with torch.cuda.stream(stream):
            venv.reset_tensors()
            # ego fwd -> partner fwd -> step (+ stats in-kernel); --closed-loop-launches 1: the step
            # kernel evaluates both policies itself (oc_step_opts.policy), 2: policy kernel + step
            one = {"auto": None, "1": True, "2": False}[args.closed_loop_launches] if fused else None
            loop = venv.closed_loop(ego, graph=False, one_launch=one)
        step_fn = lambda k: loop.enqueue()             # captured K at a time by StepBlocks below
        obs_elem = 4
    else:
        env, step_fn, _ = open_loop_workload(args, dev, seed)
        obs_elem = ELEM[args.obs_dtype]
    lv = env.level

    blocks = StepBlocks(step_fn, stream, G, use_graph)
    # A short --steps block (K < G) would be one small graph per block, and the ~10 us between two
    # graph launches would be charged to 20 steps; `bpr` consecutive K-step blocks therefore share
    # one graph replay (bpr * K <= G steps) and the replay time is divided by bpr.
    bpr = max(1, G // K) if use_graph else 1
    SB = bpr * K
    with torch.cuda.stream(stream):
        for k in range(max(1, min(args.warmup, 64))):    # touch everything before capture (module load,
            step_fn(k)                                   # first-launch work): at least once, also for --warmup 0
        stream.synchronize()
        blocks.prepare((args.warmup, SB))
        blocks.run(args.warmup)
        stream.synchronize()
        # settle: a few untimed replays of the freshly instantiated graphs on top of the `--warmup`
        # steps.  (Round 2 spent 150 ms here against a "slow stretch after host-side idling"; round 3
        # found no slow GPU work behind it at all -- shader clock, kernel-active spans and launch
        # boundaries are normal in every launch (tools/slow_stretch_probe.py); the replays of a "slow"
        # run execute at the normal rate and start at once; what is late is the host's BLOCKED wait
        # for the last event, by 50-80 ms (slow_stretch_probe4.py).  Event-timed medians never saw it;
        # wall-clock loops did.  DESIGN.md section 7.)
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            blocks.run(SB)
            stream.synchronize()

        # reps: enough blocks for MIN_TIMED_MS of GPU time, from a short calibration
        if args.reps > 0:
            replays = max(1, -(-args.reps // bpr))
        else:
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ncal = max(1, min(8, 4096 // SB))
            c0.record(stream)
            for _ in range(ncal):
                blocks.run(SB)
            c1.record(stream)
            stream.synchronize()
            replay_ms = max(c0.elapsed_time(c1) / ncal, 1e-3)
            replays = int(min(4096, max(5, math.ceil(MIN_TIMED_MS / replay_ms))))
        replays = int(ocdist.reduce_max([float(replays)], dev if args.backend == "nccl" else None)[0])
        reps = replays * bpr                             # K-step blocks in the timed region

        if env.metrics is not None:
            env.metrics.zero_()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(replays + 1)]
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0].record(stream)
        for r in range(replays):
            blocks.run(SB)                               # bpr blocks of EXACTLY K steps each
            evs[r + 1].record(stream)
        # The host learns of the end of the region by POLLING the last event, not by a blocking wait:
        # on this box a blocked hipEventSynchronize / hipDeviceSynchronize sometimes returns 50-80 ms
        # after the device has finished (tools/slow_stretch_probe4.py: first event done at 0.4 ms, GPU
        # span 5.4 ms, last event "done" on the host at 60-81 ms -- in 3 of 4 rounds that followed the
        # building and dropping of eager batches; what round 2 took for a slow stretch of the GPU), and
        # for N > 1 the wall clock over this region IS the value.
        while not evs[-1].query():
            pass
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        block_s = sorted(evs[r].elapsed_time(evs[r + 1]) / 1e3 / bpr for r in range(replays))
        med_block_s = (block_s[replays // 2] if replays % 2
                       else 0.5 * (block_s[replays // 2 - 1] + block_s[replays // 2]))
        rollout_metrics = env.metrics_vector()          # before the probe launches below

        # kernel duration cross-check: HIP events on the launch stream around individual
        # eager launches, trimmed mean (open-loop only: a closed-loop step is several kernels)
        launch_ms_bracketed = None
        if not closed:
            probe = 200
            e0 = [torch.cuda.Event(enable_timing=True) for _ in range(probe)]
            e1 = [torch.cuda.Event(enable_timing=True) for _ in range(probe)]
            for k in range(probe):
                e0[k].record(stream)
                step_fn(k)
                e1[k].record(stream)
            stream.synchronize()
            per_launch_ms = sorted(a.elapsed_time(b) for a, b in zip(e0, e1))
            launch_ms_bracketed = float(np.mean(per_launch_ms[probe // 10: probe - probe // 10]))
        decomp = None
        if args.decompose and not closed and rank == 0:
            decomp = decompose_step(args, dev, seed, stream, SB if use_graph else min(G, 240))

    # end-of-rollout metrics: the only collective on the path (RCCL all-gather, 64 B/rank)
    if dist is not None and args.backend != "nccl":
        rollout_metrics = rollout_metrics.cpu()          # gloo rehearsal: gather on host tensors
    g = ocdist.gather_rollout_metrics(rollout_metrics, med_block_s)
    med_block_s = g["elapsed_s"]                         # MAX over ranks
    wall = ocdist.reduce_max([wall], dev if args.backend == "nccl" else None)[0]
    m = [g["total"][k] for k in ocdist.METRIC_NAMES]

    if rank == 0:
        # N = 1: the median block (robust against a stray host hiccup between two replays, and what the
        # HIP events on the launch stream saw).  N > 1: a per-rank median would hide a rank that stalls
        # for part of the region, so the whole-node figure is the barrier-bracketed wall clock over the
        # region (MAX over ranks); the median stays beside it.
        med_step_s = med_block_s / K
        wall_step_s = wall / (reps * K)
        step_s = med_step_s if world == 1 else wall_step_s
        value = n * world / step_s
        A, M, S = lv.num_agents, lv.num_items, lv.num_subtasks
        rd, wr = layout_bytes_per_env_step(A, M, S, args.comm, wrapper, obs_elem)
        survey = survey_bytes_per_env_step(A, M, S, args.comm, wrapper, obs_elem)
        achieved = (rd + wr) * n / med_step_s / 1e9        # the kernel's duration: HIP events, per rank
        survey_gbps = survey * n / med_step_s / 1e9
        traffic = traffic_src = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and not closed:
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                key = "%s_a%d_n%d" % (args.level, args.agents, n)
                if args.obs_dtype != "int32":
                    key += "_" + args.obs_dtype
                ent = tj.get(key, {})
                traffic = ent.get("hbm_bytes_per_launch")
                if traffic is not None:
                    from gym_comm_amd import specialize
                    now = specialize._source_digest().hexdigest()[:16]
                    if ent.get("kernel_source_digest") != now:
                        # counters of another kernel revision say nothing about this one
                        traffic = None
                        traffic_src = ("profiles/traffic.json holds counters of kernel source %s (%s); this is %s: "
                                       "dropped" % (ent.get("kernel_source_digest"), ent.get("source"), now))
                    else:
                        traffic_src = "%s (stored rocprofv3 --pmc passes: %s, kernel source %s; not collected " \
                                      "in this run)" % ("profiles/traffic.json", ent.get("source"), now)
            except Exception:
                traffic = None
        kernel = "k_multi_step" if wrapper else "k_step"
        out = {
            "metric": "env-steps/sec (whole node)" if not closed else
                      "env-steps/sec (whole node), closed loop: ego + partner policy in the loop",
            "value": value, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "reps": reps,
            "ms_per_step": step_s * 1e3, "ms_per_step_wall": wall_step_s * 1e3,
            "ms_per_step_median_block": med_step_s * 1e3,
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": args.obs_dtype if (not closed or args.policy == "fused") else "int32", "data": "synthetic",
            "config": {"workload": "%s, %d agents, %d parallel envs per GPU, T=%d, C=%d, %s"
                                   % (args.level, args.agents, n, args.T, args.comm,
                                      ("closed loop: 2 MLP policies (hidden %d, %s) + fused multi_step + "
                                       "episode statistics" % (args.hidden, "one MFMA kernel launch for both"
                                                               if args.policy == "fused" else "torch modules"))
                                      if closed else
                                      "fused multi_step (step+obs)" if wrapper else "step only"),
                       "mode": args.mode,
                       "level": args.level, "num_agents": args.agents, "envs_per_gpu": n,
                       "max_num_timesteps": args.T, "launch": "hipgraph" if use_graph else "eager",
                       "obs_dtype": (args.obs_dtype if args.policy == "fused" else "float32") if closed else args.obs_dtype,
                       "kernel_flavour": env.kernel_flavour,
                       **({"launches_per_step": (1 if loop.one_launch else 2) if args.policy == "fused" else "torch"}
                          if closed else {}),
                       # (what the library launches: the plain step open loop, the general variant closed loop)
                       "waves_per_64_envs": env.launch_waves(general=closed) if wrapper else 1,
                       "parallelism": "env-sharded x%d" % world},
            "agent_steps_per_sec": value * lv.num_agents,
            "timing": {"what": "median over back-to-back blocks of `steps` steps (`reps` of them, "
                               "`blocks_per_replay` per hipGraph replay), HIP events on the launch stream "
                               "between replays, MAX over ranks",
                       "value_from": ("median block time (N = 1)" if world == 1 else
                                      "barrier-bracketed wall clock over the region, MAX over ranks (N > 1)"),
                       "blocks_per_replay": bpr, "replays": replays, "settle_ms_untimed": args.settle_ms,
                       "block_ms_min": block_s[0] * 1e3, "block_ms_median": med_block_s * 1e3,
                       "block_ms_max": block_s[-1] * 1e3, "region_wall_s": wall,
                       "region_gpu_span_s": evs[0].elapsed_time(evs[-1]) / 1e3},
            "rollout_metrics": {"env_steps": m[0], "episodes": m[1], "successes": m[2],
                                "reward_sum": m[3], "completed_subtasks_sum": m[4], "errors": m[5]},
            "per_rank": [{"rank": r, "env_steps": pr[0], "episodes": pr[1], "successes": pr[2],
                          "reward_sum": pr[3]} for r, pr in enumerate(g["per_rank"])],
        }
        if not closed:
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "bytes": "the bytes this layout moves per env-step (packed state in/out, actions, comm, "
                         "observation rows, timestep, rewards, done, metrics atomics)",
                "layout_bytes_per_env_step": {"read": rd, "written": wr},
                "traffic": traffic, "traffic_source": traffic_src,
                "traffic_frac": (traffic / med_step_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                # SURVEY 8(d)'s int32-per-field accounting, which this layout does not move: a
                # labelled secondary rate; as a "fraction" only while it stays below 1
                "survey_bytes_per_env_step": survey, "survey_bytes_gbps": survey_gbps,
                "frac_survey_bytes": (survey_gbps / HBM_PEAK_GBPS) if survey_gbps <= HBM_PEAK_GBPS else None,
                "kernel": kernel,
                "avg_launch_us_timed_region": med_step_s * 1e6,
                "avg_launch_us_event_bracketed": launch_ms_bracketed * 1e3}
            if decomp is not None:
                # the step as the chip saw it (timeline build): kernel-active + boundary = period, to be
                # read against ms_per_step of the product build above
                # The stamps cost the timeline build's waves an SMEM round trip at their very end (the
                # clock read has to come back before it can be stored), so ITS kernel-active span is an
                # upper bound for the product kernel's; the boundary -- store drain, end-of-kernel cache
                # work, command processor, dispatch -- does not depend on the stamps.  Hence also:
                # product step - boundary = the product kernel's active span.
                act_prod = med_step_s * 1e6 - decomp["boundary_us"]
                out["roofline"].update({
                    "kernel_active_us": decomp["kernel_active_us"], "boundary_us": decomp["boundary_us"],
                    "kernel_active_us_product": act_prod,
                    "frac_kernel_active": (rd + wr) * n / (act_prod * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                    "decompose": decomp,
                    "decompose_closure": (decomp["kernel_active_us"] + decomp["boundary_us"]) / (med_step_s * 1e6)})
        if not args.no_cpu_baseline and world == 1 and not closed:
            out["cpu_baseline"] = cpu_baseline(lv.blob, lv.num_agents, args.comm, wrapper,
                                               args.cpu_seconds)
            out["cpu_baseline"]["reference_python"] = reference_python_rate(args.level, args.agents, wrapper)
        else:
            out["cpu_baseline"] = {"skipped": "N>1" if world > 1 else
                                   "closed-loop mode" if closed else "--no-cpu-baseline"}
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

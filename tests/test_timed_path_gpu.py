"""-m gpu: the path bench.py TIMES, and the paths its N > 1 form takes, under the driver's test run.

1. The timed region replays a hipGraph of chained launches (``bench.StepBlocks`` over
   ``bench.open_loop_workload``); the rest of the parity suite steps the same kernels eagerly.  Here
   the batch is built by bench.py's own code at its own geometry (BASELINE configs[1]: 4 096 envs,
   the library's launch choice; configs[3]: tl-3 x 65 536), the 240-step graph of ``--steps 20``
   (12 blocks per replay) is captured over a FIXED action window and replayed from reset, and the
   final state, comm, the last observations / reward bits / done and the six metric totals are
   compared with the oracle stepped over the same 240 actions (first 256 envs exact, every other env
   equal to its residue-class representative) -- after the first replay and after a second one.
2. ``bench.py`` as a CHILD process: under ``python -m torch.distributed.run`` with backend nccl
   (RCCL init + device all_gather at N = 1), and ``--gpus 2 --same-gpu --backend gloo`` through its
   own ``spawn_ranks`` (two ranks, per-rank seeds, totals = sums).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from hip_util import assert_snapshots_equal, bits, scripted_then_random

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N0 = 256


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


def _tiled_equal(t, n0):
    n = t.shape[-1]
    ref = t[..., :n0]
    return bool((t.reshape(*t.shape[:-1], n // n0, n0) == ref.unsqueeze(-2)).all().item())


CASES = [("open-divider_tomato", 2, 4096), ("partial-divider_tl", 3, 65536)]


@pytest.mark.parametrize("level,A,n", CASES, ids=["%s-a%d-n%d" % c for c in CASES])
def test_graph_replay_at_bench_geometry_matches_oracle(level, A, n, oracle_lib):
    b = _bench()
    T, C, K = 60, 2, 20           # T short enough for several time-outs and auto-resets inside 240 steps
    args = b.parse(["--level", level, "--agents", str(A), "--envs", str(n), "--T", str(T),
                    "--steps", str(K), "--warmup", "5"])
    wrapper = A == 2
    dev = torch.device("cuda", 0)
    WINDOW = b.WINDOW
    rng = np.random.default_rng(20260)
    if wrapper:
        mv = scripted_then_random(rng, level, WINDOW, 2, N0, nact=4)
        cm = rng.integers(0, C, (WINDOW, 2, N0)).astype(np.int32)
        acts0 = np.stack([mv[:, 0], cm[:, 0], mv[:, 1], cm[:, 1]], axis=1).astype(np.int32)
    else:
        acts0 = scripted_then_random(rng, level, WINDOW, A, N0, nact=4)
    window = torch.from_numpy(acts0).to(dev).repeat(1, 1, n // N0).contiguous()
    env, step_fn, _ = b.open_loop_workload(args, dev, 1234, actions=window)
    if wrapper:
        if not os.environ.get("OC_LAUNCH"):        # (a forced policy is the caller's business)
            assert env.launch_waves_per_64 == 4    # what the headline launches at 4 096 envs
    lv = env.level
    stream = torch.cuda.Stream(device=dev)
    G = min(args.graph_steps, WINDOW)
    bpr = max(1, G // K)
    SB = bpr * K                                   # 240: the replay bench.py times
    assert SB == 240
    blocks = b.StepBlocks(step_fn, stream, G, True)
    with torch.cuda.stream(stream):
        # bench.py steps eagerly before it captures; do the same, then start over from reset
        for k in range(5):
            step_fn(k)
        stream.synchronize()
        blocks.prepare((SB,))
        env.reset()
        env.comm.zero_()
        env.metrics.zero_()
        stream.synchronize()

        ora = oracle_lib.OracleBatch(lv.blob, N0, threads=4)      # the path under test
        orb = oracle_lib.OracleBatch(lv.blob, N0, threads=4)      # base env beside it: sparse reward, success
        comm = np.zeros((2, N0), np.int32)
        tot = {"episodes": 0, "successes": 0, "reward_sum": 0, "completed_subtasks_sum": 0}
        move_rows = [0, 2] if wrapper else list(range(A))
        for replay in range(2):
            blocks.run(SB)
            stream.synchronize()
            for k in range(SB):
                if wrapper:
                    oo, to, ro, do = ora.multi_step(acts0[k], comm, 2, 0, C, auto_reset=True)
                else:
                    ro, do, sho = ora.step(acts0[k], auto_reset=True)
                rb, db, _ = orb.step(np.ascontiguousarray(acts0[k][move_rows]), auto_reset=False)
                assert np.array_equal(db, do)
                snap = orb.snapshot_all()
                fin = db != 0
                tot["episodes"] += int(fin.sum())
                tot["successes"] += int((fin & (snap["t"] < T)).sum())
                tot["reward_sum"] += int(rb.sum())
                tot["completed_subtasks_sum"] += int(snap["completed"][fin].sum())
                orb.reset(mask=db)
            ctx = "%s n=%d after replay %d" % (level, n, replay)
            from gym_comm_amd.state import unpack_state
            hs = unpack_state(env.state[:, :N0].cpu().numpy(), lv.num_agents, lv.num_items, lv.num_subtasks,
                              **env.unpack_kw())
            os_ = ora.snapshot_all()
            clean = (os_["error"] == 0) & (hs["error"] == 0)
            assert clean.all() or not wrapper, ctx
            assert_snapshots_equal(hs, os_, ctx, where=clean)
            assert _tiled_equal(env.state, N0), ctx
            if wrapper:
                assert np.array_equal(env.obs[:, :, :N0].cpu().numpy(), oo), ctx
                assert np.array_equal(bits(env.shaped_reward[:N0].cpu().numpy()), bits(ro)), ctx
                assert np.array_equal(bits(env.timestep[:N0].cpu().numpy()), bits(to)), ctx
                assert np.array_equal(env.done[:N0].cpu().numpy(), do), ctx
                assert np.array_equal(env.comm[:, :N0].cpu().numpy(), comm), ctx
                assert np.array_equal(env.reward[:N0].cpu().numpy(), rb), ctx
                assert _tiled_equal(env.obs, N0) and _tiled_equal(env.shaped_reward.view(torch.int64), N0), ctx
                assert _tiled_equal(env.comm, N0) and _tiled_equal(env.done, N0), ctx
            else:
                assert np.array_equal(env.reward[:N0].cpu().numpy()[clean], ro[clean]), ctx
                assert np.array_equal(env.done[:N0].cpu().numpy()[clean], do[clean]), ctx
                assert np.array_equal(bits(env.shaping[:, :N0].cpu().numpy())[:, clean], bits(sho)[:, clean]), ctx
                assert _tiled_equal(env.reward, N0) and _tiled_equal(env.shaping.view(torch.int64), N0), ctx
            m = env.read_metrics()
            reps = n // N0
            steps_done = (replay + 1) * SB
            assert m["env_steps"] == n * steps_done, ctx
            if clean.all():
                want = {k: v * reps for k, v in tot.items()}
                got = {k: m[k] for k in want}
                assert got == want, ctx
                assert m["errors"] == 0, ctx
        assert tot["episodes"] >= N0 and tot["reward_sum"] > 0      # time-outs, auto-resets and rewards happened


def _last_json(text):
    lines = [ln for ln in text.splitlines() if ln.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


def _child_env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


BENCH_SHORT = ["--steps", "20", "--warmup", "5", "--reps", "24", "--no-cpu-baseline", "--cpu-seconds", "1"]


def test_bench_child_under_torchrun_nccl_one_rank():
    """The driver's N > 1 form at N = 1: torch.distributed.run -> init_process_group("nccl") (RCCL),
    barriers, the device all_gather of the metrics vector and the MAX-reduce of the elapsed time."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--backend", "nccl"] + BENCH_SHORT
    out = subprocess.run(cmd, env=_child_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    j = _last_json(out.stdout)
    assert j["n_gpus"] == 1 and len(j["per_rank"]) == 1
    assert j["rollout_metrics"]["env_steps"] == 4096 * j["reps"] * j["steps"]
    assert j["per_rank"][0]["env_steps"] == j["rollout_metrics"]["env_steps"]
    assert j["value"] > 1e8 and j["roofline"]["frac"] > 0


def test_bench_self_spawned_two_ranks_gloo_same_gpu():
    """``bench.py --gpus 2`` without a launcher: spawn_ranks -> two ranks (both on cuda:0, gloo for the
    exchange), per-rank seeds, whole-job totals = sums over ranks."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-gpu", "--backend", "gloo",
           "--T", "60"] + BENCH_SHORT
    out = subprocess.run(cmd, env=_child_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    j = _last_json(out.stdout)
    assert j["n_gpus"] == 2 and len(j["per_rank"]) == 2
    r0, r1 = j["per_rank"]
    assert r0["env_steps"] == r1["env_steps"] == 4096 * j["reps"] * j["steps"]
    assert r0["reward_sum"] != r1["reward_sum"] or r0["episodes"] != r1["episodes"]     # per-rank seeds
    for k in ("env_steps", "episodes", "successes", "reward_sum"):
        assert j["rollout_metrics"][k] == r0[k] + r1[k], k
    assert j["cpu_baseline"] == {"skipped": "N>1"}
    assert abs(j["value"] - 2 * 4096 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6
    assert j["timing"]["value_from"].startswith("barrier-bracketed wall")


def test_timeline_build_records_every_launch_and_steps_identically():
    """The -DOC_TIMELINE flavour behind ``bench.py --decompose`` (include/oc_hip.h: oc_timeline_begin):
    same results as the product build, one record per launch with every wave counted and
    start <= issue-end per wave, no overlap between launches."""
    import ctypes
    from gym_comm_amd.batched import BatchedOvercooked
    n, steps = 4096, 64
    mk = lambda mode: BatchedOvercooked("open-divider_tomato", num_envs=n, max_num_timesteps=40, device="cuda:0",
                                        specialize_level=mode)
    prod, tl = mk("auto"), mk("timeline")
    assert prod._L.oc_timeline_begin(None, 0, 0) != 0       # the product build refuses
    acts = torch.randint(0, 2, (steps, 4, n), dtype=torch.int32, device="cuda")
    acts[:, 0] = torch.randint(0, 4, (steps, n), dtype=torch.int32, device="cuda")
    acts[:, 2] = torch.randint(0, 4, (steps, n), dtype=torch.int32, device="cuda")
    stride = 4 * (n // 64)
    rec = torch.full((steps, stride, 4), -1, dtype=torch.int32, device="cuda")
    assert tl._L.oc_timeline_begin(ctypes.c_void_p(rec.data_ptr()), steps, stride) == 0
    for k in range(steps):
        prod.multi_step(acts[k])
        tl.multi_step(acts[k])
    tl.multi_step(acts[0])                                   # past the last record: not recorded
    prod.multi_step(acts[0])
    assert tl._L.oc_timeline_begin(None, 0, 0) == 0
    torch.cuda.synchronize()
    for name in ("state", "obs", "comm", "done", "reward"):
        assert torch.equal(getattr(prod, name), getattr(tl, name)), name
    assert torch.equal(prod.shaped_reward.view(torch.int64), tl.shaped_reward.view(torch.int64))
    wrote, start, issue, drain, cycles = (t.cpu().numpy() for t in _bench().timeline_reduce(rec))
    waves = (n // 64) * tl.launch_waves_per_64
    assert wrote[:, :waves].all() and not wrote[:, waves:].any()
    start, issue, drain, cycles = (a[:, :waves] for a in (start, issue, drain, cycles))
    assert (start > 0).all() and (start <= issue).all() and (issue == drain).all()   # (drain: the =2 flavour only)
    mhz = 100.0 * cycles.sum() / (issue - start).sum()
    assert 500 < mhz < 3000, mhz                            # the shader clock the waves ran at
    first, last = start.min(axis=1), issue.max(axis=1)
    assert (first[1:] >= last[:-1]).all()                    # launches of one stream do not overlap
    assert ((last - first) < 100000).all()                   # < 1 ms at 100 MHz


def test_prepared_call_steps_like_the_full_call():
    """include/oc_hip.h: oc_multi_step_prepare / oc_call_launch = oc_multi_step with the arguments
    fixed once (the foreign call OvercookedVecEnv.step_tensors makes per step).  Two batches from the
    same seed, one stepped through ``multi_step`` and one through prepared calls -- action rows, and
    int64 ego pairs beside the kernel's own partner draw -- must hold the same bytes after every step."""
    from gym_comm_amd.batched import BatchedOvercooked
    n, K, T = 4096 + 37, 90, 40
    mk = lambda: BatchedOvercooked("open-divider_tomato", num_envs=n, max_num_timesteps=T, num_communication=2,
                                   fow_radius=2, seed=3, episode_stats=True)
    gen = torch.Generator(device="cuda").manual_seed(8)
    rows = torch.randint(0, 2, (K, 4, n), generator=gen, device="cuda", dtype=torch.int32)
    rows[:, 0] = torch.randint(0, 4, (K, n), generator=gen, device="cuda", dtype=torch.int32)
    rows[:, 2] = torch.randint(0, 4, (K, n), generator=gen, device="cuda", dtype=torch.int32)
    pairs = torch.stack([rows[:, 0], rows[:, 1]], dim=2).to(torch.int64).contiguous()        # [K][n][2]
    same = lambda a, b: all(torch.equal(getattr(a, f), getattr(b, f)) for f in
                            ("state", "comm", "obs", "done", "metrics", "ep_return", "ep_length")) and \
        torch.equal(a.shaped_reward.view(torch.int64), b.shaped_reward.view(torch.int64)) and \
        torch.equal(a.timestep.view(torch.int64), b.timestep.view(torch.int64))
    # (1) action rows: the prepared call reads the SAME buffer every launch
    a, b = mk(), mk()
    a.reset(); b.reset()
    buf = torch.empty((4, n), dtype=torch.int32, device="cuda")
    launch = b.prepare_multi_step(buf.data_ptr())
    for k in range(K):
        a.multi_step(rows[k].contiguous())
        buf.copy_(rows[k])
        launch()
        assert same(a, b), k
    assert int(a.metrics.sum()) > 0 and int(a.done.sum()) >= 0
    # (2) int64 ego pairs per launch + the kernel's own partner draw
    a, b = mk(), mk()
    a.reset(); b.reset()
    rng_a = torch.arange(n, dtype=torch.int32, device="cuda") * 7 + 1
    rng_b = rng_a.clone()
    played_a = torch.zeros((2, n), dtype=torch.int32, device="cuda")
    played_b = torch.zeros_like(played_a)
    launch = b.prepare_multi_step(0, alt_rng_ptr=rng_b.data_ptr(), alt_played_ptr=played_b.data_ptr())
    for k in range(K):
        a.multi_step(ego_pairs=pairs[k], alt_rng=rng_a, alt_played=played_a)
        launch(pairs[k].data_ptr(), True)
        assert same(a, b) and torch.equal(played_a, played_b) and torch.equal(rng_a, rng_b), k
    # a NULL call is refused, not dereferenced
    assert a._L.oc_call_launch(None, None, 0, None) == -1      # OC_E_BADARG

"""-m gpu: the hand-written HIP path (through the C ABI, gym_comm_amd.BatchedOvercooked)
against (1) the golden vectors recorded from the reference and (2) the CPU oracle on
seeded inputs.  Integer state, rewards, done flags and observations must be equal;
fp64 shaping / rewards / timesteps must have identical bits."""
import os

import numpy as np
import pytest
import torch

from conftest import compile_for, golden_files, load_golden
from hip_util import assert_snapshots_equal, bits, scripted_then_random
from spec_levels import SEEDED_CASES, SPEC_GOLDEN

pytestmark = pytest.mark.gpu

# rbase_/rwrap_: random-* levels; cbase_/cwrap_: our own maps run through the reference
BASE = golden_files("base_") + golden_files("rbase_") + golden_files("cbase_") + golden_files("pbase_")
WRAP = golden_files("wrap_") + golden_files("rwrap_") + golden_files("cwrap_") + golden_files("pwrap_")
# (c*_dup_*: maps that repeat a content type -- multiset objects, goal counts above 1, CPython's
# set order in the shaping terms; the library's "dup" kernels)


def _placement_tensor(lv, cells, n):
    """One episode's item start cells (fixture row) broadcast to all n envs."""
    packed = torch.tensor(lv.pack_placement(cells), dtype=torch.int32)
    return packed.view(-1, 1).repeat(1, n).contiguous().cuda()
# every fixture on the generic library; a subset also on the per-level specialised one
BASE_RUNS = [(p, False) for p in BASE] + [(p, True) for p in BASE if os.path.basename(p) in SPEC_GOLDEN]
WRAP_RUNS = [(p, False) for p in WRAP] + [(p, True) for p in WRAP if os.path.basename(p) in SPEC_GOLDEN]


def _rid(run):
    return os.path.basename(run[0]) + ("-spec" if run[1] else "")


def _env(lv, n, **kw):
    from gym_comm_amd.batched import BatchedOvercooked
    return BatchedOvercooked(lv, num_envs=n, **kw)


@pytest.mark.parametrize("run", BASE_RUNS, ids=[_rid(r) for r in BASE_RUNS])
def test_base_step_matches_reference_golden(run):
    """Replay every recorded tape of OvercookedEnvironment.step/reset; 96 lanes (one and
    a half waves) get the same actions and must all reproduce the reference."""
    from gym_comm_amd.state import unpack_state
    path, spec = run
    z, st = load_golden(path)
    lv = compile_for(st)
    n = 96
    env = _env(lv, n, auto_reset=False, specialize_level=spec, placement_mode="host")
    assert env.kernel_flavour == ("spec" if spec else "generic")
    K = len(z["t"])
    acts = torch.from_numpy(np.repeat(z["actions"].astype(np.int32)[:, :, None], n, axis=2)).cuda()
    hist_state, hist_r, hist_d, hist_s = [], [], [], []
    for k in range(K):
        if z["reset_before"][k]:
            if lv.random_placement:
                env.set_placement(_placement_tensor(lv, z["placements"][z["pl_index"][k]], n))
            env.reset()
        r, d, sh = env.step(acts[k])
        hist_state.append(env.state.clone())
        hist_r.append(r.clone()); hist_d.append(d.clone()); hist_s.append(sh.clone())
    S = torch.stack(hist_state).cpu().numpy()          # [K][W][n]
    R = torch.stack(hist_r).cpu().numpy()
    Dn = torch.stack(hist_d).cpu().numpy()
    Sh = torch.stack(hist_s).cpu().numpy()             # [K][2][n]
    assert (R == z["reward"][:, None]).all()
    assert (Dn == z["done"][:, None]).all()
    assert (bits(Sh) == z["shaping_bits"][:, :, None]).all()
    for lane in (0, 63, 64, 95):
        snap = unpack_state(S[:, :, lane].T, lv.num_agents, lv.num_items, lv.num_subtasks, **env.unpack_kw())
        assert (snap["items"] == z["items"][:, :, :5]).all(), lane
        assert (snap["order"] == z["order"]).all(), lane
        assert (snap["agents"] == z["agents"]).all(), lane
        assert (snap["t"] == z["t"]).all(), lane
        assert (snap["completed"] == z["completed"]).all(), lane
        assert (snap["goal_count"] == z["goal_count"]).all(), lane
        assert (snap["nobj"] == z["nobj"]).all(), lane
        assert (snap["error"] == 0).all(), lane


def _wrap_env(st, lv, n, **kw):
    return _env(lv, n, ego_config=st["ego_config"], partner_config=st["partner_config"],
                num_communication=st["num_communication"], communication_on=st["communication_on"],
                ego_led=st["ego_led"], fow_radius=st["fow_radius"],
                ego_agent_idx=st["ego_agent_idx"], **kw)


@pytest.mark.parametrize("fused", [4, 2, 1, 0], ids=["fused-split", "fused-split2", "fused-1wave", "step+obs"])
@pytest.mark.parametrize("run", WRAP_RUNS, ids=[_rid(r) for r in WRAP_RUNS])
def test_wrapper_matches_reference_golden(run, fused):
    """OvercookedMultiEnv.multi_step / multi_reset / get_observation2 tapes, through the
    fused oc_multi_step kernel -- as a split launch (four waves per 64 envs, one output each; two
    waves, two outputs each) and with one wave computing the whole step -- and through oc_step +
    oc_obs."""
    path, spec = run
    z, st = load_golden(path)
    lv = compile_for(st)
    n = 70
    env = _wrap_env(st, lv, n, auto_reset=False, specialize_level=spec, placement_mode="host",
                    waves_per_64=fused)
    C = st["num_communication"]
    if lv.random_placement:
        env.set_placement(_placement_tensor(lv, z["placements"][0], n))
        env.reset()
    # observation right after multi_reset()
    obs, ts = env.observe()
    obs_h, ts_h = obs.cpu().numpy(), ts.cpu().numpy()
    for v in range(2):
        assert (obs_h[v, :, :] == z["reset_obs"][v][:, None]).all()
    assert (bits(ts_h) == z["reset_ts_bits"][0]).all()
    K = len(z["done"])
    acts = torch.from_numpy(np.repeat(z["actions"].astype(np.int32)[:, :, None], n, axis=2)).cuda()
    H_obs, H_ts, H_r, H_d = [], [], [], []
    can = (1 if st["ego_config"]["CAN_MOVE"] else 0) | (2 if st["partner_config"]["CAN_MOVE"] else 0)
    for k in range(K):
        if z["reset_before"][k]:
            if lv.random_placement:
                env.set_placement(_placement_tensor(lv, z["placements"][z["pl_index"][k]], n))
            env.reset()
        if fused:
            o, t, r, d = env.multi_step(acts[k])
        else:
            # the wrapper's host logic spelled out with the two separate kernels
            a = acts[k]
            noop = torch.full_like(a[0], 4)
            em = a[0] if (can & 1) else noop
            am = a[2] if (can & 2) else noop
            base = torch.stack([em, am] if st["ego_agent_idx"] == 0 else [am, em]).contiguous()
            neg = torch.full_like(a[1], -1)
            env.comm[0] = a[1] if st["communication_on"] else neg
            env.comm[1] = a[3] if (st["communication_on"] and not st["ego_led"]) else neg
            rr, d, sh = env.step(base)
            r = (rr.double() - sh[0]) - sh[1]
            o, t = env.observe()
        H_obs.append(o.clone()); H_ts.append(t.clone()); H_r.append(r.clone()); H_d.append(d.clone())
    O = torch.stack(H_obs).cpu().numpy()      # [K][2][F][n]
    T = torch.stack(H_ts).cpu().numpy()
    R = torch.stack(H_r).cpu().numpy()
    Dn = torch.stack(H_d).cpu().numpy()
    assert (Dn == z["done"][:, None]).all()
    assert (bits(R) == z["rew_bits"][:, None]).all()
    assert (O == z["obs"][:, :, :, None]).all()
    assert (bits(T) == z["ts_bits"][:, 0][:, None]).all()
    assert env.F == 22 + lv.num_subtasks + 2 * C


CASES = SEEDED_CASES


@pytest.mark.parametrize("spec,waves", [(False, 1), (True, 2), (True, 1), ("structure", 2)],
                         ids=["generic", "spec-split", "spec-1wave", "structure-split"])
@pytest.mark.parametrize("level,A,T", CASES, ids=["%s-a%d" % (c[0], c[1]) for c in CASES])
def test_step_matches_oracle_seeded(level, A, T, spec, waves, oracle_lib, monkeypatch):
    """Seeded action streams, every env different, auto-reset on: full state compare with
    the oracle after every step (ragged batch: 1000 envs = 15 waves + a 40-lane tail).  The
    specialised libraries launch the step split over two waves per 64 envs (state wave + shaping
    wave) at this batch size; `1wave` forces the one-wave launch large batches get."""
    from gym_comm_amd import compiler
    monkeypatch.setenv("OC_LAUNCH", "step_split=%d" % waves)
    lv = compiler.compile_level(level, A, T)
    n, steps = 1000, 260
    rng = np.random.default_rng(1234 + A)
    acts = scripted_then_random(rng, level, steps, A, n)
    ora = oracle_lib.OracleBatch(lv.blob, n, threads=4)
    env = _env(lv, n, auto_reset=True, specialize_level=spec)
    acts_d = torch.from_numpy(acts).cuda()
    tot_r = tot_d = 0
    for k in range(steps):
        r, d, sh = env.step(acts_d[k])
        ro, do, sho = ora.step(acts[k], auto_reset=True)
        hs = env.snapshot()
        os_ = ora.snapshot_all()
        ctx = "%s step %d" % (level, k)
        # Flagged envs (states where the reference itself raises) would be excluded from the
        # compare; with these seeds the oracle flags exactly 0 env-steps in every one of the 8
        # configurations (counted on the CPU), so EVERY env is compared at EVERY step
        assert (os_["error"] == 0).all() and (hs["error"] == 0).all(), ctx
        assert np.array_equal(r.cpu().numpy(), ro), ctx
        assert np.array_equal(d.cpu().numpy(), do), ctx
        assert np.array_equal(bits(sh.cpu().numpy()), bits(sho)), ctx
        assert_snapshots_equal(hs, os_, ctx)
        tot_r += int(r.sum().item()); tot_d += int(d.sum().item())
    m = env.read_metrics()
    assert m["env_steps"] == n * steps
    assert m["episodes"] == tot_d and m["reward_sum"] == tot_r
    assert tot_d > 0 and tot_r > 0


@pytest.mark.parametrize("waves", [4, 2, 1], ids=["split", "split2", "1wave"])
@pytest.mark.parametrize("spec", [False, True, "structure"], ids=["generic", "spec", "structure"])
@pytest.mark.parametrize("level,T,C,radius", [("open-divider_tomato", 100, 2, 2),
                                              ("full-divider_salad", 120, 5, 1),
                                              ("open-divider_tl", 200, 3, 3)])
def test_multi_step_matches_oracle_seeded(level, T, C, radius, spec, waves, oracle_lib):
    from gym_comm_amd import compiler
    lv = compiler.compile_level(level, 2, T)
    n, steps = 777, 200
    rng = np.random.default_rng(99)
    mv = scripted_then_random(rng, level, steps, 2, n, nact=4)
    cm = rng.integers(0, C, (steps, 2, n)).astype(np.int32)
    acts = np.stack([mv[:, 0], cm[:, 0], mv[:, 1], cm[:, 1]], axis=1).astype(np.int32)
    ora = oracle_lib.OracleBatch(lv.blob, n, threads=4)
    comm = np.zeros((2, n), np.int32)
    env = _env(lv, n, num_communication=C, fow_radius=radius, auto_reset=True, specialize_level=spec,
               waves_per_64=waves)
    acts_d = torch.from_numpy(acts).cuda()
    tot_d = 0
    for k in range(steps):
        o, t, r, d = env.multi_step(acts_d[k])
        oo, to, ro, do = ora.multi_step(acts[k], comm, radius, 0, C, auto_reset=True)
        ctx = "%s step %d" % (level, k)
        assert np.array_equal(d.cpu().numpy(), do), ctx
        assert np.array_equal(bits(r.cpu().numpy()), bits(ro)), ctx
        assert np.array_equal(o.cpu().numpy(), oo), ctx
        assert np.array_equal(bits(t.cpu().numpy()), bits(to)), ctx
        assert np.array_equal(env.comm.cpu().numpy(), comm), ctx
        if k % 20 == 19:     # the state rows come from another wave than the observations in a split launch
            assert_snapshots_equal(env.snapshot(), ora.snapshot_all(), ctx)
        tot_d += int(do.sum())
    m = env.read_metrics()
    assert m["env_steps"] == n * steps and m["episodes"] == tot_d


RANDOM_CASES = [("random-open-divider_salad_small", 2, 80), ("random-salad-superwide", 2, 100),
                ("random-open-divider_tomato", 3, 80), ("random-partial-divider_salad", 4, 90)]


@pytest.mark.parametrize("level,A,T", RANDOM_CASES, ids=["%s-a%d" % (c[0], c[1]) for c in RANDOM_CASES])
def test_random_levels_match_oracle_with_host_placements(level, A, T, oracle_lib):
    """random-* levels, every env with its own item placement (host-supplied, so the oracle
    sees the same draw), auto-reset on: full state compare every step."""
    from gym_comm_amd import compiler
    lv = compiler.compile_level(level, A, T)
    n, steps = 600, 240
    rng = np.random.default_rng(77 + A)
    nc = len(lv.counters)
    place = np.zeros((lv.num_items, n), np.int32)
    for i in range(n):
        pick = rng.choice(nc, size=len(lv.scatter_items), replace=False)
        for k, item in enumerate(lv.scatter_items):
            x, y = lv.counters[pick[k]]
            place[item, i] = x | (y << 4)
    acts = scripted_then_random(rng, level, steps, A, n)
    ora = oracle_lib.OracleBatch(lv.blob, n, threads=4)
    ora.set_placement(place)
    ora.reset()
    env = _env(lv, n, auto_reset=True, placement_mode="host")
    env.set_placement(torch.from_numpy(place).cuda())
    env.reset()
    acts_d = torch.from_numpy(acts).cuda()
    tot = 0
    for k in range(steps):
        r, d, sh = env.step(acts_d[k])
        ro, do, sho = ora.step(acts[k], auto_reset=True)
        hs, os_ = env.snapshot(), ora.snapshot_all()
        ctx = "%s step %d" % (level, k)
        # flagged fraction: exactly 0 of the 600 x 240 env-steps in all four configurations
        assert (os_["error"] == 0).all() and (hs["error"] == 0).all(), ctx
        assert np.array_equal(r.cpu().numpy(), ro), ctx
        assert np.array_equal(d.cpu().numpy(), do), ctx
        assert np.array_equal(bits(sh.cpu().numpy()), bits(sho)), ctx
        assert_snapshots_equal(hs, os_, ctx)
        tot += int(r.sum().item())
    assert tot > 0


def test_random_levels_device_rng_placements():
    """placement_mode='rng': the in-kernel PCG32 draw puts every scattered item on its own
    Counter tile, uniformly, reproducibly per seed, and re-draws at every auto-reset."""
    from gym_comm_amd import compiler
    from gym_comm_amd.state import unpack_state
    lv = compiler.compile_level("random-open-divider_salad_small", 2, 5)
    n = 8192
    counters = {(x, y) for x, y in lv.counters}

    def cells(env):
        s = unpack_state(env.state.cpu().numpy(), 2, lv.num_items, lv.num_subtasks)
        return s["items"][:, :, :2]

    a = _env(lv, n, auto_reset=True, seed=5)
    b = _env(lv, n, auto_reset=True, seed=5)
    c = _env(lv, n, auto_reset=True, seed=6)
    ca, cb, cc = cells(a), cells(b), cells(c)
    assert (ca == cb).all() and not (ca == cc).all()
    for e in range(0, n, 97):
        got = [tuple(ca[e, i]) for i in range(lv.num_items)]
        assert set(got) <= counters and len(set(got)) == lv.num_items
    # roughly uniform over the 9 Counters
    hist = np.zeros((7, 7), np.int64)
    for i in range(lv.num_items):
        np.add.at(hist, (ca[:, i, 1], ca[:, i, 0]), 1)
    per = np.array([hist[y, x] for x, y in lv.counters]) / float(n * lv.num_items)
    assert np.abs(per - 1.0 / len(lv.counters)).max() < 0.02
    # T = 5: every env times out and is re-placed by the auto-reset
    acts = torch.full((2, n), 4, dtype=torch.int32, device="cuda")
    for _ in range(5):
        a.step(acts)
    cnew = cells(a)
    assert (a.done == 1).all() and (cnew != ca).any(axis=(1, 2)).mean() > 0.9
    for e in range(0, n, 211):
        got = [tuple(cnew[e, i]) for i in range(lv.num_items)]
        assert set(got) <= counters and len(set(got)) == lv.num_items


FOW = golden_files("fow_")


@pytest.mark.parametrize("spec", [False, True], ids=["generic", "spec"])
@pytest.mark.parametrize("path", FOW, ids=[os.path.basename(p) for p in FOW])
def test_fow_image_obs_matches_reference_golden(path, spec):
    """get_partial_observability_FOW (overcooked_env.py:161-202) through oc_obs_image."""
    z, st = load_golden(path)
    lv = compile_for(st)
    n = 80
    env = _env(lv, n, auto_reset=False, specialize_level=spec, fow_radius=st["radius"])
    K = 600
    acts = torch.from_numpy(np.repeat(z["actions"][:K].astype(np.int32)[:, :, None], n, axis=2)).cuda()
    for k in range(K):
        if z["reset_before"][k]:
            env.reset()
        env.step(acts[k])
        maps, hold = env.observe_image()
        m = maps.cpu().numpy()                            # [2][7][W][H][n]
        for lane in (0, 63, 79):
            assert (m[..., lane] == z["maps"][k]).all(), (k, lane)
        assert (hold.cpu().numpy() == z["holding"][k][:, None]).all(), k
        assert (env.completed_subtasks().cpu().numpy() == z["completed"][k][:, None]).all(), k


def test_hundred_comm_channels_like_the_reference_configs(oracle_lib):
    """spread/env_args100on.json: random-salad-superwide, num_communication = 100, T = 900."""
    from gym_comm_amd import compiler
    level, C, T = "random-salad-superwide", 100, 900
    lv = compiler.compile_level(level, 2, T)
    n, steps = 200, 120
    rng = np.random.default_rng(3)
    place = np.zeros((lv.num_items, n), np.int32)
    for i in range(n):
        pick = rng.choice(len(lv.counters), size=len(lv.scatter_items), replace=False)
        for k, item in enumerate(lv.scatter_items):
            x, y = lv.counters[pick[k]]
            place[item, i] = x | (y << 4)
    mv = rng.integers(0, 4, (steps, 2, n)).astype(np.int32)
    cm = rng.integers(0, C, (steps, 2, n)).astype(np.int32)
    acts = np.stack([mv[:, 0], cm[:, 0], mv[:, 1], cm[:, 1]], axis=1).astype(np.int32)
    ora = oracle_lib.OracleBatch(lv.blob, n, threads=4)
    ora.set_placement(place)
    ora.reset()
    comm = np.zeros((2, n), np.int32)
    env = _env(lv, n, num_communication=C, fow_radius=2, auto_reset=True, placement_mode="host")
    env.set_placement(torch.from_numpy(place).cuda())
    env.reset()
    assert env.F == 22 + 9 + 200
    a_d = torch.from_numpy(acts).cuda()
    for k in range(steps):
        o, t, r, d = env.multi_step(a_d[k])
        oo, to, ro, do = ora.multi_step(acts[k], comm, 2, 0, C, auto_reset=True)
        assert np.array_equal(o.cpu().numpy(), oo), k
        assert np.array_equal(bits(r.cpu().numpy()), bits(ro)), k
        assert np.array_equal(bits(t.cpu().numpy()), bits(to)), k


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "step+obs"])
@pytest.mark.parametrize("odt", [torch.int8, torch.float32], ids=["int8", "float32"])
@pytest.mark.parametrize("name", ["wrap_salad_open_c5.npz", "wrap_tl_full_blind_allergic.npz",
                                  "rwrap_rsuperwide_c5.npz", "cwrap_conion_r2.npz"])
def test_int8_observation_rows_match_reference_golden(name, odt, fused):
    """oc_obs_cfg.obs_int8 = 1 / 2: the same observations as int8 rows (4x fewer bytes) or as
    float32 rows (what a policy network takes without a cast); every value is a small integer,
    so both are exact."""
    path = os.path.join(os.path.dirname(BASE[0]), name)
    z, st = load_golden(path)
    lv = compile_for(st)
    n = 130
    env = _wrap_env(st, lv, n, auto_reset=False, placement_mode="host", obs_dtype=odt,
                    specialize_level=(name in SPEC_GOLDEN))
    assert env.obs.dtype == odt
    if lv.random_placement:
        env.set_placement(_placement_tensor(lv, z["placements"][0], n))
        env.reset()
    obs, ts = env.observe()
    assert (obs.cpu().numpy()[:, :, 0] == z["reset_obs"]).all()
    K = min(len(z["done"]), 900)
    acts = torch.from_numpy(np.repeat(z["actions"][:K].astype(np.int32)[:, :, None], n, axis=2)).cuda()
    can = (1 if st["ego_config"]["CAN_MOVE"] else 0) | (2 if st["partner_config"]["CAN_MOVE"] else 0)
    for k in range(K):
        if z["reset_before"][k]:
            if lv.random_placement:
                env.set_placement(_placement_tensor(lv, z["placements"][z["pl_index"][k]], n))
            env.reset()
        if fused:
            o, t, r, d = env.multi_step(acts[k])
        else:
            a = acts[k]
            noop = torch.full_like(a[0], 4)
            em = a[0] if (can & 1) else noop
            am = a[2] if (can & 2) else noop
            base = torch.stack([em, am] if st["ego_agent_idx"] == 0 else [am, em]).contiguous()
            neg = torch.full_like(a[1], -1)
            env.comm[0] = a[1] if st["communication_on"] else neg
            env.comm[1] = a[3] if (st["communication_on"] and not st["ego_led"]) else neg
            rr, d, sh = env.step(base)
            r = (rr.double() - sh[0]) - sh[1]
            o, t = env.observe()
        oh = o.cpu().numpy()
        for lane in (0, 64, 129):
            assert (oh[:, :, lane] == z["obs"][k]).all(), (k, lane)
        assert (bits(r.cpu().numpy()) == z["rew_bits"][k]).all(), k


def test_all_six_metric_counters_on_a_known_rollout():
    """Every per-wave counter (env_steps, episodes, successes, reward_sum,
    completed_subtasks_sum, errors) against a rollout whose totals are known in closed form:
    n envs play the scripted open-divider_tomato solve (rewards +1 +1 +3,
    done and successful at step 23, all 3 subtasks completed) except every 7th env, which
    stands still and times out at T = 24 (one step later: the time limit is tested before
    the deliveries, overcooked_environment.py:245-249).  n = 1000 leaves a 40-lane tail wave."""
    from hip_util import KAT1_TOMATO
    from gym_comm_amd import compiler
    n, T = 1000, len(KAT1_TOMATO) + 1
    lv = compiler.compile_level("open-divider_tomato", 2, T)
    idle = np.arange(n) % 7 == 3
    env = _env(lv, n, auto_reset=True, num_communication=2)
    for a0, a1 in KAT1_TOMATO + [(4, 4)]:
        acts = np.stack([np.where(idle, 4, a0), np.where(idle, 4, a1)]).astype(np.int32)
        env.step(torch.from_numpy(acts).cuda())
    m = env.read_metrics()
    solved = int((~idle).sum())
    assert m == {"env_steps": n * T, "episodes": n, "successes": solved, "reward_sum": 5 * solved,
                 "completed_subtasks_sum": 3 * solved, "errors": 0}, m


def _state_cells(env):
    """Packed item cells (x | y<<4) [M][n] read back from the device state."""
    w = env.state[env.A:env.A + env.M].cpu().numpy()
    return (w & 255).astype(np.int32)


RNG_CASES = [("random-open-divider_salad_small", 40), ("random-salad-superwide", 50),
             ("random-open-divider_salad_small_wide_big", 45)]


@pytest.mark.parametrize("spec", [False, True], ids=["generic", "spec"])
@pytest.mark.parametrize("fused", [0, 4, 1], ids=["step", "fused-split", "fused-1wave"])
@pytest.mark.parametrize("level,T", RNG_CASES, ids=[c[0] for c in RNG_CASES])
def test_rng_placement_path_matches_oracle(level, T, fused, spec, oracle_lib):
    """placement_mode='rng' -- the production path of every random-* level the reference trains
    on (overcooked_environment.py:157-173) -- stepped against the oracle: the cells the in-kernel
    PCG32 draw put the items on are read back from the state after the reset and after every
    auto-reset and handed to the oracle env by env (`set_placement`), then the full state
    (and, fused, both observations, reward bits, timestep bits) is compared every step."""
    from gym_comm_amd import compiler
    lv = compiler.compile_level(level, 2, T)
    n, steps, C, radius = 512, 220, 3, 2
    rng = np.random.default_rng(4242)
    mv = scripted_then_random(rng, level, steps, 2, n, nact=4)
    cm = rng.integers(0, C, (steps, 2, n)).astype(np.int32)
    env = _env(lv, n, auto_reset=True, placement_mode="rng", seed=11, specialize_level=spec,
               num_communication=C, fow_radius=radius, waves_per_64=fused)
    assert env.kernel_flavour == ("spec" if spec else "generic")
    counters = {x | (y << 4) for x, y in lv.counters}
    ora = oracle_lib.OracleBatch(lv.blob, n, threads=4)
    cells = _state_cells(env)
    assert set(np.unique(cells).tolist()) <= counters
    ora.set_placement(cells)
    ora.reset()
    assert_snapshots_equal(env.snapshot(), ora.snapshot_all(), "after reset")
    comm = np.zeros((2, n), np.int32)
    resets = 0
    for k in range(steps):
        ctx = "%s step %d" % (level, k)
        if fused:
            a = np.stack([mv[k, 0], cm[k, 0], mv[k, 1], cm[k, 1]]).astype(np.int32)
            o, t, r, d = env.multi_step(torch.from_numpy(a).cuda())
            oo, to, ro, do = ora.multi_step(a, comm, radius, 0, C, auto_reset=False)
            assert np.array_equal(bits(r.cpu().numpy()), bits(ro)), ctx
        else:
            a = mv[k].astype(np.int32)
            r, d, sh = env.step(torch.from_numpy(a).cuda())
            ro, do, sho = ora.step(a, auto_reset=False)
            assert np.array_equal(r.cpu().numpy(), ro), ctx
            assert np.array_equal(bits(sh.cpu().numpy()), bits(sho)), ctx
        dn = d.cpu().numpy()
        assert np.array_equal(dn, do), ctx
        # the envs that finished were re-placed by the kernel: give the oracle the same draw
        cells = _state_cells(env)
        if dn.any():
            assert set(np.unique(cells[:, dn != 0]).tolist()) <= counters
            ora.set_placement(cells)
            ora.reset(dn)
            resets += int(dn.sum())
        hs, os_ = env.snapshot(), ora.snapshot_all()
        assert (hs["error"] == 0).all() and (os_["error"] == 0).all(), ctx
        assert_snapshots_equal(hs, os_, ctx)
        if fused:
            oh, th = o.cpu().numpy(), t.cpu().numpy()
            live = dn == 0
            assert np.array_equal(oh[:, :, live], oo[:, :, live]), ctx
            assert np.array_equal(bits(th)[live], bits(to)[live]), ctx
            for i in np.nonzero(dn)[0]:          # first observation of the re-placed episode
                for v in range(2):
                    eo, et = ora.obs(int(i), v, radius, False, False, C, comm[:, i])
                    assert np.array_equal(oh[v, :, i], eo), (ctx, i, v)
                    assert bits(th)[i] == bits(np.array([et]))[0]
    assert resets >= 4 * n        # every env was re-placed several times


def test_error_flags_match_oracle(oracle_lib):
    """States where the reference itself raises are flagged, identically by kernel and oracle:
    OC_ERR_OOB (an agent proposes a cell outside the map -- random-open-divider_salad_small_cramped
    starts agent-0 in the (0,0) corner) and OC_ERR_ACTION (a move / comm index the reference's
    list / array lookup raises IndexError on); OC_MET_ERRORS counts the env-steps that raised one."""
    from gym_comm_amd import compiler
    from hip_util import momentum_actions
    level, T, n, steps, C = "random-open-divider_salad_small_cramped", 60, 512, 150, 3
    lv = compiler.compile_level(level, 2, T)
    rng = np.random.default_rng(5)
    nc = len(lv.counters)
    place = np.zeros((lv.num_items, n), np.int32)
    for i in range(n):
        pick = rng.choice(nc, size=len(lv.scatter_items), replace=False)
        for k, item in enumerate(lv.scatter_items):
            x, y = lv.counters[pick[k]]
            place[item, i] = x | (y << 4)
    for fused in (False, True):
        ora = oracle_lib.OracleBatch(lv.blob, n, threads=4)
        ora.set_placement(place)
        ora.reset()
        # no auto-reset: the flags are sticky, so "an env-step that raised a flag" (what
        # OC_MET_ERRORS counts) is a step after which an env's flag word differs from before
        env = _env(lv, n, auto_reset=False, placement_mode="host", num_communication=C)
        env.set_placement(torch.from_numpy(place).cuda())
        env.reset()
        acts = momentum_actions(rng, steps, 2, n, nact=4)
        cm = rng.integers(0, C, (steps, 2, n)).astype(np.int32)
        # a sprinkle of invalid indices: move 4, 7, -1; comm C, -2
        bad = rng.random((steps, 2, n)) < 0.01
        acts = np.where(bad, rng.choice([5, 7, -1], size=acts.shape), acts).astype(np.int32)
        cm = np.where(rng.random(cm.shape) < 0.01, rng.choice([C, -2, 99], size=cm.shape), cm).astype(np.int32)
        comm = np.zeros((2, n), np.int32)
        raised = oob = action = 0
        prev = np.zeros(n, np.int32)
        for k in range(steps):
            if fused:
                a = np.stack([acts[k, 0], cm[k, 0], acts[k, 1], cm[k, 1]]).astype(np.int32)
                o, t, r, d = env.multi_step(torch.from_numpy(a).cuda())
                oo, to, ro, do = ora.multi_step(a, comm, 2, 0, C, auto_reset=False)
                assert np.array_equal(env.comm.cpu().numpy(), comm), k
                assert np.array_equal(o.cpu().numpy(), oo), k
                assert np.array_equal(bits(r.cpu().numpy()), bits(ro)), k
            else:
                a = acts[k]
                r, d, sh = env.step(torch.from_numpy(a).cuda())
                ro, do, sho = ora.step(a, auto_reset=False)
                assert np.array_equal(r.cpu().numpy(), ro), k
            dn = d.cpu().numpy()
            assert np.array_equal(dn, do), k
            hs, os_ = env.snapshot(), ora.snapshot_all()
            assert np.array_equal(hs["error"], os_["error"]), k       # the same bits on the same envs
            assert_snapshots_equal(hs, os_, "cramped step %d" % k)     # flagged or not: the defined result
            err = os_["error"]
            raised += int((err != prev).sum())
            prev = err.copy()
        oob, action = int(((prev & 1) != 0).sum()), int(((prev & 4) != 0).sum())
        assert oob > n // 2 and action > n // 4 and int(((prev & 2) != 0).sum()) == 0
        m = env.read_metrics()
        assert m["env_steps"] == n * steps and m["errors"] == raised


DUP_SEEDED = [("cbase_dup_two_tomatoes_small_a2.npz", 2), ("cbase_dup_two_tomatoes_small_a3.npz", 3),
              ("cbase_dup_two_lettuces_salad_a2.npz", 2), ("cbase_dup_two_tomatoes_a3.npz", 3),
              # three of a type: goal counts reach 3 -- the 2-bit field's maximum -- in 29
              # env-steps of the 3-agent run (and in 44 steps of the a2 FIXTURE, replayed above)
              ("cbase_dup_three_tomatoes_a2.npz", 2), ("cbase_dup_three_tomatoes_a3.npz", 3)]


@pytest.mark.parametrize("spec", [False, True], ids=["generic", "spec"])
@pytest.mark.parametrize("name,A", DUP_SEEDED, ids=[c[0][10:-4] for c in DUP_SEEDED])
def test_dup_levels_match_oracle_seeded(name, A, spec, oracle_lib):
    """Levels that repeat a content type, every env with its own action stream, auto-reset on:
    full state compare (multiset objects, world order by key creation, 2-bit goal counts), sparse
    reward, done and raw shaping bits (the set-order `[0]`) against the oracle every step; for two
    agents the fused wrapper step's observations as well."""
    from hip_util import momentum_actions
    z, st = load_golden(os.path.join(os.path.dirname(BASE[0]), name))
    n, steps, T = 1500, 260, 90
    from gym_comm_amd import compiler, levels
    lv = compiler.compile_level(levels.parse_level_text(st["level"], st["level_text"]), A, T)
    assert lv.has_dup
    rng = np.random.default_rng(31 + A)
    acts = momentum_actions(rng, steps, A, n, keep=0.6)
    ora = oracle_lib.OracleBatch(lv.blob, n, threads=4)
    env = _env(lv, n, auto_reset=True, specialize_level=spec)
    assert env.kernel_flavour == ("spec" if spec else "generic") and env.W_state == A + lv.num_items + 4
    acts_d = torch.from_numpy(acts).cuda()
    tot_r = above1 = merges = flagged = at3 = 0
    was = np.zeros(n, bool)
    for k in range(steps):
        r, d, sh = env.step(acts_d[k])
        ro, do, sho = ora.step(acts[k], auto_reset=True)
        hs, os_ = env.snapshot(), ora.snapshot_all()
        ctx = "%s step %d" % (name, k)
        # the same flags on the same envs; a flagged env (three agents, two of them on one cell
        # holding same-named objects while one merges: World.remove takes the wrong one and the
        # reference's store is corrupt from there) is left out until its episode ends
        assert np.array_equal(hs["error"], os_["error"]), ctx
        clean = (os_["error"] == 0) & ~was      # ... including the step that ends it (the auto-reset
        was = os_["error"] != 0                 # clears the flag; that step ran on the corrupt state)
        flagged += int((~clean).sum())
        assert np.array_equal(r.cpu().numpy()[clean], ro[clean]), ctx
        assert np.array_equal(d.cpu().numpy()[clean], do[clean]), ctx
        assert np.array_equal(bits(sh.cpu().numpy())[:, clean], bits(sho)[:, clean]), ctx
        assert_snapshots_equal(hs, os_, ctx, where=clean)
        tot_r += int(ro.sum())
        above1 += int((os_["goal_count"] > 1).any(axis=1).sum())
        at3 += int((os_["goal_count"] >= 3).any(axis=1).sum())
        merges += int((os_["nobj"] < lv.num_items).sum())
    assert tot_r > 0 and above1 > 0 and merges > 0, (tot_r, above1, merges)
    assert at3 == {"cbase_dup_three_tomatoes_a3.npz": 29}.get(name, 0), at3
    # exact flagged fraction (counted with the oracle on the CPU): env-steps spent flagged, of 390 000
    assert flagged == {"cbase_dup_two_tomatoes_small_a3.npz": 72}.get(name, 0), flagged
    if A == 2:
        C, radius = 3, 1
        ora = oracle_lib.OracleBatch(lv.blob, n, threads=4)
        env = _env(lv, n, auto_reset=True, specialize_level=spec, num_communication=C, fow_radius=radius)
        comm = np.zeros((2, n), np.int32)
        mv = np.minimum(acts, 3)
        cm = rng.integers(0, C, (steps, 2, n)).astype(np.int32)
        for k in range(steps):
            a = np.stack([mv[k, 0], cm[k, 0], mv[k, 1], cm[k, 1]]).astype(np.int32)
            o, t, r, d = env.multi_step(torch.from_numpy(a).cuda())
            oo, to, ro, do = ora.multi_step(a, comm, radius, 0, C, auto_reset=True)
            ctx = "%s fused step %d" % (name, k)
            assert np.array_equal(d.cpu().numpy(), do), ctx
            assert np.array_equal(bits(r.cpu().numpy()), bits(ro)), ctx
            assert np.array_equal(o.cpu().numpy(), oo), ctx


@pytest.mark.parametrize("waves", [4, 1], ids=["split", "1wave"])
def test_action_sources_agree(waves, oracle_lib):
    """oc_step_opts: the same (move, comm) actions as rows of `actions`, as int32 [n][2] pairs and
    as int64 [n][2] pairs give identical steps; an int64 value outside int32 is an invalid index
    (OC_ERR_ACTION); a partner drawn in-kernel (alt_rng) plays exactly what oc_random_actions
    draws from the same PCG32 states, and reports it in alt_played."""
    import ctypes
    from gym_comm_amd import compiler, _lib
    lv = compiler.compile_level("open-divider_salad", 2, 60)
    n, steps, C = 1111, 90, 4
    rng = np.random.default_rng(17)
    mv = rng.integers(0, 4, (steps, 2, n)).astype(np.int32)
    cm = rng.integers(0, C, (steps, 2, n)).astype(np.int32)
    envs = [_env(lv, n, num_communication=C, fow_radius=1, auto_reset=True, waves_per_64=waves) for _ in range(4)]
    seeds = torch.randint(0, 2 ** 31 - 1, (n,), dtype=torch.int64).to(torch.int32).cuda()
    r_kernel, r_lib = seeds.clone(), seeds.clone()
    played = torch.zeros((2, n), dtype=torch.int32, device="cuda")
    L = _lib.load()
    for k in range(steps):
        rows = torch.from_numpy(np.stack([mv[k, 0], cm[k, 0], mv[k, 1], cm[k, 1]])).cuda()
        ego32 = torch.from_numpy(np.stack([mv[k, 0], cm[k, 0]], axis=1).copy()).cuda()
        alt32 = torch.from_numpy(np.stack([mv[k, 1], cm[k, 1]], axis=1).copy()).cuda()
        out = [envs[0].multi_step(rows),
               envs[1].multi_step(None, ego_pairs=ego32, alt_pairs=alt32),
               envs[2].multi_step(rows, ego_pairs=ego32.long())]           # int64 ego pairs + partner rows
        ref = [t.clone() for t in out[0]]
        for o in out[1:]:
            for a, b in zip(ref, o):
                assert torch.equal(a, b), k
        assert torch.equal(envs[0].state, envs[1].state) and torch.equal(envs[0].state, envs[2].state), k
        # in-kernel random partner == the library's generator kernel on the same states
        envs[3].multi_step(None, ego_pairs=ego32, alt_rng=r_kernel, alt_played=played)
        exp = torch.zeros((2, n), dtype=torch.int32, device="cuda")
        vp = ctypes.c_void_p
        assert L.oc_random_actions(vp(r_lib.data_ptr()), vp(exp[0].data_ptr()), vp(exp[1].data_ptr()), C, n,
                                   vp(torch.cuda.current_stream().cuda_stream)) == 0
        assert torch.equal(played, exp) and torch.equal(r_kernel, r_lib), k
        assert int(played[0].max()) <= 3 and int(played[1].max()) < C
    assert envs[3].read_metrics()["env_steps"] == n * steps
    # int64 values that do not fit int32 are invalid indices, not truncated ones
    e = _env(lv, n, num_communication=C, auto_reset=False)
    big = torch.zeros((n, 2), dtype=torch.int64, device="cuda")
    big[::3, 0] = (1 << 32) + 1                     # low word 1 = a valid move, were it truncated
    big[1::3, 1] = -(1 << 40)
    e.multi_step(None, ego_pairs=big, alt_pairs=torch.zeros((n, 2), dtype=torch.int64, device="cuda"))
    err = e.snapshot()["error"]
    assert (err[::3] == 4).all() and (err[1::3] == 4).all() and (err[2::3] == 0).all()
    with pytest.raises(ValueError):
        e.multi_step(None, ego_pairs=big)           # no source for the partner


def _stage(env, lv, agents, items, completed):
    """Put every env of the batch into a hand-made state of single-content objects: the twin of
    OracleEnv.debug_set (agents [x, y, held item or -1], items [x, y, state_index], completed [S]
    = completed_subtasks = goal_objects_count of the staged objects)."""
    env.reset()
    w = env.state.cpu().numpy().copy()                 # [A+M+2][n] packed words (include/oc_hip.h)
    A = lv.num_agents
    bits_ = sum(1 << env.subtask_slot[s] for s, c in enumerate(completed) if c)
    w[A + lv.num_items] = bits_
    w[A + lv.num_items + 1] = bits_
    for a, (x, y, h) in enumerate(agents):
        w[a] = (w[a] & ~0xFFF) | x | (y << 4) | ((h + 1) << 8 if h >= 0 else 0)
    for i, (x, y, st) in enumerate(items):
        holder = [a for a, ag in enumerate(agents) if ag[2] == i]
        if holder:
            x, y = agents[holder[0]][0], agents[holder[0]][1]
        w[A + i] = (w[A + i] & ~(0xFF | 0x100 | 0x7000)) | x | (y << 4) | (st << 8) | \
                   (((holder[0] + 1) if holder else 0) << 12)
    env.state.copy_(torch.from_numpy(w).cuda())


@pytest.mark.parametrize("spec", [False, True], ids=["generic", "spec"])
def test_world_remove_alias_corner_staged(spec, oracle_lib):
    """The `World.remove` alias corner (utils/world.py:239-247) on a shipped 3-agent level, staged
    because random play practically never reaches it: agents 0 and 2 stand on ONE cell, each
    holding a Plate, and agent 0 merges its plate with the chopped Tomato on the counter next to
    it.  remove() deletes by (name, location) and takes the LAST match: when agent 0 holds the
    plate that comes first in world order it deletes agent 2's plate instead -- the reference's
    store is corrupt from there on, kernel and oracle both flag OC_ERR_ALIAS; with the plates
    swapped the merge is clean and the full states agree."""
    from gym_comm_amd import compiler
    lv = compiler.compile_level("open-divider_tl", 3, 100)
    assert [t for t, _, _ in lv.items] == [0, 1, 3, 3]                 # Tomato, Lettuce, Plate, Plate
    items = [[0, 4, 1], [6, 1, 0], [6, 5, 0], [5, 6, 0]]               # the Tomato chopped, on Counter (0, 4)
    # ... which means Chop(Tomato) has been rewarded and its goal object counted
    done_ = [1 if (s.kind == compiler.KIND_CHOP and s.args == ("Tomato",)) else 0 for s in lv.subtasks]
    assert sum(done_) == 1
    n = 130
    acts = np.tile(np.array([[2], [4], [4]], np.int32), (1, n))         # agent 0 presses LEFT into the counter
    for hold0, hold2, expect_err in ((2, 3, 2), (3, 2, 0)):
        agents = [[1, 4, hold0], [3, 3, -1], [1, 4, hold2]]
        env = _env(lv, n, auto_reset=False, specialize_level=spec)
        _stage(env, lv, agents, items, done_)
        ora = oracle_lib.OracleBatch(lv.blob, n)
        for i in range(n):
            e = oracle_lib.OracleEnv.__new__(oracle_lib.OracleEnv)
            e.A, e.M, e.S, e._h = ora.A, ora.M, ora.S, ora._handles[i]
            try:
                e.debug_set(agents, items, done_, done_)
            finally:
                e._h = None
        assert_snapshots_equal(env.snapshot(), ora.snapshot_all(), "staged state")
        r, d, sh = env.step(torch.from_numpy(acts).cuda())
        ro, do, sho = ora.step(acts, auto_reset=False)
        hs, os_ = env.snapshot(), ora.snapshot_all()
        assert (hs["error"] == expect_err).all() and (os_["error"] == expect_err).all(), (hold0, hs["error"][:4])
        assert np.array_equal(r.cpu().numpy(), ro) and np.array_equal(d.cpu().numpy(), do)
        assert env.read_metrics()["errors"] == (n if expect_err else 0)
        if not expect_err:
            assert_snapshots_equal(hs, os_, "clean merge")
            assert np.array_equal(bits(sh.cpu().numpy()), bits(sho))
            assert (hs["nobj"] == 3).all() and (hs["agents"][:, 0, 2] == 0).all()   # agent 0 holds Plate-Tomato (group 0)


def test_custom_map_runs_on_the_library_of_its_structure(oracle_lib):
    """A map nobody pre-built, with the recipes / items / agent count / border kind of a shipped
    level: it loads that structure's specialised library (geometry is a run-time argument) and
    steps bit-exactly against the oracle -- base step and fused wrapper step."""
    from gym_comm_amd import compiler, levels, specialize
    text = "---t---\n/     -\n-  l  p\n*     -\n-     -\n--p----\n\nSalad\n\n1 1\n5 4"      # 7 x 6, our own
    lv = compiler.compile_level(levels.parse_level_text("custom-7x6_salad", text), 2, 80)
    assert specialize.spec_key(lv.blob, geometry=False) == \
        specialize.spec_key(compiler.compile_level("open-divider_salad", 2, 80).blob, geometry=False)
    assert not os.path.exists(specialize.spec_lib_path(lv.blob))       # no level library for this map
    n, steps, C = 900, 240, 3
    rng = np.random.default_rng(5)
    acts = scripted_then_random(rng, "custom", steps, 2, n)
    env = _env(lv, n, auto_reset=True, specialize_level="structure", num_communication=C, fow_radius=1)
    assert env.kernel_flavour == "spec" and env._L.oc_is_specialized() == 1
    ora = oracle_lib.OracleBatch(lv.blob, n, threads=4)
    tot = 0
    for k in range(steps):
        r, d, sh = env.step(torch.from_numpy(acts[k]).cuda())
        ro, do, sho = ora.step(acts[k], auto_reset=True)
        ctx = "custom map step %d" % k
        assert np.array_equal(r.cpu().numpy(), ro) and np.array_equal(d.cpu().numpy(), do), ctx
        assert np.array_equal(bits(sh.cpu().numpy()), bits(sho)), ctx
        assert_snapshots_equal(env.snapshot(), ora.snapshot_all(), ctx)
        tot += int(ro.sum())
    assert tot > 0
    env.reset(); ora.reset()
    comm = np.zeros((2, n), np.int32)
    mv = np.minimum(acts, 3)
    cm = rng.integers(0, C, (steps, 2, n)).astype(np.int32)
    for k in range(steps):
        a = np.stack([mv[k, 0], cm[k, 0], mv[k, 1], cm[k, 1]]).astype(np.int32)
        o, t, r, d = env.multi_step(torch.from_numpy(a).cuda())
        oo, to, ro, do = ora.multi_step(a, comm, 1, 0, C, auto_reset=True)
        assert np.array_equal(o.cpu().numpy(), oo) and np.array_equal(bits(r.cpu().numpy()), bits(ro)), k
        assert np.array_equal(bits(t.cpu().numpy()), bits(to)) and np.array_equal(d.cpu().numpy(), do), k


FORCED = ["split=1", "split=2", "split=4", "step_split=1", "step_split=2", "wt=0", "wt=0,split=1",
          "lds=1,block=256", "lds=1", "block=128", "block=256,step_split=1"]


@pytest.mark.parametrize("policy", FORCED)
def test_forced_launch_policies(policy, oracle_lib, monkeypatch):
    """Every launch policy the library can be forced into through its ONE knob (OC_LAUNCH, read at
    every call: csrc/oc_kernels.hip launch_policy) gives the oracle's results: a short seeded compare
    of the fused wrapper step (salad-2) and of the base step (tl-3), specialised libraries, ragged
    batch.  Under the driver's test run, not only in the builder's scratch runs."""
    from gym_comm_amd import compiler
    monkeypatch.setenv("OC_LAUNCH", policy)
    n, steps, C = 1111, 90, 3
    rng = np.random.default_rng(515)
    lv = compiler.compile_level("full-divider_salad", 2, 40)
    env = _env(lv, n, auto_reset=True, num_communication=C, fow_radius=1)
    want = dict(kv.split("=") for kv in policy.split(","))
    if "split" in want and want.get("wt", "1") == "1" and "lds" not in want:
        # (the generic library -- OC_SPECIALIZE=0 -- has no two-way split: it then launches one wave)
        two_way_missing = env.kernel_flavour == "generic" and want["split"] == "2"
        assert env.launch_waves(general=False) == (1 if two_way_missing else int(want["split"]))
    mv = scripted_then_random(rng, "full-divider_salad", steps, 2, n, nact=4)
    cm = rng.integers(0, C, (steps, 2, n)).astype(np.int32)
    acts = np.stack([mv[:, 0], cm[:, 0], mv[:, 1], cm[:, 1]], axis=1).astype(np.int32)
    ora = oracle_lib.OracleBatch(lv.blob, n, threads=4)
    comm = np.zeros((2, n), np.int32)
    a_d = torch.from_numpy(acts).cuda()
    for k in range(steps):
        o, t, r, d = env.multi_step(a_d[k])
        oo, to, ro, do = ora.multi_step(acts[k], comm, 1, 0, C, auto_reset=True)
        assert np.array_equal(o.cpu().numpy(), oo), (policy, k)
        assert np.array_equal(bits(r.cpu().numpy()), bits(ro)) and np.array_equal(d.cpu().numpy(), do), (policy, k)
        assert np.array_equal(bits(t.cpu().numpy()), bits(to)), (policy, k)
    assert_snapshots_equal(env.snapshot(), ora.snapshot_all(), policy)
    assert env.read_metrics()["env_steps"] == n * steps
    lv3 = compiler.compile_level("partial-divider_tl", 3, 40)
    env3 = _env(lv3, n, auto_reset=True)
    acts3 = scripted_then_random(rng, "partial-divider_tl", steps, 3, n)
    ora3 = oracle_lib.OracleBatch(lv3.blob, n, threads=4)
    a3 = torch.from_numpy(acts3).cuda()
    for k in range(steps):
        r, d, sh = env3.step(a3[k])
        ro, do, sho = ora3.step(acts3[k], auto_reset=True)
        clean = ora3.snapshot_all()["error"] == 0
        assert np.array_equal(r.cpu().numpy()[clean], ro[clean]) and np.array_equal(d.cpu().numpy()[clean], do[clean]), (policy, k)
        assert np.array_equal(bits(sh.cpu().numpy())[:, clean], bits(sho)[:, clean]), (policy, k)
    hs, os_ = env3.snapshot(), ora3.snapshot_all()
    assert_snapshots_equal(hs, os_, policy, where=(os_["error"] == 0) & (hs["error"] == 0))

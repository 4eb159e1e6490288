"""The CPU oracle (oracle/oc_oracle.c) against the golden vectors recorded from the
reference itself (tests/golden/make_golden.py).  Bit-exact: integers compared with ==,
fp64 values compared on their raw bits."""
import os

import numpy as np
import pytest

from conftest import compile_for, golden_files, load_golden

# rbase_/rwrap_: random-* levels; cbase_/cwrap_: our own maps (OnionSalad, shared subtasks,
# two Delivery tiles) run through the reference
BASE = golden_files("base_") + golden_files("rbase_") + golden_files("cbase_")
WRAP = golden_files("wrap_") + golden_files("rwrap_") + golden_files("cwrap_")


def _pack(cells):
    return [int(x) | (int(y) << 4) for x, y in cells]


def test_fixtures_present():
    assert len(BASE) >= 10 and len(WRAP) >= 15


@pytest.mark.parametrize("path", BASE, ids=[os.path.basename(p) for p in BASE])
def test_base_env_matches_reference(path, oracle_lib):
    z, st = load_golden(path)
    lv = compile_for(st)
    env = oracle_lib.OracleEnv(lv.blob)
    K = len(z["t"])
    sh_bits = z["shaping_bits"]
    for k in range(K):
        if z["reset_before"][k]:
            if lv.random_placement:
                env.set_placement(_pack(z["placements"][z["pl_index"][k]]))
            env.reset()
        r, d, sh = env.step(z["actions"][k])
        snap = env.snapshot()
        ctx = (os.path.basename(path), k)
        assert r == z["reward"][k], ctx
        assert d == z["done"][k], ctx
        assert snap["t"] == z["t"][k], ctx
        assert snap["nobj"] == z["nobj"][k], ctx
        assert (snap["items"] == z["items"][k][:, :5]).all(), (ctx, snap["items"], z["items"][k])
        assert (snap["order"] == z["order"][k]).all(), ctx
        assert (snap["agents"] == z["agents"][k]).all(), ctx
        assert (snap["completed"] == z["completed"][k]).all(), ctx
        assert (snap["goal_count"] == z["goal_count"][k]).all(), ctx
        assert (sh.view(np.uint64) == sh_bits[k]).all(), (ctx, sh, sh_bits[k].view(np.float64))
        assert env.error == 0, ctx


@pytest.mark.parametrize("path", WRAP, ids=[os.path.basename(p) for p in WRAP])
def test_wrapper_matches_reference(path, oracle_lib):
    z, st = load_golden(path)
    lv = compile_for(st)
    C = st["num_communication"]
    blind = (1 if st["ego_config"]["BLIND"] else 0) | (2 if st["partner_config"]["BLIND"] else 0)
    can_move = (1 if st["ego_config"]["CAN_MOVE"] else 0) | (2 if st["partner_config"]["CAN_MOVE"] else 0)
    b = oracle_lib.OracleBatch(lv.blob, 1)
    comm = np.zeros((2, 1), np.int32)          # per_agent_communications starts as one-hot(0)
    if lv.random_placement:
        b.set_placement(np.array(_pack(z["placements"][0]), np.int32).reshape(-1, 1))
        b.reset()
    # obs right after multi_reset()
    for v in range(2):
        o, ts = b.obs(0, v, st["fow_radius"], (blind >> v) & 1, blind & 1, C, comm[:, 0])
        assert (o == z["reset_obs"][v]).all(), (v, o, z["reset_obs"][v])
        assert np.float64(ts).view(np.uint64) == z["reset_ts_bits"][v]
    K = len(z["done"])
    for k in range(K):
        if z["reset_before"][k]:
            if lv.random_placement:
                b.set_placement(np.array(_pack(z["placements"][z["pl_index"][k]]), np.int32).reshape(-1, 1))
            b.reset()
        act = z["actions"][k].astype(np.int32).reshape(4, 1)
        obs, ts, rew, done = b.multi_step(
            act, comm, st["fow_radius"], blind, C, st["communication_on"], st["ego_led"],
            st["ego_agent_idx"], can_move)
        ctx = (os.path.basename(path), k)
        assert done[0] == z["done"][k], ctx
        assert rew.view(np.uint64)[0] == z["rew_bits"][k], (ctx, rew, z["rew_bits"][k:k + 1].view(np.float64))
        for v in range(2):
            assert (obs[v, :, 0] == z["obs"][k][v]).all(), (ctx, v, obs[v, :, 0], z["obs"][k][v])
            assert ts.view(np.uint64)[0] == z["ts_bits"][k][v], ctx


FOW = golden_files("fow_")


@pytest.mark.parametrize("path", FOW, ids=[os.path.basename(p) for p in FOW])
def test_fow_image_obs_matches_reference(path, oracle_lib):
    z, st = load_golden(path)
    lv = compile_for(st)
    env = oracle_lib.OracleEnv(lv.blob)
    for k in range(len(z["actions"])):
        if z["reset_before"][k]:
            env.reset()
        env.step(z["actions"][k])
        for v in range(2):
            img, hold = env.obs_image(v, st["radius"])
            assert (img == z["maps"][k][v]).all(), (k, v)
            assert (hold == z["holding"][k]).all(), (k, v)
        assert (env.snapshot()["completed"] == z["completed"][k]).all()

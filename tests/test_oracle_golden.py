"""The CPU oracle (oracle/oc_oracle.c) against the golden vectors recorded from the
reference itself (tests/golden/make_golden.py).  Bit-exact: integers compared with ==,
fp64 values compared on their raw bits."""
import os

import numpy as np
import pytest

from conftest import compile_for, golden_files, load_golden

# rbase_/rwrap_: random-* levels; cbase_/cwrap_: our own maps (OnionSalad, shared subtasks,
# two Delivery tiles) run through the reference
BASE = golden_files("base_") + golden_files("rbase_") + golden_files("cbase_") + golden_files("pbase_")
WRAP = golden_files("wrap_") + golden_files("rwrap_") + golden_files("cwrap_") + golden_files("pwrap_")


def _pack(cells):
    return [int(x) | (int(y) << 4) for x, y in cells]


def test_fixtures_present():
    assert len(BASE) >= 10 and len(WRAP) >= 15


@pytest.mark.parametrize("path", BASE, ids=[os.path.basename(p) for p in BASE])
def test_base_env_matches_reference(path, oracle_lib):
    z, st = load_golden(path)
    lv = compile_for(st)
    env = oracle_lib.OracleEnv(lv.blob)
    pl = pi = None
    if lv.random_placement:
        pl = np.array([_pack(row) for row in z["placements"]], np.int32)
        pi = z["pl_index"]
    out = env.replay_base(z["actions"], z["reset_before"], pi, pl)       # the whole tape in one C call
    name = os.path.basename(path)

    def same(a, b, what):
        a, b = np.asarray(a), np.asarray(b)
        if not np.array_equal(a, b):
            k = int(np.argwhere((a != b).reshape(len(a), -1).any(axis=1))[0, 0])
            raise AssertionError("%s: %s differs first at step %d: %s vs %s" % (name, what, k, a[k], b[k]))

    same(out["reward"], z["reward"], "reward")
    same(out["done"], z["done"], "done")
    same(out["t"], z["t"], "t")
    same(out["nobj"], z["nobj"], "number of objects")
    same(out["items"], z["items"][:, :, :5], "items")
    same(out["order"], z["order"], "world order")
    same(out["agents"], z["agents"], "agents")
    same(out["completed"], z["completed"], "completed_subtasks")
    same(out["goal_count"], z["goal_count"], "goal_objects_count")
    same(out["shaping"].view(np.uint64), z["shaping_bits"], "shaping bits")
    assert (out["error"] == 0).all()
    # the step-at-a-time API gives the same answers as the replay entry point
    env2 = oracle_lib.OracleEnv(lv.blob)
    for k in range(min(40, len(z["t"]))):
        if z["reset_before"][k]:
            if lv.random_placement:
                env2.set_placement(pl[pi[k]])
            env2.reset()
        r, d, sh = env2.step(z["actions"][k])
        snap = env2.snapshot()
        assert r == z["reward"][k] and d == z["done"][k]
        assert (snap["items"] == z["items"][k][:, :5]).all() and (snap["agents"] == z["agents"][k]).all()
        assert (sh.view(np.uint64) == z["shaping_bits"][k]).all()


@pytest.mark.parametrize("path", WRAP, ids=[os.path.basename(p) for p in WRAP])
def test_wrapper_matches_reference(path, oracle_lib):
    z, st = load_golden(path)
    lv = compile_for(st)
    C = st["num_communication"]
    blind = (1 if st["ego_config"]["BLIND"] else 0) | (2 if st["partner_config"]["BLIND"] else 0)
    can_move = (1 if st["ego_config"]["CAN_MOVE"] else 0) | (2 if st["partner_config"]["CAN_MOVE"] else 0)
    env = oracle_lib.OracleEnv(lv.blob)
    pl = pi = None
    if lv.random_placement:
        pl = np.array([_pack(row) for row in z["placements"]], np.int32)
        pi = z["pl_index"]
        env.set_placement(pl[0])
        env.reset()
    comm = np.zeros(2, np.int32)               # per_agent_communications starts as one-hot(0)
    for v in range(2):                         # obs right after multi_reset()
        o, ts = env.obs(v, st["fow_radius"], (blind >> v) & 1, blind & 1, C, comm)
        assert (o == z["reset_obs"][v]).all(), (v, o, z["reset_obs"][v])
        assert np.float64(ts).view(np.uint64) == z["reset_ts_bits"][v]
    obs, ts, rew, done = env.replay_wrapper(
        z["actions"], z["reset_before"], st["fow_radius"], blind, C, st["communication_on"],
        st["ego_led"], st["ego_agent_idx"], can_move, pi, pl, comm)
    name = os.path.basename(path)
    assert np.array_equal(done, z["done"]), name
    bad = np.argwhere(rew.view(np.uint64) != z["rew_bits"])
    assert len(bad) == 0, (name, "reward bits differ first at step", int(bad[0, 0]))
    bad = np.argwhere((obs != z["obs"]).reshape(len(obs), -1).any(axis=1))
    assert len(bad) == 0, (name, "obs differ first at step", int(bad[0, 0]))
    assert np.array_equal(ts.view(np.uint64), z["ts_bits"]), name


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


@pytest.mark.parametrize("idx", [0, 1])
def test_oracle_ascii_render_matches_reference_strings(idx, oracle_lib):
    """oracle.render_ascii (the checker's own str(env)) against the reference's recorded
    strings of the two scripted solves (tests/golden/ascii_kat.json)."""
    import json
    from conftest import GOLDEN
    from gym_comm_amd import compiler
    g = json.load(open(os.path.join(GOLDEN, "ascii_kat.json")))[idx]
    lv = compiler.compile_level(g["level"], g["num_agents"], g["T"], subtask_order=g["subtasks"])
    env = oracle_lib.OracleEnv(lv.blob)
    assert oracle_lib.render_ascii(lv.blob, env.snapshot()) == g["reset_str"]
    for acts, exp in zip(g["script"], g["steps"]):
        env.step(np.array(acts, np.int32))
        assert oracle_lib.render_ascii(lv.blob, env.snapshot()) == exp["str"]

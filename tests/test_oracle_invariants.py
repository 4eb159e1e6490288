"""Size-independent properties of the step semantics, checked on the CPU oracle over many
random episodes (the same properties hold for the HIP path by the parity tests):
conservation of items, held objects travel with their holder, unheld objects never sit on
Floor, sticky completed flags, reward bookkeeping, episode termination."""
import numpy as np
import pytest

from hip_util import scripted_then_random

CASES = [("open-divider_tomato", 2, 60), ("full-divider_salad", 2, 80), ("partial-divider_tl", 3, 70),
         ("open-divider_salad", 4, 60), ("random-open-divider_salad_small", 2, 50)]


@pytest.mark.parametrize("level,A,T", CASES, ids=["%s-a%d" % (c[0], c[1]) for c in CASES])
def test_step_invariants(level, A, T, oracle_lib):
    from gym_comm_amd import compiler
    lv = compiler.compile_level(level, A, T)
    n, steps = 256, 400
    rng = np.random.default_rng(9)
    acts = scripted_then_random(rng, level, steps, A, n)
    ora = oracle_lib.OracleBatch(lv.blob, n, threads=4)
    M, S = lv.num_items, lv.num_subtasks
    floor = (lv.cells == 0)
    deliver = np.array([s.kind == 2 for s in lv.subtasks])
    prev = ora.snapshot_all()
    for k in range(steps):
        r, d, sh = ora.step(acts[k], auto_reset=False)
        s = ora.snapshot_all()
        ok = s["error"] == 0
        it, ag = s["items"], s["agents"]
        # every item belongs to exactly one object whose group id is its smallest member
        grp = it[:, :, 3]
        assert (grp <= np.arange(M)[None, :]).all() and (np.take_along_axis(grp, grp, 1) == grp).all()
        assert (s["nobj"] == (grp == np.arange(M)[None, :]).sum(1)).all()
        # members of one object share cell and holder
        same = grp[:, :, None] == grp[:, None, :]
        for f in (0, 1, 4):
            assert (~same | (it[:, :, f][:, :, None] == it[:, :, f][:, None, :])).all()
        # a held object is at its holder's cell; an unheld one is never on Floor
        held = it[:, :, 4] >= 0
        hx = np.take_along_axis(ag[:, :, 0], np.maximum(it[:, :, 4], 0), 1)
        hy = np.take_along_axis(ag[:, :, 1], np.maximum(it[:, :, 4], 0), 1)
        assert ((~held) | ((it[:, :, 0] == hx) & (it[:, :, 1] == hy)))[ok].all()
        assert ((held) | ~floor[it[:, :, 1], it[:, :, 0]])[ok].all()
        # agents stand on Floor; an agent's hold field names a group held by that agent
        assert floor[ag[:, :, 1], ag[:, :, 0]].all()
        for a in range(A):
            hg = ag[:, a, 2]
            has = hg >= 0
            assert (np.take_along_axis(it[:, :, 4], np.maximum(hg, 0)[:, None], 1)[:, 0][has & ok] == a).all()
        # chopped is sticky, t counts steps, completed flags are sticky, rewards are consistent
        assert (it[:, :, 2] >= prev["items"][:, :, 2]).all()
        assert (s["t"] == prev["t"] + 1).all()
        assert (s["completed"] >= prev["completed"]).all()
        newly = (s["goal_count"] > prev["goal_count"])[:, ~deliver].sum(1)
        assert (r >= newly).all() and ((r - newly) % 3 == 0).all()
        assert ((r > 0) <= (s["completed"].sum(1) > 0)).all()
        assert (d == ((s["t"] >= T) | (d & (s["t"] < T)))).all()
        assert (sh >= 0).all() and np.isfinite(sh).all()
        # done envs start a fresh episode
        if d.any():
            ora.reset(d.astype(np.int32))
            s = ora.snapshot_all()
            assert (s["t"][d == 1] == 0).all() and (s["completed"][d == 1] == 0).all()
        prev = s


def test_oracle_set_order_matches_this_interpreter(oracle_lib):
    """World.get_all_object_locs returns list(set(...)) and the shaping terms take element [0]
    (world.py:290-291, overcooked_environment.py:287,374): the oracle restates CPython's tuple
    hash and set table.  Pinned here against the interpreter that recorded the fixtures: random
    location lists (with repeats) of every size the oracle can meet, full iteration order."""
    import random
    rng = random.Random(12)
    for trial in range(4000):
        n = rng.randrange(1, 17)
        locs = [(rng.randrange(16), rng.randrange(16)) for _ in range(n)]
        if rng.random() < 0.5:                       # repeats: the set drops them
            locs += [locs[rng.randrange(len(locs))] for _ in range(rng.randrange(3))]
        locs = locs[:16]
        assert oracle_lib.pyset_order(locs) == list(set(locs)), locs

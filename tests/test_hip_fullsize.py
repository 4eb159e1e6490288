"""-m gpu: BASELINE.json's full batch sizes.  The oracle cannot step 65 536 envs x 3 agents
in seconds, so full size is checked through a size-independent property: env i is fed the
action stream of env (i mod 256); the first 256 envs are compared with the oracle bit for
bit every step, and every other env must equal its residue-class representative (checked
on the GPU).  The opt-in LDS-table kernel variant is checked in a child process."""
import numpy as np
import pytest
import torch

from hip_util import assert_snapshots_equal, bits, scripted_then_random

pytestmark = pytest.mark.gpu

N0 = 256
FULL = [
    ("open-divider_tomato", 2, 4096, True),      # BASELINE configs[1]
    ("full-divider_salad", 2, 32768, True),      # configs[2]
    ("partial-divider_tl", 3, 65536, False),     # configs[3] (no 3-agent obs encoder in the reference)
    ("open-divider_tomato", 2, 131072, True),    # configs[4], one GPU's shard
]


def _tiled_equal(t, n0):
    """Every column block of width n0 equals the first one (t: [..., n])."""
    n = t.shape[-1]
    ref = t[..., :n0]
    return bool((t.reshape(*t.shape[:-1], n // n0, n0) == ref.unsqueeze(-2)).all().item())


@pytest.mark.parametrize("level,A,n,wrapper", FULL, ids=["%s-a%d-n%d" % (f[0], f[1], f[2]) for f in FULL])
def test_full_size_tiles_of_oracle_checked_batch(level, A, n, wrapper, oracle_lib):
    from gym_comm_amd import compiler
    from gym_comm_amd.batched import BatchedOvercooked
    from gym_comm_amd.state import unpack_state
    T, steps, C = 80, 140, 2
    lv = compiler.compile_level(level, A, T)
    rng = np.random.default_rng(4242)
    ora = oracle_lib.OracleBatch(lv.blob, N0, threads=4)
    env = BatchedOvercooked(lv, num_envs=n, num_communication=C, auto_reset=True)
    reps = n // N0
    if wrapper:
        mv = scripted_then_random(rng, level, steps, 2, N0, nact=4)
        cm = rng.integers(0, C, (steps, 2, N0)).astype(np.int32)
        acts0 = np.stack([mv[:, 0], cm[:, 0], mv[:, 1], cm[:, 1]], axis=1).astype(np.int32)
        comm = np.zeros((2, N0), np.int32)
    else:
        acts0 = scripted_then_random(rng, level, steps, A, N0)
    acts = torch.from_numpy(acts0).cuda().repeat(1, 1, reps).contiguous()
    rsum = 0
    for k in range(steps):
        ctx = "%s n=%d step %d" % (level, n, k)
        if wrapper:
            o, t, r, d = env.multi_step(acts[k])
            oo, to, ro, do = ora.multi_step(acts0[k], comm, 2, 0, C, auto_reset=True)
            assert np.array_equal(o[:, :, :N0].cpu().numpy(), oo), ctx
            assert np.array_equal(bits(r[:N0].cpu().numpy()), bits(ro)), ctx
            assert np.array_equal(bits(t[:N0].cpu().numpy()), bits(to)), ctx
            assert np.array_equal(d[:N0].cpu().numpy(), do), ctx
            assert _tiled_equal(o, N0) and _tiled_equal(r.view(torch.int64), N0), ctx
            assert _tiled_equal(d, N0) and _tiled_equal(env.comm, N0), ctx
            rsum += int(env.reward[:N0].sum().item())
        else:
            r, d, sh = env.step(acts[k])
            ro, do, sho = ora.step(acts0[k], auto_reset=True)
            clean = ora.snapshot_all()["error"] == 0
            assert np.array_equal(r[:N0].cpu().numpy()[clean], ro[clean]), ctx
            assert np.array_equal(d[:N0].cpu().numpy()[clean], do[clean]), ctx
            assert np.array_equal(bits(sh[:, :N0].cpu().numpy())[:, clean], bits(sho)[:, clean]), ctx
            assert _tiled_equal(r, N0) and _tiled_equal(d, N0) and _tiled_equal(sh.view(torch.int64), N0), ctx
            rsum += int(r[:N0].sum().item())
        assert _tiled_equal(env.state, N0), ctx
        if k % 10 == 9 or k == steps - 1:
            hs = unpack_state(env.state[:, :N0].cpu().numpy(), lv.num_agents, lv.num_items, lv.num_subtasks, **env.unpack_kw())
            os_ = ora.snapshot_all()
            assert_snapshots_equal(hs, os_, ctx, where=(os_["error"] == 0) & (hs["error"] == 0))
    m = env.read_metrics()
    assert m["env_steps"] == n * steps
    assert m["reward_sum"] == rsum * reps
    assert rsum > 0


def test_ragged_and_tiny_batches(oracle_lib):
    """n = 1, 63, 65, 4097 (tails of every kind) and n = 0 (no launch)."""
    from gym_comm_amd import compiler
    from gym_comm_amd.batched import BatchedOvercooked
    lv = compiler.compile_level("open-divider_tomato", 2, 40)
    for n in (1, 63, 65, 4097):
        rng = np.random.default_rng(n)
        steps = 60
        acts = scripted_then_random(rng, "open-divider_tomato", steps, 2, n)
        ora = oracle_lib.OracleBatch(lv.blob, n)
        env = BatchedOvercooked(lv, num_envs=n, auto_reset=True)
        guard = torch.full((3, n + 64), -7, dtype=torch.int32, device="cuda")   # canaries around outputs
        a_d = torch.from_numpy(acts).cuda()
        for k in range(steps):
            r, d, sh = env.step(a_d[k])
            ro, do, sho = ora.step(acts[k], auto_reset=True)
            assert np.array_equal(r.cpu().numpy(), ro) and np.array_equal(d.cpu().numpy(), do)
            assert np.array_equal(bits(sh.cpu().numpy()), bits(sho))
        assert (guard == -7).all()
        assert env.read_metrics()["env_steps"] == n * steps
    env0 = BatchedOvercooked(lv, num_envs=0)
    env0.step(torch.zeros((2, 0), dtype=torch.int32, device="cuda"))
    assert env0.read_metrics()["env_steps"] == 0


def test_masked_reset_and_argument_errors():
    from gym_comm_amd import compiler
    from gym_comm_amd._lib import OcError
    from gym_comm_amd.batched import BatchedOvercooked
    lv = compiler.compile_level("full-divider_salad", 2, 50)
    n = 300
    env = BatchedOvercooked(lv, num_envs=n, auto_reset=False)
    init = env.state.clone()
    acts = torch.randint(0, 4, (2, n), dtype=torch.int32, device="cuda")
    for _ in range(5):
        env.step(acts)
    moved = env.state.clone()
    assert not torch.equal(moved, init)
    mask = (torch.arange(n, device="cuda") % 3 == 0).to(torch.int32)
    env.reset(mask)
    sel = mask.bool()
    assert torch.equal(env.state[:, sel], init[:, sel]) and torch.equal(env.state[:, ~sel], moved[:, ~sel])
    with pytest.raises(ValueError):
        env.step(acts[:, :10])
    with pytest.raises(ValueError):
        env.step(acts.long())
    with pytest.raises(ValueError):
        env.step(acts.cpu())
    with pytest.raises(OcError):
        BatchedOvercooked(compiler.compile_level("partial-divider_tl", 3, 50), num_envs=8).multi_step(
            torch.zeros((4, 8), dtype=torch.int32, device="cuda"))


_LDS_SNIPPET = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
from gym_comm_amd import compiler
from gym_comm_amd.batched import BatchedOvercooked
from oracle import oracle
from hip_util import scripted_then_random
for level, A, spec in (("open-divider_salad", 2, False), ("partial-divider_tl", 3, True)):
    lv = compiler.compile_level(level, A, 90)
    n, steps = 1500, 150
    rng = np.random.default_rng(5)
    acts = scripted_then_random(rng, level, steps, A, n)
    ora = oracle.OracleBatch(lv.blob, n, threads=4)
    env = BatchedOvercooked(lv, num_envs=n, auto_reset=True, specialize_level=spec)
    a_d = torch.from_numpy(acts).cuda()
    for k in range(steps):
        r, d, sh = env.step(a_d[k])
        ro, do, sho = ora.step(acts[k], auto_reset=True)
        clean = ora.snapshot_all()["error"] == 0
        assert np.array_equal(r.cpu().numpy()[clean], ro[clean]), (level, k)
        assert np.array_equal(d.cpu().numpy()[clean], do[clean]), (level, k)
        assert np.array_equal(sh.cpu().numpy().view(np.uint64)[:, clean], sho.view(np.uint64)[:, clean]), (level, k)
print("LDS-variant ok")
"""


def test_lds_table_variant_in_subprocess():
    """The LDS-staged-table kernel variant (OC_LAUNCH=lds=1, 256-thread workgroups) is not
    the default; checked in a child process at BASELINE sizes (a short in-process compare of
    every forced policy: test_hip_parity.py::test_forced_launch_policies)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, OC_LAUNCH="lds=1,block=256")
    out = subprocess.run([sys.executable, "-c", _LDS_SNIPPET % (ROOT, os.path.join(ROOT, "tests"))],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "LDS-variant ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


_WT_SNIPPET = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
from gym_comm_amd import compiler
from gym_comm_amd.batched import BatchedOvercooked
from oracle import oracle
from hip_util import assert_snapshots_equal
bits = lambda a: a.view(np.uint64)
for level, spec, dt in (("open-divider_tomato", True, torch.int32), ("full-divider_salad", False, torch.int8)):
    lv = compiler.compile_level(level, 2, 60)
    n, steps, C = 1111, 130, 3
    rng = np.random.default_rng(11)
    acts = np.stack([rng.integers(0, 4, (steps, n)), rng.integers(0, C, (steps, n)),
                     rng.integers(0, 4, (steps, n)), rng.integers(0, C, (steps, n))], axis=1).astype(np.int32)
    env = BatchedOvercooked(lv, num_envs=n, num_communication=C, auto_reset=True, specialize_level=spec, obs_dtype=dt)
    ora = oracle.OracleBatch(lv.blob, n, threads=4)
    comm = np.zeros((2, n), np.int32)
    a_d = torch.from_numpy(acts).cuda()
    for k in range(steps):
        o, t, r, d = env.multi_step(a_d[k])
        oo, to, ro, do = ora.multi_step(acts[k], comm, 2, 0, C, auto_reset=True)
        assert np.array_equal(o.cpu().numpy(), oo), (level, k)
        assert np.array_equal(d.cpu().numpy(), do), (level, k)
        assert np.array_equal(bits(r.cpu().numpy()), bits(ro)), (level, k)
        assert np.array_equal(bits(t.cpu().numpy()), bits(to)), (level, k)
    assert_snapshots_equal(env.snapshot(), ora.snapshot_all(), level)
print("store-policy variant ok")
"""


@pytest.mark.parametrize("wt", ["0", "1"])
def test_store_policy_variants_in_subprocess(wt):
    """The launcher picks default-policy stores below 8192 envs and write-through (sc1)
    stores from there on; OC_LAUNCH=wt=0/1 forces one of them.  Both variants
    of the fused kernel (int32 and int8 observation rows) against the oracle at a size where
    the forced variant is NOT the default one (wt=1 at n=1111); wt=0 doubles as the check
    that the override is harmless."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, OC_LAUNCH="wt=" + wt)
    out = subprocess.run([sys.executable, "-c", _WT_SNIPPET % (ROOT, os.path.join(ROOT, "tests"))],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "store-policy variant ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_largest_batch_one_call_can_address(oracle_lib):
    """Maximum size: rows are addressed with 32-bit byte offsets, so one call takes at most
    2^31 / (4 * rows) envs.  8 388 608 envs (1.95 GB of int32 observation rows, 268 MB of
    state) is inside that bound for tomato-2: the first 256 envs against the oracle, every other
    tile of 256 equal to the first; 9 300 000 envs is outside and must be refused before any launch."""
    from gym_comm_amd import compiler
    from gym_comm_amd._lib import OcError
    from gym_comm_amd.batched import BatchedOvercooked
    level, n, T, steps, C = "open-divider_tomato", 1 << 23, 30, 36, 2
    lv = compiler.compile_level(level, 2, T)
    rng = np.random.default_rng(99)
    ora = oracle_lib.OracleBatch(lv.blob, N0, threads=4)
    env = BatchedOvercooked(lv, num_envs=n, num_communication=C, auto_reset=True, track_metrics=True)
    mv = scripted_then_random(rng, level, steps, 2, N0, nact=4)
    cm = rng.integers(0, C, (steps, 2, N0)).astype(np.int32)
    acts0 = np.stack([mv[:, 0], cm[:, 0], mv[:, 1], cm[:, 1]], axis=1).astype(np.int32)
    comm = np.zeros((2, N0), np.int32)
    reps = n // N0
    for k in range(steps):
        ctx = "n=%d step %d" % (n, k)
        a = torch.from_numpy(acts0[k]).cuda().repeat(1, reps).contiguous()
        o, t, r, d = env.multi_step(a)
        oo, to, ro, do = ora.multi_step(acts0[k], comm, 2, 0, C, auto_reset=True)
        assert np.array_equal(o[:, :, :N0].cpu().numpy(), oo), ctx
        assert np.array_equal(bits(r[:N0].cpu().numpy()), bits(ro)), ctx
        assert np.array_equal(d[:N0].cpu().numpy(), do), ctx
        if k % 6 == 5 or k == steps - 1:          # the tile comparison reads 2 GB: not every step
            assert _tiled_equal(o, N0) and _tiled_equal(env.state, N0), ctx
            assert _tiled_equal(d, N0) and _tiled_equal(r.view(torch.int64), N0), ctx
    assert env.read_metrics()["env_steps"] == n * steps
    del env
    torch.cuda.empty_cache()
    with pytest.raises(OcError, match="too large"):
        BatchedOvercooked(lv, num_envs=9_300_000, num_communication=C).multi_step(
            torch.zeros((4, 9_300_000), dtype=torch.int32, device="cuda"))

"""The fused MLP policy kernel (include/oc_policy.h, csrc/oc_policy.hip) against a plain PyTorch
fp32 evaluation of the same network.  Not part of the environment's semantics (no oracle): the
tolerance is that of fp16 operands with fp32 accumulation -- logits within 2e-2 -- and the
sampled / greedy actions are checked through properties (greedy = argmax wherever the fp32 gap is
clear; sample frequencies = softmax; streams advance; ragged batches; every observation type)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _stepped_env(level, n, C, odt, steps=25, seed=3):
    from gym_comm_amd.batched import BatchedOvercooked
    env = BatchedOvercooked(level, num_agents=2, num_envs=n, max_num_timesteps=60, num_communication=C,
                            communication_on=True, fow_radius=2, obs_dtype=odt, auto_reset=True)
    g = torch.Generator(device="cuda").manual_seed(seed)
    hi = torch.tensor([4, C, 4, C], device="cuda").view(4, 1)
    for _ in range(steps):
        env.multi_step((torch.rand((4, n), generator=g, device="cuda") * hi).to(torch.int32))
    return env


def _reference_logits(policy, rows, timestep):
    """The module itself, fp32 (vec_env.MLPPolicy.forward on the rows as they lie)."""
    from gym_comm_amd.vec_env import ObsView
    v = ObsView()
    v.rows, v.timestep = rows, timestep
    with torch.no_grad():
        mv, cm = policy(v)
    return torch.cat([mv, cm], dim=0)


@pytest.mark.parametrize("level,C,n,odt", [("open-divider_tomato", 2, 4096, torch.int32),
                                           ("full-divider_salad", 5, 1000, torch.int32),     # ragged: 31 waves + 8 envs
                                           ("open-divider_tl", 16, 777, torch.int8),
                                           ("full-divider_salad", 1, 33, torch.float32),
                                           ("open-divider_tomato", 3, 1, torch.int32)])
def test_logits_and_greedy_actions_match_the_fp32_module(level, C, n, odt):
    from gym_comm_amd.vec_env import FusedMLPPartner, MLPPolicy
    env = _stepped_env(level, n, C, odt)
    for viewer in (0, 1):
        pol = MLPPolicy(env.S, C, hidden=64, seed=11 + viewer).cuda()
        fused = FusedMLPPartner(pol, sample=False, keep_logits=True)
        rows = env.obs[viewer]
        FusedMLPPartner.launch([fused], [rows], env.timestep)
        ref = _reference_logits(pol, rows, env.timestep)
        got = fused.logits
        assert got.shape == ref.shape == (4 + C, n)
        assert float((got - ref).abs().max()) < 2e-2
        # greedy action = the module's argmax wherever its top-2 gap is clear of the fp16 error
        for lo, hi, col in ((0, 4, 0), (4, 4 + C, 1)):
            blk = ref[lo:hi]
            arg = blk.argmax(dim=0)
            if hi - lo > 1:
                top2 = blk.topk(2, dim=0).values
                clear = (top2[0] - top2[1]) > 5e-2
            else:
                clear = torch.ones(n, dtype=torch.bool, device="cuda")
            assert clear.float().mean() > 0.5
            assert torch.equal(fused.pairs[:, col][clear].long(), arg[clear])
            assert int(fused.pairs[:, col].min()) >= 0 and int(fused.pairs[:, col].max()) < hi - lo


def test_two_players_in_one_launch_equal_two_launches():
    from gym_comm_amd.vec_env import FusedMLPPartner, MLPPolicy
    env = _stepped_env("open-divider_tomato", 2500, 2, torch.int32)
    pols = [MLPPolicy(env.S, 2, seed=s).cuda() for s in (1, 2)]
    a = [FusedMLPPartner(p, sample=True, seed=40 + k, keep_logits=True) for k, p in enumerate(pols)]
    b = [FusedMLPPartner(p, sample=True, seed=40 + k, keep_logits=True) for k, p in enumerate(pols)]
    for _ in range(3):      # the streams advance identically
        FusedMLPPartner.launch(a, [env.obs[0], env.obs[1]], env.timestep)
        for k in range(2):
            FusedMLPPartner.launch([b[k]], [env.obs[k]], env.timestep)
        for k in range(2):
            assert torch.equal(a[k].pairs, b[k].pairs) and torch.equal(a[k].logits, b[k].logits)
            assert torch.equal(a[k]._rng, b[k]._rng)
    assert not torch.equal(a[0].pairs, a[1].pairs)


def test_samples_follow_the_softmax_and_streams_advance():
    """Zero weights, chosen biases: every env has the same logits, so the empirical frequencies
    over 65536 envs must be softmax(b2) (4-sigma bands), for the move head and a 7-way comm head."""
    from gym_comm_amd.vec_env import FusedMLPPartner, MLPPolicy
    C, n = 7, 65536
    env = _stepped_env("open-divider_tomato", n, C, torch.int32, steps=3)
    pol = MLPPolicy(env.S, C, seed=0).cuda()
    with torch.no_grad():
        for t in (pol.w1, pol.b1, pol.wt, pol.w2):
            t.zero_()
        pol.b2.copy_(torch.tensor([0.0, 1.0, -1.0, 0.5, 2.0, 0.0, 0.0, 1.0, -2.0, 0.5, 0.0]).view(-1, 1))
    fused = FusedMLPPartner(pol, sample=True, seed=5)
    before = None
    for rep in range(2):
        FusedMLPPartner.launch([fused], [env.obs[1]], env.timestep)
        if before is not None:
            assert not torch.equal(before[0], fused._rng) and not torch.equal(before[1], fused.pairs)
        before = (fused._rng.clone(), fused.pairs.clone())
        for lo, hi, col in ((0, 4, 0), (4, 4 + C, 1)):
            prob = torch.softmax(pol.b2[lo:hi, 0].detach().double(), 0).cpu().numpy()
            cnt = np.bincount(fused.pairs[:, col].cpu().numpy(), minlength=hi - lo)
            sigma = np.sqrt(n * prob * (1 - prob))
            assert (np.abs(cnt - n * prob) < 4 * sigma + 1).all(), (cnt, n * prob)
    # greedy: the argmax of the biases, everywhere
    greedy = FusedMLPPartner(pol, sample=False)
    FusedMLPPartner.launch([greedy], [env.obs[0]], env.timestep)
    assert (greedy.pairs[:, 0] == 1).all() and (greedy.pairs[:, 1] == 0).all()


def test_fused_policies_drive_the_closed_loop():
    """ego + partner FusedMLPPartner inside OvercookedVecEnv.closed_loop, two launches per step
    (both policies, then the fused env step; `one_launch=False`): the hipGraph replay equals the
    eager loop, and what the env executed is what the policies wrote (a twin env stepped with the
    same pairs)."""
    from types import SimpleNamespace
    from gym_comm_amd.vec_env import FusedMLPPartner, MLPPolicy, OvercookedVecEnv
    from gym_comm_amd.batched import BatchedOvercooked
    arg = SimpleNamespace(level="open-divider_tomato", num_agents=2, max_num_timesteps=40, ego_config={},
                          partner_config={}, num_communication=2, communication_on=True, ego_led=False,
                          fow_radius=2)
    n = 1500

    def make(graph):
        S = 3
        ego = FusedMLPPartner(MLPPolicy(S, 2, seed=1).cuda(), sample=True, seed=7)
        alt = FusedMLPPartner(MLPPolicy(S, 2, seed=2).cuda(), sample=True, seed=8)
        venv = OvercookedVecEnv(arg, n, partner=alt, seed=1)
        venv.reset_tensors()
        return venv, ego, alt, venv.closed_loop(ego, graph=graph, steps=4 if graph else 1, one_launch=False)

    ve, ego_e, alt_e, loop_e = make(False)
    vg, ego_g, alt_g, loop_g = make(True)
    twin = BatchedOvercooked("open-divider_tomato", num_agents=2, num_envs=n, max_num_timesteps=40,
                             num_communication=2, communication_on=True, fow_radius=2, auto_reset=True,
                             episode_stats=True)
    twin.observe()
    for k in range(12):
        loop_e.step()
        twin.multi_step(None, ego_pairs=ego_e.pairs.clone(), alt_pairs=alt_e.pairs.clone())
        assert torch.equal(twin.state, ve._b.state) and torch.equal(twin.obs, ve._b.obs), k
        if k % 4 == 3:
            loop_g.step()
            assert torch.equal(vg._b.state, ve._b.state) and torch.equal(vg._b.obs, ve._b.obs), k
            assert torch.equal(ego_g.pairs, ego_e.pairs) and torch.equal(alt_g._rng, alt_e._rng), k
    assert int(ve._b.done.sum()) >= 0 and ve._b.read_metrics()["env_steps"] == 12 * n


def test_refresh_repacks_in_place_and_a_captured_graph_sees_the_new_weights():
    """After an optimiser step the module's weights change: refresh() re-packs them into the SAME
    device buffers, so a hipGraph captured earlier evaluates the new network."""
    from gym_comm_amd.vec_env import FusedMLPPartner, MLPPolicy
    env = _stepped_env("open-divider_tomato", 640, 2, torch.int32)
    pol = MLPPolicy(env.S, 2, seed=4).cuda()
    fused = FusedMLPPartner(pol, sample=False, keep_logits=True)
    rows = env.obs[0]
    FusedMLPPartner.launch([fused], [rows], env.timestep)       # warm-up (module load) before the capture
    torch.cuda.synchronize()
    ptrs = [t.data_ptr() for t in fused._w]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        FusedMLPPartner.launch([fused], [rows], env.timestep)
    g.replay()
    before = fused.logits.clone()
    assert float((before - _reference_logits(pol, rows, env.timestep)).abs().max()) < 2e-2
    with torch.no_grad():
        pol.w2.mul_(-1.5)
        pol.b1.add_(0.25)
    fused.refresh()
    assert [t.data_ptr() for t in fused._w] == ptrs
    g.replay()
    after = fused.logits.clone()
    assert float((after - _reference_logits(pol, rows, env.timestep)).abs().max()) < 2e-2
    assert float((after - before).abs().max()) > 0.1


@pytest.mark.parametrize("level,C,odt", [("open-divider_tomato", 2, torch.int32),
                                         ("full-divider_salad", 4, torch.int8),      # 39 rows: three k-steps
                                         ("open-divider_tl", 3, torch.float32)],
                         ids=["tomato-c2-int32", "salad-c4-int8", "tl-c3-float32"])
@pytest.mark.parametrize("n", [1500, 40000], ids=["split-launch", "one-wave-launch"])
def test_closed_loop_in_one_launch_equals_policy_kernel_plus_step(n, level, C, odt):
    """oc_step_opts.policy: the step kernel evaluates both MLP policies itself, behind the step
    (a split workgroup: one (viewer, half) pass per wave behind a second barrier; at 40 000
    envs: the lone wave runs all four passes), and leaves the next step's pairs in place.  Same
    network, same packed weights, same random streams => bit-identical to one launch of the policy
    kernel followed by one of the step, eagerly and as a hipGraph."""
    from types import SimpleNamespace
    from gym_comm_amd.vec_env import FusedMLPPartner, MLPPolicy, OvercookedVecEnv
    arg = SimpleNamespace(level=level, num_agents=2, max_num_timesteps=40, ego_config={},
                          partner_config={}, num_communication=C, communication_on=True, ego_led=False,
                          fow_radius=2)

    def make(one_launch, graph):
        venv = OvercookedVecEnv(arg, n, seed=1, obs_dtype=odt)
        S = venv._b.S
        ego = FusedMLPPartner(MLPPolicy(S, C, seed=1).cuda(), sample=True, seed=7)
        venv.partner = alt = FusedMLPPartner(MLPPolicy(S, C, seed=2).cuda(), sample=True, seed=8)
        venv.reset_tensors()
        if one_launch and venv._b.kernel_flavour != "spec":
            pytest.skip("the fused policies are built into the specialised libraries only (OC_SPECIALIZE=0 forces the generic one)")
        loop = venv.closed_loop(ego, graph=graph, steps=3 if graph else 1, one_launch=one_launch)
        assert loop.one_launch == one_launch
        return venv, ego, alt, loop

    two = make(False, False)
    one = make(True, False)
    oneg = make(True, True)
    assert one[0]._b.kernel_flavour == "spec"
    if "OC_LAUNCH" not in os.environ:       # (a forced launch mode runs both cases the same way)
        assert one[0]._b.launch_waves(general=True) == (4 if n <= 32768 else 1)
    executed = []
    for k in range(9):
        two[3].step()
        executed.append((two[1].pairs.clone(), two[2].pairs.clone()))     # what step k executed
        before = (one[1].pairs.clone(), one[2].pairs.clone())              # what step k is about to execute
        one[3].step()
        assert torch.equal(before[0], executed[k][0]) and torch.equal(before[1], executed[k][1]), k
        for name in ("state", "obs", "shaped_reward", "done", "ep_return", "ep_length", "comm"):
            assert torch.equal(getattr(two[0]._b, name), getattr(one[0]._b, name)), (k, name)
        if k % 3 == 2:
            oneg[3].step()
            for name in ("state", "obs", "shaped_reward", "done", "ep_return"):
                assert torch.equal(getattr(oneg[0]._b, name), getattr(one[0]._b, name)), (k, name)
            assert torch.equal(oneg[1].pairs, one[1].pairs) and torch.equal(oneg[2]._rng, one[2]._rng), k
    assert two[0]._b.read_metrics()["env_steps"] == one[0]._b.read_metrics()["env_steps"] == 9 * n
    # a reset in mid-run: the one-launch loop's pending pairs were sampled for the
    # pre-reset observations; reset_tensors() re-primes every live loop from the NEW observations
    # (with the draw the two-launch form uses next), so the two forms stay bit-identical
    for v in (two, one, oneg):
        v[0].reset_tensors()
    for k in range(6):
        two[3].step()
        one[3].step()
        for name in ("state", "obs", "shaped_reward", "done", "ep_return", "ep_length", "comm"):
            assert torch.equal(getattr(two[0]._b, name), getattr(one[0]._b, name)), ("after reset", k, name)
        if k % 3 == 2:
            oneg[3].step()
            for name in ("state", "obs", "shaped_reward", "done", "ep_return"):
                assert torch.equal(getattr(oneg[0]._b, name), getattr(one[0]._b, name)), ("after reset", k, name)


def test_closed_loop_built_before_the_first_reset_is_primed_by_it():
    """``closed_loop()`` before ``reset_tensors()``: the pairs primed from the all-zero observations
    are replaced by the reset (documented order: closed_loop, reset_tensors, step ...)."""
    from types import SimpleNamespace
    from gym_comm_amd.vec_env import FusedMLPPartner, MLPPolicy, OvercookedVecEnv
    arg = SimpleNamespace(level="open-divider_tomato", num_agents=2, max_num_timesteps=40, ego_config={},
                          partner_config={}, num_communication=2, communication_on=True, ego_led=False, fow_radius=2)
    n = 2048

    def make(loop_first):
        venv = OvercookedVecEnv(arg, n, seed=1)
        ego = FusedMLPPartner(MLPPolicy(venv._b.S, 2, seed=1).cuda(), sample=True, seed=7)
        venv.partner = FusedMLPPartner(MLPPolicy(venv._b.S, 2, seed=2).cuda(), sample=True, seed=8)
        if venv._b.kernel_flavour != "spec":
            pytest.skip("the fused policies are built into the specialised libraries only")
        if loop_first:
            loop = venv.closed_loop(ego, graph=False, one_launch=True)
            venv.reset_tensors()
        else:
            venv.reset_tensors()
            loop = venv.closed_loop(ego, graph=False, one_launch=True)
        return venv, loop

    a, b = make(True), make(False)
    for k in range(5):
        a[1].step()
        b[1].step()
        for name in ("state", "obs", "shaped_reward", "done"):
            assert torch.equal(getattr(a[0]._b, name), getattr(b[0]._b, name)), (k, name)

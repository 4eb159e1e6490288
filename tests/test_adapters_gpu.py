"""-m gpu: the single-env adapters that mirror the reference's object API
(gym_comm_amd.envs.OvercookedEnvironment / OvercookedMultiEnv) against golden vectors."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu

NAV = [(0, 1), (0, -1), (-1, 0), (1, 0), (0, 0)]


def _arglist(level, A, T, **kw):
    d = dict(level=level, num_agents=A, max_num_timesteps=T, max_num_subtasks=14,
             ego_config={"ALLERGIC": False, "BLIND": False, "CAN_MOVE": True},
             partner_config={"ALLERGIC": False, "BLIND": False, "CAN_MOVE": True},
             num_communication=2, communication_on=True, ego_led=False, fow_radius=2)
    d.update(kw)
    return SimpleNamespace(**d)


@pytest.mark.parametrize("idx", [0, 1])
def test_base_env_adapter_matches_reference_strings(idx):
    """step(action_dict) -> (reward, done, info), attrs and str(env) on the scripted solves."""
    from gym_comm_amd.envs import OvercookedEnvironment
    g = json.load(open(os.path.join(GOLDEN, "ascii_kat.json")))[idx]
    env = OvercookedEnvironment(_arglist(g["level"], g["num_agents"], g["T"]), subtask_order=g["subtasks"])
    assert str(env) == g["reset_str"]
    assert env.all_subtasks == ["%s(%s)" % (k, ", ".join(a)) for k, a in g["subtasks"]]
    tot = 0
    for acts, exp in zip(g["script"], g["steps"]):
        r, d, info = env.step({"agent-%d" % a: NAV[c] for a, c in enumerate(acts)})
        assert (r, d) == (exp["reward"], exp["done"])
        assert str(env) == exp["str"]
        assert [a.get_holding() for a in env.sim_agents] == exp["holding"]
        assert [list(a.location) for a in env.sim_agents] == exp["locations"]
        assert env.termination_info == exp["termination_info"] and env.successful == exp["successful"]
        assert info["t"] == env.t and info["done"] == d and isinstance(r, int) and isinstance(d, bool)
        tot += r
    assert tot in (5, 9)
    with pytest.raises(KeyError):
        env.step({"agent-0": (0, 1)})                     # missing agent name (reference :217)
    frame = env.render_rgb(scale=16)
    assert frame.shape == (env.world.height * 16, env.world.width * 16, 3) and frame.dtype == np.uint8
    assert frame.reshape(-1, 3).std(axis=0).min() > 0          # not a blank image
    env.reset()
    assert env.t == 0 and str(env) == g["reset_str"]


def test_base_env_adapter_shaping_bits():
    from gym_comm_amd.envs import OvercookedEnvironment
    z, st = load_golden(os.path.join(GOLDEN, "base_open-divider_tomato_a2.npz"))
    env = OvercookedEnvironment(_arglist(st["level"], 2, st["max_num_timesteps"]), subtask_order=st["subtasks"])
    for k in range(23):
        r, d, info = env.step({"agent-%d" % a: NAV[int(c)] for a, c in enumerate(z["actions"][k])})
        sh = np.array([info["agent_0_reward_shaping"], info["agent_1_reward_shaping"]])
        assert (sh.view(np.uint64) == z["shaping_bits"][k]).all()
        assert env.completed_subtasks == list(z["completed"][k])
        assert env.goal_objects_count == list(z["goal_count"][k])


class _TapePartner:
    """Stand-in for a pantheonrl Agent: replays recorded (move, comm) actions."""

    def __init__(self, tape):
        self.tape, self.k, self.updates = tape, 0, []

    def get_action(self, obs):
        a = self.tape[self.k]
        self.k += 1
        return np.array(a)

    def update(self, reward, done):
        self.updates.append((reward, done))


def test_multi_env_adapter_matches_reference_wrapper():
    """multi_step / multi_reset return values (dtypes, shapes, values, reward bits) and the
    MultiAgentEnv-style step()/reset() plumbing (multiagentenv.py:172-243)."""
    from gym_comm_amd.envs import OvercookedMultiEnv
    z, st = load_golden(os.path.join(GOLDEN, "wrap_tomato_r2.npz"))
    arg = _arglist(st["level"], 2, st["max_num_timesteps"], fow_radius=st["fow_radius"])
    env = OvercookedMultiEnv(arg, subtask_order=st["subtasks"])
    S, C = len(st["subtasks"]), 2
    sizes = [("object_encodings_x", 4), ("object_encodings_y", 4), ("state_encodings", 4),
             ("is_hidden", 4), ("completed_subtasks", S), ("agent1_location", 2),
             ("agent2_location", 2), ("agent_is_holding", 2), ("agent1_comm", C), ("agent2_comm", C)]

    def flat(o):
        return np.concatenate([np.asarray(o[k]).astype(np.int64).reshape(-1) for k, _ in sizes])

    o0, o1 = env.multi_reset()
    assert (flat(o0) == z["reset_obs"][0]).all() and (flat(o1) == z["reset_obs"][1]).all()
    assert list(o0.keys())[0] == "timestep" and o0["timestep"].dtype == np.float64
    for k, n in sizes:
        assert o0[k].shape == (n,), k
        assert str(o0[k].dtype) == st["obs_dtypes"][k], (k, o0[k].dtype)
    K = 400
    for k in range(K):
        if z["reset_before"][k] and k:
            env.multi_reset()
        a = z["actions"][k]
        (o0, o1), (r0, r1), d, info = env.multi_step((int(a[0]), int(a[1])), (int(a[2]), int(a[3])))
        assert r0 == r1 and np.float64(r0).view(np.uint64) == z["rew_bits"][k]
        assert d == bool(z["done"][k]) and info == {}
        assert (flat(o0) == z["obs"][k][0]).all() and (flat(o1) == z["obs"][k][1]).all()
        assert np.float64(o0["timestep"][0]).view(np.uint64) == z["ts_bits"][k][0]
    assert env.base_env.t == int(round(float(o0["timestep"][0]) * st["max_num_timesteps"]))
    with pytest.raises(IndexError):
        env.multi_step((4, 0), (0, 0))

    # ego-perspective step()/reset() with a tape-driven partner
    env2 = OvercookedMultiEnv(arg, subtask_order=st["subtasks"])
    partner = _TapePartner([(int(a[2]), int(a[3])) for a in z["actions"]])
    env2.add_partner_agent(partner)
    ob = env2.reset()
    ob = getattr(ob, "obs", ob)
    assert (flat(ob) == z["reset_obs"][0]).all()
    prev = None
    for k in range(150):
        a = z["actions"][k]
        ob, rew, done, info = env2.step(np.array([int(a[0]), int(a[1])]))
        ob = getattr(ob, "obs", ob)
        assert np.float64(rew).view(np.uint64) == z["rew_bits"][k] or k == 0
        assert info["_partnerid"] == [0]
        if done:
            assert (flat(ob) == prev).all()                # previous ego obs on done (:206-208)
            break
        assert (flat(ob) == z["obs"][k][0]).all()
        prev = flat(ob)
    assert len(partner.updates) > 0


def test_spaces_description():
    from gym_comm_amd.envs import make_spaces
    obs, act = make_spaces(7, 7, 3, 2)
    spaces = obs.spaces if hasattr(obs, "spaces") else obs
    keys = list(spaces.keys()) if hasattr(spaces, "keys") else list(spaces.spaces.keys())
    assert keys == ["timestep", "object_encodings_x", "object_encodings_y", "state_encodings",
                    "is_hidden", "completed_subtasks", "agent1_location", "agent2_location",
                    "agent_is_holding", "agent1_comm", "agent2_comm"]
    assert list(getattr(act, "nvec")) == [4, 2]


def test_sprite_frame_geometry_and_draw_order(tmp_path):
    """render_frame(): compositing of misc/game/game.py:55-158 with sprites read from a
    directory.  The reference's PNGs do not travel; solid-colour stand-ins (one colour per
    sprite name, the Plate half transparent) pin the geometry: tile fills and outline, an
    unheld item at tile size, what an agent holds in the bottom-right quarter, plated
    contents at 0.7 of the plate, draw order tiles < items < agents < held."""
    from PIL import Image
    from hip_util import KAT1_TOMATO
    from gym_comm_amd.envs import OvercookedEnvironment
    rgb = {"delivery": (10, 10, 10), "cutboard": (20, 20, 20), "Plate": (200, 200, 255),
           "FreshTomato": (255, 0, 0), "ChoppedTomato": (128, 0, 0), "FreshLettuce": (0, 255, 0),
           "ChoppedLettuce": (0, 128, 0), "agent-blue": (0, 0, 255), "agent-magenta": (255, 0, 255)}
    for name, c in rgb.items():
        a = 128 if name == "Plate" else 255
        Image.new("RGBA", (8, 8), c + (a,)).save(str(tmp_path / (name + ".png")))
    env = OvercookedEnvironment(_arglist("open-divider_tomato", 2, 100))
    S = 40
    f0 = env.render_frame(str(tmp_path), scale=S)
    assert f0.shape == (7 * S, 7 * S, 3) and f0.dtype == np.uint8
    px = lambda f, x, y, dx=S // 2, dy=S // 2: tuple(int(v) for v in f[y * S + dy, x * S + dx])
    assert px(f0, 1, 2) == (245, 230, 210)                      # Floor (Color.FLOOR)
    assert px(f0, 0, 0, 0, 0) == (114, 93, 51)                  # Counter outline
    assert px(f0, 0, 0) == (220, 170, 110)                      # Counter fill
    a0, a1 = env.sim_agents[0].location, env.sim_agents[1].location
    assert px(f0, *a0) == rgb["agent-blue"] and px(f0, *a1) == rgb["agent-magenta"]
    items = {o.full_name: o.location for o in env.world.objects_in_order}
    assert px(f0, *items["FreshTomato"]) == rgb["FreshTomato"]
    plate = items["Plate"]
    blend = tuple((c * 128 + b * 127 + 127) // 255 for c, b in zip(rgb["Plate"], (220, 170, 110)))
    assert px(f0, *plate) == blend                               # half-transparent sprite over the Counter
    # play the scripted solve until agent-1 holds Plate-ChoppedTomato (after step 16)
    for a0c, a1c in KAT1_TOMATO[:16]:
        env.step({"agent-0": NAV[a0c], "agent-1": NAV[a1c]})
    ag = env.sim_agents[1]
    assert ag.holding is not None and ag.holding.full_name == "Plate-ChoppedTomato"
    f = env.render_frame(str(tmp_path), scale=S)
    x, y = ag.location
    assert px(f, x, y, S // 4, S // 4) == rgb["agent-magenta"]   # top-left quarter: the agent
    # bottom-right quarter: the held plate (0.5 tile, blended over the agent), its contents at 0.7 of that
    held_plate = tuple((c * 128 + b * 127 + 127) // 255 for c, b in zip(rgb["Plate"], rgb["agent-magenta"]))
    assert px(f, x, y, S // 2 + 1, S // 2 + 1) == held_plate     # plate corner, outside the contents inset
    assert px(f, x, y, 3 * S // 4, 3 * S // 4) == rgb["ChoppedTomato"]
    # without a sprite directory the sprite-free frame comes back
    assert env.render_frame(None, scale=8).shape == (56, 56, 3)

"""Import harness for the *reference* (kyle-he/gym-comm at /root/reference).

TEST INFRASTRUCTURE, build-container only.  This module is used solely by
``make_golden.py`` to run the reference's own Python code and record golden
input/output vectors.  It is never imported by the product package, by
``bench.py`` or by any ``-m gpu`` test: ``/root/reference`` does not exist on the
GPU box.  Only the ``.npz`` fixtures it produces travel.

What it does:
  * writes tiny stand-in modules for ``gym``, ``termcolor`` and ``wandb`` into a
    temp dir (they are absent from this image and only provide names the
    reference imports at module level; none of their behaviour is on the path);
  * registers a bare ``pantheonrl.common.multiagentenv.SimultaneousEnv`` so that
    ``gym_comm.envs.overcooked_env`` can be imported without stable_baselines3;
  * chdirs to the reference root (level files are opened CWD-relative,
    gym_cooking/envs/overcooked_environment.py:103).

Nothing from the reference is copied; the stubs are a few lines of our own.
"""
import contextlib
import io
import os
import struct
import sys
import tempfile
import types
from types import SimpleNamespace

REF_ROOT = os.environ.get("OC_REFERENCE_ROOT", "/root/reference")

_STUBS = {
    "gym/__init__.py": (
        "class Env(object):\n    pass\n"
        "class Wrapper(Env):\n    def __init__(self, env=None):\n        self.env = env\n"
        "from . import error, spaces, utils, envs\n"
    ),
    "gym/error.py": "",
    "gym/spaces.py": (
        "class Space(object):\n"
        "    def __init__(self, *a, **k):\n        self.args = a; self.kwargs = k\n"
        "class Box(Space): pass\n"
        "class Discrete(Space): pass\n"
        "class MultiBinary(Space): pass\n"
        "class MultiDiscrete(Space): pass\n"
        "class Tuple(Space): pass\n"
        "class Dict(Space): pass\n"
    ),
    "gym/utils/__init__.py": "from . import seeding\n",
    "gym/utils/seeding.py": "",
    "gym/envs/__init__.py": "from . import registration\n",
    "gym/envs/registration.py": "def register(*a, **k):\n    pass\n",
    "termcolor.py": "def colored(s, *a, **k):\n    return s\n",
    "wandb.py": "",
}

_ready = False


def setup():
    """Make the reference importable.  Idempotent."""
    global _ready
    if _ready:
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference checkout not found at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    stub_dir = tempfile.mkdtemp(prefix="oc_ref_stubs_")
    for rel, src in _STUBS.items():
        p = os.path.join(stub_dir, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(src)
    sys.path[:0] = [stub_dir, REF_ROOT, os.path.join(REF_ROOT, "gym_cooking")]
    for name in ("pantheonrl", "pantheonrl.common", "pantheonrl.common.multiagentenv"):
        sys.modules[name] = types.ModuleType(name)

    class SimultaneousEnv(object):
        def __init__(self, partners=None):
            pass

    sys.modules["pantheonrl.common.multiagentenv"].SimultaneousEnv = SimultaneousEnv
    os.chdir(REF_ROOT)
    _ready = True


def use_custom_levels(levels):
    """Make extra level files visible to the reference without touching its tree: the
    reference opens 'gym_cooking/utils/levels/<name>.txt' relative to the CWD
    (overcooked_environment.py:103), so chdir into a temp dir that has that path."""
    setup()
    root = tempfile.mkdtemp(prefix="oc_custom_levels_")
    d = os.path.join(root, "gym_cooking", "utils", "levels")
    os.makedirs(d)
    for name, text in levels.items():
        with open(os.path.join(d, name + ".txt"), "w") as f:
            f.write(text)
    os.chdir(root)
    return root


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def make_arglist(level, num_agents, T, ego_config=None, partner_config=None,
                 num_communication=2, communication_on=True, ego_led=False,
                 fow_radius=2, play=False):
    cfg = {"ALLERGIC": False, "BLIND": False, "CAN_MOVE": True}
    e = dict(cfg); e.update(ego_config or {})
    p = dict(cfg); p.update(partner_config or {})
    return SimpleNamespace(
        level=level, num_agents=num_agents, max_num_timesteps=T,
        max_num_subtasks=14, seed=1, with_image_obs=False,
        beta=1.3, alpha=0.01, tau=2, cap=75, main_cap=100,
        play=play, record=False,
        model1=None, model2=None, model3=None, model4=None,
        ego_config=e, partner_config=p,
        num_communication=num_communication, communication_on=communication_on,
        ego_led=ego_led, fow_radius=fow_radius)


def base_env(arglist):
    setup()
    from gym_cooking.envs.overcooked_environment import OvercookedEnvironment
    with quiet():
        return OvercookedEnvironment(arglist)


def wrapper_env(arglist, ego_agent_idx=0):
    setup()
    from gym_comm.envs.overcooked_env import OvercookedMultiEnv
    with quiet():
        return OvercookedMultiEnv(arglist, ego_agent_idx=ego_agent_idx)


def f64_bits(x):
    """Raw IEEE-754 bits of a Python float/int as an unsigned 64-bit int."""
    return struct.unpack("<Q", struct.pack("<d", float(x)))[0]


# --------------------------------------------------------------------------
# canonical snapshots of the reference's object graph
# --------------------------------------------------------------------------
TYPE_ID = {"Tomato": 0, "Lettuce": 1, "Onion": 2, "Plate": 3}
CELL_ID = {"Floor": 0, "Counter": 1, "Cutboard": 2, "Delivery": 3}


def world_objects(env):
    """Movable Object instances in world.objects iteration order."""
    from gym_cooking.utils.core import Object
    out = []
    for lst in env.world.objects.values():
        for o in lst:
            if isinstance(o, Object):
                out.append(o)
    return out


def base_items(env):
    """Identity list of base contents (Tomato/Lettuce/Plate instances) right after
    reset, in world iteration order.  Index in this list = item id."""
    items = []
    for o in world_objects(env):
        for c in o.contents:
            items.append(c)
    return items


def static_tables(env):
    """Static per-level facts as plain Python data."""
    w, h = env.world.width, env.world.height
    cells = [[-1] * w for _ in range(h)]
    from gym_cooking.utils.core import GridSquare
    for lst in env.world.objects.values():
        for o in lst:
            if isinstance(o, GridSquare):
                x, y = o.location
                cells[y][x] = CELL_ID[o.name]
    dist = [[0] * (w * h) for _ in range(w * h)]
    for ay in range(h):
        for ax in range(w):
            for by in range(h):
                for bx in range(w):
                    dist[ay * w + ax][by * w + bx] = int(
                        env.world.get_path_distance_between((ax, ay), (bx, by)))
    items = []
    for o in world_objects(env):
        for c in o.contents:
            items.append([TYPE_ID[c.name], o.location[0], o.location[1]])
    subtasks = [[st.name, list(st.args)] for st in env.all_subtasks]
    return {
        "width": w, "height": h, "cells": cells, "dist": dist,
        "items": items,
        "agents": [list(a.location) for a in env.sim_agents],
        "subtasks": subtasks,
        "recipes": [type(r).__name__ for r in env.recipes],
        "recipe0_names": [c.name for c in env.recipes[0].contents],
        "object_keys": list(env.world.objects.keys()),
    }


def snapshot(env, items):
    """Dynamic state after a step/reset, as flat int lists.

    items  : per base item [x, y, state_index, group, holder_agent]
             group = smallest item id sharing its Object; holder = -1 if not held.
    order  : groups in world.objects iteration order, -1 padded to len(items)
    agents : per agent [x, y, holding_group(-1 none)]
    """
    objs = world_objects(env)
    ident = {id(c): i for i, c in enumerate(items)}
    group_of_obj = {}
    rows = [None] * len(items)
    for o in objs:
        ids = [ident[id(c)] for c in o.contents]
        g = min(ids)
        group_of_obj[id(o)] = g
        holder = -1
        for ai, a in enumerate(env.sim_agents):
            if a.holding is o:
                holder = ai
        for c in o.contents:
            st = getattr(c, "state_index", 0)
            rows[ident[id(c)]] = [o.location[0], o.location[1], int(st), g,
                                  holder, 1 if o.is_held else 0]
    order = [group_of_obj[id(o)] for o in objs]
    order += [-1] * (len(items) - len(order))
    agents = []
    for a in env.sim_agents:
        hg = -1
        if a.holding is not None:
            hg = group_of_obj.get(id(a.holding), -2)
        agents.append([a.location[0], a.location[1], hg])
    return rows, order, agents, len(objs)

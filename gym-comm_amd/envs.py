"""Single-env adapters with the reference's object API, backed by the HIP path.

``OvercookedEnvironment``  mirrors gym_cooking.envs.OvercookedEnvironment
                           (gym_cooking/envs/overcooked_environment.py:37-241)
``OvercookedMultiEnv``     mirrors gym_comm.envs.OvercookedMultiEnv
                           (gym_comm/envs/overcooked_env.py:15-297), including the
                           pantheonrl SimultaneousEnv / MultiAgentEnv step/reset plumbing
                           (pantheonrl/common/multiagentenv.py:172-243,383-409)

so ``trainer.py`` / ``tester.py`` / ``episode_recorder.py`` keep working when the gym
registry id ``OvercookedMultiCommEnv-v0`` points here.  Every step is
one kernel launch on a 1-env batch; the batched API (batched.BatchedOvercooked, vec_env)
is where the throughput is.

Deviations, all of them deliberate and stated:
  * a missing ``CAN_MOVE`` key in ego_config / partner_config means True (the reference
    raises KeyError on the first step, overcooked_env.py:254; most of its own
    spread/*.json configs lack the key);
  * the path's many ``print`` calls are dropped;
  * the subtask order is canonical (compiler.canonical_subtasks) unless
    ``subtask_order=`` is given -- the reference's own order changes with PYTHONHASHSEED.
"""
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import levels as L
from .batched import BatchedOvercooked, obs_layout

NAV_ACTIONS = [(0, 1), (0, -1), (-1, 0), (1, 0)]        # utils/world.py:16
_CODE_OF = {(0, 1): 0, (0, -1): 1, (-1, 0): 2, (1, 0): 3, (0, 0): 4}


def _arg(arglist, name, default=None):
    if isinstance(arglist, dict):
        return arglist.get(name, default)
    return getattr(arglist, name, default)


class _Content:
    """A Food / Plate inside a held object (gym_cooking/utils/core.py:270-377)."""

    def __init__(self, type_id, state_index):
        self.name = L.TYPE_NAME[type_id]
        self.state_index = state_index
        if type_id == L.PLATE:
            self.full_name = "Plate"
        else:
            self.full_name = ("Chopped" if state_index else "Fresh") + self.name

    def __str__(self):
        if self.name == "Plate":
            return "p"
        return "%d%s" % (self.state_index + 1, self.name[0].lower())


class _Object:
    """Read-only view of an Object (core.py:149-237)."""

    def __init__(self, contents, location, is_held):
        self.contents = contents
        self.location = location
        self.is_held = is_held
        srt = sorted(contents, key=lambda c: c.name)
        self.name = "-".join(c.name for c in srt)
        self.full_name = "-".join(c.full_name for c in srt)

    def __str__(self):
        return "-".join(str(c) for c in sorted(self.contents, key=lambda c: c.name))


class _SimAgent:
    """Read-only view of a SimAgent (utils/agent.py:258-314)."""

    def __init__(self, idx, config):
        self.name = "agent-%d" % idx
        self.location = (0, 0)
        self.holding = None
        self.action = (0, 0)
        self.config = config

    def __str__(self):
        return self.name[-1]

    def get_holding(self):
        return "None" if self.holding is None else self.holding.full_name


class OvercookedEnvironment:
    """``OvercookedEnvironment(arglist)``: reset() -> None;
    step({"agent-i": (dx, dy)}) -> (reward: int, done: bool, info: dict)."""

    def __init__(self, arglist, device="cuda", subtask_order=None, placements=None,
                 level_dir=None, _batch=None, _index=0, _view=False):
        """``_batch`` / ``_index`` / ``_view``: bind to env ``_index`` of an existing
        ``BatchedOvercooked`` instead of creating a 1-env batch; with ``_view`` the object is a
        read-only mirror of that env (what ``EpisodeRecorder`` / ``ParallelEpisodeRecorder`` read:
        ``t``, ``world``, ``sim_agents``, ``completed_subtasks``, ``display()``, ``str()``,
        episode_recorder.py:15-85) and the batch is stepped by its owner."""
        self.arglist = arglist
        ego = dict(_arg(arglist, "ego_config", {}) or {})
        partner = dict(_arg(arglist, "partner_config", {}) or {})
        self._index, self._view = int(_index), bool(_view)
        self._b = _batch or BatchedOvercooked(
            _arg(arglist, "level"), num_agents=_arg(arglist, "num_agents"), num_envs=1,
            max_num_timesteps=_arg(arglist, "max_num_timesteps", 100),
            max_num_subtasks=_arg(arglist, "max_num_subtasks", 14),
            ego_config=ego, partner_config=partner,
            num_communication=_arg(arglist, "num_communication", 10),
            communication_on=_arg(arglist, "communication_on", False),
            ego_led=_arg(arglist, "ego_led", False), fow_radius=_arg(arglist, "fow_radius", 2),
            device=device, subtask_order=subtask_order, placements=placements,
            level_dir=level_dir, auto_reset=False, track_metrics=False,
            play=bool(_arg(arglist, "play", False)))     # interact.py:44-47,52,66-67
        lv = self._b.level
        self._stale, self._pending_words = False, None
        self._v_world = SimpleNamespace(width=lv.width, height=lv.height,
                                        perimeter=2 * (lv.width + lv.height), arglist=arglist)
        self._v_world.get_object_list = lambda: list(self._v_world.objects_in_order)
        self.recipes = list(lv.recipes)
        self.all_subtasks = [s.name for s in lv.subtasks]
        self._v_sim_agents = [_SimAgent(i, self._b.ego_config if i == 0 else self._b.partner_config)
                              for i in range(lv.num_agents)]
        self.termination_info = ""
        self.successful = False
        self.collisions = []
        self.agent_actions = {}
        if self._view:
            self._stale = True              # materialised from the device on first read
        else:
            self._act = torch.zeros((lv.num_agents, 1), dtype=torch.int32, device=self._b.device)
            self.reset()

    def mark_dirty(self):
        """The batch has moved on: rebuild the mirror from the device at the next read."""
        self._pending_words = None
        self._stale = True

    # -- state mirror ------------------------------------------------------------
    # The reference's attributes (t, completed_subtasks, goal_objects_count, world, sim_agents,
    # rep) are rebuilt from the packed state words only when somebody reads them: the wrapper's
    # multi_step() hands over the words it fetched anyway (_mark_stale) and a training loop
    # that never looks at base_env pays nothing for the mirror.
    _LAZY = ("t", "completed_subtasks", "goal_objects_count", "rep", "sim_agents", "world", "_error")

    def __getattr__(self, name):
        if name in OvercookedEnvironment._LAZY:
            self._materialize()
            try:
                return self.__dict__["_v_" + name]
            except KeyError:
                pass
        raise AttributeError(name)

    def _mark_stale(self, words):
        """words: host copy of this env's packed state, int32 [A+M+2][1]."""
        self._pending_words = np.array(words, copy=True)
        self._stale = True

    def _materialize(self):
        if self.__dict__.get("_stale"):
            self._stale = False
            self._sync(self._pending_words)

    def _sync(self, words=None):
        from .state import unpack_state
        lv = self._b.level
        self._stale = False
        if words is None:                   # one column of the [A+M+2][n] state tensor
            i = self._index
            words = self._b.state[:, i:i + 1].cpu().numpy()
        s = unpack_state(words, lv.num_agents, lv.num_items, lv.num_subtasks, **self._b.unpack_kw())
        self._v_t = int(s["t"][0])
        self._v_completed_subtasks = [int(v) for v in s["completed"][0]]
        self._v_goal_objects_count = [int(v) for v in s["goal_count"][0]]
        items, agents, order = s["items"][0], s["agents"][0], s["order"][0]
        objs = {}
        for g in order:
            if g < 0:
                continue
            members = [i for i in range(lv.num_items) if items[i][3] == g]
            contents = [_Content(lv.items[i][0], int(items[i][2])) for i in members]
            objs[int(g)] = _Object(contents, (int(items[g][0]), int(items[g][1])), items[g][4] >= 0)
        self._v_world.objects_in_order = [objs[int(g)] for g in order if g >= 0]
        self._v_world.objects = {}          # name -> list, as World.objects (utils/world.py:21), movable objects only
        for o in self._v_world.objects_in_order:
            self._v_world.objects.setdefault(o.name, []).append(o)
        for a, ag in enumerate(self._v_sim_agents):
            ag.location = (int(agents[a][0]), int(agents[a][1]))
            ag.holding = objs.get(int(agents[a][2])) if agents[a][2] >= 0 else None
        self._v__error = int(s["error"][0])
        self._v_rep = self._render()

    def _render(self):
        """world.update_display + agents (overcooked_environment.py:442-446, world.py:38-48)."""
        lv = self._b.level
        glyph = [" ", "-", "/", "*"]
        rep = [[glyph[int(lv.cells[y][x])] for x in range(lv.width)] for y in range(lv.height)]
        objs = self.world.objects_in_order
        for o in objs:
            rep[o.location[1]][o.location[0]] = str(o)
        for o in objs:
            if o.name == "Tomato":
                rep[o.location[1]][o.location[0]] = str(o)
        for ag in self.sim_agents:
            rep[ag.location[1]][ag.location[0]] = str(ag)
        return rep

    def __str__(self):
        return "\n".join("".join(c + " " for c in row) for row in self.rep)

    # sprite-free frame for video logging (episode_recorder.py feeds pygame sprite frames of
    # misc/game/gameimage.py to wandb; those PNG assets are not reproduced here)
    _TILE_RGB = {0: (232, 220, 196), 1: (150, 108, 72), 2: (120, 120, 128), 3: (96, 176, 96)}
    _TYPE_RGB = {"Tomato": (214, 40, 40), "Lettuce": (60, 170, 60), "Onion": (200, 160, 210),
                 "Plate": (250, 250, 250)}
    _AGENT_RGB = [(40, 90, 200), (230, 140, 30), (150, 60, 170), (40, 170, 170)]

    def render_rgb(self, scale=24):
        """H*scale x W*scale x 3 uint8 image of the current state: tiles, items (a dot per
        content, darker when chopped) and agents.  Not a pixel copy of the reference's
        pygame renderer."""
        lv = self._b.level
        img = np.zeros((lv.height * scale, lv.width * scale, 3), np.uint8)
        for y in range(lv.height):
            for x in range(lv.width):
                img[y * scale:(y + 1) * scale, x * scale:(x + 1) * scale] = self._TILE_RGB[int(lv.cells[y][x])]
        q = max(2, scale // 4)
        for ag_i, ag in enumerate(self.sim_agents):
            x, y = ag.location
            img[y * scale + 2:(y + 1) * scale - 2, x * scale + 2:(x + 1) * scale - 2] = \
                self._AGENT_RGB[ag_i % len(self._AGENT_RGB)]
        for o in self.world.objects_in_order:
            x, y = o.location
            for k, c in enumerate(sorted(o.contents, key=lambda c: c.name)):
                col = np.array(self._TYPE_RGB[c.name], np.int32)
                if c.name != "Plate" and c.state_index:
                    col = col * 2 // 3
                ox = x * scale + 2 + (k % 2) * (q + 2)
                oy = y * scale + 2 + (k // 2) * (q + 2)
                img[oy:oy + q, ox:ox + q] = col.astype(np.uint8)
        return img

    # -- sprite frames ------------------------------------------------
    # misc/game/utils.py:4-9 (Color) and utils/agent.py:25 (COLORS)
    _GAME_RGB = {"FLOOR": (245, 230, 210), "COUNTER": (220, 170, 110),
                 "COUNTER_BORDER": (114, 93, 51), "DELIVERY": (96, 96, 96)}
    _AGENT_COLOR = ["blue", "magenta", "yellow", "green"]
    _sprite_cache = {}

    @classmethod
    def _sprite(cls, sprite_dir, name, size):
        """RGBA sprite `name`.png scaled to size x size (nearest neighbour, as
        pygame.transform.scale), cached."""
        key = (sprite_dir, name, size)
        if key not in cls._sprite_cache:
            from PIL import Image
            with Image.open(os.path.join(sprite_dir, name + ".png")) as im:
                cls._sprite_cache[key] = np.asarray(
                    im.convert("RGBA").resize((size, size), Image.NEAREST), dtype=np.uint8)
        return cls._sprite_cache[key]

    @staticmethod
    def _blit(img, sprite, x, y):
        """Alpha-blend an RGBA sprite onto the RGB frame with its top-left corner at (x, y),
        clipped to the frame (pygame Surface.blit of a per-pixel-alpha image)."""
        h, w = sprite.shape[:2]
        x0, y0 = max(x, 0), max(y, 0)
        x1, y1 = min(x + w, img.shape[1]), min(y + h, img.shape[0])
        if x1 <= x0 or y1 <= y0:
            return
        sp = sprite[y0 - y:y1 - y, x0 - x:x1 - x].astype(np.int32)
        a = sp[..., 3:4]
        dst = img[y0:y1, x0:x1].astype(np.int32)
        img[y0:y1, x0:x1] = ((sp[..., :3] * a + dst * (255 - a) + 127) // 255).astype(np.uint8)

    def render_frame(self, sprite_dir=None, scale=80):
        """(H*scale) x (W*scale) x 3 uint8 frame composed as the reference's pygame renderer
        does (misc/game/game.py:55-158, Game.on_render): floor fill; Counter / Delivery /
        Cutboard tiles; objects nobody holds (a Plate at tile size with its plated contents at
        0.7 of it, centred); agents, and what each holds in the bottom-right quarter (0.5 of the
        tile, plated contents at 0.7 of that).  The PNG sprites are the reference's own files
        (gym_cooking/misc/game/graphics/*.png, not redistributed here): pass their directory as
        `sprite_dir` or in OC_SPRITE_DIR.  Without sprites the result is render_rgb(scale).
        pygame is absent from the build image, so pixel parity with the reference's frames is
        unpinned; geometry and draw order follow the cited lines."""
        sprite_dir = sprite_dir or os.environ.get("OC_SPRITE_DIR")
        if not sprite_dir or not os.path.isdir(sprite_dir):
            return self.render_rgb(scale)
        lv = self._b.level
        C = self._GAME_RGB
        img = np.empty((lv.height * scale, lv.width * scale, 3), np.uint8)
        img[:] = C["FLOOR"]                                                   # game.py:56
        hold = int(0.5 * scale)                     # holding_size   (game.py:31,37)
        cont = int(0.7 * scale)                     # container_size (game.py:32,38)
        hold_cont = int(0.7 * hold)                 # holding_container_size (game.py:39)
        for y in range(lv.height):                                           # draw_gridsquare, :79-96
            for x in range(lv.width):
                t = int(lv.cells[y][x])
                if t == L.FLOOR:
                    continue
                px, py = x * scale, y * scale
                tile = img[py:py + scale, px:px + scale]
                if t == L.DELIVERY:
                    tile[:] = C["DELIVERY"]
                    self._blit(img, self._sprite(sprite_dir, "delivery", scale), px, py)
                else:
                    tile[:] = C["COUNTER"]
                    tile[0, :] = tile[-1, :] = C["COUNTER_BORDER"]            # 1-pixel outline
                    tile[:, 0] = tile[:, -1] = C["COUNTER_BORDER"]
                    if t == L.CUTBOARD:
                        self._blit(img, self._sprite(sprite_dir, "cutboard", scale), px, py)

        def draw_object(o, size, inner, at, inner_at):                       # draw_object / draw_agent_object
            names = [c.full_name for c in sorted(o.contents, key=lambda c: c.name)]
            if "Plate" in names:
                self._blit(img, self._sprite(sprite_dir, "Plate", size), *at)
                rest = [n for n in names if n != "Plate"]
                if rest:
                    self._blit(img, self._sprite(sprite_dir, "-".join(rest), inner), *inner_at)
            else:
                self._blit(img, self._sprite(sprite_dir, "-".join(names), size), *at)

        for o in self.world.objects_in_order:                                # :70-72
            if not o.is_held:
                x, y = o.location
                off = int(scale * (1 - 0.7) / 2)                              # container_location, :142-145
                draw_object(o, scale, cont, (x * scale, y * scale), (x * scale + off, y * scale + off))
        for i, ag in enumerate(self.sim_agents):                             # draw_agent, :103-106
            x, y = ag.location
            self._blit(img, self._sprite(sprite_dir, "agent-" + self._AGENT_COLOR[i % 4], scale),
                       x * scale, y * scale)
            if ag.holding is not None:
                off = int(scale * (1 - 0.5))                                  # holding_location, :137-140
                off2 = int(scale * ((1 - 0.5) + (1 - 0.7) / 2 * 0.5))         # holding_container_location, :147-151
                draw_object(ag.holding, hold, hold_cont, (x * scale + off, y * scale + off),
                            (x * scale + off2, y * scale + off2))
        return img

    def display(self):
        self._materialize()
        self._v_rep = self._render()

    def get_agent_names(self):
        return [a.name for a in self.sim_agents]

    # -- API ---------------------------------------------------------------------
    def reset(self):
        if self._view:
            raise RuntimeError("this is a view of env %d of a batch; reset the batch" % self._index)
        self._b.reset()
        self.termination_info = ""
        self.successful = False
        self.collisions = []
        self._sync()

    def step(self, action_dict):
        if self._view:
            raise RuntimeError("this is a view of env %d of a batch; step the batch" % self._index)
        for a, ag in enumerate(self.sim_agents):
            act = tuple(int(v) for v in action_dict[ag.name])        # KeyError like the reference (:217)
            if act not in _CODE_OF:
                raise ValueError("action %r is not a unit move" % (act,))
            ag.action = act
            self._act[a, 0] = _CODE_OF[act]
        reward, done, shaping = self._b.step(self._act, auto_reset=False)
        r, d = int(reward.item()), bool(done.item())
        sh = shaping.cpu().numpy()
        self._sync()
        T = self._b.level.max_num_timesteps
        if d and T and self.t >= T:
            self.termination_info = "Terminating because passed {} timesteps".format(T)
            self.successful = False
        elif d:
            self.termination_info = "Terminating because all deliveries were completed"
            self.successful = True
        else:
            self.termination_info = ""
            self.successful = False
        info = {"t": self.t, "repr_obs": self.rep, "done": d,
                "termination_info": self.termination_info,
                "agent_0_reward_shaping": float(sh[0, 0]),
                "agent_1_reward_shaping": float(sh[1, 0])}
        return r, d, info

    def close(self):
        return


# ---------------------------------------------------------------------------------
# spaces: gym / gymnasium if importable, else a plain description with the same facts
# ---------------------------------------------------------------------------------
class SpaceSpec:
    """Stand-in for gym.spaces.* when neither gym nor gymnasium is installed."""

    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)

    def __repr__(self):
        return "SpaceSpec(%s, %s)" % (self.kind, {k: v for k, v in self.__dict__.items() if k != "kind"})


def _spaces_module():
    for name in ("gym", "gymnasium"):
        try:
            mod = __import__(name)
            return mod.spaces
        except Exception:
            continue
    return None


def make_spaces(width, height, S, C):
    """observation_space / action_space exactly as declared at overcooked_env.py:41-85
    (including the y-bound quirk low=high=height at :62)."""
    sp = _spaces_module()
    if sp is not None:
        loc = sp.Box(low=np.array([0, 0]), high=np.array([width - 1, height - 1]), dtype=np.float32)
        obs = sp.Dict({
            "timestep": sp.Box(low=0.0, high=1.0, shape=(1,), dtype=np.float32),
            "object_encodings_x": sp.Box(low=-1 * width, high=width, shape=(4,), dtype=np.int64),
            "object_encodings_y": sp.Box(low=height, high=height, shape=(4,), dtype=np.int64),
            "state_encodings": sp.MultiBinary(4), "is_hidden": sp.MultiBinary(4),
            "completed_subtasks": sp.MultiBinary(S),
            "agent1_location": loc, "agent2_location": loc,
            "agent_is_holding": sp.MultiBinary(2),
            "agent1_comm": sp.MultiBinary(C), "agent2_comm": sp.MultiBinary(C)})
        return obs, sp.MultiDiscrete([4, C])
    box = lambda lo, hi, shape, dt: SpaceSpec("Box", low=lo, high=hi, shape=shape, dtype=dt)
    mb = lambda n: SpaceSpec("MultiBinary", n=n)
    loc = box([0, 0], [width - 1, height - 1], (2,), "float32")
    obs = SpaceSpec("Dict", spaces={
        "timestep": box(0.0, 1.0, (1,), "float32"),
        "object_encodings_x": box(-width, width, (4,), "int64"),
        "object_encodings_y": box(height, height, (4,), "int64"),
        "state_encodings": mb(4), "is_hidden": mb(4), "completed_subtasks": mb(S),
        "agent1_location": loc, "agent2_location": loc, "agent_is_holding": mb(2),
        "agent1_comm": mb(C), "agent2_comm": mb(C)})
    return obs, SpaceSpec("MultiDiscrete", nvec=[4, C])


# ---------------------------------------------------------------------------------
# pantheonrl plumbing: use the real SimultaneousEnv when importable, else a stand-in
# that replays MultiAgentEnv.step/reset (multiagentenv.py:149-243) for 2 players
# ---------------------------------------------------------------------------------
def _simultaneous_base():
    try:
        from pantheonrl.common.multiagentenv import SimultaneousEnv
        return SimultaneousEnv
    except Exception:
        return _SimultaneousEnvStandIn


class _SimultaneousEnvStandIn:
    def __init__(self, partners=None):
        self.ego_ind = 0
        self.n_players = 2
        self.partners = [list(partners)] if partners else [[]]
        self.partnerids = [0]
        self._players = tuple()
        self._obs = tuple()
        self._old_ego_obs = None
        self.should_update = [False]
        self.total_rews = [0, 0]
        self.ego_moved = False
        self.ego_extractor = lambda ob: ob

    def getDummyEnv(self, player_num):
        return self

    def set_ego_extractor(self, fn):
        self.ego_extractor = fn

    def add_partner_agent(self, agent, player_num=1):
        if player_num == self.ego_ind:
            raise ValueError("Ego agent is not set by the environment")
        self.partners[0].append(agent)

    def set_partnerid(self, agent_id, player_num=1):
        assert 0 <= agent_id < len(self.partners[0])
        self.partnerids[0] = agent_id

    def resample_partner(self):
        self.partnerids = [(self.partnerids[0] + 1) % len(self.partners[0])]   # round robin (:117-124)

    def _get_actions(self, players, obs, ego_act=None):
        actions = []
        for player, ob in zip(players, obs):
            if player == self.ego_ind:
                actions.append(ego_act)
            else:
                agent = self.partners[0][self.partnerids[0]]
                actions.append(agent.get_action(ob))
                if not self.should_update[0]:
                    agent.update(self.total_rews[player], False)
                self.should_update[0] = True
        return np.array(actions)

    def _update_players(self, rews, done):
        if self.should_update[0]:
            self.partners[0][self.partnerids[0]].update(rews[1], done)
        for i in range(2):
            self.total_rews[i] += rews[i]

    def n_step(self, actions):
        (o0, o1), r, d, i = self.multi_step(actions[0], actions[1])
        return (0, 1), (o0, o1), r, d, i

    def n_reset(self):
        o0, o1 = self.multi_reset()
        return (0, 1), (o0, o1)

    def step(self, action):
        ego_rew = 0.0
        acts = self._get_actions(self._players, self._obs, action)
        self._players, self._obs, rews, done, info = self.n_step(acts)
        info["_partnerid"] = self.partnerids
        self._update_players(rews, done)
        ego_rew += rews[self.ego_ind] if self.ego_moved else self.total_rews[self.ego_ind]
        self.ego_moved = True
        if done:
            return self.ego_extractor(self._old_ego_obs), ego_rew, done, info   # previous obs (:206-208)
        ego_obs = self._obs[self._players.index(self.ego_ind)]
        self._old_ego_obs = ego_obs
        return self.ego_extractor(ego_obs), ego_rew, done, info

    def reset(self):
        self.resample_partner()
        self._players, self._obs = self.n_reset()
        self.should_update = [False]
        self.total_rews = [0, 0]
        self.ego_moved = False
        ego_obs = self._obs[self._players.index(self.ego_ind)]
        self._old_ego_obs = ego_obs
        return self.ego_extractor(ego_obs)


def _make_multi_env_class():
    Base = _simultaneous_base()

    class OvercookedMultiEnv(Base):
        """``OvercookedMultiEnv(arglist, ego_agent_idx=0, baselines=False)``."""

        def __init__(self, arglist, ego_agent_idx=0, baselines=False, device="cuda",
                     subtask_order=None, placements=None, level_dir=None):
            super().__init__()
            self.arglist = arglist
            self.ego_agent_idx = ego_agent_idx
            if baselines:
                np.random.seed(0)                       # overcooked_env.py:34
            if _arg(arglist, "num_agents") != 2:
                raise ValueError("the gym_comm wrapper drives exactly 2 agents "
                                 "(overcooked_env.py:250-262 sets agent-0 and agent-1 only)")
            self._b = BatchedOvercooked(
                _arg(arglist, "level"), num_agents=2, num_envs=1,
                max_num_timesteps=_arg(arglist, "max_num_timesteps", 100),
                max_num_subtasks=_arg(arglist, "max_num_subtasks", 14),
                ego_config=dict(_arg(arglist, "ego_config", {}) or {}),
                partner_config=dict(_arg(arglist, "partner_config", {}) or {}),
                num_communication=_arg(arglist, "num_communication", 10),
                communication_on=_arg(arglist, "communication_on", False),
                ego_led=_arg(arglist, "ego_led", False),
                fow_radius=_arg(arglist, "fow_radius", 2), ego_agent_idx=ego_agent_idx,
                device=device, subtask_order=subtask_order, placements=placements,
                level_dir=level_dir, auto_reset=False, track_metrics=False,
                play=bool(_arg(arglist, "play", False)))
            self.base_env = OvercookedEnvironment(arglist, _batch=self._b)
            lv = self._b.level
            self.lA = len(NAV_ACTIONS)
            self.observation_space, self.action_space = make_spaces(
                lv.width, lv.height, lv.num_subtasks, self._b.C)
            self._layout = obs_layout(lv.num_subtasks, self._b.C)
            self._act = torch.zeros((4, 1), dtype=torch.int32, device=self._b.device)
            self._act_host = torch.zeros((4, 1), dtype=torch.int32).pin_memory()
            self.multi_reset()

        @property
        def per_agent_communications(self):
            c = self._b.comm.cpu().numpy()[:, 0]
            out = []
            for k in range(2):
                v = np.zeros(self._b.C)
                if c[k] >= 0:
                    v[c[k]] = 1
                out.append(v)
            return out

        def _obs_dicts(self, obs, ts):
            """obs [2][F][1], ts [1]: device tensors or their host copies (BatchedOvercooked.fetch)."""
            if isinstance(obs, torch.Tensor):
                obs, ts = obs.cpu().numpy(), ts.cpu().numpy()
            o = obs[:, :, 0].astype(np.int64)       # a fresh array per call: the keys are views of it
            t = ts
            ego_blind = bool(self._b.ego_config["BLIND"])
            out = []
            for v in range(2):
                d = {"timestep": np.array((t[0],))}
                for k, (a, b) in self._layout.items():
                    d[k] = o[v, a:b]
                if not ego_blind:                       # bool pair unless the EGO is BLIND (:154)
                    d["agent_is_holding"] = d["agent_is_holding"].astype(bool)
                d["agent1_comm"] = d["agent1_comm"].astype(np.float64)
                d["agent2_comm"] = d["agent2_comm"].astype(np.float64)
                out.append(d)
            return out[0], out[1]

        def get_observation2(self, agent_idx, radius=1000):
            if radius != self._b._obs_cfg.fow_radius:
                saved = self._b._obs_cfg.fow_radius
                self._b._obs_cfg.fow_radius = int(radius)
                try:
                    obs, ts = self._b.observe()
                    return self._obs_dicts(obs, ts)[agent_idx]
                finally:
                    self._b._obs_cfg.fow_radius = saved
            obs, ts = self._b.observe()
            return self._obs_dicts(obs, ts)[agent_idx]

        def multi_step(self, ego_action, alt_action):
            for v in (ego_action[0], alt_action[0]):
                if not 0 <= int(v) < 4:
                    raise IndexError("list index out of range")        # NAV_ACTIONS[idx] (:248)
            # one pinned host->device copy in, one launch, ONE device->host copy out
            h = self._act_host
            h[0, 0], h[1, 0] = int(ego_action[0]), int(ego_action[1])
            h[2, 0], h[3, 0] = int(alt_action[0]), int(alt_action[1])
            self._act.copy_(h, non_blocking=True)
            self._b.multi_step(self._act, auto_reset=False)
            out = self._b.fetch()
            o0, o1 = self._obs_dicts(out["obs"], out["timestep"])
            r = float(out["shaped_reward"][0])
            self.base_env._mark_stale(out["state"])     # the mirror is rebuilt only if somebody reads it
            return (o0, o1), (r, r), bool(out["done"][0]), {}

        def multi_reset(self):
            self.base_env.reset()
            obs, ts = self._b.observe()
            return self._obs_dicts(obs, ts)

        def cost_fn(self):
            return 1

        def render(self, mode="human", close=False):
            print(str(self.base_env))

    return OvercookedMultiEnv


OvercookedMultiEnv = _make_multi_env_class()


def register(gym_module=None):
    """Point the reference's registry ids at these classes
    (gym_comm/__init__.py:3-5, gym_cooking/__init__.py:3-6)."""
    if gym_module is None:
        import gym as gym_module
    reg = gym_module.envs.registration.register
    reg(id="OvercookedMultiCommEnv-v0", entry_point="gym_comm_amd.envs:OvercookedMultiEnv")
    reg(id="overcookedEnv-v0", entry_point="gym_comm_amd.envs:OvercookedEnvironment")

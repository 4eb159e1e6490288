"""Level compiler: LevelSpec -> static tables -> the int32 "level blob" that the HIP
library (and the test oracle) consume.

Everything the reference recomputes on every ``reset()`` but that never changes for
a given level is computed here once, on the host
(gym_cooking/envs/overcooked_environment.py:180-206):

  * the cell-type grid and item / agent start positions (``load_level`` :100-178);
  * the subtask list (``run_recipes`` :452-459 -> recipe_planner/stripsworld.py:24-79
    over the action sets of recipe_planner/recipe.py:5-97) together with the goal
    object of every subtask (navigation_planner/utils.py:161-209);
  * the path-distance table behind ``World.get_path_distance_between``
    (utils/world.py:61-93,114-131), which reward shaping reads.

Blob layout: see include/oc_hip.h (``OC_LV_*`` word offsets).
"""
from collections import deque
from dataclasses import dataclass, field
from itertools import combinations
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import levels as L

MAGIC = 0x4F434C56          # 'OCLV'
VERSION = 2
HEADER_WORDS = 32
MAX_AGENTS = 4
MAX_ITEMS = 8
MAX_SUBTASKS = 32
MAX_CELLS = 128

KIND_CHOP, KIND_MERGE, KIND_DELIVER = 0, 1, 2
KIND_NAME = ["Chop", "Merge", "Deliver"]
NAV_ACTIONS = [(0, 1), (0, -1), (-1, 0), (1, 0)]      # utils/world.py:16


# --------------------------------------------------------------------------
# recipes -> subtasks (own restatement of the STRIPS planner's result)
# --------------------------------------------------------------------------
RECIPE_INGREDIENTS = {          # recipe_planner/recipe.py:68-97
    "SimpleTomato": ["Tomato"],
    "SimpleLettuce": ["Lettuce"],
    "Salad": ["Tomato", "Lettuce"],
    "OnionSalad": ["Tomato", "Lettuce", "Onion"],
}


def _join(names):
    return "-".join(sorted(names))


class _Action:
    __slots__ = ("kind", "args", "pre", "post")

    def __init__(self, kind, args, pre, post):
        self.kind, self.args, self.pre, self.post = kind, tuple(args), list(pre), list(post)

    @property
    def key(self):
        return (self.kind, self.args)

    def __repr__(self):
        return "%s(%s)" % (self.kind, ", ".join(self.args))


def _holding_predicate(names):
    """The STRIPS predicate under which the planner tracks a pile of contents: a bare Plate is
    ``Fresh(Plate)`` (plates are never chopped), a single food must be ``Chopped``, anything
    with two or more contents is ``Merged(<names joined>)``.  Every Merge precondition of the
    reference's action table is this function of the two argument piles (the explicit
    preconditions at recipe.py:24-25,46-47,58-59,64-65 and the default ``[Chopped(arg1),
    Merged(arg2)]`` of recipe_planner/utils.py:131-141 all agree with it)."""
    names = tuple(names)
    if names == ("Plate",):
        return "Fresh(Plate)"
    if len(names) == 1:
        return "Chopped(%s)" % names[0]
    return "Merged(%s)" % _join(names)


def recipe_actions(recipe_name: str) -> Tuple[Dict[tuple, _Action], str]:
    """Action set of one recipe and its goal predicate, derived from the rule behind the
    reference's table (recipe_planner/recipe.py:5-66) rather than from its construction order:

      * ``Get(Plate)`` and, per ingredient, ``Get`` and ``Chop``;
      * ``Deliver`` of the fully plated dish;
      * a ``Merge(a, b)`` for every way the table lets two piles meet, each with the
        preconditions ``_holding_predicate`` gives the piles and ``Merged(a + b)`` as effect:
        any pile of ingredients U goes onto the bare Plate; and for |U| >= 2 a single
        ingredient x meets the rest of U -- bare, with the rest already plated, or with x itself
        already plated.  In that last family the table names a lone food first and the plated x
        second, but the plated x first when the rest is a pile.

    Actions are identified by (name, args) (recipe_planner/utils.py:83-89); since the
    preconditions are a function of the args, no two constructions of the same key disagree."""
    if recipe_name not in RECIPE_INGREDIENTS:
        raise ValueError("unknown recipe class %r" % recipe_name)
    ings = sorted(RECIPE_INGREDIENTS[recipe_name])
    plate = ("Plate",)
    acts: Dict[tuple, _Action] = {}

    def put(kind, args, pre, post):
        acts.setdefault((kind, tuple(args)), _Action(kind, args, pre, post))

    for x in ["Plate"] + ings:
        put("Get", (x,), ["None"], ["Fresh(%s)" % x, "None"])
    for x in ings:
        put("Chop", (x,), ["Fresh(%s)" % x], ["Chopped(%s)" % x])
    dish = _join(ings + ["Plate"])
    put("Deliver", (dish,), ["Merged(%s)" % dish], ["Delivered(%s)" % dish])

    meetings = []                               # (pile a, pile b) in the table's argument order
    for size in range(1, len(ings) + 1):
        for pile in combinations(ings, size):
            meetings.append((pile, plate))
            if size == 1:
                continue
            for x in pile:
                rest = tuple(n for n in pile if n != x)
                meetings.append(((x,), rest))
                meetings.append(((x,), rest + plate))
                meetings.append((rest, (x,) + plate) if len(rest) == 1 else ((x,) + plate, rest))
    for a, b in meetings:
        put("Merge", (_join(a), _join(b)), [_holding_predicate(a), _holding_predicate(b)],
            ["Merged(%s)" % _join(a + b)])
    return acts, "Delivered(%s)" % dish


def plan_subtasks(recipe_name: str, item_types: Sequence[int], max_path_length: int = 14):
    """All actions that lie on some shortest plan for the recipe
    (stripsworld.py:24-79): breadth-first search over predicate multisets from the
    initial state (one ``Fresh(X)`` per world object that contains X, :19-22) until a
    state with the recipe's ``Delivered`` predicate appears, then the union of the
    edge labels over all shortest paths to it.

    Two actions with the same effect label the same edge; networkx keeps whichever
    was added last, which in the reference depends on ``set`` iteration order
    (PYTHONHASHSEED).  We keep the one whose args sort last -- an arbitrary but
    fixed choice; both have the same goal object."""
    acts, goal = recipe_actions(recipe_name)
    init = ["None"] + ["Fresh(%s)" % L.TYPE_NAME[t] for t in item_types]
    start = tuple(sorted(init))
    depth = {start: 0}
    edges: Dict[Tuple[tuple, tuple], _Action] = {}
    frontier = [start]
    goal_state = None
    for d in range(max_path_length):
        nxt = []
        for st in frontier:
            for a in acts.values():
                pool = list(st)
                ok = True
                for p in a.pre:
                    if p in pool:
                        pool.remove(p)
                    else:
                        ok = False
                        break
                if not ok:
                    continue
                ns = tuple(sorted(pool + a.post))
                if ns not in depth:
                    depth[ns] = d + 1
                    nxt.append(ns)
                if depth[ns] == d + 1:
                    old = edges.get((st, ns))
                    if old is None or a.args > old.args:
                        edges[(st, ns)] = a
                    if goal in ns and goal_state is None:
                        goal_state = ns
        if goal_state is not None:
            break
        frontier = nxt
    if goal_state is None:
        # stripsworld.py:53-56 prints and sys.exit(0)s here
        raise RuntimeError("goal state could not be found, try increasing max_num_subtasks")
    # backward sweep: edges (u,v) with depth[v]==depth[u]+1 and v on a shortest path
    on_path = {goal_state}
    used = {}
    by_target: Dict[tuple, List[tuple]] = {}
    for (u, v) in edges:
        by_target.setdefault(v, []).append(u)
    q = deque([goal_state])
    while q:
        v = q.popleft()
        for u in by_target.get(v, []):
            if depth[u] + 1 == depth[v]:
                used[edges[(u, v)].key] = edges[(u, v)]
                if u not in on_path:
                    on_path.add(u)
                    q.append(u)
    return list(used.values())


@dataclass
class Subtask:
    kind: int                     # KIND_*
    args: Tuple[str, ...]
    goal_types: Tuple[int, ...]   # sorted type ids of the goal object's contents
    food: int = -1                # Chop only: the food type

    @property
    def name(self):
        return "%s(%s)" % (KIND_NAME[self.kind], ", ".join(self.args))

    @property
    def goal_sig(self):
        """Per-type content counts packed in nibbles (T | L<<4 | O<<8 | P<<12)."""
        s = 0
        for t in self.goal_types:
            s += 1 << (4 * t)
        return s


def _subtask_from_action(a: _Action) -> Subtask:
    """Goal object per subtask (navigation_planner/utils.py:161-209): Chop(X) ->
    chopped X alone; Merge(a, b) -> all names of a and b, foods chopped;
    Deliver(X) -> names of X, foods chopped."""
    tid = {n: i for i, n in enumerate(L.TYPE_NAME)}
    if a.kind == "Chop":
        return Subtask(KIND_CHOP, a.args, (tid[a.args[0]],), food=tid[a.args[0]])
    if a.kind == "Merge":
        names = a.args[0].split("-") + a.args[1].split("-")
        return Subtask(KIND_MERGE, a.args, tuple(sorted(tid[n] for n in names)))
    if a.kind == "Deliver":
        return Subtask(KIND_DELIVER, a.args, tuple(sorted(tid[n] for n in a.args[0].split("-"))))
    raise ValueError("unexpected subtask %r" % (a,))


_KIND_RANK = {"Chop": 0, "Merge": 1, "Deliver": 2}


def canonical_subtasks(recipes: Sequence[str], item_types: Sequence[int],
                       max_path_length: int = 14) -> List[Subtask]:
    """Subtasks of all recipes, flattened recipe by recipe
    (overcooked_environment.py:456-457).  Within a recipe the reference's order is
    ``set`` iteration order, i.e. it changes with PYTHONHASHSEED; ours is Chop <
    Merge < Deliver, then by goal object, then by args."""
    out = []
    for r in recipes:
        subs = [_subtask_from_action(a) for a in plan_subtasks(r, item_types, max_path_length)]
        # (kind, goal object, food) is also how the HIP library orders subtask bits internally
        # (include/oc_hip.h: oc_level_subtask_info), so for a one-recipe level the canonical
        # order needs no permutation on the device
        subs.sort(key=lambda t: (t.kind, t.goal_sig, t.food, t.args))
        out.extend(subs)
    return out


def order_subtasks(subtasks: List[Subtask], order) -> List[Subtask]:
    """Re-order to an explicit order: a permutation of indices, or a list of
    ``[kind_name, [args...]]`` / ``"Kind(a, b)"`` entries (e.g. the order a golden
    fixture was recorded with).  Merge entries match on the goal object, so
    ``Merge(Tomato, Lettuce)`` and ``Merge(Lettuce, Tomato)`` are the same."""
    if order is None:
        return subtasks
    order = list(order)
    if all(isinstance(o, (int, np.integer)) for o in order):
        if sorted(order) != list(range(len(subtasks))):
            raise ValueError("subtask_order is not a permutation")
        return [subtasks[i] for i in order]
    pool = list(subtasks)
    out = []
    tid = {n: i for i, n in enumerate(L.TYPE_NAME)}
    for o in order:
        if isinstance(o, str):
            kind, rest = o.split("(", 1)
            args = [a.strip() for a in rest.rstrip(")").split(",")]
        else:
            kind, args = o[0], list(o[1])
        k = _KIND_RANK[kind]
        names = [n for a in args for n in a.split("-")]
        goal = tuple(sorted(tid[n] for n in names))
        hit = None
        for st in pool:
            if st.kind == k and st.goal_types == goal and (k != KIND_MERGE or tuple(args) == st.args):
                hit = st
                break
        if hit is None:
            for st in pool:
                if st.kind == k and st.goal_types == goal and k == KIND_MERGE and \
                        sorted(args) == sorted(st.args):
                    hit = st
                    break
        if hit is None:
            raise ValueError("subtask %r not produced by the planner (have %s)"
                             % (o, [s.name for s in pool]))
        pool.remove(hit)
        out.append(Subtask(hit.kind, tuple(args), hit.goal_types, hit.food))
    if pool:
        raise ValueError("subtask_order misses %s" % [s.name for s in pool])
    return out


# --------------------------------------------------------------------------
# path-distance table
# --------------------------------------------------------------------------
def distance_table(cells: List[List[int]]) -> np.ndarray:
    """D[a][b] = ``World.get_path_distance_between(a, b)`` (utils/world.py:114-131)
    for every ordered pair of cells, a and b as ``y*W + x``.

    The reachability graph (world.py:61-93) has one node per Floor cell and one node
    per (non-Floor cell, direction whose in-bounds-clamped neighbour is Floor);
    edges join Floor neighbours and each (cell, direction) node to that Floor
    neighbour.  Hence: D = MAX_PATH if a is not Floor (no source node, the lookup
    raises and is skipped); the Floor-to-Floor hop count if b is Floor; otherwise
    1 + the smallest hop count to a Floor 4-neighbour of b; MAX_PATH when
    unreachable.  MAX_PATH = perimeter + 1 (:117)."""
    h, w = len(cells), len(cells[0])
    max_path = 2 * (w + h) + 1
    n = w * h
    D = np.full((n, n), max_path, dtype=np.int32)

    def floor(x, y):
        return 0 <= x < w and 0 <= y < h and cells[y][x] == L.FLOOR

    for sy in range(h):
        for sx in range(w):
            if not floor(sx, sy):
                continue
            hop = {(sx, sy): 0}
            q = deque([(sx, sy)])
            while q:
                cx, cy = q.popleft()
                for dx, dy in NAV_ACTIONS:
                    nx_, ny_ = cx + dx, cy + dy
                    if floor(nx_, ny_) and (nx_, ny_) not in hop:
                        hop[(nx_, ny_)] = hop[(cx, cy)] + 1
                        q.append((nx_, ny_))
            a = sy * w + sx
            for by in range(h):
                for bx in range(w):
                    if floor(bx, by):
                        d = hop.get((bx, by))
                    else:
                        best = None
                        for dx, dy in NAV_ACTIONS:
                            # world.py:75 clamps the neighbour in-bounds; a clamped
                            # neighbour is the cell itself, which is not Floor
                            nb = (bx + dx, by + dy)
                            if floor(*nb) and nb in hop:
                                c = hop[nb] + 1
                                best = c if best is None or c < best else best
                        d = best
                    if d is not None and d < max_path:
                        D[a, by * w + bx] = d
    return D


# --------------------------------------------------------------------------
# compiled level
# --------------------------------------------------------------------------
@dataclass
class CompiledLevel:
    name: str
    width: int
    height: int
    num_agents: int
    max_num_timesteps: int
    cells: np.ndarray                 # [H][W] int32
    dist: np.ndarray                  # [HW][HW] int32
    agents: List[Tuple[int, int]]
    items: List[Tuple[int, int, int]]  # (type, x, y) in world.objects iteration order
    subtasks: List[Subtask]
    recipes: List[str]
    pair_types: List[int]             # Plate + recipe[0] ingredients sorted by name
    delivery: List[Tuple[int, int]]
    allergic_mask: int
    counters: List[Tuple[int, int]] = field(default_factory=list)   # Counter tiles, world order
    scatter_items: List[int] = field(default_factory=list)          # item ids placed at random per reset
    play: bool = False                                              # arglist.play (interact.py:44-47,52,66-67)
    blob: np.ndarray = field(default=None, repr=False)

    @property
    def max_path(self):
        return 2 * (self.width + self.height) + 1

    @property
    def num_items(self):
        return len(self.items)

    @property
    def num_subtasks(self):
        return len(self.subtasks)

    @property
    def random_placement(self):
        """True for random-* levels: item start cells differ per env and per episode."""
        return len(self.scatter_items) > 0

    def pack_placement(self, cells):
        """[(x, y)] per item (world order) -> packed x | y<<4 words."""
        return [int(x) | (int(y) << 4) for x, y in cells]

    @property
    def has_dup(self):
        """Some content type occurs more than once (among the items, or inside a goal object):
        an Object is then a multiset of types and the HIP library runs its "dup" kernels
        (include/oc_hip.h).  No shipped level does this."""
        types = [t for t, _, _ in self.items]
        if len([t for t in types if t != L.PLATE]) != len({t for t in types if t != L.PLATE}):
            return True
        return any(len(set(s.goal_types)) != len(s.goal_types) for s in self.subtasks)

    @property
    def hip_supported(self):
        """Limits of the HIP path's packed item words: at most three items of one type, at most
        16 distinct merged names in a level that repeats a type."""
        types = [t for t, _, _ in self.items]
        if any(types.count(t) > 3 for t in set(types)):
            return False
        if self.has_dup:
            names = 1
            for t in (L.TOMATO, L.LETTUCE, L.ONION):
                names *= types.count(t) + 1
            names *= 2 if L.PLATE in types else 1
            singles = sum(1 for t in set(types)) + 1          # the empty multiset and the single contents
            if names - singles > 16:
                return False
        return True


def world_order_items(spec: L.LevelSpec, placements=None) -> List[Tuple[int, int, int]]:
    """Items in ``world.objects`` iteration order: dict keys are created the first
    time a name is inserted (row-major map scan, then the scattered items in line
    order, overcooked_environment.py:113-122,157-173) and each key's list keeps
    insertion order, so items come grouped by type in first-appearance order."""
    seq = list(spec.map_items)
    if spec.scatter:
        if placements is None:
            # nominal start cells (the first Counters); the real ones are per env and per
            # episode and are supplied / sampled at reset time
            placements = spec.counters[:len(spec.scatter)]
        if len(placements) != len(spec.scatter):
            raise ValueError("need %d placements" % len(spec.scatter))
        for ch, (x, y) in zip(spec.scatter, placements):
            if spec.cells[y][x] != L.COUNTER:
                raise ValueError("placement (%d,%d) is not a Counter" % (x, y))
            seq.append((L.TYPE_OF_CHAR[ch], x, y))
        if len({(x, y) for _, x, y in seq}) != len(seq):
            raise ValueError("two items on one Counter")
    first = {}
    for t, _, _ in seq:
        first.setdefault(t, len(first))
    order = sorted(range(len(seq)), key=lambda k: first[seq[k][0]])   # stable: scan order per type
    return [seq[k] for k in order]


def scatter_item_ids(spec: L.LevelSpec) -> List[int]:
    """World-order item id of every scattered letter, in the level file's letter order."""
    seq = [t for t, _, _ in spec.map_items] + [L.TYPE_OF_CHAR[ch] for ch in spec.scatter]
    first = {}
    for t in seq:
        first.setdefault(t, len(first))
    order = sorted(range(len(seq)), key=lambda k: first[seq[k]])
    pos = {k: i for i, k in enumerate(order)}
    n_map = len(spec.map_items)
    return [pos[n_map + j] for j in range(len(spec.scatter))]


def compile_level(level, num_agents: int, max_num_timesteps: int = 100,
                  max_num_subtasks: int = 14, ego_allergic: bool = False,
                  partner_allergic: bool = False, subtask_order=None, placements=None,
                  level_dir: Optional[str] = None, play: bool = False) -> CompiledLevel:
    spec = level if isinstance(level, L.LevelSpec) else L.load_level(level, level_dir)
    if not (1 <= num_agents <= MAX_AGENTS):
        raise ValueError("num_agents must be 1..%d" % MAX_AGENTS)
    if num_agents > len(spec.agent_starts):
        raise ValueError("level %r lists only %d agent starts" % (spec.name, len(spec.agent_starts)))
    if spec.width * spec.height > MAX_CELLS or spec.width > 16 or spec.height > 16:
        raise ValueError("level %r too large (%dx%d)" % (spec.name, spec.width, spec.height))
    items = world_order_items(spec, placements)
    if len(items) > MAX_ITEMS:
        raise ValueError("too many items")
    subtasks = canonical_subtasks(spec.recipes, [t for t, _, _ in items], max_num_subtasks)
    subtasks = order_subtasks(subtasks, subtask_order)
    if len(subtasks) > MAX_SUBTASKS:
        raise ValueError("too many subtasks")
    if not any(s.kind == KIND_DELIVER for s in subtasks):
        raise AssertionError("no delivery subtask")        # overcooked_environment.py:251
    cells = np.array(spec.cells, dtype=np.int32)
    dist = distance_table(spec.cells)
    delivery = [(x, y) for y in range(spec.height) for x in range(spec.width)
                if spec.cells[y][x] == L.DELIVERY]
    # pair term of reward shaping uses Plate + the FIRST recipe's ingredient names,
    # sorted by name (overcooked_environment.py:319-321, recipe.py:28)
    tid = {n: i for i, n in enumerate(L.TYPE_NAME)}
    pair_types = [L.PLATE] + [tid[n] for n in sorted(RECIPE_INGREDIENTS[spec.recipes[0]])]
    allergic = (1 if ego_allergic else 0)
    if partner_allergic:
        allergic |= ((1 << num_agents) - 1) & ~1
    lv = CompiledLevel(
        name=spec.name, width=spec.width, height=spec.height, num_agents=num_agents,
        max_num_timesteps=int(max_num_timesteps), cells=cells, dist=dist,
        agents=list(spec.agent_starts[:num_agents]), items=items, subtasks=subtasks,
        recipes=list(spec.recipes), pair_types=pair_types, delivery=delivery,
        allergic_mask=allergic, counters=list(spec.counters), scatter_items=scatter_item_ids(spec),
        play=bool(play))
    if len(lv.counters) > 64:
        raise ValueError("too many Counter tiles")
    lv.blob = build_blob(lv)
    return lv


def build_blob(lv: CompiledLevel) -> np.ndarray:
    n = lv.width * lv.height
    sec = {}
    off = HEADER_WORDS
    for key, size in (("cells", n), ("dist", n * n), ("agents", 2 * lv.num_agents),
                      ("items", 3 * lv.num_items), ("subtasks", 4 * lv.num_subtasks),
                      ("pair", len(lv.pair_types)), ("delivery", 2 * len(lv.delivery)),
                      ("counters", 2 * len(lv.counters)), ("scatter", len(lv.scatter_items))):
        sec[key] = off
        off += size
    b = np.zeros(off, dtype=np.int32)
    b[0:16] = [MAGIC, VERSION, lv.width, lv.height, lv.num_agents, lv.num_items,
               lv.num_subtasks, lv.max_num_timesteps, lv.max_path, lv.allergic_mask,
               len(lv.pair_types), len(lv.delivery), len(lv.counters), len(lv.scatter_items),
               1 if lv.play else 0, 0]
    b[16:26] = [sec["cells"], sec["dist"], sec["agents"], sec["items"], sec["subtasks"],
                sec["pair"], sec["delivery"], off, sec["counters"], sec["scatter"]]
    b[sec["cells"]:sec["cells"] + n] = lv.cells.reshape(-1)
    b[sec["dist"]:sec["dist"] + n * n] = lv.dist.reshape(-1)
    b[sec["agents"]:sec["agents"] + 2 * lv.num_agents] = np.array(lv.agents).reshape(-1)
    b[sec["items"]:sec["items"] + 3 * lv.num_items] = np.array(lv.items).reshape(-1)
    st = []
    for s in lv.subtasks:
        st += [s.kind, s.goal_sig, s.food, len(s.goal_types)]
    b[sec["subtasks"]:sec["subtasks"] + len(st)] = st
    b[sec["pair"]:sec["pair"] + len(lv.pair_types)] = lv.pair_types
    if lv.delivery:
        b[sec["delivery"]:sec["delivery"] + 2 * len(lv.delivery)] = np.array(lv.delivery).reshape(-1)
    if lv.counters:
        b[sec["counters"]:sec["counters"] + 2 * len(lv.counters)] = np.array(lv.counters).reshape(-1)
    if lv.scatter_items:
        b[sec["scatter"]:sec["scatter"] + len(lv.scatter_items)] = lv.scatter_items
    return b

"""Level catalogue and level-text parser.

The reference names a level by the stem of a text file under
``gym_cooking/utils/levels/`` and parses it in
``OvercookedEnvironment.load_level`` (gym_cooking/envs/overcooked_environment.py:100-178):
map rows / blank / recipe class names / blank / agent ``x y`` lines / optional
blank + a line of item letters that are dropped on random Counters.

This module keeps our own transcription of the 19 maps (data, not code) so the
package works on a machine without the reference checkout, and a parser for the
same text format so a user can also point ``level_dir=`` at any directory of
level files.

Map alphabet (gym_cooking/utils/core.py:18-26): ``' '`` Floor, ``'-'`` Counter,
``'/'`` Cutboard, ``'*'`` Delivery, ``t l o p`` = a Counter with a Tomato /
Lettuce / Onion / Plate on it.
"""
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

FLOOR, COUNTER, CUTBOARD, DELIVERY = 0, 1, 2, 3
CELL_OF_CHAR = {" ": FLOOR, "-": COUNTER, "/": CUTBOARD, "*": DELIVERY}
CELL_NAME = ["Floor", "Counter", "Cutboard", "Delivery"]

TOMATO, LETTUCE, ONION, PLATE = 0, 1, 2, 3          # = observation channels, core.py:383-388
TYPE_OF_CHAR = {"t": TOMATO, "l": LETTUCE, "o": ONION, "p": PLATE}
TYPE_NAME = ["Tomato", "Lettuce", "Onion", "Plate"]

_AGENTS_7x7 = [(2, 1), (4, 1), (4, 4), (2, 4)]


def _fixed(rows, recipes, agents=_AGENTS_7x7, scatter=""):
    return {"map": rows, "recipes": recipes, "agents": list(agents), "scatter": scatter}


_OPEN = ["-----t-", "/     l", "/     -", "*     -", "-     -", "-     p", "-----p-"]
_PARTIAL = ["-----t-", "/  -  l", "/  -  -", "*  -  -", "-  -  -", "-     p", "-----p-"]
_FULL = ["-----t-", "/  -  l", "/  -  -", "*  -  -", "-  -  -", "-  -  p", "-----p-"]
_R_OPEN = [" ----- ", "/     -", "/     -", "*     -", "-     -", "-     -", " ----- "]
_R_PARTIAL = [" -- -- ", "/  -  -", "/  -  -", "*  -  -", "-  -  -", "-     -", " ----- "]
_R_FULL = [" -- -- ", "/  -  -", "/  -  -", "*  -  -", "-  -  -", "-  -  -", " -- -- "]

BUILTIN = {}
for _shape, _rows in (("open", _OPEN), ("partial", _PARTIAL), ("full", _FULL)):
    BUILTIN["%s-divider_tomato" % _shape] = _fixed(_rows, ["SimpleTomato"])
    BUILTIN["%s-divider_tl" % _shape] = _fixed(_rows, ["SimpleTomato", "SimpleLettuce"])
    BUILTIN["%s-divider_salad" % _shape] = _fixed(_rows, ["Salad"])
_R_AGENTS = [(2, 1), (4, 2), (4, 4), (2, 4)]
BUILTIN.update({
    "random-full-divider_salad": _fixed(_R_FULL, ["Salad"], scatter="plt"),
    "random-full-divider_tomato": _fixed(_R_FULL, ["SimpleTomato"], scatter="plt"),
    "random-partial-divider_salad": _fixed(_R_PARTIAL, ["Salad"], scatter="plt"),
    "random-open-divider_salad": _fixed(_R_OPEN, ["Salad"], _R_AGENTS, "plt"),
    "random-open-divider_tomato": _fixed(_R_OPEN, ["SimpleTomato"], _R_AGENTS, "plt"),
    "random-open-divider_salad_small": _fixed(
        [" -/- ", "-   -", "/   -", "*   -", " --- "], ["Salad"], [(1, 1), (3, 1)], "plt"),
    "random-open-divider_salad_small_cramped": _fixed(
        [" -/-- ", "-    -", "- -- -", "-    -", " -*-- "], ["Salad"], [(0, 0), (3, 1)], "plt"),
    "random-open-divider_salad_small_wide": _fixed(
        [" --/---- ", "-       -", "- ----- -", "-       -", " --*---- "],
        ["Salad"], [(3, 1), (3, 3)], "plt"),
    "random-open-divider_salad_small_wide_big": _fixed(
        [" --//-- ", "-      -", "- ---- -", "- -  - -", "- ---- -", "-      -", " --*--- "],
        ["Salad"], [(3, 1), (3, 5)], "plt"),
    "random-salad-superwide": _fixed(
        [" --- ----- ", "/   -     -", "/   -     *", " --- ----- "],
        ["Salad"], [(3, 1), (6, 1)], "plt"),
})


@dataclass
class LevelSpec:
    """A parsed level: what ``load_level`` would have put into the World."""
    name: str
    width: int
    height: int
    cells: List[List[int]]                    # [y][x] -> FLOOR/COUNTER/CUTBOARD/DELIVERY
    map_items: List[Tuple[int, int, int]]     # (type, x, y) in row-major scan order
    recipes: List[str]                        # recipe class names
    agent_starts: List[Tuple[int, int]]       # every start listed in the file
    scatter: str = ""                         # item letters placed on random Counters
    counters: List[Tuple[int, int]] = field(default_factory=list)  # world order of Counter tiles


def parse_level_text(name: str, text: str) -> LevelSpec:
    """Parse the reference's level file format (overcooked_environment.py:100-178).

    ``width`` is the length of the last map row and ``height`` the number of map
    rows (:176-177)."""
    phase = 1
    rows, recipes, agents, scatter = [], [], [], ""
    for line in text.split("\n"):
        if line == "":
            phase += 1
        elif phase == 1:
            rows.append(line)
        elif phase == 2:
            recipes.append(line.strip())
        elif phase == 3:
            parts = line.split(" ")
            agents.append((int(parts[0]), int(parts[1])))
        elif phase == 4:
            scatter += "".join(c for c in line if c in "tlop")
    return _build(name, rows, recipes, agents, scatter)


def _build(name, rows, recipes, agents, scatter) -> LevelSpec:
    if not rows:
        raise ValueError("level %r has no map rows" % name)
    width = len(rows[-1])
    height = len(rows)
    cells = [[FLOOR] * width for _ in range(height)]
    items, counters = [], []
    for y, row in enumerate(rows):
        if len(row) != width:
            raise ValueError("level %r: ragged map row %d" % (name, y))
        for x, ch in enumerate(row):
            if ch in TYPE_OF_CHAR:
                cells[y][x] = COUNTER
                counters.append((x, y))
                items.append((TYPE_OF_CHAR[ch], x, y))
            elif ch in CELL_OF_CHAR:
                cells[y][x] = CELL_OF_CHAR[ch]
                if cells[y][x] == COUNTER:
                    counters.append((x, y))
            else:
                cells[y][x] = FLOOR      # unknown glyph -> Floor (:127-130)
    return LevelSpec(name=name, width=width, height=height, cells=cells, map_items=items,
                     recipes=list(recipes), agent_starts=list(agents), scatter=scatter,
                     counters=counters)


def load_level(name: str, level_dir: Optional[str] = None) -> LevelSpec:
    """Built-in transcription by default; ``level_dir/<name>.txt`` if given."""
    if level_dir is not None:
        with open(os.path.join(level_dir, name + ".txt"), "r") as f:
            return parse_level_text(name, f.read())
    if name not in BUILTIN:
        raise FileNotFoundError(
            "unknown level %r (built-ins: %s); pass level_dir= to load a level file"
            % (name, ", ".join(sorted(BUILTIN))))
    b = BUILTIN[name]
    return _build(name, b["map"], b["recipes"], b["agents"], b["scatter"])

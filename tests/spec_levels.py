"""Levels whose specialised kernel libraries the GPU tests exercise; __graft_entry__.build()
pre-compiles them (plus the BASELINE configs) so a fresh GPU box needs no JIT."""
import os

from conftest import GOLDEN, compile_for, load_golden

# (level, num_agents, T) compiled with the canonical subtask order
SEEDED_CASES = [
    ("open-divider_tomato", 2, 100), ("full-divider_salad", 2, 120), ("partial-divider_tl", 3, 100),
    ("open-divider_salad", 2, 150), ("open-divider_tl", 3, 150), ("partial-divider_tomato", 4, 60),
    ("open-divider_tl", 2, 200), ("partial-divider_salad", 4, 90),
]
# golden fixtures replayed on the specialised path as well (recorded subtask order)
SPEC_GOLDEN = ["base_open-divider_tomato_a2.npz", "base_full-divider_salad_a2.npz",
               "base_partial-divider_tl_a3.npz", "base_full-divider_tl_a4.npz",
               "wrap_tomato_r2.npz", "wrap_salad_open_c5.npz", "wrap_tl_full_blind_allergic.npz",
               "rbase_random-open-divider_salad_small_a2.npz", "rwrap_rsuperwide_c5.npz",
               "fow_tomato_r2.npz", "fow_salad_r3.npz",
               "cbase_custom-onion_salad_a3.npz", "cwrap_conion_r2.npz", "cbase_custom-two_deliveries_a2.npz",
               # levels that repeat a content type (dup mode)
               "cbase_dup_two_tomatoes_a2.npz", "cbase_dup_two_tomatoes_small_a3.npz",
               "cbase_dup_two_lettuces_salad_a2.npz", "cwrap_dup_two_lettuces_salad_c3.npz",
               "cwrap_dup_two_tomatoes_small_r1.npz",
               # ... three times (the count == 3 paths: 2-bit goal counts at their maximum, five items)
               "cbase_dup_three_tomatoes_a2.npz", "cbase_dup_three_tomatoes_a3.npz", "cwrap_dup_three_tomatoes_r2.npz",
               # arglist.play = True (a run-time flag: the same libraries)
               "pbase_open-divider_tomato_a2.npz", "pbase_partial-divider_tl_a3.npz", "pwrap_play_salad_c3.npz"]


# fixtures whose level (canonical subtask order) the seeded dup-mode test steps
DUP_SEEDED = [("cbase_dup_two_tomatoes_small_a2.npz", 2), ("cbase_dup_two_tomatoes_small_a3.npz", 3),
              ("cbase_dup_two_lettuces_salad_a2.npz", 2), ("cbase_dup_two_tomatoes_a3.npz", 3),
              ("cbase_dup_three_tomatoes_a2.npz", 2), ("cbase_dup_three_tomatoes_a3.npz", 3)]


def all_spec_levels():
    from gym_comm_amd import compiler, levels
    out = [compiler.compile_level(l, a, t) for l, a, t in SEEDED_CASES]
    for f in SPEC_GOLDEN:
        _, st = load_golden(os.path.join(GOLDEN, f))
        out.append(compile_for(st))
    for f, a in DUP_SEEDED:
        _, st = load_golden(os.path.join(GOLDEN, f))
        out.append(compiler.compile_level(levels.parse_level_text(st["level"], st["level_text"]), a, 90))
    return out

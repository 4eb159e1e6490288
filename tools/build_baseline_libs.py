#!/usr/bin/env python3
"""Build only the libraries the BASELINE bench configs load (level libraries of tomato-2, salad-2,
tl-3 + their two timeline flavours), side by side: a kernel experiment's turnaround is ~1 minute
instead of the ~10 of __graft_entry__.build().  (The generic library is NOT rebuilt here: the ABI
must be unchanged, or run build() instead.)"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_comm_amd import compiler, specialize

levels = [compiler.compile_level(n, a, 500) for n, a in
          (("open-divider_tomato", 2), ("full-divider_salad", 2), ("partial-divider_tl", 3))]
jobs = [(lv, v) for lv in levels for v in ("", "timeline", "timeline-drain")]
with ThreadPoolExecutor(max_workers=8) as pool:
    for path in pool.map(lambda j: specialize.ensure(j[0].blob, geometry=True, variant=j[1]), jobs):
        print(path)

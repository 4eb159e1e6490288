import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_golden(path):
    z = np.load(path)
    st = json.loads(str(z["static_json"]))
    return z, st


def compile_for(st, **kw):
    """Compile the level of a fixture with the subtask order the fixture was recorded
    with (the reference's order depends on PYTHONHASHSEED, see compiler.py)."""
    from gym_comm_amd import compiler, levels
    e, p = st.get("ego_config", {}), st.get("partner_config", {})
    level = st["level"]
    if "level_text" in st:        # a level that is not built in (tests/golden custom maps)
        level = levels.parse_level_text(st["level"], st["level_text"])
    return compiler.compile_level(
        level, st["num_agents"], st["max_num_timesteps"],
        ego_allergic=bool(e.get("ALLERGIC")), partner_allergic=bool(p.get("ALLERGIC")),
        subtask_order=st["subtasks"], play=bool(st.get("play", False)), **kw)


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle
    oracle.build()
    return oracle
collect_ignore = ["soak.py", "soak_wrapper.py"]

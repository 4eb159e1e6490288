"""Run-config loader with the reference's JSON schema (env_args.json, spread/*.json).

The reference turns the JSON into a synthetic argv and parses it with argparse
(arglist.py:96-121); the environment only ever reads the fields below.  Differences, both
lenient: a missing ``ego_config`` / ``partner_config`` is ``{}`` (the reference raises
KeyError at arglist.py:104 for its own spread/env_args100on.json), and a missing
``CAN_MOVE`` means True (see envs.py).
"""
import json
from types import SimpleNamespace

DEFAULTS = {                      # arglist.py:40-92 defaults
    "max_num_timesteps": 100, "max_num_subtasks": 14, "seed": 1, "with_image_obs": False,
    "play": False, "record": False, "communication_on": False, "num_communication": 10,
    "ego_led": False, "fow_radius": 2, "total_timesteps": 20000000, "record_interval": 500,
    "log": False, "wandb": False, "notes": "XXX notes", "hyperparams": {},
}


def load_env_args(src):
    """src: path of a JSON file, or an already-parsed dict.  Returns a namespace usable as
    the ``arglist`` of OvercookedEnvironment / OvercookedMultiEnv / OvercookedVecEnv."""
    if isinstance(src, dict):
        raw = dict(src)
    else:
        with open(src, "r") as f:
            raw = json.load(f)
    for key in ("level", "num_agents"):
        if key not in raw:
            raise KeyError(key)                       # required by the reference's parser too
    out = dict(DEFAULTS)
    out.update({k: v for k, v in raw.items() if k not in ("env_config",)})
    out["ego_config"] = dict(raw.get("ego_config") or {})
    out["partner_config"] = dict(raw.get("partner_config") or {})
    out["num_agents"] = int(out["num_agents"])
    out["max_num_timesteps"] = int(out["max_num_timesteps"])
    for m in ("model1", "model2", "model3", "model4"):
        out.setdefault(m, None)
    return SimpleNamespace(**out)

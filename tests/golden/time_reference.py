"""Time the REFERENCE's own Python hot path on this container's host cores.

TEST INFRASTRUCTURE, build-container only (same rules as make_golden.py: it imports and runs the
reference at /root/reference through ref_harness.py; nothing of the reference is copied and this
script is never imported by the product, by bench.py's timed region or by any -m gpu test).  The
reference cannot travel to the GPU box, so its CPU rate is a RECORDED figure:
this script writes it, with the CPU model and core count, to ``profiles/reference_cpu_rate.json``;
``bench.py`` reads that file like ``profiles/traffic.json`` and carries it as
``cpu_baseline.reference_python``.

Measured per config, one process per core, every process its own env and its own seeded action
stream (the reference is single-threaded Python; N cores = N independent processes):
  * tomato-2 (BASELINE configs[1]): ``OvercookedMultiEnv.multi_step`` + ``multi_reset`` on done
    (gym_comm/envs/overcooked_env.py:207-297), T = 500, C = 2, fow_radius 2;
  * salad-2 / tl-3: the base ``OvercookedEnvironment.step`` + ``reset`` on done
    (gym_cooking/envs/overcooked_environment.py:180-241), T = 500.

    PYTHONHASHSEED=0 python tests/golden/time_reference.py [--seconds 10] [--procs 1,8]
"""
import argparse
import json
import os
import platform
import random
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)

CONFIGS = {
    "open-divider_tomato_a2_wrapper": dict(level="open-divider_tomato", agents=2, wrapper=True),
    "full-divider_salad_a2_base": dict(level="full-divider_salad", agents=2, wrapper=False),
    "partial-divider_tl_a3_base": dict(level="partial-divider_tl", agents=3, wrapper=False),
}
NAV = [(0, 1), (0, -1), (-1, 0), (1, 0)]          # world.py:16


def worker(cfg_name, seconds, seed):
    """One process: steps until `seconds` of wall clock are used; prints 'steps elapsed resets'."""
    import ref_harness as rh
    cfg = CONFIGS[cfg_name]
    rng = random.Random(seed)
    arglist = rh.make_arglist(cfg["level"], cfg["agents"], 500, num_communication=2, fow_radius=2)
    steps = resets = 0
    if cfg["wrapper"]:
        env = rh.wrapper_env(arglist)
        with rh.quiet():
            env.multi_reset()
            t0 = time.perf_counter()
            while True:
                for _ in range(50):
                    _, _, done, _ = env.multi_step([rng.randrange(4), rng.randrange(2)],
                                                   [rng.randrange(4), rng.randrange(2)])
                    steps += 1
                    if done:
                        env.multi_reset()
                        resets += 1
                el = time.perf_counter() - t0
                if el >= seconds:
                    break
    else:
        env = rh.base_env(arglist)
        names = [a.name for a in env.sim_agents]
        with rh.quiet():
            env.reset()
            t0 = time.perf_counter()
            while True:
                for _ in range(50):
                    _, done, _ = env.step({nm: NAV[rng.randrange(4)] for nm in names})
                    steps += 1
                    if done:
                        env.reset()
                        resets += 1
                el = time.perf_counter() - t0
                if el >= seconds:
                    break
    sys.__stdout__.write("RESULT %d %.6f %d\n" % (steps, el, resets))
    sys.__stdout__.flush()


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def run(cfg_name, procs, seconds):
    env = dict(os.environ, PYTHONHASHSEED="0", OMP_NUM_THREADS="1")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", cfg_name,
                            "--seconds", str(seconds), "--seed", str(1000 + k)],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
          for k in range(procs)]
    rates, steps_total, resets = [], 0, 0
    for p in ps:
        out, _ = p.communicate()
        line = [ln for ln in out.splitlines() if ln.startswith("RESULT")]
        if p.returncode != 0 or not line:
            raise RuntimeError("worker failed for %s" % cfg_name)
        s, el, r = line[-1].split()[1:]
        rates.append(int(s) / float(el))
        steps_total += int(s)
        resets += int(r)
    return {"procs": procs, "env_steps_per_s": sum(rates), "per_proc_min": min(rates),
            "per_proc_max": max(rates), "steps": steps_total, "resets": resets, "seconds_per_proc": seconds}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--procs", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "reference_cpu_rate.json"))
    a = ap.parse_args()
    if a.worker:
        worker(a.worker, a.seconds, a.seed)
        return
    cores = len(os.sched_getaffinity(0))
    plist = [int(x) for x in a.procs.split(",")] if a.procs else [1, cores]
    result = {"what": "the reference's own Python hot path (kyle-he/gym-comm, imported through "
                      "tests/golden/ref_harness.py), seeded uniform-random actions, T = 500, reset on done; "
                      "N procs = N independent single-threaded processes",
              "script": "tests/golden/time_reference.py", "hardware": cpu_model(), "cores_available": cores,
              "python": platform.python_version(), "recorded": time.strftime("%Y-%m-%d"),
              "where": "build container (the reference cannot travel to the GPU box)", "configs": {}}
    for name, cfg in CONFIGS.items():
        ent = dict(level=cfg["level"], num_agents=cfg["agents"],
                   call=("OvercookedMultiEnv.multi_step + multi_reset (gym_comm/envs/overcooked_env.py:207-297)"
                         if cfg["wrapper"] else
                         "OvercookedEnvironment.step + reset (gym_cooking/envs/overcooked_environment.py:180-241)"),
                   runs=[])
        for procs in plist:
            r = run(name, procs, a.seconds)
            ent["runs"].append(r)
            print("%-34s %2d procs: %9.1f env-steps/s (%d steps, %d resets)"
                  % (name, procs, r["env_steps_per_s"], r["steps"], r["resets"]), flush=True)
        result["configs"][name] = ent
    with open(a.out, "w") as f:
        json.dump(result, f, indent=1)
        f.write("\n")
    print("wrote", a.out)


if __name__ == "__main__":
    main()

"""Helpers for the -m gpu parity tests (HIP path vs the CPU oracle / golden vectors)."""
import numpy as np

LETTER = {"D": 0, "U": 1, "L": 2, "R": 3, "N": 4}
# scripted solves, as (agent0, agent1) action codes
KAT1_TOMATO = list(zip([LETTER[c] for c in "DDDRR" + "N" * 18],
                       [LETTER[c] for c in "RULLLLLDDDDRRRRDLLLLUUL"]))
KAT2_SALAD = [(LETTER[a], LETTER[b]) for a, b in [
    "NR", "NU", "NL", "NL", "RR", "LR", "LL", "UL", "RD", "RD", "LD", "LD",
    "UR", "DR", "DL", "DL", "DN", "RN", "RN", "LN", "UN", "UN", "LN"]]
SCRIPTS = {"open-divider_tomato": KAT1_TOMATO, "open-divider_tl": KAT1_TOMATO,
           "open-divider_salad": KAT1_TOMATO, "full-divider_salad": KAT2_SALAD}


def momentum_actions(rng, steps, A, n, keep=0.7, nact=5):
    """[steps][A][n] action codes: each agent keeps its direction with prob `keep`."""
    out = np.zeros((steps, A, n), np.int32)
    cur = rng.integers(0, nact, (A, n))
    for k in range(steps):
        new = rng.integers(0, nact, (A, n))
        cur = np.where(rng.random((A, n)) < keep, cur, new)
        out[k] = cur
    return out


def scripted_then_random(rng, level, steps, A, n, nact=5):
    """Every env plays a random-length prefix of the level's solve script (if one is
    known) from reset, then momentum-random actions; so the batch visits deep states
    (held / chopped / merged / delivered items) from which random actions branch."""
    acts = momentum_actions(rng, steps, A, n, nact=nact)
    script = SCRIPTS.get(level)
    if script is not None:
        L = len(script)
        pre = rng.integers(0, L + 1, n)
        for i in range(n):
            p = min(int(pre[i]), steps)
            for k in range(p):
                for a in range(A):
                    c = script[k][a] if a < 2 else 4
                    acts[k, a, i] = c if (c < nact) else (acts[k, a, i] % nact)
    return acts


def assert_snapshots_equal(hip, ora, ctx, where=None):
    """Compare unpack_state() of the HIP state with OracleBatch.snapshot_all()."""
    keys = ["items", "order", "agents", "t", "completed", "goal_count", "nobj"]
    for k in keys:
        a, b = np.asarray(hip[k]), np.asarray(ora[k])
        if where is not None:
            a, b = a[where], b[where]
        if not np.array_equal(a, b):
            bad = np.argwhere(np.asarray(a != b).reshape(a.shape[0], -1).any(axis=1))[:4].ravel()
            raise AssertionError("%s: field %r differs at envs %s\nhip=%s\noracle=%s"
                                 % (ctx, k, bad, a[bad[0]], b[bad[0]]))


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)

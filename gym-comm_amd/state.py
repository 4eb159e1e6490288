"""Packed env-state words <-> named fields (layout: include/oc_hip.h)."""
import numpy as np


def unpack_state(words, A, M, S, slot=None, goal_index=None, deliver=None):
    """words: int array [A+M+2, n] (numpy).  Returns the canonical snapshot used by the
    golden fixtures and the oracle:
      items  [n][M][5]  x, y, state_index, group, holder agent (-1)
      order  [n][M]     groups in world.objects iteration order, -1 padded
      agents [n][A][3]  x, y, held group (-1)
      t, completed [n][S], goal_count [n][S], merge_counter, error
    slot (per subtask: its bit in the state's subtask words; BatchedOvercooked.unpack_kw()).
    goal_index / deliver (per subtask: index of its distinct goal object, is-a-Deliver flag):
    given for a level in dup mode, where the state keeps 2-bit counts per distinct goal; the
    seq field of such a level is kseq<<4 | seq, still a world-order sort key.
    """
    w = np.asarray(words).astype(np.int64)
    n = w.shape[1]
    ag = w[:A]
    agents = np.stack([ag & 15, (ag >> 4) & 15, ((ag >> 8) & 15) - 1], axis=-1).transpose(1, 0, 2)
    it = w[A:A + M]
    items = np.stack([it & 15, (it >> 4) & 15, (it >> 8) & 1, (it >> 9) & 7,
                      ((it >> 12) & 7) - 1], axis=-1).transpose(1, 0, 2)
    seq = ((it >> 16) & 255).T                       # [n][M]
    group = items[:, :, 3]
    is_rep = group == np.arange(M)[None, :]          # one representative item per Object
    key = np.where(is_rep, seq, 1 << 20)
    idx = np.argsort(key, axis=1, kind="stable")
    order = np.where(np.take_along_axis(is_rep, idx, axis=1), idx, -1)
    m0, m1 = w[A + M] & 0xFFFFFFFF, w[A + M + 1] & 0xFFFFFFFF
    # the kernels keep subtask bits in a canonical order of their own (include/oc_hip.h,
    # oc_level_subtask_info): slot[s] = the bit of the caller's subtask s (default: bit s)
    bits = np.arange(S) if slot is None else np.asarray(slot)
    if goal_index is None:
        goal_count = (m1[:, None] >> bits) & 1
    else:
        # dup mode (a level that repeats a content type): two bits per DISTINCT goal object;
        # a Deliver subtask's own count is never updated by the reference (:404-415) and stays 0
        gi = np.asarray(goal_index)
        goal_count = (m1[:, None] >> (2 * gi)[None, :]) & 3
        goal_count = np.where(np.asarray(deliver, bool)[None, :], 0, goal_count)
    return {
        "items": items.astype(np.int32), "order": order.astype(np.int32),
        "agents": agents.astype(np.int32), "t": ((w[0] >> 16) & 0xFFFF).astype(np.int32),
        "completed": ((m0[:, None] >> bits) & 1).astype(np.int32),
        "goal_count": goal_count.astype(np.int32),
        "merge_counter": ((w[1] >> 16) & 255).astype(np.int32),
        "error": ((w[1] >> 24) & 255).astype(np.int32),
        "nobj": is_rep.sum(axis=1).astype(np.int32),
    }

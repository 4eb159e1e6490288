#!/usr/bin/env python3
"""Scan device assembly for the stall tools/issue_probe.hip measured on gfx950: a SCALAR instruction
reading an SGPR (or vcc) that a VECTOR instruction wrote shortly before (v_cmp -> s_and_b64,
v_readlane -> s_*, v_cmp -> s_cbranch_vcc*): ~16 extra cycles for a lone wave unless ~4 other
instructions sit in between.   python tools/hazard_scan.py /tmp/oc_isa.s <mangled-name-substring> [window]"""
import re
import sys


def regs(tok):
    """SGPR numbers named by an operand token (s5, s[4:5], vcc, exec)."""
    tok = tok.strip().rstrip(",")
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return {"vcc"}
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main():
    path, name = sys.argv[1], sys.argv[2]
    window = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    text = open(path).read()
    m = re.search(r"\n(\S*%s\S*):\s" % re.escape(name), text)
    start = m.start()
    end = text.index(".Lfunc_end", start)
    lines = [ln for ln in text[start:end].splitlines() if re.match(r"\t[a-z]", ln) and not ln.startswith("\t.")]
    recent = []          # (age, set of sgprs written by a VALU instruction, text)
    hits = 0
    for idx, ln in enumerate(lines):
        parts = ln.strip().split(None, 1)
        op = parts[0]
        ops = [o.strip() for o in (parts[1].split(",") if len(parts) > 1 else [])]
        if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_load", "s_barrier")):
            reads = set()
            srcs = ops[1:] if len(ops) > 1 and not op.startswith(("s_cmp", "s_bitcmp", "s_cbranch")) else ops
            for o in srcs:
                reads |= regs(o)
            if op.startswith(("s_cbranch_vcc",)):
                reads.add("vcc")
            if op.startswith("s_and_saveexec") or "exec" in ln:
                pass
            for age_idx, wr, wtxt in recent:
                if idx - age_idx <= window and reads & wr:
                    hits += 1
                    print("%5d  %-60s <- %d back: %s" % (idx, ln.strip()[:60], idx - age_idx, wtxt.strip()[:70]))
                    break
        if op.startswith("v_"):
            wr = set()
            if op.startswith(("v_cmp", "v_cmpx")):
                if op.endswith("_e32") or (ops and ops[0] == "vcc"):
                    wr = {"vcc"} if op.endswith("_e32") or ops[0] == "vcc" else regs(ops[0])
                else:
                    wr = regs(ops[0])
            elif op.startswith(("v_readlane", "v_readfirstlane")):
                wr = regs(ops[0])
            elif op.startswith(("v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co")) and len(ops) > 1:
                wr = regs(ops[1])
            if wr:
                recent.append((idx, wr, ln))
        recent = [r for r in recent if idx - r[0] <= window]
    print("%d scalar reads of a freshly vector-written SGPR within %d instructions, %d instructions scanned" % (hits, window, len(lines)))


if __name__ == "__main__":
    main()

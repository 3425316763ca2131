#!/usr/bin/env python3
"""List, per kernel of a hipcc -S listing, every packed-fp32 VALU instruction (v_pk_{mul,add,fma}_f32) whose SOURCE registers are
overwritten by a load (ds_read* / buffer_load* / global_load*) issued within the next `window` instructions -- the pattern behind
round 4's pool_lrn_bwd finding.  usage: isa_pk_war.py file.s [kernel-substring] [window=3]"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
window = int(sys.argv[3]) if len(sys.argv) > 3 else 3
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M)]


def vregs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


grand = 0
for idx, (pos, name) in enumerate(starts):
    if pat not in name:
        continue
    end = starts[idx + 1][0] if idx + 1 < len(starts) else len(s)
    lines = [l.split(";")[0].rstrip() for l in s[pos:end].split("\n")]
    ins = [(i, l.strip()) for i, l in enumerate(lines) if l.startswith("\t") and l.strip() and not l.strip().startswith(".")]
    npk = sum(1 for _, t in ins if t.startswith("v_pk_") and "_f32" in t.split()[0])
    hits = []
    for k, (ln, t) in enumerate(ins):
        op = t.split()[0]
        if not (op.startswith("v_pk_") and op.endswith("_f32")):
            continue
        ops_ = t.split(None, 1)[1].split(",")
        src = set()
        for o in ops_[1:]:
            src |= vregs(o)
        for d, (ln2, t2) in enumerate(ins[k + 1:k + 1 + window]):
            op2 = t2.split()[0]
            if re.match(r"(ds_read|buffer_load|global_load|flat_load)", op2) and " lds" not in t2:
                dst = vregs(t2.split(None, 1)[1].split(",")[0])
                if dst & src:
                    hits.append((ln, t, ln2, t2, d))
    grand += len(hits)
    if npk or hits:
        print("%-90s packed-f32 %4d  source overwritten by a load within %d: %d" % (name[:90], npk, window, len(hits)))
    for ln, t, ln2, t2, d in hits:
        print("      line %5d: %-70s <- line %5d (+%d): %s" % (ln, t, ln2, d + 1, t2))
print("total", grand)

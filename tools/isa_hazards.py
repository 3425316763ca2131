#!/usr/bin/env python3
"""Scan a hipcc -S listing for hazards the compiler does not guard around INLINE-ASM vector-memory instructions (gfx9 family):
  * VALU writes an SGPR (v_readlane / v_readfirstlane / v_cmp ... sdst) -> VMEM reads that SGPR (descriptor or soffset) needs 5 wait states;
  * SALU writes M0 -> LDS-DMA (buffer_load ... lds) needs 1 wait state.
usage: isa_hazards.py file.s [kernel-name-substring]"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M)]


def sregs(tok):
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


total = 0
for idx, (pos, name) in enumerate(starts):
    if pat not in name:
        continue
    end = starts[idx + 1][0] if idx + 1 < len(starts) else len(s)
    lines = [l.split(";")[0].rstrip() for l in s[pos:end].split("\n")]
    ins = [(i, l.strip()) for i, l in enumerate(lines) if l.startswith("\t") and not l.strip().startswith(".") and l.strip()]
    found = 0
    for k, (ln, t) in enumerate(ins):
        op = t.split()[0]
        dst = None
        if op in ("v_readlane_b32", "v_readfirstlane_b32"):
            dst = sregs(t.split(",")[0])
        elif op.startswith("v_cmp") and re.search(r"\bs\[\d+:\d+\]", t.split(",")[0]):
            dst = sregs(t.split(",")[0])
        if dst:
            ws = 0
            for (ln2, t2) in ins[k + 1:k + 8]:
                op2 = t2.split()[0]
                if re.match(r"(buffer|global|flat|scratch)_", op2):
                    srcs = sregs(t2.split(None, 1)[1]) if " " in t2 else set()
                    if (srcs & dst) and ws < 5:
                        print("%s: line %d: %s  ->  %s  (%d wait states)" % (name[:60], ln2, t, t2, ws))
                        found += 1
                if op2 == "s_nop":
                    ws += int(t2.split()[1]) + 1
                else:
                    ws += 1
                if ws >= 5:
                    break
        if op == "s_mov_b32" and t.split()[1].startswith("m0"):
            nxt = ins[k + 1][1] if k + 1 < len(ins) else ""
            if re.match(r"buffer_load.* lds", nxt):
                print("%s: line %d: M0 write directly before %s" % (name[:60], ln, nxt))
                found += 1
    total += found
    print("%-70s hazards: %d" % (name[:70], found))
print("total", total)

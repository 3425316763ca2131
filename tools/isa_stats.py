#!/usr/bin/env python3
"""Per-kernel instruction mix / register use from a hipcc -S listing.  usage: isa_stats.py file.s [name-substring]"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):\s*; @", s, re.M)]
for idx, (pos, name) in enumerate(starts):
    if pat not in name:
        continue
    end = starts[idx + 1][0] if idx + 1 < len(starts) else len(s)
    body = s[pos:end]
    lines = body.split("\n")
    cnt = lambda p: sum(1 for l in lines if re.search(p, l))
    print(name[:90])
    print("  mfma %d  lds-dma %d  vmem-load %d  ds_read %d  ds_write %d  valu~ %d  salu~ %d  s_load %d  waitcnt %d (vmcnt %d)  barrier %d  scratch %d"
          % (cnt(r"\bv_mfma"), cnt(r"buffer_load.* lds"), cnt(r"\b(buffer|global)_load"), cnt(r"\bds_read"), cnt(r"\bds_write"),
             cnt(r"^\s+v_(?!mfma)"), cnt(r"^\s+s_(?!waitcnt|barrier|load|nop)"), cnt(r"\bs_load"), cnt(r"s_waitcnt"),
             cnt(r"s_waitcnt.*vmcnt"), cnt(r"s_barrier"), cnt(r"scratch_")))
    for l in lines:
        if re.search(r"; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize)", l):
            print("   ", l.strip())

#!/usr/bin/env python3
"""Lint of the BUILT device code (every gfx950 code object embedded in libvltf_hip.so) for instruction forms that give wrong results on
MI355X under co-residency:

  v_pk_{add,mul,fma}_f32 ... op_sel:[..1..]   packed fp32 arithmetic whose LOW result takes a HIGH source register.  Measured
      (tools/ubench/pk_opsel_raw.hip, pk_cross_kernel.hip): with op_sel[1] set the low result comes out as src0.lo + 0 in lanes 48..63
      while waves of a kernel mixing LDS reads, v_cvt_pk_bf16_f32 / v_pk_add_f32 and bf16 MFMAs share the CU -- hipcc's SLP vectoriser
      forms it from ordinary scalar code (the build uses -fno-slp-vectorize); any op_sel bit on these three opcodes is refused.

usage: isa_lint.py [path/to/libvltf_hip.so]     exit code 1 when a refused form is present"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
BAD = re.compile(r"\bv_pk_(add|mul|fma)_f32\b.*\bop_sel:\[[0-9,]*1[0-9,]*\]")


def code_objects(path):
    """the gfx950 ELF images of every offload bundle in the file"""
    d = open(path, "rb").read()
    pos = 0
    while True:
        i = d.find(MAGIC, pos)
        if i < 0:
            return
        pos = i + len(MAGIC)
        (num,) = struct.unpack_from("<Q", d, i + 24)
        if not 0 < num < 16:
            continue
        off = i + 32
        for _ in range(num):
            o, s, tl = struct.unpack_from("<QQQ", d, off)
            off += 24
            triple = d[off:off + tl].decode("ascii", "replace")
            off += tl
            if "gfx950" in triple and s:
                yield d[i + o:i + o + s]


def lint(path):
    """[(kernel, instruction)] of refused forms, and the number of instructions looked at"""
    hits, seen = [], 0
    for img in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            out = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
        kernel = "?"
        for line in out.split("\n"):
            m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
            if m:
                kernel = m.group(1)
                continue
            t = line.split("//")[0].strip()
            if not t:
                continue
            seen += 1
            if BAD.search(t):
                hits.append((kernel, t))
    return hits, seen


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "video-learning-tf_amd", "libvltf_hip.so")
    hits, seen = lint(lib)
    for k, t in hits[:40]:
        print("%s: %s" % (k[:80], t))
    print("%s: %d instructions, %d refused" % (lib, seen, len(hits)))
    sys.exit(1 if hits else 0)

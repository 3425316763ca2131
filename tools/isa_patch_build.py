#!/usr/bin/env python3
"""Build a variant of libvltf_hip.so whose pointwise.hip DEVICE code is a hand-patched ISA listing (experiments of the round-4
determinism hunt): re-runs hipcc's own steps (device compile -> .s, [patch], assemble, lld, bundle, host compile with the bundle) and
links against the in-tree objects.  usage: isa_patch_build.py <name> <patch> ;  patch = "none" | "pkgap:<wait states>"
  pkgap:N  in pool_lrn_bwd_stream_kernel<5,3,true,0>: N wait states (s_nop) between every packed-fp32 instruction and a load issued
           within the next 3 instructions that overwrites one of its source registers."""
import os
import re
import shlex
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "video-learning-tf_amd", "csrc")
out = os.path.join(root, "scratch", "plbv")
os.makedirs(out, exist_ok=True)
name, patch = sys.argv[1], sys.argv[2]
hipcc = "/opt/rocm/bin/hipcc"
base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm", "-I" + os.path.join(root, "include"), "-c",
        os.path.join(src, "pointwise.hip"), "-o", os.path.join(out, "pw_%s.o" % name)]
p = subprocess.run(base + ["-###"], capture_output=True, text=True)
steps = [shlex.split(l.strip()) for l in p.stderr.split("\n") if l.startswith(' "')]
assert len(steps) == 4, len(steps)
dev, lld, bundler, host = steps
dev_o = [a for a in dev if a.endswith(".o") and "gfx950" in a][0]
hsaco = [a for a in lld if a.endswith(".out")][0]
fb = [a for a in host if a.endswith(".hipfb")][0]
tmp = os.path.join(out, "tmp_" + name)
os.makedirs(tmp, exist_ok=True)
s_path, o_path, co_path, fb_path = (os.path.join(tmp, f) for f in ("dev.s", "dev.o", "dev.out", "dev.hipfb"))
# 1. device compile to assembly
cmd = [a if a != dev_o else s_path for a in dev]
cmd = ["-S" if a == "-emit-obj" else a for a in cmd]
subprocess.run(cmd, check=True)
text = open(s_path).read()


def vregs(tok):
    o = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        o.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        o.add(int(m.group(1)))
    return o


if patch.startswith("pkgap:"):
    n = int(patch.split(":")[1])
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z26pool_lrn_bwd_stream_kernelILi5ELi3ELb1ELi0E") and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    code = [i for i in range(start, end) if lines[i].startswith("\t") and lines[i].strip() and not lines[i].strip().startswith((".", ";"))]
    ins_at = []
    for k, i in enumerate(code):
        t = lines[i].split(";")[0].strip()
        op = t.split()[0]
        if not (op.startswith("v_pk_") and op.endswith("_f32")):
            continue
        srcs = set()
        for o in t.split(None, 1)[1].split(",")[1:]:
            srcs |= vregs(o)
        for j in code[k + 1:k + 4]:
            t2 = lines[j].split(";")[0].strip()
            if re.match(r"(ds_read|buffer_load|global_load)", t2) and " lds" not in t2 and vregs(t2.split(None, 1)[1].split(",")[0]) & srcs:
                ins_at.append(j)
                break
    nops = []
    left = n
    while left > 0:
        c = min(left, 16)
        nops.append("\ts_nop %d" % (c - 1))
        left -= c
    for j in sorted(set(ins_at), reverse=True):
        lines[j:j] = nops
    print("pkgap: %d load(s) delayed by %d wait states" % (len(set(ins_at)), n))
    text = "\n".join(lines)
elif patch.startswith("scalarize:"):
    # v_pk_{mul,add,fma}_f32 -> two scalar instructions on the same registers (everything else byte for byte the same).
    # scalarize:all = every one in pool_lrn_bwd_stream_kernel<5,3,true,0>; scalarize:war = only those whose sources a load issued
    # within the next 3 instructions overwrites; scalarize:notwar = all the others
    which = patch.split(":")[1]
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z26pool_lrn_bwd_stream_kernelILi5ELi3ELb1ELi0E") and ":" in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    code = [i for i in range(start, end) if lines[i].startswith("\t") and lines[i].strip() and not lines[i].strip().startswith((".", ";"))]
    done = kept = 0
    repl = {}
    for k, i in enumerate(code):
        t = lines[i].split(";")[0].strip()
        m = re.match(r"v_pk_(mul|add|fma)_f32 (.*)$", t)
        if not m:
            continue
        kind, rest = m.groups()
        mods = dict((a, [int(x) for x in b.split(",")]) for a, b in re.findall(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[([0-9,]+)\]", rest))
        opnds = [o.strip() for o in re.sub(r"\s*(op_sel|op_sel_hi|neg_lo|neg_hi):\[[0-9,]+\]", "", rest).split(",")]
        pairs = [re.match(r"v\[(\d+):(\d+)\]$", o) for o in opnds]
        if not all(pairs) or "neg_lo" in mods or "neg_hi" in mods:
            kept += 1
            continue
        regs = [(int(q.group(1)), int(q.group(2))) for q in pairs]
        nsrc = len(regs) - 1
        osl = mods.get("op_sel", [0] * nsrc) + [0] * nsrc
        osh = mods.get("op_sel_hi", [1] * nsrc) + [1] * nsrc
        lo = (regs[0][0], [regs[1 + j][osl[j]] for j in range(nsrc)])
        hi = (regs[0][1], [regs[1 + j][osh[j]] for j in range(nsrc)])
        srcs = set(r for pr in regs[1:] for r in pr)
        war = False
        for j in code[k + 1:k + 4]:
            t2 = lines[j].split(";")[0].strip()
            if re.match(r"(ds_read|buffer_load|global_load)", t2) and " lds" not in t2 and vregs(t2.split(None, 1)[1].split(",")[0]) & srcs:
                war = True
        if (which == "war" and not war) or (which == "notwar" and war):
            continue
        pk_index = getattr(sys.modules[__name__], "_pk_i", 0)
        sys.modules[__name__]._pk_i = pk_index + 1
        if which.startswith("keep=") and pk_index in [int(x) for x in which[5:].split("+")]:
            print("   kept packed: #%d line %d: %s" % (pk_index, i - start, t))
            continue
        if lo[0] not in hi[1]:
            order = [lo, hi]
        elif hi[0] not in lo[1]:
            order = [hi, lo]
        else:
            kept += 1
            continue
        opname = {"mul": "v_mul_f32", "add": "v_add_f32", "fma": "v_fma_f32"}[kind]
        repl[i] = ["\t%s v%d, %s" % (opname, d, ", ".join("v%d" % r for r in ss)) for d, ss in order]
        done += 1
    for i in sorted(repl, reverse=True):
        lines[i:i + 1] = repl[i]
    print("scalarize:%s: %d packed instruction(s) replaced, %d left packed" % (which, done, kept))
    text = "\n".join(lines)
elif patch != "none":
    raise SystemExit("unknown patch " + patch)
open(s_path, "w").write(text)
# 2. assemble, link, bundle, host compile
clang = dev[0]
subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s_path, "-o", o_path], check=True)
subprocess.run([a if a != dev_o else o_path for a in [b if b != hsaco else co_path for b in lld]], check=True)
subprocess.run([a.replace(hsaco, co_path).replace(fb, fb_path) for a in bundler], check=True)
subprocess.run([a if a != fb else fb_path for a in host], check=True)
lib = os.path.join(out, "libvltf_hip_%s.so" % name)
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [os.path.join(src, "obj", f + ".o") for f in ("api", "mfma_gemm")] +
               [os.path.join(out, "pw_%s.o" % name)] + [os.path.join(src, "obj", f + ".o") for f in ("lstm_cluster", "resize", "conv_c8")], check=True)
print("built", lib)

"""The built device code must not contain the packed-fp32 instruction form that gives wrong results on MI355X under co-residency
(round 4): `v_pk_{add,mul,fma}_f32` with an op_sel bit set -- with op_sel[1] the low result comes out as src0.lo + 0 in lanes
48..63 while waves of a kernel mixing v_cvt_pk_bf16_f32 / v_pk_add_f32 and bf16 MFMAs (the split-bf16 convolutions) share the CU
(tools/ubench/pk_opsel_raw.hip, pk_cross_kernel.hip; found through conv2's pool / LRN backward, whose SLP-vectorised code held two of
them).  csrc/build.sh compiles with -fno-slp-vectorize; this test disassembles every gfx950 code object of the built library and
refuses the form wherever it comes from.  No GPU needed."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

LIB = os.path.join(ROOT, "video-learning-tf_amd", "libvltf_hip.so")


@pytest.mark.skipif(not os.path.exists(isa_lint.OBJDUMP), reason="llvm-objdump of the ROCm image not found")
def test_built_library_holds_no_refused_packed_fp32_form():
    assert os.path.exists(LIB), "build the library first (__graft_entry__.build())"
    hits, seen = isa_lint.lint(LIB)
    assert seen > 100000, "the disassembly looks empty (%d instructions): did the bundle layout change?" % seen
    assert not hits, "refused instruction forms in the built library:\n" + "\n".join("%s: %s" % h for h in hits[:20])


def test_the_lint_recognises_the_form():
    bad = ["v_pk_add_f32 v[18:19], v[18:19], v[100:101] op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]",
           "v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,0,1] op_sel_hi:[1,1,0]"]
    good = ["v_pk_add_f32 v[0:1], v[2:3], v[4:5]", "v_pk_add_f32 v[0:1], v[2:3], v[4:5] neg_lo:[0,1] neg_hi:[0,1]",
            "v_pk_mul_f32 v[96:97], v[34:35], v[34:35] op_sel_hi:[0,1]", "v_pk_fma_f32 v[18:19], v[18:19], v[32:33], v[12:13] op_sel_hi:[1,1,0]",
            "v_pk_mov_b32 v[18:19], v[88:89], v[96:97] op_sel:[1,0]", "v_add_f32_e32 v1, v2, v3"]
    assert all(isa_lint.BAD.search(t) for t in bad)
    assert not any(isa_lint.BAD.search(t) for t in good)

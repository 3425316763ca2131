"""The bf16 conv PATH (csrc/conv_c8.hip; include/vltf.h "c8"): operands packed bf16 in memory, v_mfma_f32_32x32x16_bf16, fp32
accumulation.  The arithmetic is exactly "round both operands to bf16 (nearest even), multiply-accumulate in fp32", so the oracle
evaluated on bf16-ROUNDED inputs must agree to fp32 summation-order error -- the tolerances below are the fp32 ones, not bf16 ones.
Parity unpinned: the reference has no bf16 path; this is BASELINE config 5's reduced-precision conv arithmetic."""
import math

import numpy as np
import pytest
import torch

from oracle import lrcn_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import vltf_amd.ops as ops_
    return ops_


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().float().numpy()


def nchw(a):
    return np.ascontiguousarray(np.transpose(a, (0, 3, 1, 2)))


def nhwc(a):
    return np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1)))


def to_c8(a_nhwc, halo):
    """NHWC fp32 (values already bf16-representable) -> device bf16 [n][cb][h + 2 halo][w + 2 halo][8]."""
    n, h, w, c = a_nhwc.shape
    cb = (c + 7) // 8
    full = np.zeros((n, h + 2 * halo, w + 2 * halo, cb * 8), np.float32)
    full[:, halo:halo + h, halo:halo + w, :c] = a_nhwc
    t = torch.from_numpy(np.ascontiguousarray(full.reshape(n, h + 2 * halo, w + 2 * halo, cb, 8).transpose(0, 3, 1, 2, 4)))
    return t.to(DEV).bfloat16().contiguous()


def from_c8(t, c, halo):
    """device c8 -> host NHWC fp32 interior, + the halo / channel-padding values (must be zero)."""
    torch.cuda.synchronize()
    a = t.float().cpu().numpy()                                   # [n][cb][hp][wp][8]
    n, cb, hp, wp, _ = a.shape
    full = a.transpose(0, 2, 3, 1, 4).reshape(n, hp, wp, cb * 8)
    inner = full[:, halo:hp - halo, halo:wp - halo, :c]
    outside = full.copy()
    outside[:, halo:hp - halo, halo:wp - halo, :c] = 0
    return np.ascontiguousarray(inner), outside


def close(got, want, rtol=3e-5, atol_rel=3e-5, msg=""):
    want = np.asarray(want, np.float64)
    scale = float(np.abs(want).max()) or 1.0
    np.testing.assert_allclose(np.asarray(got, np.float64), want, rtol=rtol, atol=atol_rel * scale, err_msg=msg)


def pad_nchw(a_nchw, halo):
    return np.pad(a_nchw, ((0, 0), (0, 0), (halo, halo), (halo, halo)))


def test_pack_c8(ops):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 7, 9, 20)).astype(np.float32)                      # 20 channels: the third block is half empty
    xd = torch.from_numpy(pad_nchw(nchw(x), 2)).to(DEV)
    xb = torch.zeros(ops.c8_shape(3, 20, 7, 9, 1), dtype=torch.bfloat16, device=DEV)
    ops.pack_c8(xd, xb, 2, 1)
    inner, outside = from_c8(xb, 20, 1)
    assert np.array_equal(inner, bf16_round(x))
    assert not outside.any()


# n, h, w, cin, cout, k, groups: AlexNet's stride-1 layers (conv2 .. conv5; conv4's 192-channel groups take the 192-wide tile), a
# narrow layer (24 output channels per group: the 64-wide tile; 18 taps -> a padded last stage) a batch that ends inside a pixel tile, and
# 3x3 planes (wgrad's sweep of an image is shorter than a 32-position stage)
C8_CASES = [(3, 13, 13, 256, 384, 3, 1), (2, 27, 27, 96, 256, 5, 2), (2, 13, 13, 384, 384, 3, 2), (2, 13, 13, 384, 256, 3, 2),
            (5, 9, 11, 32, 48, 3, 2), (21, 13, 13, 64, 128, 3, 1), (40, 3, 3, 16, 32, 3, 1)]


@pytest.mark.parametrize("n,h,w,cin,cout,k,g", C8_CASES)
def test_conv_c8_fwd_dgrad_wgrad(ops, n, h, w, cin, cout, k, g):
    rng = np.random.default_rng(h * 100 + cin)
    x = bf16_round(np.maximum(rng.standard_normal((n, h, w, cin)), 0))             # post-ReLU-like input (zeros included)
    wt = (rng.standard_normal((k, k, cin // g, cout)) / math.sqrt(k * k * cin / g)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    dy = bf16_round(rng.standard_normal((n, h, w, cout)))
    wr = bf16_round(wt)
    conv = ops.Conv(cin, h, w, cout, k, k, 1, g)
    pad = conv.same_pad()
    conv.set_halo(pad, 1, pad, 2)                                                  # x / dy: the SAME padding; y: 1; dx: 2
    xb, dyb = to_c8(x, pad), to_c8(dy, pad)
    wd = torch.from_numpy(wt).to(DEV)
    wb = torch.zeros(conv.c8_w_bytes(False), dtype=torch.uint8, device=DEV)
    wbt = torch.zeros(conv.c8_w_bytes(True), dtype=torch.uint8, device=DEV)
    conv.c8_pack_w(wd, wb, False)
    conv.c8_pack_w(wd, wbt, True)

    # forward: fp32 NCHW and c8 outputs of the same launch
    z = O.grouped_conv(x, wr, b, 1, g)
    y = torch.zeros((n, cout, h + 2, w + 2), device=DEV)
    yb = torch.zeros(ops.c8_shape(n, cout, h, w, 1), dtype=torch.bfloat16, device=DEV)
    conv.c8_fwd(xb, wb, torch.from_numpy(b).to(DEV), y=y, yb=yb, relu=False)
    torch.cuda.synchronize()
    yh = y.cpu().numpy()
    close(nhwc(yh[:, :, 1:-1, 1:-1]), z, msg="c8 fwd (fp32 out)")
    assert not yh[:, :, 0].any() and not yh[:, :, :, -1].any()
    inner, outside = from_c8(yb, cout, 1)
    assert np.array_equal(inner, bf16_round(nhwc(yh[:, :, 1:-1, 1:-1]))), "c8 output = bf16 rounding of the fp32 output"
    assert not outside.any()
    conv.c8_fwd(xb, wb, torch.from_numpy(b).to(DEV), y=y, relu=True)
    close(nhwc(y.cpu().numpy()[:, :, 1:-1, 1:-1]), np.maximum(z, 0), msg="c8 fwd + relu")
    conv.set_halo(pad, 0, pad, 2)                                                  # dense y: the LDS-staged 16-byte store path
    yd = torch.full((n, cout, h, w), 7.0, device=DEV)
    conv.c8_fwd(xb, wb, torch.from_numpy(b).to(DEV), y=yd, relu=False)
    close(nhwc(yd.cpu().numpy()), z, msg="c8 fwd (dense fp32 out)")
    conv.set_halo(pad, 1, pad, 2)

    dxo, dwo, _ = O.grouped_conv_grad(x, wr, dy, 1, g, need_dx=True)
    # dgrad
    dx = torch.zeros((n, cin, h + 4, w + 4), device=DEV)
    dxb = torch.zeros(ops.c8_shape(n, cin, h, w, 2), dtype=torch.bfloat16, device=DEV)
    conv.c8_dgrad(dyb, wbt, dx=dx, dxb=dxb)
    torch.cuda.synchronize()
    dxh = dx.cpu().numpy()
    close(nhwc(dxh[:, :, 2:-2, 2:-2]), dxo, msg="c8 dgrad")
    inner, outside = from_c8(dxb, cin, 2)
    assert np.array_equal(inner, bf16_round(nhwc(dxh[:, :, 2:-2, 2:-2]))) and not outside.any()
    mask = rng.standard_normal(x.shape).astype(np.float32)
    conv.c8_dgrad(dyb, wbt, dx=dx, relu_mask=torch.from_numpy(pad_nchw(nchw(mask), 2)).to(DEV))
    close(nhwc(dx.cpu().numpy()[:, :, 2:-2, 2:-2]), dxo * (mask > 0), msg="c8 dgrad + mask")
    conv.c8_dgrad(dyb, wbt, dx=dx, relu_mask_c8=xb)                                # ReluGrad from the layer's own packed input
    close(nhwc(dx.cpu().numpy()[:, :, 2:-2, 2:-2]), dxo * (x > 0), msg="c8 dgrad + packed mask")
    db = torch.empty(cout, device=DEV)
    ops.bias_grad_c8(dyb, db, torch.empty(64 * 8 * ((cout + 7) // 8), device=DEV), cout, pad)
    close(db.cpu().numpy(), dy.reshape(-1, cout).sum(0), msg="bias grad from the packed gradient")
    # wgrad (twice: bitwise reproducible)
    dw = torch.empty_like(wd)
    ws = torch.empty(max(conv.c8_wgrad_ws_bytes(n) // 4, 1), device=DEV)
    conv.c8_wgrad(xb, dyb, dw, ws)
    close(dw.cpu().numpy(), dwo, msg="c8 wgrad")
    dw2 = torch.empty_like(wd)
    conv.c8_wgrad(xb, dyb, dw2, ws)
    assert torch.equal(dw, dw2)


def test_c8_matches_the_in_loop_bf16_mode(ops):
    """The c8 path and vl_set_conv_math(1) (fp32 operands rounded inside the loop) are the same arithmetic."""
    rng = np.random.default_rng(3)
    n, h, w, cin, cout, k, g = 4, 13, 13, 256, 384, 3, 1
    x = np.maximum(rng.standard_normal((n, h, w, cin)), 0).astype(np.float32)
    wt = (rng.standard_normal((k, k, cin, cout)) / math.sqrt(k * k * cin)).astype(np.float32)
    conv = ops.Conv(cin, h, w, cout, k, k, 1, g)
    conv.set_halo(1, 0, 1, 0)
    xd = torch.from_numpy(pad_nchw(nchw(x), 1)).to(DEV)
    wd, bd = torch.from_numpy(wt).to(DEV), torch.zeros(cout, device=DEV)
    y1 = torch.empty((n, cout, h, w), device=DEV)
    ops.set_conv_math("bf16")
    try:
        conv.fwd(xd, wd, bd, y1, relu=False)
    finally:
        ops.set_conv_math("f32")
    xb = torch.zeros(ops.c8_shape(n, cin, h, w, 1), dtype=torch.bfloat16, device=DEV)
    ops.pack_c8(xd, xb, 1, 1)
    wb = torch.zeros(conv.c8_w_bytes(False), dtype=torch.uint8, device=DEV)
    conv.c8_pack_w(wd, wb, False)
    y2 = torch.empty_like(y1)
    conv.c8_fwd(xb, wb, bd, y=y2, relu=False)
    torch.cuda.synchronize()
    close(y2.cpu().numpy(), y1.cpu().numpy(), rtol=1e-5, atol_rel=1e-5)


@pytest.mark.parametrize("n,h,w,cout,phase", [(2, 67, 67, 96, True), (3, 37, 41, 24, False), (2, 227, 227, 96, True)])
def test_strided_first_layer_as_a_stride1_layer(ops, n, h, w, cout, phase):
    """conv1 (11x11 / 4 over 3 channels) through vl_s2d_*: the packed space-to-depth input and the rearranged weights run the
    stride-1 c8 kernels and must reproduce the oracle's strided convolution and its weight gradient (bf16-rounded operands)."""
    rng = np.random.default_rng(h)
    k, s, cin = 11, 4, 3
    x = bf16_round(rng.standard_normal((n, h, w, cin)) * 50)
    wt = (rng.standard_normal((k, k, cin, cout)) / math.sqrt(k * k * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    conv = ops.Conv(cin, h, w, cout, k, k, s, 1)
    pad = conv.same_pad()
    conv.set_halo(pad, 0, 0, 0)
    ph = conv.set_x_phase_split(True) if phase else 1
    xp = pad_nchw(nchw(x), pad)
    if ph > 1:
        wq = -(-xp.shape[3] // ph)
        xp = np.pad(xp, ((0, 0), (0, 0), (0, 0), (0, wq * ph - xp.shape[3])))
        xp = np.ascontiguousarray(xp.reshape(n, cin, xp.shape[2], wq, ph).transpose(0, 1, 4, 2, 3)).reshape(n, cin * ph, xp.shape[2], wq)
    x0 = torch.from_numpy(np.ascontiguousarray(xp)).to(DEV)
    eq = conv.s2d_layer()
    assert (eq.cin, eq.kh, eq.stride, eq.oh, eq.ow) == (cin * s * s, 3, 1, conv.oh, conv.ow)
    xb = torch.zeros(ops.c8_shape(n, eq.cin, eq.h, eq.w, 1), dtype=torch.bfloat16, device=DEV)
    conv.s2d_c8_from_x0(x0, xb)
    wd = torch.from_numpy(wt).to(DEV)
    ws2d = torch.empty(eq.w_shape, device=DEV)
    conv.s2d_weights(wd, ws2d)
    wb = torch.zeros(eq.c8_w_bytes(False), dtype=torch.uint8, device=DEV)
    eq.c8_pack_w(ws2d, wb, False)
    y = torch.empty((n, cout, conv.oh, conv.ow), device=DEV)
    eq.c8_fwd(xb, wb, torch.from_numpy(b).to(DEV), y=y, relu=False)
    torch.cuda.synchronize()
    wr = bf16_round(wt)
    close(nhwc(y.cpu().numpy()), O.grouped_conv(x, wr, b, s, 1), msg="strided conv through space-to-depth")
    dy = bf16_round(rng.standard_normal((n, conv.oh, conv.ow, cout)))
    _, dwo, _ = O.grouped_conv_grad(x, wr, dy, s, 1, need_dx=False)
    dws2d, dw = torch.empty(eq.w_shape, device=DEV), torch.empty_like(wd)
    eq.c8_wgrad(xb, to_c8(dy, 1), dws2d, torch.empty(max(eq.c8_wgrad_ws_bytes(n) // 4, 1), device=DEV))
    conv.s2d_weights(dws2d, dw, grad=True)
    close(dw.cpu().numpy(), dwo, msg="strided wgrad through space-to-depth")


def test_frames_straight_into_the_packed_input(ops):
    """vl_input_prep_u8_s2d == vl_input_prep_u8 followed by vl_s2d_c8_from_x0, bit for bit (crop offsets, mirror, mean)."""
    rng = np.random.default_rng(11)
    n, rh, rw, h, w = 5, 80, 90, 67, 71
    frames = torch.from_numpy(rng.integers(0, 256, (n, rh, rw, 3), dtype=np.uint8)).to(DEV)
    cy = torch.from_numpy(rng.integers(0, rh - h + 1, n).astype(np.int32)).to(DEV)
    cx = torch.from_numpy(rng.integers(0, rw - w + 1, n).astype(np.int32)).to(DEV)
    mir = torch.from_numpy(rng.integers(0, 2, n).astype(np.uint8)).to(DEV)
    mean = torch.tensor([99.2, 105.3, 109.5], device=DEV)
    conv = ops.Conv(3, h, w, 96, 11, 11, 4, 1)
    pad = conv.same_pad()
    conv.set_halo(pad, 0, 0, 0)
    ph = conv.set_x_phase_split(True)
    x0 = torch.zeros(ops.phase_split_shape(n, 3, h, w, pad, ph), device=DEV)
    ops.input_prep_u8(frames, x0, cy, cx, mir, mean, halo=pad, phase=ph, out_hw=(h, w))
    shape = ops.c8_shape(n, 48, conv.oh, conv.ow, 1)
    a, b = torch.zeros(shape, dtype=torch.bfloat16, device=DEV), torch.zeros(shape, dtype=torch.bfloat16, device=DEV)
    conv.s2d_c8_from_x0(x0, a)
    conv.input_prep_u8_s2d(frames, b, cy, cx, mir, mean)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and float(a.float().abs().max()) > 50


@pytest.mark.parametrize("n,h,w,c,halo", [(2, 9, 7, 96, 1), (3, 13, 13, 256, 2), (2, 27, 27, 16, 1), (1, 55, 55, 96, 1)])
@pytest.mark.parametrize("relu,ranges", [(True, ""), (False, ""), (True, "1"), (True, "3")])
def test_pool_lrn_bwd_packed_output(ops, request, n, h, w, c, halo, relu, ranges):
    """vl_pool_lrn_bwd_c8 = vl_pool_lrn_bwd with the result rounded to bf16 into the c8 layout (same arithmetic before the rounding);
    `ranges`: the forced count of channel ranges both kernels split a (band, image) into (multiples of 8 channels for the packed one)."""
    ops.pool_lrn_bwd_test_ranges(int(ranges or 0))
    request.addfinalizer(lambda: ops.pool_lrn_bwd_test_ranges(0))
    rng = np.random.default_rng(c + h)
    x = torch.from_numpy(np.maximum(rng.standard_normal((n, c, h, w)) * 30, 0).astype(np.float32)).to(DEV)
    oh, ow = ops.pool_out(h), ops.pool_out(w)
    p = torch.zeros((n, c, oh + 2, ow + 2), device=DEV)
    arg = torch.zeros((n, c, oh + 2, ow + 2), dtype=torch.uint8, device=DEV)
    ops.lrn_pool_fwd(x, p, arg, p_halo=1)
    dp = torch.zeros_like(p)
    dp[:, :, 1:-1, 1:-1] = torch.from_numpy(rng.standard_normal((n, c, oh, ow)).astype(np.float32)).to(DEV)
    dx = torch.zeros((n, c, h + 2 * halo, w + 2 * halo), device=DEV)
    ops.pool_lrn_bwd(x, dp, arg, dx, p_halo=1, dx_halo=halo, relu_fused=relu)
    dxb = torch.zeros(ops.c8_shape(n, c, h, w, halo), dtype=torch.bfloat16, device=DEV)
    ops.pool_lrn_bwd_c8(x, dp, arg, dxb, p_halo=1, dxb_halo=halo, relu_fused=relu)
    want = torch.zeros_like(dxb)
    ops.pack_c8(dx, want, halo, halo)
    torch.cuda.synchronize()
    # the two instantiations (5- and 8-channel chunks) may contract an fma differently: the last fp32 bit, hence now and then the
    # bf16 rounding, can differ -- every element within one bf16 ulp of the fp32 kernel's, all but a handful identical
    a, b = dxb.float(), want.float()
    assert float((a - b).abs().max()) <= 2.0 ** -7 * float(b.abs().max())
    assert torch.all((a - b).abs() <= 2.0 ** -7 * b.abs() + 1e-30)
    assert float((a != b).float().mean()) < 1e-3 and float(a.abs().max()) > 0


@pytest.mark.parametrize("n,h,w,c,halo", [(2, 9, 7, 96, 1), (3, 13, 13, 256, 2), (1, 55, 55, 96, 2)])
def test_lrn_pool_fwd_packed_output(ops, n, h, w, c, halo):
    """vl_lrn_pool_fwd_c8 = vl_lrn_pool_fwd with the pooled output rounded to bf16 into the c8 layout; same arg-max map."""
    rng = np.random.default_rng(c + w)
    x = torch.from_numpy(np.maximum(rng.standard_normal((n, c, h, w)) * 30, 0).astype(np.float32)).to(DEV)
    oh, ow = ops.pool_out(h), ops.pool_out(w)
    p = torch.zeros((n, c, oh + 2 * halo, ow + 2 * halo), device=DEV)
    arg, arg2 = torch.zeros(p.shape, dtype=torch.uint8, device=DEV), torch.zeros(p.shape, dtype=torch.uint8, device=DEV)
    ops.lrn_pool_fwd(x, p, arg, p_halo=halo)
    pb = torch.zeros(ops.c8_shape(n, c, oh, ow, halo), dtype=torch.bfloat16, device=DEV)
    ops.lrn_pool_fwd_c8(x, pb, arg2, p_halo=halo)
    want = torch.zeros_like(pb)
    ops.pack_c8(p, want, halo, halo)
    torch.cuda.synchronize()
    assert torch.equal(pb, want) and torch.equal(arg, arg2) and float(pb.float().abs().max()) > 0


@pytest.mark.parametrize("n,h,w,c", [(2, 9, 7, 96), (3, 28, 28, 256), (1, 57, 57, 96), (2, 13, 11, 20)])
def test_pool_kernels_read_the_packed_conv_output(ops, n, h, w, c):
    """x_packed: the LRN input handed over as the conv epilogue wrote it (bf16 c8, no halo).  Same results, bit for bit, as the fp32
    input holding the same (bf16-representable) values."""
    rng = np.random.default_rng(c * h)
    x = bf16_round(np.maximum(rng.standard_normal((n, h, w, c)) * 30, 0))
    xf = torch.from_numpy(nchw(x)).to(DEV)
    xp = to_c8(x, 0)
    oh, ow = ops.pool_out(h), ops.pool_out(w)
    shape = (n, c, oh + 2, ow + 2)
    pb1, pb2 = (torch.zeros(ops.c8_shape(n, c, oh, ow, 1), dtype=torch.bfloat16, device=DEV) for _ in range(2))
    a1, a2 = (torch.zeros(shape, dtype=torch.uint8, device=DEV) for _ in range(2))
    ops.lrn_pool_fwd_c8(xf, pb1, a1, p_halo=1)
    ops.lrn_pool_fwd_c8(xp, pb2, a2, p_halo=1, channels=c)
    dp = torch.zeros(shape, device=DEV)
    dp[:, :, 1:-1, 1:-1] = torch.from_numpy(rng.standard_normal((n, c, oh, ow)).astype(np.float32)).to(DEV)
    d1, d2 = (torch.zeros(ops.c8_shape(n, c, h, w, 2), dtype=torch.bfloat16, device=DEV) for _ in range(2))
    ops.pool_lrn_bwd_c8(xf, dp, a1, d1, p_halo=1, dxb_halo=2)
    ops.pool_lrn_bwd_c8(xp, dp, a1, d2, p_halo=1, dxb_halo=2)
    torch.cuda.synchronize()
    assert torch.equal(pb1, pb2) and torch.equal(a1, a2) and torch.equal(d1, d2)
    assert float(pb1.float().abs().max()) > 0 and float(d1.float().abs().max()) > 0


@pytest.mark.parametrize("m,n,k,bias,relu", [(256, 128, 64, False, False), (256, 256, 1000, False, False), (64, 72, 333, True, True),
                                              (1024, 512, 2304, True, False), (8, 8, 5, False, True)])
def test_dense_product_on_the_wgrad_kernel(ops, m, n, k, bias, relu):
    """vl_pack_kc8 + vl_gemm_kc8: c = a^T b (+ bias) (ReLU) with bf16 products, against fp64 on the bf16-rounded operands; operands
    packed from a matrix and from a transposed view (strides); split-k slabs, the direct (one slab) form and ragged k."""
    rng = np.random.default_rng(m + n + k)
    a = rng.standard_normal((k, m)).astype(np.float32)           # [k][m]
    bt = rng.standard_normal((n, k)).astype(np.float32)          # b stored TRANSPOSED [n][k]: packed through strides
    bv = rng.standard_normal(n).astype(np.float32) if bias else None
    ad, btd = torch.from_numpy(a).to(DEV), torch.from_numpy(bt).to(DEV)
    ak = torch.zeros(ops.kc8_shape(k, m), dtype=torch.bfloat16, device=DEV)
    bk = torch.zeros(ops.kc8_shape(k, n), dtype=torch.bfloat16, device=DEV)
    ops.pack_kc8(ad, ak, k, m, m, 1)                              # element (position kk, channel i) = a[kk][i]
    ops.pack_kc8(btd, bk, k, n, 1, k)                             # element (position kk, channel j) = bt[j][kk]
    c = torch.full((m, n), 7.0, device=DEV)
    ws = torch.empty(max(ops.gemm_kc8_ws_bytes(m, n, k) // 4, 1), device=DEV)
    ops.gemm_kc8(ak, bk, c, m, n, k, bias=None if bv is None else torch.from_numpy(bv).to(DEV), relu=relu, ws=ws)
    want = bf16_round(a).astype(np.float64).T @ bf16_round(bt).astype(np.float64).T
    if bv is not None:
        want = want + bv
    if relu:
        want = np.maximum(want, 0)
    torch.cuda.synchronize()
    close(c.cpu().numpy(), want, msg="gemm_kc8")
    c2 = torch.empty_like(c)
    ops.gemm_kc8(ak, bk, c2, m, n, k, bias=None if bv is None else torch.from_numpy(bv).to(DEV), relu=relu, ws=ws)
    assert torch.equal(c, c2)


def test_the_packed_path_refuses_what_it_cannot_run(ops):
    """Fail-fast convention of the C-ABI (utils_.error in the reference: raise): every unsupported geometry is an error, not a fallback."""
    from vltf_amd._ffi import VltfError
    dev_u8 = lambda nbytes: torch.zeros(max(nbytes, 16), dtype=torch.uint8, device=DEV)
    # channels per group not a multiple of 8
    conv = ops.Conv(12, 9, 9, 16, 3, 3, 1, 1)
    conv.set_halo(1, 0, 1, 0)
    xb = torch.zeros(ops.c8_shape(2, 12, 9, 9, 1), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(VltfError, match="multiple of 8"):
        conv.c8_fwd(xb, dev_u8(conv.c8_w_bytes(False)), torch.zeros(16, device=DEV), y=torch.zeros((2, 16, 9, 9), device=DEV))
    # input without the SAME halo: the gather would need bounds tests the packed kernels do not have
    conv = ops.Conv(16, 9, 9, 16, 3, 3, 1, 1)
    conv.set_halo(0, 0, 0, 0)
    xb = torch.zeros(ops.c8_shape(2, 16, 9, 9, 0), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(VltfError, match="padded layout"):
        conv.c8_fwd(xb, dev_u8(conv.c8_w_bytes(False)), torch.zeros(16, device=DEV), y=torch.zeros((2, 16, 9, 9), device=DEV))
    # wgrad needs x and dy in one plane geometry
    conv.set_halo(1, 0, 2, 0)
    xb = torch.zeros(ops.c8_shape(2, 16, 9, 9, 1), dtype=torch.bfloat16, device=DEV)
    dyb = torch.zeros(ops.c8_shape(2, 16, 9, 9, 2), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(VltfError, match="x_halo == dy_halo"):
        conv.c8_wgrad(xb, dyb, torch.zeros(conv.w_shape, device=DEV), torch.zeros(1 << 20, device=DEV))
    # strided layers have no packed dgrad / wgrad of their own (conv1 goes through s2d_layer())
    conv = ops.Conv(16, 9, 9, 16, 3, 3, 2, 1)
    conv.set_halo(1, 0, 1, 0)
    with pytest.raises(VltfError):
        conv.c8_dgrad(torch.zeros(ops.c8_shape(2, 16, 5, 5, 1), dtype=torch.bfloat16, device=DEV), dev_u8(conv.c8_w_bytes(True)),
                      dx=torch.zeros((2, 16, 9, 9), device=DEV))
    # wrong operand layout / dtype is caught before the call
    with pytest.raises(VltfError):
        conv.c8_fwd(torch.zeros((2, 16, 11, 11), device=DEV), dev_u8(64), torch.zeros(16, device=DEV), y=torch.zeros((2, 16, 5, 5), device=DEV))
    # dense product: m, n in whole 8-channel blocks
    with pytest.raises(VltfError):
        ops.gemm_kc8(torch.zeros(ops.kc8_shape(16, 12), dtype=torch.bfloat16, device=DEV), torch.zeros(ops.kc8_shape(16, 8), dtype=torch.bfloat16, device=DEV),
                     torch.zeros((12, 8), device=DEV), 12, 8, 16, ws=torch.zeros(1 << 16, device=DEV))

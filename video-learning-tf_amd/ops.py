"""Thin torch-tensor wrappers over the C-ABI (include/vltf.h).  Tensors are device allocations
only: every wrapper passes ``data_ptr()`` and the current HIP stream; no torch math runs here."""
import ctypes as C

import torch

from . import _ffi

F32 = torch.float32


def stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def _f32(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == F32):
            raise _ffi.VltfError("expected a CUDA/HIP float32 tensor, got %s on %s" % (t.dtype, t.device))


def _dense(*ts):
    for t in ts:
        if t is not None and not t.is_contiguous():
            raise _ffi.VltfError("expected a contiguous tensor")


# ---- input ---------------------------------------------------------------------------------------
def phase_split_shape(n, c, h, w, halo, phase):
    """Shape of an activation tensor with a zero halo, plain (phase 1) or column-phase-split (vl_conv_set_x_phase_split)."""
    if phase <= 1:
        return (n, c, h + 2 * halo, w + 2 * halo)
    return (n, c * phase, h + 2 * halo, -(-(w + 2 * halo) // phase))


def input_prep_u8(src, dst, crop_y=None, crop_x=None, mirror=None, mean_bgr=None, halo=0, phase=1, out_hw=None):
    """src uint8 [n, raw_h, raw_w, 3] (HWC, BGR) -> dst f32 [n, 3, out_h + 2 halo, out_w + 2 halo] (dataset_.py:481-501);
    phase > 1: dst is phase_split_shape(n, 3, *out_hw, halo, phase) and out_hw must be given."""
    if src.dtype != torch.uint8 or not src.is_cuda:
        raise _ffi.VltfError("input_prep_u8: src must be a device uint8 tensor")
    _f32(dst, mean_bgr)
    _dense(src, dst, crop_y, crop_x, mirror, mean_bgr)
    n, rh, rw, c = src.shape
    if out_hw is None:
        if phase > 1:
            raise _ffi.VltfError("input_prep_u8: out_hw is required with a phase-split destination")
        out_hw = (dst.shape[2] - 2 * halo, dst.shape[3] - 2 * halo)
    if c != 3 or tuple(dst.shape) != phase_split_shape(n, 3, out_hw[0], out_hw[1], halo, phase):
        raise _ffi.VltfError("input_prep_u8: shape mismatch %s -> %s" % (tuple(src.shape), tuple(dst.shape)))
    _ffi.call("vl_input_prep_u8", _p(src), _p(dst), n, rh, rw, out_hw[0], out_hw[1], _p(crop_y), _p(crop_x), _p(mirror),
              _p(mean_bgr), halo, phase, stream())


def nhwc_to_nchw(src, dst, halo=0, phase=1):
    _f32(src, dst); _dense(src, dst)
    n, h, w, c = src.shape
    if tuple(dst.shape) != phase_split_shape(n, c, h, w, halo, phase):
        raise _ffi.VltfError("nhwc_to_nchw: shape mismatch %s -> %s (halo %d)" % (tuple(src.shape), tuple(dst.shape), halo))
    _ffi.call("vl_nhwc_to_nchw", _p(src), _p(dst), n, h, w, c, halo, phase, stream())


def nchw_to_nhwc(src, dst):
    _f32(src, dst); _dense(src, dst)
    n, c, h, w = src.shape
    _ffi.call("vl_nchw_to_nhwc", _p(src), _p(dst), n, c, h, w, stream())


# ---- convolution ---------------------------------------------------------------------------------
class Conv:
    """Descriptor for one dcnn.conv layer (alexnet.py:15-31): TF 'SAME' conv per group."""

    def __init__(self, cin, h, w, cout, kh, kw, stride, groups):
        self.cin, self.h, self.w, self.cout = cin, h, w, cout
        self.kh, self.kw, self.stride, self.groups = kh, kw, stride, groups
        self._d = C.c_void_p()
        _ffi.check(_ffi.lib().vl_conv_create(C.byref(self._d), cin, h, w, cout, kh, kw, stride, groups), "vl_conv_create")
        oh, ow = C.c_int(), C.c_int()
        _ffi.check(_ffi.lib().vl_conv_out_hw(self._d, C.byref(oh), C.byref(ow)), "vl_conv_out_hw")
        self.oh, self.ow = oh.value, ow.value
        self.w_shape = (kh, kw, cin // groups, cout)
        self.x_halo = self.y_halo = self.dy_halo = self.dx_halo = 0

    def same_pad(self):
        """Largest SAME padding on any side (TF rule): the halo that makes the gather test-free."""
        def one(n, k, s):
            out = -(-n // s)
            tot = max((out - 1) * s + k - n, 0)
            return tot - tot // 2
        return max(one(self.h, self.kh, self.stride), one(self.w, self.kw, self.stride))

    def set_halo(self, x_halo=0, y_halo=0, dy_halo=0, dx_halo=0):
        """Declare the zero-halo layouts of x / y / dy / dx (include/vltf.h: vl_conv_set_halo)."""
        _ffi.call("vl_conv_set_halo", self._d, x_halo, y_halo, dy_halo, dx_halo)
        self.x_halo, self.y_halo, self.dy_halo, self.dx_halo = x_halo, y_halo, dy_halo, dx_halo
        self.x_phase = int(_ffi.lib().vl_conv_x_phase(self._d))

    def set_x_phase_split(self, on=True):
        """Store x column-phase-split (strided convs in the padded layout; include/vltf.h).  Returns the phase count."""
        _ffi.call("vl_conv_set_x_phase_split", self._d, int(on))
        self.x_phase = int(_ffi.lib().vl_conv_x_phase(self._d))
        return self.x_phase

    def _shape(self, n, c, h, w, halo):
        return (n, c, h + 2 * halo, w + 2 * halo)

    def x_shape(self, n):
        return phase_split_shape(n, self.cin, self.h, self.w, self.x_halo, getattr(self, "x_phase", 1))

    def __del__(self):
        try:
            if self._d:
                _ffi.lib().vl_conv_destroy(self._d)
                self._d = None
        except Exception:
            pass

    def fwd(self, x, w, bias, y, relu=True):
        _f32(x, w, bias, y); _dense(x, w, bias, y)
        n = x.shape[0]
        if tuple(x.shape) != self.x_shape(n) or tuple(y.shape) != self._shape(n, self.cout, self.oh, self.ow, self.y_halo):
            raise _ffi.VltfError("conv.fwd: shape mismatch x=%s y=%s (halos %d, %d)" % (tuple(x.shape), tuple(y.shape),
                                                                                        self.x_halo, self.y_halo))
        _ffi.call("vl_conv_fwd", self._d, _p(x), _p(w), _p(bias), _p(y), n, int(relu), stream())

    def wt_transpose(self, w, wt):
        _f32(w, wt); _dense(w, wt)
        _ffi.call("vl_conv_wt_transpose", self._d, _p(w), _p(wt), stream())

    def dgrad(self, dy, wt, dx, relu_mask=None):
        _f32(dy, wt, dx, relu_mask); _dense(dy, wt, dx, relu_mask)
        n = dy.shape[0]
        if tuple(dy.shape) != self._shape(n, self.cout, self.oh, self.ow, self.dy_halo) or \
                tuple(dx.shape) != self._shape(n, self.cin, self.h, self.w, self.dx_halo) or \
                (relu_mask is not None and tuple(relu_mask.shape) != self._shape(n, self.cin, self.h, self.w, self.x_halo)):
            raise _ffi.VltfError("conv.dgrad: shape mismatch dy=%s dx=%s" % (tuple(dy.shape), tuple(dx.shape)))
        _ffi.call("vl_conv_dgrad", self._d, _p(dy), _p(wt), _p(dx), _p(relu_mask), dy.shape[0], stream())

    def wgrad_ws_bytes(self, n):
        return int(_ffi.lib().vl_conv_wgrad_ws_bytes(self._d, n))

    def fuses_bias(self):
        return bool(_ffi.lib().vl_conv_wgrad_fuses_bias(self._d))

    def wgrad(self, x, dy, dw, ws, db=None):
        """dw (and, when fuses_bias(), db = sum of dy over n,h,w in the same pass)."""
        _f32(x, dy, dw, db); _dense(x, dy, dw, ws, db)
        n = x.shape[0]
        if tuple(x.shape) != self.x_shape(n) or tuple(dy.shape) != self._shape(n, self.cout, self.oh, self.ow, self.dy_halo):
            raise _ffi.VltfError("conv.wgrad: shape mismatch x=%s dy=%s" % (tuple(x.shape), tuple(dy.shape)))
        nbytes = 0 if ws is None else ws.numel() * ws.element_size()
        _ffi.call("vl_conv_wgrad", self._d, _p(x), _p(dy), _p(dw), _p(db), _p(ws), nbytes, x.shape[0], stream())


    # ---- the bf16 path: operands in the c8 layout [n][ceil(c/8)][h + 2 halo][w + 2 halo][8] bf16 (include/vltf.h) -------------
    def c8_w_bytes(self, bwd=False):
        return int(_ffi.lib().vl_conv_c8_w_bytes(self._d, int(bwd)))

    def c8_pack_w(self, w, wb, bwd=False):
        """HWIO fp32 weights -> packed bf16 operand of c8_fwd (bwd=False) / c8_dgrad (bwd=True); wb: uint8 tensor of c8_w_bytes."""
        _f32(w); _dense(w, wb)
        if wb.numel() * wb.element_size() < self.c8_w_bytes(bwd):
            raise _ffi.VltfError("conv.c8_pack_w: weight buffer too small")
        _ffi.call("vl_conv_c8_pack_w", self._d, _p(w), _p(wb), int(bwd), stream())

    def _c8_check(self, t, n, c, h, w, halo, what):
        if t is not None and (t.dtype != torch.bfloat16 or tuple(t.shape) != c8_shape(n, c, h, w, halo) or not t.is_contiguous()):
            raise _ffi.VltfError("conv: %s must be a contiguous bf16 c8 tensor %s, got %s %s" % (what, c8_shape(n, c, h, w, halo),
                                                                                                 t.dtype, tuple(t.shape)))

    def c8_fwd(self, xb, wb, bias, y=None, yb=None, relu=True):
        n = xb.shape[0]
        _f32(bias, y); _dense(bias, y, wb)
        self._c8_check(xb, n, self.cin, self.h, self.w, self.x_halo, "xb")
        self._c8_check(yb, n, self.cout, self.oh, self.ow, self.y_halo, "yb")
        if y is not None and tuple(y.shape) != self._shape(n, self.cout, self.oh, self.ow, self.y_halo):
            raise _ffi.VltfError("conv.c8_fwd: y shape %s" % (tuple(y.shape),))
        _ffi.call("vl_conv_c8_fwd", self._d, _p(xb), _p(wb), _p(bias), _p(y), _p(yb), n, int(relu), stream())

    def c8_dgrad(self, dyb, wbt, dx=None, dxb=None, relu_mask=None, relu_mask_c8=None):
        """relu_mask (fp32, dx's layout) or relu_mask_c8 (this layer's packed input xb): fused ReluGrad of the producing layer."""
        n = dyb.shape[0]
        self._c8_check(relu_mask_c8, n, self.cin, self.h, self.w, self.x_halo, "relu_mask_c8")
        _f32(dx, relu_mask); _dense(dx, relu_mask, wbt)
        self._c8_check(dyb, n, self.cout, self.oh, self.ow, self.dy_halo, "dyb")
        self._c8_check(dxb, n, self.cin, self.h, self.w, self.dx_halo, "dxb")
        for t in (dx, relu_mask):
            if t is not None and tuple(t.shape) != self._shape(n, self.cin, self.h, self.w, self.dx_halo):
                raise _ffi.VltfError("conv.c8_dgrad: dx / relu_mask shape %s" % (tuple(t.shape),))
        _ffi.call("vl_conv_c8_dgrad", self._d, _p(dyb), _p(wbt), _p(dx), _p(dxb), _p(relu_mask), _p(relu_mask_c8), n, stream())

    # a strided first layer as a stride-1 layer over its space-to-depth input (include/vltf.h: vl_s2d_*)
    def s2d_layer(self):
        """Descriptor of the equivalent stride-1 layer (halos 1 for x and dy, dense y)."""
        s, ka = self.stride, (self.kh - 1) // self.stride + 1
        eq = Conv(self.cin * s * s, self.oh, self.ow, self.cout, ka, ka, 1, 1)
        eq.set_halo((ka - 1) // 2, 0, (ka - 1) // 2, 0)
        return eq

    def s2d_c8_from_x0(self, x0, xb):
        _f32(x0); _dense(x0, xb)
        n = x0.shape[0]
        s, ka = self.stride, (self.kh - 1) // self.stride + 1
        if tuple(x0.shape) != self.x_shape(n) or xb.dtype != torch.bfloat16 or \
                tuple(xb.shape) != c8_shape(n, self.cin * s * s, self.oh, self.ow, (ka - 1) // 2):
            raise _ffi.VltfError("conv.s2d_c8_from_x0: shape mismatch x0=%s xb=%s" % (tuple(x0.shape), tuple(xb.shape)))
        _ffi.call("vl_s2d_c8_from_x0", self._d, _p(x0), _p(xb), n, stream())

    def input_prep_u8_s2d(self, src, xb, crop_y=None, crop_x=None, mirror=None, mean_bgr=None):
        """uint8 frames [n, raw_h, raw_w, 3] -> the packed space-to-depth input (= input_prep_u8 + s2d_c8_from_x0)."""
        if src.dtype != torch.uint8 or not src.is_cuda or src.dim() != 4 or src.shape[3] != self.cin:
            raise _ffi.VltfError("conv.input_prep_u8_s2d: src must be a device uint8 [n, h, w, %d] tensor" % self.cin)
        _f32(mean_bgr); _dense(src, xb, mean_bgr)
        n = src.shape[0]
        s, ka = self.stride, (self.kh - 1) // self.stride + 1
        if xb.dtype != torch.bfloat16 or tuple(xb.shape) != c8_shape(n, self.cin * s * s, self.oh, self.ow, (ka - 1) // 2):
            raise _ffi.VltfError("conv.input_prep_u8_s2d: xb shape %s" % (tuple(xb.shape),))
        for t, dt in ((crop_y, torch.int32), (crop_x, torch.int32), (mirror, torch.uint8)):
            if t is not None and (t.dtype != dt or t.numel() < n or not t.is_contiguous()):
                raise _ffi.VltfError("conv.input_prep_u8_s2d: crop / mirror must be contiguous int32 / uint8 tensors of n entries")
        _ffi.call("vl_input_prep_u8_s2d", self._d, _p(src), _p(xb), n, src.shape[1], src.shape[2], _p(crop_y), _p(crop_x), _p(mirror),
                  _p(mean_bgr), stream())

    def s2d_weights(self, src, dst, grad=False):
        """grad=False: w [k][k][cin][cout] -> [ka][ka][cin s^2][cout]; grad=True: the stride-1 layer's dw -> dw."""
        _f32(src, dst); _dense(src, dst)
        s, ka = self.stride, (self.kh - 1) // self.stride + 1
        big, small = (ka, ka, self.cin * s * s, self.cout), self.w_shape
        if (tuple(src.shape), tuple(dst.shape)) != ((big, small) if grad else (small, big)):
            raise _ffi.VltfError("conv.s2d_weights: shape mismatch %s -> %s" % (tuple(src.shape), tuple(dst.shape)))
        _ffi.call("vl_s2d_weights", self._d, _p(src), _p(dst), int(grad), stream())

    def c8_wgrad_ws_bytes(self, n):
        return int(_ffi.lib().vl_conv_c8_wgrad_ws_bytes(self._d, n))

    def c8_wgrad(self, xb, dyb, dw, ws):
        n = xb.shape[0]
        _f32(dw); _dense(dw, ws)
        self._c8_check(xb, n, self.cin, self.h, self.w, self.x_halo, "xb")
        self._c8_check(dyb, n, self.cout, self.oh, self.ow, self.dy_halo, "dyb")
        if tuple(dw.shape) != self.w_shape:
            raise _ffi.VltfError("conv.c8_wgrad: dw shape %s" % (tuple(dw.shape),))
        _ffi.call("vl_conv_c8_wgrad", self._d, _p(xb), _p(dyb), _p(dw), _p(ws), ws.numel() * ws.element_size(), n, stream())


def bias_grad_c8(dyb, db, ws, c, halo):
    """db[c] = sum over images and pixels of the packed gradient dyb (c8 with `halo`)."""
    _f32(db, ws); _dense(dyb, db, ws)
    n, cb, hp, wp, _ = dyb.shape
    if dyb.dtype != torch.bfloat16 or cb != (c + 7) // 8 or ws.numel() < 64 * 8 * cb:
        raise _ffi.VltfError("bias_grad_c8: dyb must be bf16 c8 of %d channels, ws >= 64 * 8 * blocks floats" % c)
    _ffi.call("vl_bias_grad_c8", _p(dyb), _p(db), _p(ws), n, c, hp - 2 * halo, wp - 2 * halo, halo, stream())


def kc8_shape(positions, channels):
    """Reduction-major packed operand of gemm_kc8: [ceil(channels / 8)][positions][8] bf16."""
    return ((channels + 7) // 8, positions, 8)


def pack_kc8(src, dst, positions, channels, pos_stride, ch_stride):
    """dst (kc8) = src[position * pos_stride + channel * ch_stride]: a row-major fp32 matrix (or its transpose, by the strides)."""
    _f32(src); _dense(dst)
    if dst.dtype != torch.bfloat16 or tuple(dst.shape) != kc8_shape(positions, channels):
        raise _ffi.VltfError("pack_kc8: dst must be bf16 %s, got %s %s" % (kc8_shape(positions, channels), dst.dtype, tuple(dst.shape)))
    if (positions - 1) * pos_stride + (channels - 1) * ch_stride >= src.numel():
        raise _ffi.VltfError("pack_kc8: strides reach past the source")
    _ffi.call("vl_pack_kc8", _p(src), _p(dst), positions, channels, pos_stride, ch_stride, stream())


def gemm_kc8_ws_bytes(m, n, k):
    return int(_ffi.lib().vl_gemm_kc8_ws_bytes(m, n, k))


def gemm_kc8(a, b, c, m, n, k, bias=None, relu=False, ws=None):
    """c[m][n] = sum_k a[k][m] b[k][n] (+bias) (relu): bf16 products of kc8 operands, fp32 accumulation and output."""
    _f32(c, bias); _dense(a, b, c, bias, ws)
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or tuple(a.shape) != kc8_shape(k, m) or tuple(b.shape) != kc8_shape(k, n) \
            or c.numel() < m * n:
        raise _ffi.VltfError("gemm_kc8: operand shapes a=%s b=%s for m=%d n=%d k=%d" % (tuple(a.shape), tuple(b.shape), m, n, k))
    _ffi.call("vl_gemm_kc8", _p(a), _p(b), _p(c), m, n, k, _p(bias), int(relu), _p(ws), 0 if ws is None else ws.numel() * ws.element_size(),
              stream())


def c8_shape(n, c, h, w, halo):
    return (n, (c + 7) // 8, h + 2 * halo, w + 2 * halo, 8)


def pack_c8(x, xb, x_halo, xb_halo):
    """fp32 NCHW (x_halo) -> bf16 c8 (xb_halo); xb zero-initialised once by the caller (only interiors are written)."""
    _f32(x); _dense(x, xb)
    n, c = x.shape[0], x.shape[1]
    h, w = x.shape[2] - 2 * x_halo, x.shape[3] - 2 * x_halo
    if xb.dtype != torch.bfloat16 or tuple(xb.shape) != c8_shape(n, c, h, w, xb_halo):
        raise _ffi.VltfError("pack_c8: xb must be bf16 %s, got %s %s" % (c8_shape(n, c, h, w, xb_halo), xb.dtype, tuple(xb.shape)))
    _ffi.call("vl_pack_c8", _p(x), _p(xb), n, c, h, w, x_halo, xb_halo, stream())


def bias_grad_nchw(dy, db, ws):
    _f32(dy, db, ws); _dense(dy, db, ws)
    n, c = dy.shape[0], dy.shape[1]
    hw = dy.numel() // (n * c)
    if ws.numel() < 64 * c:
        raise _ffi.VltfError("bias_grad_nchw: workspace needs 64*c floats")
    _ffi.call("vl_bias_grad_nchw", _p(dy), _p(db), _p(ws), n, c, hw, stream())


# ---- LRN / pool ----------------------------------------------------------------------------------
def lrn_fwd(x, y, radius=2, alpha=2e-5, beta=0.75, bias=1.0):
    _f32(x, y); _dense(x, y)
    n, c = x.shape[0], x.shape[1]
    _ffi.call("vl_lrn_fwd", _p(x), _p(y), n, c, x.numel() // (n * c), radius, alpha, beta, bias, stream())


def lrn_bwd(x, dy, dx, radius=2, alpha=2e-5, beta=0.75, bias=1.0, relu_fused=False, dx_halo=0):
    _f32(x, dy, dx); _dense(x, dy, dx)
    n, c, h, w = x.shape
    if tuple(dx.shape) != (n, c, h + 2 * dx_halo, w + 2 * dx_halo):
        raise _ffi.VltfError("lrn_bwd: dx shape %s does not match x %s with halo %d" % (tuple(dx.shape), tuple(x.shape), dx_halo))
    _ffi.call("vl_lrn_bwd", _p(x), _p(dy), _p(dx), n, c, h * w, radius, alpha, beta, bias, int(relu_fused), w, dx_halo, stream())


def pool_out(h, k=3, s=2):
    return (h - k) // s + 1


def _pool_layout(c, oh, ow, hwc, halo):
    """(element offset of the interior origin, strides n/c/h/w) of a pool output: NCHW with halo, or the
    (h, w, c)-flat order fc6 expects (alexnet.py:228)."""
    if hwc:
        return 0, (oh * ow * c, 1, ow * c, c)
    wp = ow + 2 * halo
    pp = (oh + 2 * halo) * wp
    return halo * wp + halo, (c * pp, pp, wp, 1)


def maxpool_fwd(x, y, argmax, k=3, s=2, hwc=False, y_halo=0):
    _f32(x, y); _dense(x, y, argmax)
    n, c, h, w = x.shape
    o, st = _pool_layout(c, pool_out(h, k, s), pool_out(w, k, s), hwc, y_halo)
    _ffi.call("vl_maxpool_fwd", _p(x), _p(y) + 4 * o, None if argmax is None else _p(argmax) + o, n, c, h, w, k, s, st[0], st[1],
              st[2], st[3], stream())


def maxpool_bwd(dy, argmax, dx, relu_mask=None, k=3, s=2, hwc=False, dy_halo=0, dx_halo=0):
    """dy / argmax have the pool OUTPUT layout (halo dy_halo); dx is NCHW with dx_halo."""
    _f32(dy, dx, relu_mask); _dense(dy, dx, argmax, relu_mask)
    n, c = dx.shape[0], dx.shape[1]
    h, w = dx.shape[2] - 2 * dx_halo, dx.shape[3] - 2 * dx_halo
    o, st = _pool_layout(c, pool_out(h, k, s), pool_out(w, k, s), hwc, dy_halo)
    _ffi.call("vl_maxpool_bwd", _p(dy) + 4 * o, _p(argmax) + o, _p(dx), _p(relu_mask), n, c, h, w, k, s, st[0], st[1], st[2], st[3],
              dx_halo, stream())


def lrn_pool_fwd(x, p, argmax, p_halo=0, radius=2, alpha=2e-5, beta=0.75, bias=1.0):
    """Fused LRN + maxpool(3,2) forward; p / argmax have the pool-output layout (p_halo); the LRN output is not stored."""
    _f32(x, p); _dense(x, p, argmax)
    n, c, h, w = x.shape
    want = (n, c, pool_out(h) + 2 * p_halo, pool_out(w) + 2 * p_halo)
    if tuple(p.shape) != want or tuple(argmax.shape) != want:
        raise _ffi.VltfError("lrn_pool_fwd: shape mismatch x=%s p=%s argmax=%s" % (tuple(x.shape), tuple(p.shape), tuple(argmax.shape)))
    _ffi.call("vl_lrn_pool_fwd", _p(x), _p(p), _p(argmax), n, c, h, w, p_halo, radius, alpha, beta, bias, stream())


def _x_or_packed(x, c):
    """(packed?, n, h, w) of a pool / LRN input: fp32 NCHW, or bf16 c8 without a halo ([n][c/8][h][w][8]) of c channels."""
    if x.dtype == torch.bfloat16:
        if x.dim() != 5 or x.shape[4] != 8 or c is None or x.shape[1] != (c + 7) // 8 or not x.is_contiguous():
            raise _ffi.VltfError("packed pool input must be a contiguous bf16 [n][c/8][h][w][8] tensor and needs channels=")
        return 1, x.shape[0], x.shape[2], x.shape[3]
    _f32(x)
    return 0, x.shape[0], x.shape[2], x.shape[3]


def lrn_pool_fwd_c8(x, pb, argmax, p_halo=0, radius=2, alpha=2e-5, beta=0.75, bias=1.0, channels=None):
    """lrn_pool_fwd with the pooled output written as packed bf16 (c8 layout, p_halo): the next conv's operand on the bf16 path.
    x: fp32 NCHW, or the packed conv output (bf16 c8, no halo; pass channels=)."""
    _dense(x, pb, argmax)
    c = channels if channels is not None else x.shape[1]
    packed, n, h, w = _x_or_packed(x, c)
    oh, ow = pool_out(h), pool_out(w)
    if pb.dtype != torch.bfloat16 or tuple(pb.shape) != c8_shape(n, c, oh, ow, p_halo) or \
            tuple(argmax.shape) != (n, c, oh + 2 * p_halo, ow + 2 * p_halo):
        raise _ffi.VltfError("lrn_pool_fwd_c8: shape mismatch x=%s pb=%s argmax=%s" % (tuple(x.shape), tuple(pb.shape), tuple(argmax.shape)))
    _ffi.call("vl_lrn_pool_fwd_c8", _p(x), packed, _p(pb), _p(argmax), n, c, h, w, p_halo, radius, alpha, beta, bias, stream())


def pool_lrn_bwd(x, dp, argmax, dx, p_halo=0, dx_halo=0, radius=2, alpha=2e-5, beta=0.75, bias=1.0, relu_fused=True):
    """Fused maxpool(3,2) backward + LRN backward (+ReluGrad); dp/argmax have the pool-output layout (p_halo)."""
    _f32(x, dp, dx); _dense(x, dp, argmax, dx)
    n, c, h, w = x.shape
    oh, ow = pool_out(h), pool_out(w)
    if tuple(dp.shape) != (n, c, oh + 2 * p_halo, ow + 2 * p_halo) or tuple(argmax.shape) != tuple(dp.shape) or \
            tuple(dx.shape) != (n, c, h + 2 * dx_halo, w + 2 * dx_halo):
        raise _ffi.VltfError("pool_lrn_bwd: shape mismatch x=%s dp=%s dx=%s" % (tuple(x.shape), tuple(dp.shape), tuple(dx.shape)))
    _ffi.call("vl_pool_lrn_bwd", _p(x), _p(dp), _p(argmax), _p(dx), n, c, h, w, p_halo, radius, alpha, beta, bias, int(relu_fused),
              dx_halo, stream())


def pool_lrn_bwd_test_ranges(ranges=0):
    """Test hook: force the channel-range count of pool_lrn_bwd / pool_lrn_bwd_c8 launches (1..4); 0 = the launcher's own choice."""
    _ffi.call("vl_pool_lrn_bwd_test_ranges", int(ranges))


def pool_lrn_bwd_c8(x, dp, argmax, dxb, p_halo=0, dxb_halo=0, radius=2, alpha=2e-5, beta=0.75, bias=1.0, relu_fused=True):
    """pool_lrn_bwd with the gradient written as packed bf16 (c8 layout, dxb_halo): the bf16 conv path's operand.
    x: fp32 NCHW, or the packed conv output (bf16 c8, no halo)."""
    _f32(dp); _dense(x, dp, argmax, dxb)
    c = dp.shape[1]
    packed, n, h, w = _x_or_packed(x, c)
    oh, ow = pool_out(h), pool_out(w)
    if tuple(dp.shape) != (n, c, oh + 2 * p_halo, ow + 2 * p_halo) or tuple(argmax.shape) != tuple(dp.shape) or \
            dxb.dtype != torch.bfloat16 or tuple(dxb.shape) != c8_shape(n, c, h, w, dxb_halo):
        raise _ffi.VltfError("pool_lrn_bwd_c8: shape mismatch x=%s dp=%s dxb=%s" % (tuple(x.shape), tuple(dp.shape), tuple(dxb.shape)))
    _ffi.call("vl_pool_lrn_bwd_c8", _p(x), packed, _p(dp), _p(argmax), _p(dxb), n, c, h, w, p_halo, radius, alpha, beta, bias, int(relu_fused),
              dxb_halo, stream())


# ---- dense ---------------------------------------------------------------------------------------
def gemm(a, b, c, m, n, k, transa=False, transb=False, lda=None, ldb=None, ldc=None, bias=None, relu=False,
         relu_mask=None, ws=None):
    """c[m,n] = op(a) @ op(b) (+bias) (relu) (mask).  a/b/c may be views: only data_ptr + ld are used."""
    _f32(a, b, c, bias, relu_mask)
    lda = lda if lda is not None else (m if transa else k)
    ldb = ldb if ldb is not None else (k if transb else n)
    ldc = ldc if ldc is not None else n
    nbytes = 0 if ws is None else ws.numel() * ws.element_size()
    _ffi.call("vl_gemm", int(transa), int(transb), m, n, k, _p(a), lda, _p(b), ldb, _p(c), ldc, _p(bias), int(relu),
              _p(relu_mask), _p(ws), nbytes, stream())


def gemm_split_ws_bytes(m, n, k):
    """Workspace with which gemm() runs in the split-bf16 arithmetic of set_conv_math (vl_gemm_split_ws_bytes)."""
    return int(_ffi.lib().vl_gemm_split_ws_bytes(int(m), int(n), int(k)))


def colsum(a, out, ws, m, n, lda=None):
    _f32(a, out, ws)
    if ws.numel() < 64 * n:
        raise _ffi.VltfError("colsum: workspace needs 64*n floats")
    _ffi.call("vl_colsum", _p(a), lda if lda is not None else n, _p(out), _p(ws), m, n, stream())


# ---- LSTM ----------------------------------------------------------------------------------------
def lstm_step_fwd(gx, gh, act, cseq, hseq, hprev, batch, T, t, H, forget_bias=1.0):
    _f32(gx, gh, act, cseq, hseq, hprev)
    _ffi.call("vl_lstm_step_fwd", _p(gx), _p(gh), _p(act), _p(cseq), _p(hseq), _p(hprev), batch, T, t, H, forget_bias,
              stream())


def lstm_step_bwd(dout, dh_next, act, cseq, dc, dz, batch, T, t, H):
    _f32(dout, dh_next, act, cseq, dc, dz)
    _ffi.call("vl_lstm_step_bwd", _p(dout), _p(dh_next), _p(act), _p(cseq), _p(dc), _p(dz), batch, T, t, H, stream())


def lstm_seq_ws(batch, T, H, device):
    """Scratch for lstm_seq_fwd / lstm_seq_bwd (vl_lstm_seq_ws_bytes): the exchange words of the cluster form."""
    n = int(_ffi.lib().vl_lstm_seq_ws_bytes(int(batch), int(T), int(H)))
    return torch.zeros((n + 3) // 4, dtype=torch.float32, device=device)


def lstm_seq_fwd(gx, kh, act, cseq, hseq, hprev, batch, T, H, forget_bias=1.0, ws=None, h0=None, c0=None):
    """All T steps in one launch; kh = kernel[D:] view ([H, 4H]); h0 / c0: initial state [batch, H] (None = zeros)."""
    _f32(gx, kh, act, cseq, hseq, hprev, h0, c0, ws)
    if ws is None:
        raise _ffi.VltfError("lstm_seq_fwd: a workspace from lstm_seq_ws() is required")
    _ffi.call("vl_lstm_seq_fwd", _p(gx), _p(kh), _p(h0), _p(c0), _p(act), _p(cseq), _p(hseq), _p(hprev), batch, T, H, forget_bias,
              _p(ws), ws.numel() * 4, stream())


def lstm_seq_bwd(dout, kh, act, cseq, dz, batch, T, H, ws=None, c0=None, dh0=None, dc0=None):
    """BPTT in one launch; kh = kernel[D:] view (not transposed); dh0 / dc0: optional outputs [batch, H]."""
    _f32(dout, kh, act, cseq, dz, c0, dh0, dc0, ws)
    if ws is None:
        raise _ffi.VltfError("lstm_seq_bwd: a workspace from lstm_seq_ws() is required")
    _ffi.call("vl_lstm_seq_bwd", _p(dout), _p(kh), _p(act), _p(cseq), _p(c0), _p(dz), _p(dh0), _p(dc0), batch, T, H, _p(ws),
              ws.numel() * 4, stream())


def lstm_seq_timed_out(ws):
    """True if a workgroup of any cluster-form launch on `ws` since the last call gave up waiting for its peers; the flag is
    sticky across launches and reset by this read (synchronises)."""
    v = C.c_int(0)
    _ffi.call("vl_lstm_seq_status", _p(ws), C.byref(v))
    return bool(v.value)


def lstm_seq_check(*workspaces):
    """Raises VltfError when a cluster-form LSTM launch on one of the workspaces timed out (results of that step are invalid)."""
    bad = [i for i, ws in enumerate(workspaces) if ws is not None and lstm_seq_timed_out(ws)]
    if bad:
        raise _ffi.VltfError("LSTM cluster kernel timed out waiting for its peer workgroups (workspace %s): the recurrence needs "
                             "every workgroup resident at once and another kernel was holding compute units; this step's results "
                             "are invalid" % bad)


def lstm_seq_test_hooks(spin_limit=0, mute_workgroup=-1):
    """Test hooks (vl_lstm_seq_test_hooks): shorter spin limit / one workgroup that never publishes.  Defaults restore production."""
    _ffi.call("vl_lstm_seq_test_hooks", int(spin_limit), int(mute_workgroup))


def transpose(src, dst, rows, cols, ld=None):
    _f32(src, dst)
    _ffi.call("vl_transpose", _p(src), ld if ld is not None else cols, _p(dst), rows, cols, stream())


FUSION_CODE = {"avg": 0, "last": 1}


def temporal_fusion_fwd(x, y, batch, T, H, method):
    _f32(x, y)
    _ffi.call("vl_temporal_fusion_fwd", _p(x), _p(y), batch, T, H, FUSION_CODE[method], stream())


def temporal_fusion_bwd(dy, dx, batch, T, H, method):
    _f32(dy, dx)
    _ffi.call("vl_temporal_fusion_bwd", _p(dy), _p(dx), batch, T, H, FUSION_CODE[method], stream())


def dropout_fwd(x, y, mask, keep, seed):
    _f32(x, y)
    _ffi.call("vl_dropout_fwd", _p(x), _p(y), _p(mask), x.numel(), keep, seed, stream())


def dropout_bwd(dy, mask, dx, keep):
    _f32(dy, dx)
    _ffi.call("vl_dropout_bwd", _p(dy), _p(mask), _p(dx), dy.numel(), keep, stream())


# ---- loss / optimizer ----------------------------------------------------------------------------
def softmax_xent(logits, labels, dlogits, stats, grad_scale, rows=None):
    """rows: float32 workspace of >= 2*batch elements (per-row losses and hits); without it one workgroup walks the batch."""
    _f32(logits, dlogits, stats); _dense(logits, labels, dlogits)
    if labels.dtype != torch.int32:
        raise _ffi.VltfError("softmax_xent: labels must be int32 one-hot")
    b, c = logits.shape
    if rows is not None:
        _f32(rows)
        if rows.numel() < 2 * b:
            raise _ffi.VltfError("softmax_xent: rows workspace needs 2*batch floats")
    _ffi.call("vl_softmax_xent", _p(logits), _p(labels), _p(dlogits), _p(stats), _p(rows), b, c, grad_scale, stream())


def sumsq(g, out, ws, accumulate=False):
    _f32(g, out, ws)
    if ws.numel() < 1024:
        raise _ffi.VltfError("sumsq: workspace needs 1024 floats")
    _ffi.call("vl_sumsq", _p(g), g.numel(), _p(out), _p(ws), int(accumulate), stream())


def _skip_word(skip):
    if skip is not None and not (skip.is_cuda and skip.dtype == torch.int32 and skip.numel() >= 1):
        raise _ffi.VltfError("skip must be a CUDA/HIP int32 tensor (one word)")
    return _p(skip)


def sgd_apply(w, g, lr, clip_norm=0.0, sumsq_t=None, gscale=1.0, skip=None):
    """skip: optional device word (int32); non-zero at execution time = the update is dropped (step_guard)."""
    _f32(w, g, sumsq_t)
    _ffi.call("vl_sgd_apply", _p(w), _p(g), w.numel(), lr, clip_norm, _p(sumsq_t), gscale, _skip_word(skip), stream())


def adam_apply(w, g, m, v, lr, step, clip_norm=0.0, sumsq_t=None, gscale=1.0, skip=None):
    _f32(w, g, m, v, sumsq_t)
    _ffi.call("vl_adam_apply", _p(w), _p(g), _p(m), _p(v), w.numel(), lr, clip_norm, _p(sumsq_t), gscale, step, _skip_word(skip), stream())


def step_guard(skip, *lstm_workspaces):
    """skip[0] = 1 if a cluster-form LSTM launch on any of the workspaces has timed out since its status was last read (the word is
    sticky, lstm_seq_check reads and resets it), else 0 -- on the stream, no host round trip.  Hand `skip` to sgd_apply / adam_apply:
    a step whose recurrence gave up (its results are invalid by the kernel's own contract) must not reach the weights."""
    _skip_word(skip)
    first = True                                  # no workspace: the word keeps the 0 it was created with (nothing ever sets it)
    for ws in lstm_workspaces:
        if ws is not None:
            _ffi.call("vl_status_or", _p(skip), _p(ws), int(first), stream())
            first = False
    return skip


def fill(t, value):
    _f32(t)
    _ffi.call("vl_fill", _p(t), t.numel(), value, stream())


CONV_MATH = {"f32": 0, "bf16": 1, "bf16x3": 3, "bf16x6": 6}


def set_conv_math(name):
    """Arithmetic of the conv contractions (vl_set_conv_math): "f32" (default), "bf16x3" (split bf16 products) or "bf16" (plain
    bf16 products: reduced precision), both opt-in."""
    if name not in CONV_MATH:
        raise _ffi.VltfError("conv math must be one of %s, not %r" % (sorted(CONV_MATH), name))
    _ffi.call("vl_set_conv_math", CONV_MATH[name])


def conv_math():
    return {v: k for k, v in CONV_MATH.items()}[int(_ffi.lib().vl_conv_math())]


def relu_grad(d, y, count=None):
    """d = y > 0 ? d : 0 in place over the first `count` elements."""
    _f32(d, y); _dense(d, y)
    _ffi.call("vl_relu_grad", _p(d), _p(y), d.numel() if count is None else int(count), stream())


# ---- tensor-list plumbing (tf_util.py:99-192) ----------------------------------------------------------------------------------
def copy2d(src, dst, rows, cols, src_ld=None, dst_ld=None):
    """dst[r, :cols] = src[r, :cols]; src / dst may be views (data_ptr + row strides are used); src_ld = 0 repeats one row."""
    _f32(src, dst)
    _ffi.call("vl_copy2d", _p(src), cols if src_ld is None else int(src_ld), _p(dst), cols if dst_ld is None else int(dst_ld),
              int(rows), int(cols), stream())


ELTWISE_OP = {"add": 0, "avg": 1, "maximum": 2}


def eltwise2(a, b, out, op, count=None):
    _f32(a, b, out)
    _ffi.call("vl_eltwise2", _p(a), _p(b), _p(out), a.numel() if count is None else int(count), ELTWISE_OP[op], stream())


def max2_grad(a, b, d, da, db, count=None):
    _f32(a, b, d, da, db)
    _ffi.call("vl_max2_grad", _p(a), _p(b), _p(d), _p(da), _p(db), a.numel() if count is None else int(count), stream())


FUSE_OP = {"avg": 0, "maximum": 1}


def _ptr_list(ts):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def fuse_n(ins, out, method, count=None):
    """out = mean | maximum over the list `ins` of equally shaped tensors (apply_tensor_list_fusion avg | maximum, tf_util.py:142-145)."""
    _f32(out, *ins)
    if not 1 <= len(ins) <= 8:
        raise _ffi.VltfError("fuse_n: 1..8 inputs, got %d" % len(ins))
    _ffi.call("vl_fuse_n", _ptr_list(ins), len(ins), _p(out), ins[0].numel() if count is None else int(count), FUSE_OP[method], stream())


def fuse_n_grad(ins, d, dins, method, count=None):
    """dins[i] (None = not wanted) = gradient of fuse_n w.r.t. input i: d / n, or d shared evenly among the inputs equal to the maximum."""
    _f32(d, *[t for t in list(ins) + list(dins) if t is not None])
    if len(ins) != len(dins) or not 1 <= len(ins) <= 8:
        raise _ffi.VltfError("fuse_n_grad: 1..8 inputs and as many gradient slots")
    _ffi.call("vl_fuse_n_grad", _ptr_list(ins), len(ins), _p(d), _ptr_list(dins), d.numel() if count is None else int(count),
              FUSE_OP[method], stream())


# ---- imresize (dataset_.py:481-495) ---------------------------------------------------------------------------------------------
class Resize:
    """scipy.misc.imresize(image, (oh, ow, 3)) = PIL bilinear on uint8 [n, h, w, 3] device images (vl_resize_*), bit-exact."""

    def __init__(self, h, w, oh, ow):
        self.h, self.w, self.oh, self.ow = int(h), int(w), int(oh), int(ow)
        self._d = C.c_void_p()
        _ffi.check(_ffi.lib().vl_resize_create(C.byref(self._d), self.h, self.w, self.oh, self.ow, 3), "vl_resize_create")
        self._tmp = None

    def __del__(self):
        try:
            if self._d:
                _ffi.lib().vl_resize_destroy(self._d)
                self._d = None
        except Exception:
            pass

    def __call__(self, src, dst=None):
        if src.dtype != torch.uint8 or not src.is_cuda or not src.is_contiguous() or tuple(src.shape[1:]) != (self.h, self.w, 3):
            raise _ffi.VltfError("resize: expected contiguous device uint8 [n, %d, %d, 3], got %s" % (self.h, self.w, tuple(src.shape)))
        n = src.shape[0]
        if dst is None:
            dst = torch.empty((n, self.oh, self.ow, 3), dtype=torch.uint8, device=src.device)
        elif dst.dtype != torch.uint8 or tuple(dst.shape) != (n, self.oh, self.ow, 3) or not dst.is_contiguous():
            raise _ffi.VltfError("resize: destination must be contiguous uint8 %s" % ((n, self.oh, self.ow, 3),))
        need = int(_ffi.lib().vl_resize_tmp_bytes(self._d, n))
        if need and (self._tmp is None or self._tmp.numel() < need):
            self._tmp = torch.empty(need, dtype=torch.uint8, device=src.device)
        _ffi.call("vl_resize_u8", self._d, _p(src), _p(self._tmp) if need else None, _p(dst), n, stream())
        return dst

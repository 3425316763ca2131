// scipy.misc.imresize of the reference's host image processing (dataset_.py:481-495: `raw_resize` to the raw shape, `resize` to
// the network input size) on the device, bit-exact on uint8.
//
// imresize(arr, shape) = PIL Image.resize((w, h), BILINEAR) on the uint8 image: Pillow's separable resample with a triangle filter
// whose support widens with the down-scale factor (anti-aliasing), evaluated in 22-bit fixed point with a uint8 intermediate:
//   pass 1 (horizontal): tmp[y][xo] = clip8((2^21 + sum_k src[y][xmin(xo) + k] * kh[xo][k]) >> 22)
//   pass 2 (vertical)  : dst[yo][x] = clip8((2^21 + sum_k tmp[ymin(yo) + k][x] * kv[yo][k]) >> 22)
// (libImaging/Resample.c; an axis whose size is unchanged is skipped).  The coefficient tables are computed on the host in double
// precision exactly as precompute_coeffs / normalize_coeffs_8bpc do and uploaded once per (in, out) size pair (vl_resize_create).
// HBM-bound byte work: one thread per output pixel (3 channels), ~5 taps per pass for 240x320 -> 227x227.
#include <math.h>

#include <vector>

#include "common.h"

static constexpr int RS_BITS = 32 - 8 - 2;

struct AxisTab {
    int* bounds = nullptr;   // [out][2]: first input index, tap count
    int* kk = nullptr;       // [out][ksize] coefficients in 2^-22 units
    int ksize = 0;
};

struct vl_resize_desc {
    int h, w, oh, ow, c;
    AxisTab hor, ver;
};

static int build_axis(AxisTab& t, int in_size, int out_size) {
    // precompute_coeffs(inSize, in0 = 0, in1 = inSize, outSize, BILINEAR) + normalize_coeffs_8bpc
    const double scale = (double)((float)in_size - 0.0f) / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)ceil(support) * 2 + 1;
    std::vector<int> bounds((size_t)out_size * 2), kk((size_t)out_size * ksize, 0);
    std::vector<double> k((size_t)ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = 0.0 + (xx + 0.5) * scale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        for (int x = 0; x < xmax; ++x) {
            double v = (x + xmin - center + 0.5) * ss;
            if (v < 0.0) v = -v;
            const double wgt = v < 1.0 ? 1.0 - v : 0.0;
            k[x] = wgt;
            ww += wgt;
        }
        for (int x = 0; x < xmax; ++x) {
            if (ww != 0.0) k[x] /= ww;
            kk[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << RS_BITS)) : (int)(0.5 + k[x] * (1 << RS_BITS));
        }
        bounds[(size_t)xx * 2] = xmin;
        bounds[(size_t)xx * 2 + 1] = xmax;
    }
    t.ksize = ksize;
    VL_HIP(hipMalloc(&t.bounds, bounds.size() * sizeof(int)));
    VL_HIP(hipMalloc(&t.kk, kk.size() * sizeof(int)));
    VL_HIP(hipMemcpy(t.bounds, bounds.data(), bounds.size() * sizeof(int), hipMemcpyHostToDevice));
    VL_HIP(hipMemcpy(t.kk, kk.data(), kk.size() * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int vl_resize_create(vl_resize_desc** out, int h, int w, int oh, int ow, int channels) {
    VL_CHECK(out && h > 0 && w > 0 && oh > 0 && ow > 0 && channels == 3, "vl_resize_create: bad shape (3-channel uint8 images only)");
    vl_resize_desc* d = new vl_resize_desc();
    d->h = h; d->w = w; d->oh = oh; d->ow = ow; d->c = channels;
    if ((ow != w && build_axis(d->hor, w, ow)) || (oh != h && build_axis(d->ver, h, oh))) {
        delete d;
        return 2;
    }
    *out = d;
    return 0;
}

extern "C" void vl_resize_destroy(vl_resize_desc* d) {
    if (!d) return;
    for (AxisTab* t : {&d->hor, &d->ver}) {
        if (t->bounds) (void)hipFree(t->bounds);
        if (t->kk) (void)hipFree(t->kk);
    }
    delete d;
}

/* bytes of the uint8 intermediate ([n][h][ow][3]) vl_resize_u8 needs for n images; 0 when only one axis changes */
extern "C" size_t vl_resize_tmp_bytes(const vl_resize_desc* d, int n) {
    if (!d || n <= 0 || d->ow == d->w || d->oh == d->h) return 0;
    return (size_t)n * d->h * d->ow * 3;
}

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= RS_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// one pass along the axis with `len_in` -> `len_out` entries; `inner` = bytes between consecutive entries of that axis
// (3 for the horizontal pass, row bytes for the vertical one); `lines` = independent lines per image; line_in / line_out = their byte strides
__global__ void resample_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, const int* __restrict__ bounds,
                                   const int* __restrict__ kk, int ksize, int n, int lines, int len_out, int64_t img_in, int64_t img_out,
                                   int64_t line_in, int64_t line_out, int64_t step_in, int64_t step_out) {
    const int64_t total = (int64_t)n * lines * len_out;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        // horizontal: consecutive threads = consecutive output pixels of a row; vertical: consecutive pixels of an output row too
        int64_t r = e;
        int xo, line;
        if (step_in == 3) {                       // horizontal pass: e = (img, line = y, xo)
            xo = (int)(r % len_out); r /= len_out;
            line = (int)(r % lines); r /= lines;
        } else {                                  // vertical pass: e = (img, xo = yo, line = x) so that x is fastest
            line = (int)(r % lines); r /= lines;
            xo = (int)(r % len_out); r /= len_out;
        }
        const int img = (int)r;
        const int x0 = bounds[2 * xo], cnt = bounds[2 * xo + 1];
        const int* k = kk + (int64_t)xo * ksize;
        const uint8_t* p = src + img * img_in + line * line_in + x0 * step_in;
        int s0 = 1 << (RS_BITS - 1), s1 = s0, s2 = s0;
        for (int t = 0; t < cnt; ++t) {
            const int kv = k[t];
            s0 += p[0] * kv; s1 += p[1] * kv; s2 += p[2] * kv;
            p += step_in;
        }
        uint8_t* q = dst + img * img_out + line * line_out + xo * step_out;
        q[0] = clip8(s0); q[1] = clip8(s1); q[2] = clip8(s2);
    }
}

extern "C" int vl_resize_u8(const vl_resize_desc* d, const uint8_t* src, uint8_t* tmp, uint8_t* dst, int n, vl_stream_t stream) {
    VL_CHECK(d && src && dst && n > 0, "vl_resize_u8: null argument");
    hipStream_t s = (hipStream_t)stream;
    const bool hor = d->ow != d->w, ver = d->oh != d->h;
    const int64_t row_in = (int64_t)d->w * 3, row_mid = (int64_t)d->ow * 3;
    if (!hor && !ver) {                                        // PIL returns a copy of an image already at the target size
        VL_HIP(hipMemcpyAsync(dst, src, (size_t)n * d->h * row_in, hipMemcpyDeviceToDevice, s));
        return 0;
    }
    VL_CHECK(!(hor && ver) || tmp, "vl_resize_u8: the two-pass resize needs the intermediate buffer (vl_resize_tmp_bytes)");
    const uint8_t* cur = src;
    if (hor) {
        uint8_t* out = ver ? tmp : dst;
        const int64_t total = (int64_t)n * d->h * d->ow;
        const int grid = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
        hipLaunchKernelGGL(resample_u8_kernel, dim3(grid), dim3(256), 0, s, cur, out, d->hor.bounds, d->hor.kk, d->hor.ksize, n, d->h, d->ow,
                           d->h * row_in, d->h * row_mid, row_in, row_mid, (int64_t)3, (int64_t)3);
        VL_LAUNCH_CHECK();
        cur = out;
    }
    if (ver) {
        const int64_t total = (int64_t)n * d->oh * d->ow;
        const int grid = (int)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
        hipLaunchKernelGGL(resample_u8_kernel, dim3(grid), dim3(256), 0, s, cur, dst, d->ver.bounds, d->ver.kk, d->ver.ksize, n, d->ow, d->oh,
                           d->h * row_mid, d->oh * row_mid, (int64_t)3, (int64_t)3, row_mid, row_mid);
        VL_LAUNCH_CHECK();
    }
    return 0;
}

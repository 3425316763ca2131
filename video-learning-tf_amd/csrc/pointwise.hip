// HBM-bound kernels of the LRCN path: input prep, LRN, max-pool, bias/column sums, LSTM gate
// pointwise, temporal fusion, dropout, softmax cross-entropy, global-norm + SGD/Adam.
// All activations NCHW fp32; lanes always walk the contiguous (w / hw / column) dimension.
#include <stdlib.h>

#include <stdio.h>
#include <stdlib.h>

#include "common.h"

// A/B switches for measurements, read once: VL_POOL_LRN_CHUNKED=1 runs the fused pool+LRN backward on the older chunked
// kernel, VL_POOL_LRN_CHK16=1 forces the 16-channel chunk of the channel-stream kernel.
static const bool kPoolLrnChunked = vl_exp_env("VL_POOL_LRN_CHUNKED") != nullptr;
static const bool kPoolLrnChk16 = vl_exp_env("VL_POOL_LRN_CHK16") != nullptr;
static const bool kMaxpoolGeneric = vl_exp_env("VL_MAXPOOL_GENERIC") != nullptr;   // A/B: pool5 backward, element-per-thread kernel

static inline int grid_for(int64_t work, int threads, int cap) {
    int64_t b = (work + threads - 1) / threads;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- input prep (dataset_.py:481-501) ---------------------------------------------------------
// Threads walk the DESTINATION plane by plane: thread -> (phase plane ph, row y, column q of the plane), an image per blockIdx.y.  The
// interior rows of a (channel, phase) plane are one contiguous run when every column of a row is written -- the halo / padding columns
// too, as the 0.0 they hold by contract -- so a wave's stores are whole 256-byte runs in each of the three channel planes.  (Round 1 -
// 3: (row, phase, column) order with the halo columns skipped: 236-byte row pieces, every piece's end sectors written partially, and
// three 64-bit divisions per element; 0.25 ms per 1024 frames = 3.1 TB/s.)  The source bytes of a wave are `phase` pixels apart (12 B
// at phase 4), all within a few cache lines.
__global__ __launch_bounds__(256) void input_prep_u8_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int rh, int rw, int oh,
                                                            int ow, const int32_t* __restrict__ cy, const int32_t* __restrict__ cx,
                                                            const uint8_t* __restrict__ mir, const float* __restrict__ mean, int halo, int phase,
                                                            int wp, FastDiv d_plane, FastDiv d_wp) {
    // phase > 1: column-phase-split destination [c][phase][oh + 2 halo][wp = ceil((ow + 2 halo) / phase)] (vl_conv_set_x_phase_split)
    const int img = blockIdx.y;
    const uint32_t plane_in = (uint32_t)oh * wp, per_img = plane_in * phase;      // interior rows of one plane; of all phase planes
    const int64_t pp = (int64_t)(oh + 2 * halo) * wp;
    const float m0 = mean ? mean[0] : 0.f, m1 = mean ? mean[1] : 0.f, m2 = mean ? mean[2] : 0.f;
    const int oy = cy ? cy[img] : 0, ox = cx ? cx[img] : 0;
    const bool flip = mir && mir[img];
    const uint8_t* simg = src + (int64_t)img * rh * rw * 3;
    float* dimg = dst + (int64_t)img * 3 * phase * pp + (int64_t)halo * wp;
    for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < per_img; e += gridDim.x * 256u) {
        const uint32_t ph = fd_div(e, d_plane), r = e - ph * plane_in;
        const uint32_t y = fd_div(r, d_wp), q = r - y * wp;
        const int x = (int)(q * phase + ph) - halo;                   // physical column q * phase + ph of the haloed row
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;                           // halo / padding columns: the zeros they hold
        if (x >= 0 && x < ow) {
            const uint8_t* sp = simg + ((int64_t)(y + oy) * rw + (flip ? ow - 1 - x : x) + ox) * 3;
            v0 = (float)sp[0] - m0;
            v1 = (float)sp[1] - m1;
            v2 = (float)sp[2] - m2;
        }
        float* d = dimg + (int64_t)ph * pp + r;                        // r = y * wp + q: the plane's interior rows are contiguous
        d[0] = v0;
        d[(int64_t)phase * pp] = v1;
        d[2 * (int64_t)phase * pp] = v2;
    }
}

extern "C" int vl_input_prep_u8(const uint8_t* src, float* dst, int n, int raw_h, int raw_w, int out_h, int out_w,
                                const int32_t* crop_y, const int32_t* crop_x, const uint8_t* mirror, const float* mean_bgr,
                                int dst_halo, int dst_phase, vl_stream_t stream) {
    VL_CHECK(src && dst && dst_halo >= 0 && dst_phase >= 1, "vl_input_prep_u8: bad argument");
    VL_CHECK(n > 0 && out_h > 0 && out_w > 0 && out_h <= raw_h && out_w <= raw_w, "vl_input_prep_u8: bad shape");
    VL_CHECK(n <= 65535, "vl_input_prep_u8: batch %d exceeds the grid limit", n);
    const int wp = (out_w + 2 * dst_halo + dst_phase - 1) / dst_phase;
    const int64_t per_img = (int64_t)out_h * dst_phase * wp;
    VL_CHECK(per_img < (1ll << 31), "vl_input_prep_u8: image too large");
    int bx = (int)((per_img + 1023) / 1024);                          // ~4 elements per thread
    if ((int64_t)bx * n < 8ll * vl_device_cus()) bx = (int)((per_img + 255) / 256);   // few frames: one element per thread
    hipLaunchKernelGGL(input_prep_u8_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, src, dst, raw_h, raw_w, out_h, out_w,
                       crop_y, crop_x, mirror, mean_bgr, dst_halo, dst_phase, wp, make_fastdiv((uint32_t)out_h * wp), make_fastdiv((uint32_t)wp));
    VL_LAUNCH_CHECK();
    return 0;
}

// generic 3-level strided copy: dst[(a*nb + b)*nc + c] = src[a*sa + b*sb + c*sc]
__global__ void permute3_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t na, int nb, int nc, int64_t sa,
                                int64_t sb, int64_t sc) {
    const int64_t total = na * nb * nc;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e % nc);
        const int b = (int)((e / nc) % nb);
        const int64_t a = e / ((int64_t)nc * nb);
        dst[e] = src[a * sa + b * sb + c * sc];
    }
}

__global__ void nhwc_to_nchw_halo_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int h, int w, int c,
                                         int halo, int phase) {
    const int64_t total = (int64_t)n * c * h * w;
    const int wp = (w + 2 * halo + phase - 1) / phase;
    const int64_t pp = (int64_t)(h + 2 * halo) * wp;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % w);
        const int y = (int)((e / w) % h);
        const int ch = (int)((e / ((int64_t)w * h)) % c);
        const int64_t img = e / ((int64_t)w * h * c);
        const int xc = x + halo;
        dst[((img * c + ch) * phase + xc % phase) * pp + (int64_t)(y + halo) * wp + xc / phase] = src[((img * h + y) * w + x) * c + ch];
    }
}

extern "C" int vl_nhwc_to_nchw(const float* src, float* dst, int n, int h, int w, int c, int dst_halo, int dst_phase,
                               vl_stream_t stream) {
    VL_CHECK(src && dst && n > 0 && h > 0 && w > 0 && c > 0 && dst_halo >= 0 && dst_phase >= 1, "vl_nhwc_to_nchw: bad argument");
    hipLaunchKernelGGL(nhwc_to_nchw_halo_kernel, dim3(grid_for((int64_t)n * h * w * c, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                       src, dst, n, h, w, c, dst_halo, dst_phase);
    VL_LAUNCH_CHECK();
    return 0;
}

extern "C" int vl_nchw_to_nhwc(const float* src, float* dst, int n, int c, int h, int w, vl_stream_t stream) {
    VL_CHECK(src && dst && n > 0 && h > 0 && w > 0 && c > 0, "vl_nchw_to_nhwc: bad argument");
    const int64_t hw = (int64_t)h * w;
    hipLaunchKernelGGL(permute3_kernel, dim3(grid_for(n * hw * c, 256, 8192)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       (int64_t)n, (int)hw, c, hw * c, (int64_t)1, hw);
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- LRN across channels (alexnet.py:79-89) ---------------------------------------------------
// One thread = one (image, pixel) column x CH consecutive channels; lanes walk pixels, so every
// load is a coalesced row of a channel plane.  The window lives in registers (static unroll).
__device__ __forceinline__ float pow_neg(float v, float beta) {
    // v >= bias > 0.  beta = 0.75 (the only value the reference uses) maps to two 1-ulp ops.
    if (beta == 0.75f) {
        const float r = __builtin_amdgcn_rsqf(v);
        return r * __builtin_amdgcn_sqrtf(r);
    }
    return powf(v, -beta);
}

template <int CH, int R>
__global__ void lrn_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int C, int HW, float alpha, float beta,
                               float bias) {
    const int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= (int64_t)n * HW) return;
    const int img = (int)(pos / HW);
    const int p = (int)(pos - (int64_t)img * HW);
    const int c0 = blockIdx.y * CH;
    const float* xp = x + (int64_t)img * C * HW + p;
    float* yp = y + (int64_t)img * C * HW + p;
    float v[CH + 2 * R];
#pragma unroll
    for (int i = 0; i < CH + 2 * R; ++i) {
        const int c = c0 - R + i;
        v[i] = (c >= 0 && c < C) ? xp[(int64_t)c * HW] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if (c0 + i < C) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d <= 2 * R; ++d) s += v[i + d] * v[i + d];
            yp[(int64_t)(c0 + i) * HW] = v[i + R] * pow_neg(bias + alpha * s, beta);
        }
    }
}

template <int CH, int R>
__global__ void lrn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int n, int C,
                               int HW, float alpha, float beta, float bias, int relu_fused, int W, int halo) {
    const int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >= (int64_t)n * HW) return;
    const int img = (int)(pos / HW);
    const int p = (int)(pos - (int64_t)img * HW);
    const int c0 = blockIdx.y * CH;
    const int64_t off = (int64_t)img * C * HW + p;
    const float* xp = x + off;
    const float* gp = dy + off;
    float xv[CH + 4 * R];  // channels c0-2R .. c0+CH+2R-1
#pragma unroll
    for (int i = 0; i < CH + 4 * R; ++i) {
        const int c = c0 - 2 * R + i;
        xv[i] = (c >= 0 && c < C) ? xp[(int64_t)c * HW] : 0.f;
    }
    // dx may carry a halo: plane pitch (H + 2 halo)(W + 2 halo), interior origin at (halo, halo)
    const int py = p / W, px = p - py * W;
    const int wp = W + 2 * halo;
    const int64_t dpp = (int64_t)(HW / W + 2 * halo) * wp;
    float* dxp = dx + (int64_t)img * C * dpp + (int64_t)(py + halo) * wp + px + halo;
    float t[CH + 2 * R];  // dy_c * x_c * s_c^(-beta-1) for c0-R .. c0+CH+R-1
    float u[CH];          // s_c^-beta for the owned channels
#pragma unroll
    for (int i = 0; i < CH + 2 * R; ++i) {
        const int c = c0 - R + i;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d <= 2 * R; ++d) s += xv[i + d] * xv[i + d];
        const float sc = bias + alpha * s;
        const float pw = pow_neg(sc, beta);
        const float g = (c >= 0 && c < C) ? gp[(int64_t)c * HW] : 0.f;
        t[i] = g * xv[i + R] * pw * __builtin_amdgcn_rcpf(sc);
        if (i >= R && i < CH + R) u[i - R] = g * pw;
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        if (c0 + i < C) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d <= 2 * R; ++d) a += t[i + d];
            const float xc = xv[i + 2 * R];
            float r = u[i] - 2.f * alpha * beta * xc * a;
            if (relu_fused) r = xc > 0.f ? r : 0.f;
            dxp[(int64_t)(c0 + i) * dpp] = r;
        }
    }
}

// Fused max-pool(3x3/2 VALID) backward + LRN backward (+ReluGrad): the gradient wrt the LRN output is
// never materialised.  One thread = one (image, pixel) column x CH channels.  The <= 4 pooling windows
// that contain the pixel are the same for every channel, so they are decoded once; per channel the
// routed gradient is a <= 4-term gather from the (tiny, cache resident) pooled gradient + arg-max maps.
// One workgroup = (image, CH-channel chunk, band of 2*PR input rows).  The pooled gradient and the arg-max
// bytes that band can touch ((PR+1) pooled rows x CH+2R channels: ~20 KB) are staged in LDS with coalesced
// loads; every per-pixel gather then hits LDS, never the texture path (a per-pixel global gather version
// was TA-bound at 3.5-6 ms for lrn1).
template <int CH, int R, int NPIX>
__global__ __launch_bounds__(256) void pool_lrn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dp,
                                                           const uint8_t* __restrict__ arg, float* __restrict__ dx, int C, int H,
                                                           int W, int OH, int OW, int64_t ps_n, int ps_c, int ps_h, float alpha,
                                                           float beta, float bias, int relu_fused, int halo) {
    constexpr int NC = CH + 2 * R;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int HW = H * W;
    const int img = blockIdx.z, c0 = blockIdx.y * CH;
    const int p0 = blockIdx.x * NPIX;                                 // band = NPIX consecutive pixels of the plane
    const int npix = min(NPIX, HW - p0);
    const int r0 = p0 / W, r1 = (p0 + npix - 1) / W;                  // input rows the band touches
    const int oh0 = (r0 >> 1) - 1;                                    // first pooled row that can reach the band
    const int prow = (r1 >> 1) - oh0 + 1;                             // pooled rows staged
    const int rowlen = prow * OW;
    float* sdp = reinterpret_cast<float*>(smem);                      // [NC][prow][OW]
    uint8_t* sarg = smem + (size_t)NC * rowlen * sizeof(float);       // [NC][prow][OW]
    const float* dpp = dp + (int64_t)img * ps_n;
    const uint8_t* ap = arg + (int64_t)img * ps_n;
    for (int e = threadIdx.x; e < NC * rowlen; e += 256) {
        const int ci = e / rowlen, rem = e - ci * rowlen;
        const int rr = rem / OW, ow = rem - rr * OW;
        const int c = c0 - R + ci, oh = oh0 + rr;
        const bool ok = c >= 0 && c < C && oh >= 0 && oh < OH;
        const int o = ok ? c * ps_c + oh * ps_h + ow : 0;
        sdp[e] = ok ? dpp[o] : 0.f;
        sarg[e] = ok ? ap[o] : (uint8_t)255;                           // 255 never equals a window-local index
    }
    __syncthreads();
    const float* xim = x + (int64_t)img * C * HW;
    const int wp = W + 2 * halo;
    const int64_t dplane = (int64_t)(H + 2 * halo) * wp;
    float* dxim = dx + (int64_t)img * C * dplane;
    for (int pix = threadIdx.x; pix < npix; pix += 256) {
        const int py = (p0 + pix) / W, px = (p0 + pix) - py * W;
        // the <= 4 windows (k = 3, s = 2) containing (py, px), as LDS offsets within a channel slab
        int woff[4], wloc[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oh = (py >> 1) - a, lr = (py & 1) + 2 * a;
            const bool rok = oh >= 0 && oh < OH && lr <= 2;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ow = (px >> 1) - b, lc = (px & 1) + 2 * b;
                const bool ok = rok && ow >= 0 && ow < OW && lc <= 2;
                woff[a * 2 + b] = ok ? (oh - oh0) * OW + ow : 0;
                wloc[a * 2 + b] = ok ? lr * 3 + lc : 254;              // 254: matches neither a real index nor the 255 filler
            }
        }
        const float* xp = xim + (int64_t)py * W + px;
        float xv[CH + 4 * R];
#pragma unroll
        for (int i = 0; i < CH + 4 * R; ++i) {
            const int c = c0 - 2 * R + i;
            xv[i] = (c >= 0 && c < C) ? xp[(int64_t)c * HW] : 0.f;
        }
        float t[NC], u[CH];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            float g = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = i * rowlen + woff[q];
                g += ((int)sarg[o] == wloc[q]) ? sdp[o] : 0.f;
            }
            float s = 0.f;
#pragma unroll
            for (int d = 0; d <= 2 * R; ++d) s += xv[i + d] * xv[i + d];
            const float sc = bias + alpha * s;
            const float pw = pow_neg(sc, beta);
            t[i] = g * xv[i + R] * pw * __builtin_amdgcn_rcpf(sc);
            if (i >= R && i < CH + R) u[i - R] = g * pw;
        }
        float* dxp = dxim + (int64_t)(py + halo) * wp + px + halo;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (c0 + i < C) {
                float a = 0.f;
#pragma unroll
                for (int d = 0; d <= 2 * R; ++d) a += t[i + d];
                const float xc = xv[i + 2 * R];
                float r = u[i] - 2.f * alpha * beta * xc * a;
                if (relu_fused) r = xc > 0.f ? r : 0.f;
                dxp[(int64_t)(c0 + i) * dplane] = r;
            }
        }
    }
}

// ---- the same fused backward as a CHANNEL STREAM (the form the benchmark layers run) ----------------------------------
// One thread = one pixel of one image; it walks ALL channels in order, carrying the LRN windows (5 inputs, 5 t-terms,
// 3 u-terms) in registers, so no channel is re-read or re-computed at chunk edges (the chunked form above reads 24 and
// evaluates 20 channels per 16 outputs), and the x loads of the next 16 channels are in flight while this 16 are computed:
// 16 KB per workgroup outstanding at all times -- the chunked form alternated load-wait and compute phases and reached
// ~40 % of the HBM rate.  The pooled gradient / arg-max of each 16-channel chunk go through LDS (two buffers, one barrier
// per chunk) as 8-byte {dp, arg} entries; the <= 4 pooling windows of a pixel are two adjacent entries in each of two
// pooled rows, i.e. two addresses + an immediate.  x loads / dx stores are buffer accesses with a per-thread constant
// pixel offset and a scalar channel offset: pixels past the plane fail the range check (loads 0, stores dropped).
// Channel cc enters the x window at iteration cc; t[cc-2] is then computable, and the output of channel cc-4.
// (band, image) of a workgroup in XCD-aware order (round 4).  Workgroup ids are dealt round-robin over the 8 XCDs, each with its own L2;
// with the natural order (band fastest) the bands of one image -- which share input rows / pooled rows at their seams, and for the
// backward kernel re-read each pooled row 2.25 times between them -- land on 8 different L2s and every re-read is an HBM fetch
// (measured: 1.44 x the algorithmic bytes fetched, tools/pw_pmc_probe.sh).  Here eight consecutive images take the 8 XCDs and an
// image's bands follow each other on ITS XCD, eight ids apart, i.e. dispatched together.  Grid = (bands, images rounded up to 8, z);
// false = padding.  Placement affects speed only.
__device__ __forceinline__ bool xcd_band_image(int nimg, int& band, int& img) {
    const int nb = gridDim.x;
    const int L = blockIdx.y * nb + blockIdx.x;
    const int g = L / (nb * 8), r = L - g * (nb * 8);
    band = r >> 3;
    img = g * 8 + (r & 7);
    return img < nimg;
}

static constexpr uint32_t PW_OOB = 0xF0000000u;   // fails the range check of every resource built below (sizes < 2^31)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pw_rsrc(const void* base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(uint32_t)bytes, 0x00020000);
}

// C8: dx is the packed-bf16 "c8" gradient [image][C / 8][H + 2 halo][W + 2 halo][8] (conv_c8.hip) instead of fp32 NCHW.  The chunk is
// then 8 channels starting at a multiple of 8: its outputs are channels c0 - 4 .. c0 + 3, i.e. the upper half of one 16-byte chunk and
// the lower half of the next, each stored as ONE 8-byte word per pixel (4 bf16, rounded to nearest even).
// C8 == 2: x (the LRN input = the conv's ReLU output) is packed too, [image][C / 8][H][W][8] bf16 without a halo: one 16-byte load
// per pixel brings a chunk's 8 channels.
template <int CHK, int NST, bool RELU, int C8 = 0>
__global__ __launch_bounds__(256, (C8 ? 3 : 1)) void pool_lrn_bwd_stream_kernel(const float* __restrict__ x, const float* __restrict__ dp,
                                                                  const uint8_t* __restrict__ arg, float* __restrict__ dx, int C,
                                                                  int H, int W, int OH, int OW, int64_t ps_n, int ps_c, int ps_h,
                                                                  int64_t pooled_bytes_f, float alpha, float bias, int halo, int cper, int nimg) {
    constexpr int BUF = NST * 256 + 16;                               // entries per LDS buffer (+ pad: entries -1 and one-past-the-end are read)
    // blockIdx.z owns the OUTPUT channels [R0, R1) (cper channels; the whole tensor when the grid is flat).  Its walk starts LEAD
    // channels earlier with empty windows: what it computes for channels below R0 there is wrong and dropped, and by channel R0 the x
    // window (4 back), the t window (2 more) are complete -- 4 (8: whole packed chunks) extra iterations per range buy workgroups
    // for launches whose (band, image) grid fills the chip 1.5 times (27 x 27 x 256 at 1024 frames) or not at all (128 frames).
    constexpr int LEAD = C8 ? 8 : 4;
    const int R0 = blockIdx.z * cper, R1 = min(C, R0 + cper);
    const int cs = max(R0 - LEAD, 0);                                 // never below channel 0: windows that start there ARE empty
    __shared__ __attribute__((aligned(16))) uint2 stage[2 * BUF];
    const int HW = H * W;
    int band, img;
    if (!xcd_band_image(nimg, band, img)) return;
    // fp32 output in a halo layout: the lanes enumerate WHOLE rows of that layout, halo columns included -- those lanes load nothing
    // (their x and routed gradient are 0, so the value they compute IS the 0.0 a halo holds) and a wave's stores are runs of full
    // sectors instead of 112-byte pieces at a 128-byte pitch (round 4: lrn_pool_fwd's lesson, profiles/r04_lrn_pool_fwd_experiments.txt)
    const int Wq = C8 == 0 ? W + 2 * halo : W, HWq = H * Wq;
    const int p0 = band * 256, p = p0 + threadIdx.x;
    const bool inplane = p < HWq;
    const int pc = inplane ? p : HWq - 1;
    const int py = pc / Wq, pxq = pc - py * Wq, pxr = pxq - (Wq - W) / 2;
    const bool valid = inplane && pxr >= 0 && pxr < W;
    const int px = min(max(pxr, 0), W - 1);
    const int r0 = p0 / Wq, r1 = min(p0 + 255, HWq - 1) / Wq;         // input rows of the band
    const int oh0 = (r0 >> 1) - 1;                                    // first pooled row that can reach the band
    const int prow = (r1 >> 1) - oh0 + 1;
    const int rowlen = prow * OW;

    // ---- per-thread constants of the staging pass: entry e = tid + 256 j  <->  (slab ci, pooled row, pooled column),
    // packed as ci << 24 | (row * pitch + column); -1 = nothing to fetch
    int st[NST];
#pragma unroll
    for (int j = 0; j < NST; ++j) {
        const int e = threadIdx.x + 256 * j;
        const int ci = e / rowlen, rem = e - ci * rowlen;
        const int rr = rem / OW, ow = rem - rr * OW;
        const int oh = oh0 + rr;
        st[j] = (ci < CHK && oh >= 0 && oh < OH) ? (ci << 24) | (oh * ps_h + ow) : -1;
    }
    // ---- the pixel's pooling windows (k = 3, s = 2): rows oh = py>>1 (a = 0) and oh - 1 (a = 1), columns ow = px>>1 (b = 0)
    // and ow - 1 (b = 1).  Addresses: entry (row, ow - 1) of slab 0, +1 entry for b = 0; wloc = the window-local index the
    // arg-max byte must equal for the window to route its gradient to this pixel (254 = never).
    int wloc[4];
    uint32_t rowaddr[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oh = (py >> 1) - a, lr = (py & 1) + 2 * a;
        const bool rok = valid && oh >= 0 && oh < OH && lr <= 2;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ow = (px >> 1) - b, lc = (px & 1) + 2 * b;
            wloc[a * 2 + b] = (rok && ow >= 0 && ow < OW && lc <= 2) ? lr * 3 + lc : 254;
        }
        const int rr = min(max(oh - oh0, 0), prow - 1);               // clamped: unreachable windows still read inside the buffer
        rowaddr[a] = (uint32_t)(8 + rr * OW + (px >> 1) - 1) * 8u;    // byte offset in a buffer; entry 8 = slab 0, row 0, column 0
    }
    const uint32_t slab_bytes = (uint32_t)rowlen * 8u;

    const __amdgpu_buffer_rsrc_t rs_x = C8 == 2 ? pw_rsrc(reinterpret_cast<const char*>(x) + (int64_t)img * ((C + 7) / 8) * HW * 16,
                                                          (int64_t)((C + 7) / 8) * HW * 16)
                                                : pw_rsrc(x + (int64_t)img * C * HW, (int64_t)C * HW * 4);
    const int wp = W + 2 * halo;
    const int dplane = (H + 2 * halo) * wp;
    static_assert(C8 == 0 || CHK == 8, "packed output: 8-channel chunks");
    const int CB = (C + 7) / 8;
    const __amdgpu_buffer_rsrc_t rs_dx = C8 ? pw_rsrc(reinterpret_cast<const char*>(dx) + (int64_t)img * CB * dplane * 16, (int64_t)CB * dplane * 16)
                                            : pw_rsrc(dx + (int64_t)img * C * dplane, (int64_t)C * dplane * 4);
    const __amdgpu_buffer_rsrc_t rs_dp = pw_rsrc(dp + (int64_t)img * ps_n, pooled_bytes_f);
    const __amdgpu_buffer_rsrc_t rs_arg = pw_rsrc(arg + (int64_t)img * ps_n, pooled_bytes_f / 4);
    const uint32_t voff_x = valid ? (uint32_t)(py * W + px) * (C8 == 2 ? 16u : 4u) : PW_OOB;
    const uint32_t voff_dx = C8 == 0 ? (inplane ? (uint32_t)((py + halo) * wp + pxq) * 4u : PW_OOB)
                                     : (valid ? (uint32_t)((py + halo) * wp + px + halo) * 16u : PW_OOB);
    const int x_cs = HW * 4, dx_cs = dplane * 4;                      // channel strides in bytes

    float dpv[NST];
    uint32_t av[NST];
    auto stage_load = [&](int c0) {                                   // slab ci holds channel c0 - 2 + ci
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int c = c0 - 2 + (st[j] >> 24);
            const bool ok = st[j] >= 0 && c >= 0 && c < C;
            const int off = c * ps_c + (st[j] & 0xffffff);
            dpv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_dp, ok ? off * 4 : (int)PW_OOB, 0, 0));
            av[j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs_arg, ok ? off : (int)PW_OOB, 0, 0);
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NST; ++j) stage[buf * BUF + 8 + threadIdx.x + 256 * j] = make_uint2(__builtin_bit_cast(uint32_t, dpv[j]), av[j]);
    };
    float xa[CHK], xb[CHK];
    auto x_load = [&](int c0, float (&v)[CHK]) {
        if constexpr (C8 == 2) {                                      // c0 is a multiple of 8: one chunk; blocks past C answer 0
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            const u4 w = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(c0 < C ? voff_x : PW_OOB), (c0 >> 3) * HW * 16, 0));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[2 * i] = __uint_as_float(w[i] << 16);
                v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < CHK; ++i) {
                const int cc = c0 + i;                                    // uniform; channels past C: the range check answers 0
                v[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)(cc >= 0 && cc < C ? voff_x : PW_OOB), max(cc, 0) * x_cs, 0));
            }
        }
    };

    float xw0 = 0.f, xw1 = 0.f, xw2 = 0.f, xw3 = 0.f, xw4 = 0.f;      // x[cc-4 .. cc]
    float tw0 = 0.f, tw1 = 0.f, tw2 = 0.f, tw3 = 0.f, tw4 = 0.f;      // t[cc-6 .. cc-2]
    float uw0 = 0.f, uw1 = 0.f, uw2 = 0.f;                            // u[cc-4 .. cc-2]
    const float k2ab = 2.f * alpha * 0.75f;
    const unsigned char* sbytes = reinterpret_cast<const unsigned char*>(stage);

    // one chunk: no branch inside, so the LDS reads and the math of neighbouring channels interleave freely
    auto chunk = [&](int c0, int buf, const float (&xin)[CHK]) {
        const uint32_t b0 = (uint32_t)(buf * BUF) * 8u;
        uint32_t a0 = b0 + rowaddr[0], a1 = b0 + rowaddr[1];
        float rr[4];
#pragma unroll
        for (int i = 0; i < CHK; ++i) {
            const int cc = c0 + i;
            // routed pooling gradient of channel cc - 2 (slab i)
            const uint2 e00 = *reinterpret_cast<const uint2*>(sbytes + a0 + 8), e01 = *reinterpret_cast<const uint2*>(sbytes + a0);
            const uint2 e10 = *reinterpret_cast<const uint2*>(sbytes + a1 + 8), e11 = *reinterpret_cast<const uint2*>(sbytes + a1);
            a0 += slab_bytes;
            a1 += slab_bytes;
            float g = 0.f;
            g += ((int)e00.y == wloc[0]) ? __builtin_bit_cast(float, e00.x) : 0.f;
            g += ((int)e01.y == wloc[1]) ? __builtin_bit_cast(float, e01.x) : 0.f;
            g += ((int)e10.y == wloc[2]) ? __builtin_bit_cast(float, e10.x) : 0.f;
            g += ((int)e11.y == wloc[3]) ? __builtin_bit_cast(float, e11.x) : 0.f;
            xw0 = xw1; xw1 = xw2; xw2 = xw3; xw3 = xw4; xw4 = xin[i];
            float s = 0.f;
            s += xw0 * xw0; s += xw1 * xw1; s += xw2 * xw2; s += xw3 * xw3; s += xw4 * xw4;
            const float sc = bias + alpha * s;
            const float rq = __builtin_amdgcn_rsqf(sc);
            const float pw = rq * __builtin_amdgcn_sqrtf(rq);        // sc^-0.75 (pow_neg's beta = 0.75 form)
            tw0 = tw1; tw1 = tw2; tw2 = tw3; tw3 = tw4;
            tw4 = g * xw2 * pw * (rq * rq);                           // sc^-1 = (sc^-1/2)^2: a multiply instead of a third quarter-rate op
            uw0 = uw1; uw1 = uw2; uw2 = g * pw;
            const int oc = cc - 4;                                    // uniform; outside [0, C): the store's range check drops it
            float a = 0.f;
            a += tw0; a += tw1; a += tw2; a += tw3; a += tw4;
            float r = uw0 - k2ab * xw0 * a;
            if (RELU) r = xw0 > 0.f ? r : 0.f;
            if constexpr (C8 != 0) {
                rr[i & 3] = r;
                if ((i & 3) == 3) {                                   // channels oc - 3 .. oc: half a chunk of block (oc - 3) >> 3
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
                    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
                    const u2 w = {__builtin_bit_cast(uint32_t, __builtin_convertvector(f2{rr[0], rr[1]}, b2)),
                                  __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{rr[2], rr[3]}, b2))};
                    const int blk = (oc - 3) >> 3;                    // uniform; outside [0, CB): dropped by the range check
                    // (round 4, measured and dropped: holding the lower half of a chunk in registers and storing the whole 16-byte chunk
                    // at once -- 0.54 -> 0.75 ms on layer 1)
                    __builtin_amdgcn_raw_buffer_store_b64(w, rs_dx, (int)((oc - 3 >= R0 && oc - 3 < R1 && blk < CB) ? voff_dx + (((oc - 3) & 7) * 2) : PW_OOB),
                                                          blk * dplane * 16, 0);
                }
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, r), rs_dx, (int)((oc >= R0 && oc < R1) ? voff_dx : PW_OOB),
                                                      oc * dx_cs, 0);
            }
        }
    };

    const int nchunks = (R1 + 4 - cs + CHK - 1) / CHK;
    // The pad entries of both buffers (8 in front: "entry -1" of a band's first pooled row; 8 behind) are READ by pixels whose window
    // does not exist and rejected only by their arg-max word never equalling wloc = 254 -- so they must not hold stale LDS bytes:
    // beside a kernel that leaves 0xfe words in LDS (the split-bf16 wgrad on a second stream) one step in three had ~170 of 67 M
    // gradient elements pick up a stale "dp" (round 3).  {0, 255} matches no window.
    if (threadIdx.x < 32) {
        const int t = threadIdx.x & 15, b2 = threadIdx.x >> 4;
        stage[b2 * BUF + (t < 8 ? t : NST * 256 + t)] = make_uint2(0u, 255u);
    }
    stage_load(cs);
    x_load(cs, xa);
    stage_write(0);
    __syncthreads();
    for (int k = 0; k < nchunks; k += 2) {                            // unrolled by two: the x registers alternate without copies
        stage_load(cs + (k + 1) * CHK);                               // next chunk's traffic goes out before this chunk's math;
        x_load(cs + (k + 1) * CHK, xb);                               // past the last channel every lane fails the channel test
        chunk(cs + k * CHK, 0, xa);
        stage_write(1);
        __syncthreads();
        if (k + 1 >= nchunks) break;
        stage_load(cs + (k + 2) * CHK);
        x_load(cs + (k + 2) * CHK, xa);
        chunk(cs + (k + 1) * CHK, 1, xb);
        stage_write(0);
        __syncthreads();
    }
}

// Channel ranges per (band, image) of the channel-stream backward: the count (<= 4) with the fewest rounds x channels walked, a
// round = 8 workgroups per CU (56 VGPRs, 12.5 KB of LDS).  `unit`: ranges start at multiples of it (8 for the packed output).
static int g_plb_forced_ranges = 0;
static int g_lpf_forced_ranges = 0;            // tests: vl_pool_lrn_bwd_test_ranges forces the forward's channel-range count as well
static int plb_channel_ranges(int64_t workgroups, int c, int unit, int lead, int* cper) {
    const int64_t slots = 8ll * vl_device_cus();
    int best = 1;
    int64_t best_cost = 0;
    for (int sp = 1; sp <= 4; ++sp) {
        const int per = (int)(((c + sp - 1) / sp + unit - 1) / unit * unit);
        if (sp > 1 && (per * (sp - 1) >= c || per < lead)) continue;  // the last range would be empty / a range shorter than the lead-in
        const int64_t cost = ((workgroups * sp + slots - 1) / slots) * (per + 12);
        if (sp == 1 || cost < best_cost) { best = sp; best_cost = cost; *cper = per; }
    }
    if (g_plb_forced_ranges >= 1 && g_plb_forced_ranges <= 4) {            // tests: vl_pool_lrn_bwd_test_ranges
        best = g_plb_forced_ranges;
        *cper = (int)(((c + best - 1) / best + unit - 1) / unit * unit);
        while (best > 1 && (*cper * (best - 1) >= c || *cper < lead)) {   // a range must start >= `lead` channels in (the kernel's walk
            --best;                                                        // restarts that far below its first output channel)
            *cper = (int)(((c + best - 1) / best + unit - 1) / unit * unit);
        }
    }
    return best;
}

/* Test hook (not part of the operator surface): force the number of channel ranges per (band, image) of vl_pool_lrn_bwd /
 * vl_pool_lrn_bwd_c8 to 1..4 (fewer when the channel count does not allow it); 0 = the cost model.  Process-wide, not thread-safe. */
extern "C" int vl_pool_lrn_bwd_test_ranges(int ranges) {
    VL_CHECK(ranges >= 0 && ranges <= 4, "vl_pool_lrn_bwd_test_ranges: 0 (automatic) .. 4");
    g_plb_forced_ranges = ranges;
    g_lpf_forced_ranges = ranges;                 // vl_lrn_pool_fwd / _c8 split their channel walk the same way (at most 8 ranges on their own)
    return 0;
}

extern "C" int vl_pool_lrn_bwd(const float* x, const float* dp, const uint8_t* argmax, float* dx, int n, int c, int h, int w,
                               int p_halo, int radius, float alpha, float beta, float bias, int relu_fused, int dx_halo,
                               vl_stream_t stream) {
    VL_CHECK(x && dp && argmax && dx && n > 0 && c > 0 && h >= 3 && w >= 3 && p_halo >= 0 && dx_halo >= 0, "vl_pool_lrn_bwd: bad argument");
    VL_CHECK(radius == 2, "vl_pool_lrn_bwd: only depth_radius 2 is built (alexnet.py:81); got %d", radius);
    const int oh = (h - 3) / 2 + 1, ow = (w - 3) / 2 + 1;
    const int owp = ow + 2 * p_halo;
    const int64_t pplane = (int64_t)(oh + 2 * p_halo) * owp;
    const int64_t origin = (int64_t)p_halo * owp + p_halo;
    VL_CHECK(n <= 65528, "vl_pool_lrn_bwd: batch %d exceeds the grid limit", n);
    {   // channel-stream form: a 256-pixel band x one channel chunk of pooled entries must fit NST * 256 staging slots
        const int rows = (255 + w - 1) / w + 1;                        // input rows a 256-pixel band can touch
        const int max_prow = rows / 2 + 2;
        const int64_t bytes_x = (int64_t)(c + 24) * h * w * 4, bytes_dx = (int64_t)(c + 24) * (h + 2 * dx_halo) * (w + 2 * dx_halo) * 4;
        const int64_t pooled_f = ((int64_t)c * pplane - origin) * 4;
        const bool ok = beta == 0.75f && bytes_x < (1ll << 31) && bytes_dx < (1ll << 31) && pooled_f < (1ll << 31) && pplane < (1 << 24) &&
                        !kPoolLrnChunked;
        int cper = c;
        const int nz = plb_channel_ranges((int64_t)ceil_div((int64_t)h * (w + 2 * dx_halo), 256) * n, c, 1, 4, &cper);
        const dim3 grid(ceil_div((int64_t)h * (w + 2 * dx_halo), 256), (n + 7) / 8 * 8, nz);   // bands over halo-layout rows; images padded to the 8 XCDs
        const int64_t psn = (int64_t)c * pplane;
#define VL_PLB_LAUNCH(CHK, NST)                                                                                                     \
    do {                                                                                                                            \
        if (relu_fused)                                                                                                             \
            hipLaunchKernelGGL((pool_lrn_bwd_stream_kernel<CHK, NST, true, 0>), grid, dim3(256), 0, (hipStream_t)stream, x, dp + origin, \
                               argmax + origin, dx, c, h, w, oh, ow, psn, (int)pplane, owp, pooled_f, alpha, bias, dx_halo, cper, n);  \
        else                                                                                                                        \
            hipLaunchKernelGGL((pool_lrn_bwd_stream_kernel<CHK, NST, false, 0>), grid, dim3(256), 0, (hipStream_t)stream, x, dp + origin, \
                               argmax + origin, dx, c, h, w, oh, ow, psn, (int)pplane, owp, pooled_f, alpha, bias, dx_halo, cper, n);  \
        VL_LAUNCH_CHECK();                                                                                                          \
        return 0;                                                                                                                   \
    } while (0)
        // chunk = 5 channels (110 registers, 12.5 KB of LDS: four workgroups per CU instead of two) when C + 4 is a multiple of it (96
        // and 256 are: no idle tail iterations): layer 1 0.79 -> 0.71 ms, layer 2 0.57 -> 0.51 ms against the 20-channel chunks of
        // round 1 (VL_POOL_LRN_BWD_R1=1)
        static const bool r1 = vl_exp_env("VL_POOL_LRN_BWD_R1") != nullptr;
        if (ok && !r1 && (c + 4) % 5 == 0 && (int64_t)5 * max_prow * ow <= 3 * 256) VL_PLB_LAUNCH(5, 3);
        // chunk = 20 channels when C + 4 is a multiple of it (96 and 256 are): no idle tail iterations
        if (ok && (c + 4) % 20 == 0 && (int64_t)20 * max_prow * ow <= 11 * 256 && !kPoolLrnChk16) VL_PLB_LAUNCH(20, 11);
        if (ok && (int64_t)16 * max_prow * ow <= 9 * 256) VL_PLB_LAUNCH(16, 9);
#undef VL_PLB_LAUNCH
    }
    constexpr int CH = 16, NPIX = 512;
    const int max_prow = ((NPIX + w - 1) / w + 1) / 2 + 3;             // pooled rows one band can touch
    const size_t lds = (((size_t)(CH + 4) * max_prow * ow * (sizeof(float) + 1)) + 15) & ~(size_t)15;
    VL_CHECK(lds <= 64 * 1024, "vl_pool_lrn_bwd: pooled plane too wide (%d) for the LDS staging", ow);
    dim3 grid(ceil_div((int64_t)h * w, NPIX), ceil_div(c, CH), n);
    hipLaunchKernelGGL((pool_lrn_bwd_kernel<CH, 2, NPIX>), grid, dim3(256), lds, (hipStream_t)stream, x, dp + origin, argmax + origin, dx,
                       c, h, w, oh, ow, (int64_t)c * pplane, (int)pplane, owp, alpha, beta, bias, relu_fused, dx_halo);
    VL_LAUNCH_CHECK();
    return 0;
}

/* vl_pool_lrn_bwd writing the packed-bf16 gradient dxb ("c8" layout of conv_c8.hip, dxb_halo) instead of fp32 dx: what the bf16 conv
 * path's wgrad / dgrad / bias gradient read.  Channel-stream form only (beta 0.75, planes that fit its staging). */
extern "C" int vl_pool_lrn_bwd_c8(const void* x, int x_packed, const float* dp, const uint8_t* argmax, void* dxb, int n, int c, int h, int w,
                                  int p_halo, int radius, float alpha, float beta, float bias, int relu_fused, int dxb_halo, vl_stream_t stream) {
    VL_CHECK(x && dp && argmax && dxb && n > 0 && c > 0 && h >= 3 && w >= 3 && p_halo >= 0 && dxb_halo >= 0, "vl_pool_lrn_bwd_c8: bad argument");
    VL_CHECK(radius == 2 && beta == 0.75f, "vl_pool_lrn_bwd_c8: only depth_radius 2, beta 0.75 are built (alexnet.py:81-84)");
    VL_CHECK(n <= 65528, "vl_pool_lrn_bwd_c8: batch %d exceeds the grid limit", n);
    const int oh = (h - 3) / 2 + 1, ow = (w - 3) / 2 + 1;
    const int owp = ow + 2 * p_halo;
    const int64_t pplane = (int64_t)(oh + 2 * p_halo) * owp;
    const int64_t origin = (int64_t)p_halo * owp + p_halo;
    const int rows = (255 + w - 1) / w + 1, max_prow = rows / 2 + 2;
    const int64_t bytes_x = (int64_t)(c + 24) * h * w * 4, bytes_dx = (int64_t)((c + 7) / 8 + 3) * (h + 2 * dxb_halo) * (w + 2 * dxb_halo) * 16;
    const int64_t pooled_f = ((int64_t)c * pplane - origin) * 4;
    VL_CHECK(bytes_x < (1ll << 31) && bytes_dx < (1ll << 31) && pooled_f < (1ll << 31) && pplane < (1 << 24) && (int64_t)8 * max_prow * ow <= 5 * 256,
             "vl_pool_lrn_bwd_c8: plane too large for the channel-stream form");
    int cper = c;
    const int nz = plb_channel_ranges((int64_t)ceil_div((int64_t)h * w, 256) * n, c, 8, 8, &cper);
    const dim3 grid(ceil_div((int64_t)h * w, 256), (n + 7) / 8 * 8, nz);
    const int64_t psn = (int64_t)c * pplane;
    const float* xf = (const float*)x;
#define VL_PLBC(RELU, MODE)                                                                                                             \
    hipLaunchKernelGGL((pool_lrn_bwd_stream_kernel<8, 5, RELU, MODE>), grid, dim3(256), 0, (hipStream_t)stream, xf, dp + origin, argmax + origin, \
                       (float*)dxb, c, h, w, oh, ow, psn, (int)pplane, owp, pooled_f, alpha, bias, dxb_halo, cper, n)
    if (relu_fused && x_packed) VL_PLBC(true, 2);
    else if (relu_fused) VL_PLBC(true, 1);
    else if (x_packed) VL_PLBC(false, 2);
    else VL_PLBC(false, 1);
#undef VL_PLBC
    VL_LAUNCH_CHECK();
    return 0;
}

extern "C" int vl_lrn_fwd(const float* x, float* y, int n, int c, int hw, int radius, float alpha, float beta, float bias,
                          vl_stream_t stream) {
    VL_CHECK(x && y && n > 0 && c > 0 && hw > 0, "vl_lrn_fwd: bad argument");
    VL_CHECK(radius == 2, "vl_lrn_fwd: only depth_radius 2 is built (alexnet.py:81); got %d", radius);
    constexpr int CH = 32;
    dim3 grid(ceil_div((int64_t)n * hw, 256), ceil_div(c, CH));
    hipLaunchKernelGGL((lrn_fwd_kernel<CH, 2>), grid, dim3(256), 0, (hipStream_t)stream, x, y, n, c, hw, alpha, beta, bias);
    VL_LAUNCH_CHECK();
    return 0;
}

extern "C" int vl_lrn_bwd(const float* x, const float* dy, float* dx, int n, int c, int hw, int radius, float alpha, float beta,
                          float bias, int relu_fused, int w, int dx_halo, vl_stream_t stream) {
    VL_CHECK(x && dy && dx && n > 0 && c > 0 && hw > 0, "vl_lrn_bwd: bad argument");
    VL_CHECK(w > 0 && hw % w == 0 && dx_halo >= 0, "vl_lrn_bwd: plane width %d does not divide hw %d", w, hw);
    VL_CHECK(radius == 2, "vl_lrn_bwd: only depth_radius 2 is built (alexnet.py:81); got %d", radius);
    constexpr int CH = 16;
    dim3 grid(ceil_div((int64_t)n * hw, 256), ceil_div(c, CH));
    hipLaunchKernelGGL((lrn_bwd_kernel<CH, 2>), grid, dim3(256), 0, (hipStream_t)stream, x, dy, dx, n, c, hw, alpha, beta, bias,
                       relu_fused, w, dx_halo);
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- fused forward of [LRN -> max_pool 3x3/2 VALID] (alexnet.py:79-98, 120-139) -------------------------------------
// The LRN output is only ever the pool's input (the backward pass reads the LRN INPUT, vl_pool_lrn_bwd), so it is never
// written to HBM: lrn1 + pool1 moved 2 x 1.28 + 1.28 + 0.39 GB per step through two kernels; fused it is 1.28 + 0.39 GB.
// One workgroup = one image x a band of PRB pooled rows (= 2 PRB + 1 input rows); a thread owns PPT pixels of the band and
// walks ALL channels with the 5-wide LRN window in registers (channel cc enters at iteration cc, LRN output cc - 2 leaves),
// the next chunk's loads in flight behind the current chunk's math.  Each chunk of CHK LRN outputs goes to LDS (two buffers,
// one barrier per chunk); the chunk's pooled outputs (<= NSL per thread, their window origins decoded once) take the
// strict-> first maximum of 9 LDS reads, exactly vl_maxpool_fwd's scan order, and are stored in the pool-output halo layout.
// C8: pout is the packed-bf16 "c8" tensor [image][C / 8][pooled plane][8] (conv_c8.hip) instead of fp32 NCHW: each pooled output is
// stored as one bf16 (nearest even) at its channel's slot of the pixel's chunk; the arg-max map is unchanged.
// C8 == 2: x is packed too ([image][C / 8][H][W][8] bf16, no halo): a 16-byte load per pixel brings 8 channels = four 2-channel chunks.
template <int CHK, int PPT, int NSL, int C8 = 0>
__global__ __launch_bounds__(512) void lrn_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ pout,
                                                           uint8_t* __restrict__ argout, int C, int H, int W, int OH, int OW, int prb,
                                                           int pplane, int owp, int p_halo, float alpha, float bias, int cper, int nimg) {
    extern __shared__ __attribute__((aligned(16))) float lbuf[];      // [2][CHK][npix]
    // blockIdx.z owns the OUTPUT channels [R0, R1) (cper of them, a multiple of CHK; the whole tensor when the grid is flat): its walk
    // starts two channels (one chunk) early so that the LRN window of channel R0 is complete, and ends two channels late.  Few-frame
    // launches only (round 4): 128 frames of layer 2 were 512 two-wave workgroups walking 256 channels each, 79 us where 1/8 of the
    // 1024-frame time is 37.
    const int R0 = blockIdx.z * cper, R1 = min(C, R0 + cper);
    const int cs = max(R0 - CHK, 0);                                  // CHK >= 2 = the LRN radius
    const int T = blockDim.x;
    int band, img;
    if (!xcd_band_image(nimg, band, img)) return;                     // bands of an image share an input row at every seam
    const int oh_a = band * prb;
    const int nprow = min(prb, OH - oh_a);                            // pooled rows of this band
    const int npix = (2 * nprow + 1) * W;                             // input pixels of this band
    const int p_base = 2 * oh_a * W;
    const int HW = H * W;
    const int CB = (C + 7) / 8;
    const __amdgpu_buffer_rsrc_t rs_x = C8 == 2 ? pw_rsrc(reinterpret_cast<const char*>(x) + (int64_t)img * CB * HW * 16, (int64_t)CB * HW * 16)
                                                : pw_rsrc(x + (int64_t)img * C * HW, (int64_t)C * HW * 4);
    const __amdgpu_buffer_rsrc_t rs_p = C8 != 0 ? pw_rsrc(reinterpret_cast<const char*>(pout) + (int64_t)img * CB * pplane * 16, (int64_t)CB * pplane * 16)
                                           : pw_rsrc(pout + (int64_t)img * C * pplane, (int64_t)C * pplane * 4);
    const __amdgpu_buffer_rsrc_t rs_a = pw_rsrc(argout + (int64_t)img * C * pplane, (int64_t)C * pplane);
    uint32_t voff_x[PPT];
    int lpix[PPT];                                                    // band-local pixel, -1 = none
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int pix = threadIdx.x + q * T;
        lpix[q] = pix < npix ? pix : -1;                              // (-1: its LRN values go to the buffer's spare slot)
        voff_x[q] = pix < npix ? (uint32_t)(p_base + pix) * (C8 == 2 ? 16u : 4u) : PW_OOB;
    }
    // pooled outputs of a chunk: o = tid + s T  <->  (slab ci, pooled row, pooled column)
    // fp32 output: the slots enumerate WHOLE rows of the halo layout, halo columns included (they store 0.0 = what the halo holds by
    // contract; the arg-max map's halo is never read), so that a wave's stores are runs of full 32-byte sectors.  Round 4, parts of the
    // kernel compiled out (VL_LRN_POOL_FORM, experiment build): loads + LRN + LDS writes alone ran at 5.4 - 5.9 TB/s, the two stores per
    // pooled output -- 112-byte value rows and 28-byte arg-max rows at a 128- / 32-byte pitch: every sector of the arg-max map written
    // partially -- cost a third of the kernel.  The packed output (C8) keeps interior slots (2-byte stores into 16-byte chunks).
    constexpr bool ROWS = C8 == 0;
    int s_lds[NSL], s_out[NSL];                                       // LDS float offset of the window origin | ci << 24 | output element offset
    uint32_t s_halo = 0;                                              // bit sl: slot sl is a halo column (stores 0)
    const int row_w = ROWS ? owp : OW;
    const int per_slab = nprow * row_w;
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl) {
        const int o = threadIdx.x + sl * T;
        const int ci = o / per_slab, rem = o - ci * per_slab;
        const int ohl = rem / row_w, col = rem - ohl * row_w;
        const int ow = ROWS ? col - p_halo : col;
        const bool ok = ci < CHK;
        const bool inner = ow >= 0 && ow < OW;
        s_lds[sl] = (ok && inner) ? ci * npix + 2 * ohl * W + 2 * ow : 0;
        s_out[sl] = ok ? (ci << 24) | ((oh_a + ohl + p_halo) * owp + ow + p_halo) : -1;
        if (ok && !inner) s_halo |= 1u << sl;
    }
    const int x_cs = HW * 4;
    float xa[PPT][CHK], xb[PPT][CHK];
    auto x_load = [&](int c0, float (&v)[PPT][CHK]) {
#pragma unroll
        for (int i = 0; i < CHK; ++i) {
            const int cc = c0 + i;                                    // uniform; channels past C: the range check answers 0
#pragma unroll
            for (int q = 0; q < PPT; ++q)
                v[q][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)(cc < C ? voff_x[q] : PW_OOB), cc * x_cs, 0));
        }
    };
    float xw[PPT][5];
#pragma unroll
    for (int q = 0; q < PPT; ++q)
#pragma unroll
        for (int d = 0; d < 5; ++d) xw[q][d] = 0.f;

    auto chunk = [&](int c0, int buf, const float (&xin)[PPT][CHK]) {
        float* lb = lbuf + buf * (CHK * npix + 1);                    // + 1: the spare slot idle lanes write to (no exec-mask branch)
        // LRN outputs of channels c0 - 2 .. c0 + CHK - 3 -> slabs 0 .. CHK - 1
#pragma unroll
        for (int i = 0; i < CHK; ++i) {
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                xw[q][0] = xw[q][1]; xw[q][1] = xw[q][2]; xw[q][2] = xw[q][3]; xw[q][3] = xw[q][4]; xw[q][4] = xin[q][i];
                float sq = 0.f;
#pragma unroll
                for (int d = 0; d < 5; ++d) sq += xw[q][d] * xw[q][d];
                const float sc = bias + alpha * sq;
                const float rq = __builtin_amdgcn_rsqf(sc);
                const float l = xw[q][2] * (rq * __builtin_amdgcn_sqrtf(rq));   // x * sc^-0.75 (pow_neg's beta = 0.75 form)
                lb[lpix[q] >= 0 ? i * npix + lpix[q] : CHK * npix] = l;
            }
        }
        __syncthreads();
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl) {
            const int c = c0 - 2 + (s_out[sl] >> 24);
            const bool live = s_out[sl] >= 0 && c >= R0 && c < R1;    // dead lanes scan slab 0's first window and store out of range
            {
                const float* wp = lb + s_lds[sl];
                float best = -INFINITY;
                int bi = 0;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const float v = wp[i * W + j];
                        if (v > best) {                               // strict: the first maximum in scan order wins
                            best = v;
                            bi = i * 3 + j;
                        }
                    }
                const int off = c * pplane + (s_out[sl] & 0xffffff);
                if (ROWS && ((s_halo >> sl) & 1u)) {                  // halo column: dead lanes scanned slab 0's first window
                    best = 0.f;
                    bi = 0;
                }
                if constexpr (C8 != 0) {
                    // (round 4, measured and dropped: gathering a pixel's 8-channel block in registers and storing it as ONE 16-byte
                    // word -- the pooling phase then runs on half the threads with two windows each and the kernel got 10 % slower)
                    const __bf16 hb = (__bf16)best;
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(uint16_t, hb), rs_p,
                                                          live ? ((c >> 3) * pplane + (s_out[sl] & 0xffffff)) * 16 + (c & 7) * 2 : (int)PW_OOB, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, best), rs_p, live ? off * 4 : (int)PW_OOB, 0, 0);
                }
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)bi, rs_a, live ? off : (int)PW_OOB, 0, 0);
            }
        }
    };
    const int nchunks = (R1 + 2 - cs + CHK - 1) / CHK;
    if constexpr (C8 == 2) {
        static_assert(C8 != 2 || CHK == 2, "packed input: 2-channel chunks (one dword of a pixel's 16-byte block)");
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        u4 cur[PPT], nxt[PPT];
        auto blk_load = [&](int kb, u4 (&v)[PPT]) {                   // channels 8 kb .. 8 kb + 7; blocks past C answer 0
#pragma unroll
            for (int q = 0; q < PPT; ++q)
                v[q] = __builtin_bit_cast(u4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(kb < CB ? voff_x[q] : PW_OOB), kb * HW * 16, 0));
        };
        // (channel ranges of the packed form are multiples of 8: cs = R0 - 2 lies in the block before R0's; the walk starts at that
        // block's first chunk and the chunks below cs only fill the window -- they compute channels the `live` test drops)
        const int kb0 = cs >> 3, kend = (R1 + 2 + CHK - 1) / CHK;     // first block; chunk index (from channel 0) behind the last one
        blk_load(kb0, cur);
        for (int kb = kb0; kb * 4 < kend; ++kb) {
            blk_load(kb + 1, nxt);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (kb * 4 + j >= kend) break;
                float xin[PPT][CHK];
#pragma unroll
                for (int q = 0; q < PPT; ++q) {
                    xin[q][0] = __uint_as_float(cur[q][j] << 16);
                    xin[q][1] = __uint_as_float(cur[q][j] & 0xffff0000u);
                }
                chunk((kb * 4 + j) * CHK, j & 1, xin);
            }
#pragma unroll
            for (int q = 0; q < PPT; ++q) cur[q] = nxt[q];
        }
        return;
    }
    x_load(cs, xa);
    for (int k = 0; k < nchunks; k += 2) {                            // unrolled by two: the x registers alternate without copies
        x_load(cs + (k + 1) * CHK, xb);
        chunk(cs + k * CHK, 0, xa);
        if (k + 1 >= nchunks) break;
        x_load(cs + (k + 2) * CHK, xa);
        chunk(cs + (k + 1) * CHK, 1, xb);
    }
}

template <int CHK, int PPT, int NSL, int C8 = 0>
static int launch_lrn_pool_fwd(const float* x, float* p, uint8_t* argmax, int n, int c, int h, int w, int p_halo, float alpha, float bias,
                               hipStream_t stream) {
    const int oh = (h - 3) / 2 + 1, ow = (w - 3) / 2 + 1;
    const int owp = ow + 2 * p_halo;
    const int64_t pplane = (int64_t)(oh + 2 * p_halo) * owp;
    // band = prb pooled rows: the largest that fits PPT pixels per thread of a <= 512-thread workgroup, NSL outputs per thread
    // and 64 KB of LDS; then the smallest thread count (multiple of 64) that still covers it
    int prb = 0, threads = 0;
    for (int cand = oh; cand >= 1 && !prb; --cand) {
        const int npix = (2 * cand + 1) * w;
        const int need_px = ceil_div(npix, PPT), need_out = ceil_div((int64_t)CHK * cand * (C8 == 0 ? owp : ow), NSL);
        int t = ceil_div(need_px > need_out ? need_px : need_out, 64) * 64;
        if (t <= 512 && (size_t)2 * (CHK * npix + 1) * sizeof(float) <= 64 * 1024) {
            prb = cand;
            threads = t < 64 ? 64 : t;
        }
    }
    VL_CHECK(prb > 0, "vl_lrn_pool_fwd: plane too wide (%d) for one band of pooled rows", w);
    // balance the bands (e.g. 28 pooled rows, at most 5 per band -> 6 bands of 5,5,5,5,4,4 rather than 5,5,5,5,5,3); few frames (one
    // rank's 128-frame shard of the 8-GPU job: layer 2 was ONE band x 128 images = half the CUs, 98 us where 1/8 of the 1024-frame time
    // is 37): cut more bands, up to two workgroups per CU, at the price of their shared input rows (round 4)
    int bands = ceil_div(oh, prb);
    const int want = ceil_div(2 * (int64_t)vl_device_cus(), n);
    if (bands < want) bands = want < oh ? want : oh;
    prb = ceil_div(oh, bands);
    bands = ceil_div(oh, prb);
    {
        const int need_px = ceil_div((2 * prb + 1) * w, PPT), need_out = ceil_div((int64_t)CHK * prb * (C8 == 0 ? owp : ow), NSL);
        threads = ceil_div(need_px > need_out ? need_px : need_out, 64) * 64;
    }
    const size_t lds = (size_t)2 * (CHK * (2 * prb + 1) * w + 1) * sizeof(float);       // two buffers, each with a spare slot
    static const bool verbose = vl_exp_env("VL_LRN_POOL_VERBOSE") != nullptr;
    if (verbose) fprintf(stderr, "lrn_pool_fwd<%d,%d,%d>: bands %d prb %d threads %d lds %zu\n", CHK, PPT, NSL, bands, prb, threads, lds);
    // channel ranges (grid z) until the launch holds eight workgroups per CU, in units of 8 channels (whole packed blocks, whole chunks),
    // at least 32 channels each: every range re-reads 2 + 2 channels of its neighbours
    int nz = 1, cper = c;
    {
        const int64_t wgs = (int64_t)bands * n, slots = 8ll * vl_device_cus();
        const int can = c / 32 > 0 ? c / 32 : 1;
        int wantz = (int)((slots + wgs - 1) / wgs);
        wantz = wantz > 8 ? 8 : wantz;
        wantz = wantz > can ? can : wantz;
        if (g_lpf_forced_ranges >= 1) wantz = g_lpf_forced_ranges;
        if (wantz > 1) {
            cper = ((c + wantz - 1) / wantz + 7) / 8 * 8;
            nz = (c + cper - 1) / cper;
        }
    }
    hipLaunchKernelGGL((lrn_pool_fwd_kernel<CHK, PPT, NSL, C8>), dim3(bands, (n + 7) / 8 * 8, nz), dim3(threads), lds, stream, x, p, argmax, c, h,
                       w, oh, ow, prb, (int)pplane, owp, p_halo, alpha, bias, cper, n);
    VL_LAUNCH_CHECK();
    return 0;
}

extern "C" int vl_lrn_pool_fwd(const float* x, float* p, uint8_t* argmax, int n, int c, int h, int w, int p_halo, int radius,
                               float alpha, float beta, float bias, vl_stream_t stream) {
    VL_CHECK(x && p && argmax && n > 0 && c > 0 && h >= 3 && w >= 3 && p_halo >= 0, "vl_lrn_pool_fwd: bad argument");
    VL_CHECK(radius == 2 && beta == 0.75f, "vl_lrn_pool_fwd: only depth_radius 2, beta 0.75 are built (alexnet.py:81-84)");
    VL_CHECK(n <= 65528, "vl_lrn_pool_fwd: batch %d exceeds the grid limit", n);
    const int oh = (h - 3) / 2 + 1, ow = (w - 3) / 2 + 1;
    const int owp = ow + 2 * p_halo;
    const int64_t pplane = (int64_t)(oh + 2 * p_halo) * owp;
    VL_CHECK((int64_t)(c + 16) * h * w * 4 < (1ll << 31) && (int64_t)c * pplane * 4 < (1ll << 31) && pplane < (1 << 24),
             "vl_lrn_pool_fwd: image too large for 32-bit buffer offsets");
    // Chunk of 2 channels, 2 pixels and 1 pooled output per thread: 46 registers, 13 KB of LDS -> the CU holds 4-5 workgroups of 448
    // threads.  The round-1 shape <8, 3, 6> (8-channel chunks, 214 registers, 55 KB) left ONE 320-thread workgroup = 5 waves on a CU and
    // ran LRN's ~13 VALU per element at that occupancy: layer 1 0.63 -> 0.45 ms, layer 2 0.33 -> 0.28 ms (sweep of 14 shapes on MI355X,
    // tools/pw_probe.py lrn_pool_fwd; VL_LRN_POOL_R1=1 runs the old shape for comparison).
    static const bool r1 = vl_exp_env("VL_LRN_POOL_R1") != nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (r1) return launch_lrn_pool_fwd<8, 3, 6>(x, p, argmax, n, c, h, w, p_halo, alpha, bias, s);
    return launch_lrn_pool_fwd<2, 2, 1>(x, p, argmax, n, c, h, w, p_halo, alpha, bias, s);
}

/* vl_lrn_pool_fwd with the pooled output written as packed bf16 (pb: "c8" layout of the bf16 conv path, p_halo; nearest even) instead of
 * fp32 p -- the next conv's operand; argmax keeps the fp32 form's NCHW layout with p_halo. */
extern "C" int vl_lrn_pool_fwd_c8(const void* x, int x_packed, void* pb, uint8_t* argmax, int n, int c, int h, int w, int p_halo, int radius,
                                  float alpha, float beta, float bias, vl_stream_t stream) {
    VL_CHECK(x && pb && argmax && n > 0 && c > 0 && h >= 3 && w >= 3 && p_halo >= 0, "vl_lrn_pool_fwd_c8: bad argument");
    VL_CHECK(radius == 2 && beta == 0.75f, "vl_lrn_pool_fwd_c8: only depth_radius 2, beta 0.75 are built (alexnet.py:81-84)");
    VL_CHECK(n <= 65528, "vl_lrn_pool_fwd_c8: batch %d exceeds the grid limit", n);
    const int oh = (h - 3) / 2 + 1, ow = (w - 3) / 2 + 1;
    const int64_t pplane = (int64_t)(oh + 2 * p_halo) * (ow + 2 * p_halo);
    VL_CHECK((int64_t)(c + 16) * h * w * 4 < (1ll << 31) && (int64_t)((c + 7) / 8) * pplane * 16 < (1ll << 31) && pplane < (1 << 24),
             "vl_lrn_pool_fwd_c8: image too large for 32-bit buffer offsets");
    if (x_packed) return launch_lrn_pool_fwd<2, 2, 1, 2>((const float*)x, (float*)pb, argmax, n, c, h, w, p_halo, alpha, bias, (hipStream_t)stream);
    return launch_lrn_pool_fwd<2, 2, 1, 1>((const float*)x, (float*)pb, argmax, n, c, h, w, p_halo, alpha, bias, (hipStream_t)stream);
}

// ---- max-pool VALID (alexnet.py:91-98) --------------------------------------------------------
// Index decode uses 32-bit magic-number division (total element counts are < 2^31, checked on the host);
// K, S > 0 are compile-time for the 3x3/2 case the reference uses, 0 = runtime k, s.
template <int K, int S>
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ arg, uint32_t total,
                                   int C, int H, int W, int OH, int OW, int k_, int s_, FastDiv dOHW, FastDiv dOW, FastDiv dC,
                                   int64_t ysn, int64_t ysc, int64_t ysh, int64_t ysw) {
    const int k = K ? K : k_, s = S ? S : s_;
    const int OHW = OH * OW;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const uint32_t plane = fd_div(e, dOHW);
        const uint32_t p = e - plane * OHW;
        const uint32_t oh = fd_div(p, dOW), ow = p - oh * OW;
        const uint32_t img = fd_div(plane, dC), c = plane - img * C;
        const float* xp = x + ((int64_t)plane * H + oh * s) * W + ow * s;
        float best = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int i = 0; i < (K ? K : 1); ++i)
#pragma unroll
            for (int j = 0; j < (K ? K : 1); ++j) {
                if (K) {
                    const float v = xp[i * W + j];
                    if (v > best) {  // strict: the first maximum in scan order wins
                        best = v;
                        bi = i * K + j;
                    }
                }
            }
        if (!K) {
            for (int i = 0; i < k; ++i)
                for (int j = 0; j < k; ++j) {
                    const float v = xp[i * W + j];
                    if (v > best) {
                        best = v;
                        bi = i * k + j;
                    }
                }
        }
        const int64_t o = img * ysn + c * ysc + oh * ysh + ow * ysw;
        y[o] = best;
        if (arg) arg[o] = (uint8_t)bi;
    }
}

template <int K, int S>
__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ arg, float* __restrict__ dx,
                                   const float* __restrict__ mask, uint32_t total, int C, int H, int W, int OH, int OW, int k_,
                                   int s_, FastDiv dHW, FastDiv dW, FastDiv dC, int64_t ysn, int64_t ysc, int64_t ysh, int64_t ysw,
                                   int halo) {
    const int k = K ? K : k_, s = S ? S : s_;
    const int HW = H * W;
    const int wp = W + 2 * halo;
    const int64_t pp = (int64_t)(H + 2 * halo) * wp;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const uint32_t plane = fd_div(e, dHW);
        const uint32_t p = e - plane * HW;
        const int ih = (int)fd_div(p, dW), iw = (int)(p - ih * W);
        const uint32_t img = fd_div(plane, dC), c = plane - img * C;
        float acc = 0.f;
        if (!mask || mask[e] > 0.f) {
            int oh_lo = (ih - k + s) / s;  // ceil((ih - k + 1) / s) for ih-k+1 >= 0
            if (ih - k + 1 < 0) oh_lo = 0;
            int ow_lo = (iw - k + s) / s;
            if (iw - k + 1 < 0) ow_lo = 0;
            const int oh_hi = min(ih / s, OH - 1), ow_hi = min(iw / s, OW - 1);
            const int64_t base = img * ysn + c * ysc;
            for (int oh = oh_lo; oh <= oh_hi; ++oh)
                for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                    const int64_t o = base + oh * ysh + ow * ysw;
                    if ((int)arg[o] == (ih - oh * s) * k + (iw - ow * s)) acc += dy[o];
                }
        }
        dx[(int64_t)plane * pp + (int64_t)(ih + halo) * wp + iw + halo] = acc;
    }
}

extern "C" int vl_maxpool_fwd(const float* x, float* y, uint8_t* argmax, int n, int c, int h, int w, int k, int s, int64_t ys_n,
                              int64_t ys_c, int64_t ys_h, int64_t ys_w, vl_stream_t stream) {
    VL_CHECK(x && y && n > 0 && c > 0 && k > 0 && s > 0 && h >= k && w >= k && k * k <= 255, "vl_maxpool_fwd: bad argument");
    const int oh = (h - k) / s + 1, ow = (w - k) / s + 1;
    const int64_t total = (int64_t)n * c * oh * ow;
    VL_CHECK(total < (1ll << 31) && (int64_t)n * c * h * w < (1ll << 40), "vl_maxpool_fwd: tensor too large");
    const dim3 grid(grid_for(total, 256, 16384));
    const FastDiv d1 = make_fastdiv(oh * ow), d2 = make_fastdiv(ow), d3 = make_fastdiv(c);
    if (k == 3 && s == 2)
        hipLaunchKernelGGL((maxpool_fwd_kernel<3, 2>), grid, dim3(256), 0, (hipStream_t)stream, x, y, argmax, (uint32_t)total, c, h, w, oh,
                           ow, k, s, d1, d2, d3, ys_n, ys_c, ys_h, ys_w);
    else
        hipLaunchKernelGGL((maxpool_fwd_kernel<0, 0>), grid, dim3(256), 0, (hipStream_t)stream, x, y, argmax, (uint32_t)total, c, h, w, oh,
                           ow, k, s, d1, d2, d3, ys_n, ys_c, ys_h, ys_w);
    VL_LAUNCH_CHECK();
    return 0;
}

// pool5's backward (3x3 / 2, pooled tensor stored (h, w, c)-flat for fc6, input gradient NCHW): in the generic kernel above a
// wave walks input columns, so its <= 4 pooled neighbours are c * 4 bytes apart -- every byte / float of the pooled tensors is
// its own cache line.  Here a workgroup stages the pooled gradient and arg-max of CB channels of one image through LDS
// (coalesced along c, which is contiguous in the source), then writes the NCHW planes coalesced along the pixels.
// Round 3: lane = pixel, channels in the loop (was: one flat element index per thread with two divisions per element): 0.355 ->
// 0.176 ms at 1024 frames (tools/pool5_probe.py; VL_MAXPOOL_GENERIC=1 runs the element-per-thread kernel).  The same treatment
// of the FORWARD (64 planes copied to LDS, lane = channel) was slower than the element-per-thread kernel (0.105 vs 0.071 ms) and
// was dropped: its strided writes are absorbed by the L2, the serial copy phase is not.
template <int CB, bool MASK>
__global__ __launch_bounds__(256) void maxpool_bwd_hwc_k3s2_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ arg,
                                                                   float* __restrict__ dx, const float* __restrict__ mask, int C, int H,
                                                                   int W, int OH, int OW, int halo) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pl_smem[];
    const int OHW = OH * OW, HW = H * W, SP = OHW + 1;
    uint2* sp = reinterpret_cast<uint2*>(pl_smem);                     // [CB][OHW + 1] entries {dy bits, arg-max}
    const int img = blockIdx.y, c0 = blockIdx.x * CB;
    const int nch = min(CB, C - c0);
    const float* dyp = dy + (int64_t)img * OHW * C;
    const uint8_t* ap = arg + (int64_t)img * OHW * C;
    // Round 4: every loop of this kernel was ONE load, its wait, its use -- hipcc had turned the conditional loads into branches
    // (exec-mask tests around each), 9 + 64 dependent round trips per thread.  Now: clamped addresses instead of conditions, the
    // loads of four entries / four channels issued before the first is used, selects instead of branches.
    constexpr int EU = 4;
    for (int e0 = threadIdx.x; e0 < CB * OHW; e0 += 256 * EU) {
        uint32_t dv[EU], av[EU];
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int e = min(e0 + 256 * u, CB * OHW - 1);
            const int p = e / CB, cl = min(e - p * CB, nch - 1);       // lanes walk c: contiguous in the (h, w, c) source
            dv[u] = __float_as_uint(dyp[(int64_t)p * C + c0 + cl]);
            av[u] = (uint32_t)ap[(int64_t)p * C + c0 + cl];
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int e = e0 + 256 * u;
            const int p = e / CB, cl = e - p * CB;
            if (e < CB * OHW) sp[cl * SP + p] = cl < nch ? make_uint2(dv[u], av[u]) : make_uint2(0u, 255u);
        }
    }
    if ((int)threadIdx.x < CB) sp[threadIdx.x * SP + OHW] = make_uint2(0u, 255u);   // the pad entry unreachable windows read
    __syncthreads();
    // lane = input pixel (64 at a time), wave w takes channels w, w + 4, ...: a pixel's <= 4 windows (rows ih>>1 and one above,
    // columns iw>>1 and one to the left) are decoded ONCE per pixel chunk, the channel loop is 4 LDS reads, 4 compare-selects, the
    // mask load and the store -- no division per element
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wp = W + 2 * halo;
    const int64_t pp = (int64_t)(H + 2 * halo) * wp;
    // (round 4) the lanes enumerate whole rows of dx's halo layout, halo columns included (they store the 0.0 a halo holds): 13-float
    // rows at a 15-float pitch left every sector partially written -- pool_lrn_bwd gained 22 % from the same change
    const int HWq = H * wp;
    for (int pc = 0; pc * 64 < HWq; ++pc) {
        const int pq = pc * 64 + lane;
        const bool inplane = pq < HWq;
        const int pcl = inplane ? pq : HWq - 1;
        const int ih = pcl / wp, iwq = pcl - ih * wp, iwr = iwq - halo;
        const bool valid = inplane && iwr >= 0 && iwr < W;
        const int iw = min(max(iwr, 0), W - 1);
        const int p = ih * W + iw;                                     // (clamped: always a pixel of the plane)
        int off[4], wl[4];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oh = (ih >> 1) - a, lr = (ih & 1) + 2 * a;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int ow = (iw >> 1) - b, lc = (iw & 1) + 2 * b;
                const bool ok = valid && oh >= 0 && oh < OH && lr <= 2 && ow >= 0 && ow < OW && lc <= 2;
                off[a * 2 + b] = ok ? oh * OW + ow : OHW;              // entry OHW of a row: the pad, never a match
                wl[a * 2 + b] = ok ? lr * 3 + lc : 254;
            }
        }
        const int64_t dxo = (int64_t)(ih + halo) * wp + iwq;
        constexpr int CU = 4;                                          // channels per pass
        for (int cb = wave; cb < nch; cb += 4 * CU) {
            float mk[CU];
            uint2 en[CU][4];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int cl = min(cb + 4 * u, nch - 1);               // (clamped: a repeated channel is computed and not stored)
                if constexpr (MASK) mk[u] = mask[((int64_t)img * C + c0 + cl) * HW + p];
                else mk[u] = 1.f;
                const uint2* row = sp + cl * SP;
#pragma unroll
                for (int k = 0; k < 4; ++k) en[u][k] = row[off[k]];
            }
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int cl = cb + 4 * u;
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc += ((int)en[u][k].y == wl[k]) ? __uint_as_float(en[u][k].x) : 0.f;
                if (!(mk[u] > 0.f)) acc = 0.f;
                if (inplane && cl < nch) dx[((int64_t)img * C + c0 + cl) * pp + dxo] = acc;   // halo columns: no window matched, acc = 0
            }
        }
    }
}

extern "C" int vl_maxpool_bwd(const float* dy, const uint8_t* argmax, float* dx, const float* relu_mask, int n, int c, int h,
                              int w, int k, int s, int64_t ys_n, int64_t ys_c, int64_t ys_h, int64_t ys_w, int dx_halo,
                              vl_stream_t stream) {
    VL_CHECK(dy && argmax && dx && n > 0 && c > 0 && k > 0 && s > 0 && h >= k && w >= k && dx_halo >= 0, "vl_maxpool_bwd: bad argument");
    const int oh = (h - k) / s + 1, ow = (w - k) / s + 1;
    const int64_t total = (int64_t)n * c * h * w;
    VL_CHECK(total < (1ll << 31), "vl_maxpool_bwd: tensor too large");
    if (k == 3 && s == 2 && ys_c == 1 && ys_w == c && ys_h == (int64_t)ow * c && ys_n == (int64_t)oh * ow * c && n <= 65535 &&
        (size_t)64 * (oh * ow + 1) * 8 <= 48 * 1024 && !kMaxpoolGeneric) {
        // the (h, w, c)-flat pooled layout (pool5 -> fc6): LDS-transposed form
        // 64 channels per workgroup; 16 when that grid is under four workgroups per CU (128 frames: 512 workgroups of 48 serial
        // iterations per wave took 45 us where 1/8 of the 1024-frame time is 18; round 4)
        if ((int64_t)ceil_div(c, 64) * n >= 4ll * vl_device_cus()) {
            constexpr int CB = 64;
            auto k = relu_mask ? maxpool_bwd_hwc_k3s2_kernel<CB, true> : maxpool_bwd_hwc_k3s2_kernel<CB, false>;
            hipLaunchKernelGGL(k, dim3(ceil_div(c, CB), n), dim3(256), (size_t)CB * (oh * ow + 1) * 8, (hipStream_t)stream,
                               dy, argmax, dx, relu_mask, c, h, w, oh, ow, dx_halo);
        } else {
            constexpr int CB = 16;
            auto k = relu_mask ? maxpool_bwd_hwc_k3s2_kernel<CB, true> : maxpool_bwd_hwc_k3s2_kernel<CB, false>;
            hipLaunchKernelGGL(k, dim3(ceil_div(c, CB), n), dim3(256), (size_t)CB * (oh * ow + 1) * 8, (hipStream_t)stream,
                               dy, argmax, dx, relu_mask, c, h, w, oh, ow, dx_halo);
        }
        VL_LAUNCH_CHECK();
        return 0;
    }
    const dim3 grid(grid_for(total, 256, 16384));
    const FastDiv d1 = make_fastdiv(h * w), d2 = make_fastdiv(w), d3 = make_fastdiv(c);
    if (k == 3 && s == 2)
        hipLaunchKernelGGL((maxpool_bwd_kernel<3, 2>), grid, dim3(256), 0, (hipStream_t)stream, dy, argmax, dx, relu_mask, (uint32_t)total,
                           c, h, w, oh, ow, k, s, d1, d2, d3, ys_n, ys_c, ys_h, ys_w, dx_halo);
    else
        hipLaunchKernelGGL((maxpool_bwd_kernel<0, 0>), grid, dim3(256), 0, (hipStream_t)stream, dy, argmax, dx, relu_mask, (uint32_t)total,
                           c, h, w, oh, ow, k, s, d1, d2, d3, ys_n, ys_c, ys_h, ys_w, dx_halo);
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- bias gradients ---------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* sm) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sm[wv] = v;
    __syncthreads();
    const float r = sm[0] + sm[1] + sm[2] + sm[3];
    __syncthreads();
    return r;
}

// grid (C, S): block (c, s) sums channel c over images [s*per, (s+1)*per) -> ws[s*C + c]
__global__ void bias_grad_stage1(const float* __restrict__ dy, float* __restrict__ ws, int n, int C, int HW, int per) {
    __shared__ float sm[4];
    const int c = blockIdx.x, s = blockIdx.y;
    const int n0 = s * per, n1 = min(n, n0 + per);
    float acc = 0.f;
    // eight images' loads in flight per pass (one image per pass was one dependent round trip per image: conv3's 15 x 15 planes, 36
    // images per workgroup, ran at 3 TB/s -- latency, not bandwidth; round 4)
    const int64_t istride = (int64_t)C * HW;
    for (int img = n0; img < n1; img += 8) {
        const float* p = dy + ((int64_t)img * C + c) * HW;
        const int cnt = min(8, n1 - img);
        for (int i = threadIdx.x; i < HW; i += 256) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = e < cnt ? p[e * istride + i] : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += v[e];
        }
    }
    acc = block_sum_256(acc, sm);
    if (threadIdx.x == 0) ws[s * C + c] = acc;
}

__global__ void sum_partials_kernel(const float* __restrict__ ws, float* __restrict__ out, int count, int S) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= count) return;
    float a = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {                                       // eight loads in flight, added in slice order
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ws[(int64_t)(s + e) * count + c];
#pragma unroll
        for (int e = 0; e < 8; ++e) a += v[e];
    }
    for (; s < S; ++s) a += ws[(int64_t)s * count + c];
    out[c] = a;
}

extern "C" int vl_bias_grad_nchw(const float* dy, float* db, float* ws, int n, int c, int hw, vl_stream_t stream) {
    VL_CHECK(dy && db && ws && n > 0 && c > 0 && hw > 0, "vl_bias_grad_nchw: bad argument");
    // slices of >= 8192 elements per channel, so that the second stage adds a handful of partials (S = 64 for any n cost 15 us there)
    const int64_t want = ((int64_t)n * hw + 8191) / 8192;
    const int S = want < 1 ? 1 : want > 64 ? 64 : want > n ? n : (int)want;
    const int per = ceil_div(n, S);
    const int S2 = ceil_div(n, per);
    hipLaunchKernelGGL(bias_grad_stage1, dim3(c, S2), dim3(256), 0, (hipStream_t)stream, dy, S2 == 1 ? db : ws, n, c, hw, per);
    VL_LAUNCH_CHECK();
    if (S2 > 1) {
        hipLaunchKernelGGL(sum_partials_kernel, dim3(ceil_div(c, 256)), dim3(256), 0, (hipStream_t)stream, ws, db, c, S2);
        VL_LAUNCH_CHECK();
    }
    return 0;
}

// grid (ceil(ncol / 32), S): a workgroup sums 32 columns over rows [s*per, (s+1)*per) -- 8 row groups of 32 lanes (whole 128-byte
// lines), combined through LDS in a fixed order -> dst[s*ncol + j].  With S = 1 dst is the result itself (one launch: the
// classifier / LSTM / fc6 bias gradients of a 128-frame shard were two launches of 6 + 17 us each).
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ a, int64_t lda, float* __restrict__ dst, int m, int ncol, int per) {
    __shared__ float part[8][32];
    const int lx = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + lx;
    const int s = blockIdx.y;
    const int r1 = min(m, (s + 1) * per);
    float acc = 0.f;
    if (j < ncol) {
        int r = s * per + g;
        for (; r + 24 < r1; r += 32) {
            const float v0 = a[(int64_t)r * lda + j], v1 = a[(int64_t)(r + 8) * lda + j], v2 = a[(int64_t)(r + 16) * lda + j],
                        v3 = a[(int64_t)(r + 24) * lda + j];
            acc += v0; acc += v1; acc += v2; acc += v3;
        }
        for (; r < r1; r += 8) acc += a[(int64_t)r * lda + j];
    }
    part[g][lx] = acc;
    __syncthreads();
    if (g == 0 && j < ncol) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += part[i][lx];
        dst[(int64_t)s * ncol + j] = t;
    }
}

extern "C" int vl_colsum(const float* a, int64_t lda, float* out, float* ws, int m, int n, vl_stream_t stream) {
    VL_CHECK(a && out && ws && m > 0 && n > 0 && lda >= n, "vl_colsum: bad argument");
    const int S = ceil_div(m, 256) < 64 ? ceil_div(m, 256) : 64;
    const int per = ceil_div(m, S);
    const int S2 = ceil_div(m, per);
    hipLaunchKernelGGL(colsum_stage1, dim3(ceil_div(n, 32), S2), dim3(256), 0, (hipStream_t)stream, a, lda, S2 == 1 ? out : ws, m, n, per);
    VL_LAUNCH_CHECK();
    if (S2 > 1) {
        hipLaunchKernelGGL(sum_partials_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, ws, out, n, S2);
        VL_LAUNCH_CHECK();
    }
    return 0;
}

// ---- LSTM gate pointwise (TF BasicLSTMCell; lstm.py:17) ---------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void lstm_step_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ gh, float* __restrict__ act,
                                     float* __restrict__ cseq, float* __restrict__ hseq, float* __restrict__ hprev, int batch,
                                     int T, int t, int H, float forget_bias) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= batch * H) return;
    const int b = e / H, u = e - b * H;
    const int64_t r = (int64_t)b * T + t;
    const float* zx = gx + r * 4 * H;
    float zi = zx[u], zj = zx[H + u], zf = zx[2 * H + u], zo = zx[3 * H + u];
    if (gh) {
        const float* zh = gh + (int64_t)b * 4 * H;
        zi += zh[u];
        zj += zh[H + u];
        zf += zh[2 * H + u];
        zo += zh[3 * H + u];
    }
    const float gi = sigmoidf_(zi), gj = tanhf(zj), gf = sigmoidf_(zf + forget_bias), go = sigmoidf_(zo);
    const float cp = t > 0 ? cseq[(r - 1) * H + u] : 0.f;
    const float hp = t > 0 ? hseq[(r - 1) * H + u] : 0.f;
    const float c = cp * gf + gi * gj;
    const float h = tanhf(c) * go;
    float* a = act + r * 4 * H;
    a[u] = gi;
    a[H + u] = gj;
    a[2 * H + u] = gf;
    a[3 * H + u] = go;
    cseq[r * H + u] = c;
    hseq[r * H + u] = h;
    hprev[r * H + u] = hp;
}

__global__ void lstm_step_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ dh_next,
                                     const float* __restrict__ act, const float* __restrict__ cseq, float* __restrict__ dc,
                                     float* __restrict__ dz, int batch, int T, int t, int H) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= batch * H) return;
    const int b = e / H, u = e - b * H;
    const int64_t r = (int64_t)b * T + t;
    float dh = dout ? dout[r * H + u] : 0.f;
    if (dh_next) dh += dh_next[(int64_t)b * H + u];
    const float* a = act + r * 4 * H;
    const float gi = a[u], gj = a[H + u], gf = a[2 * H + u], go = a[3 * H + u];
    const float c = cseq[r * H + u];
    const float cp = t > 0 ? cseq[(r - 1) * H + u] : 0.f;
    const float tc = tanhf(c);
    const float d_o = dh * tc;
    const float dcv = dc[e] + dh * go * (1.f - tc * tc);
    float* z = dz + r * 4 * H;
    z[u] = dcv * gj * gi * (1.f - gi);
    z[H + u] = dcv * gi * (1.f - gj * gj);
    z[2 * H + u] = dcv * cp * gf * (1.f - gf);
    z[3 * H + u] = d_o * go * (1.f - go);
    dc[e] = dcv * gf;
}

extern "C" int vl_lstm_step_fwd(const float* gx, const float* gh, float* act, float* cseq, float* hseq, float* hprev, int batch,
                                int T, int t, int H, float forget_bias, vl_stream_t stream) {
    VL_CHECK(gx && act && cseq && hseq && hprev, "vl_lstm_step_fwd: null argument");
    VL_CHECK(batch > 0 && H > 0 && T > 0 && t >= 0 && t < T, "vl_lstm_step_fwd: bad shape");
    VL_CHECK(gh || t == 0, "vl_lstm_step_fwd: gh required for t > 0");
    hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3(ceil_div((int64_t)batch * H, 256)), dim3(256), 0, (hipStream_t)stream, gx, gh,
                       act, cseq, hseq, hprev, batch, T, t, H, forget_bias);
    VL_LAUNCH_CHECK();
    return 0;
}

extern "C" int vl_lstm_step_bwd(const float* dout, const float* dh_next, const float* act, const float* cseq, float* dc,
                                float* dz, int batch, int T, int t, int H, vl_stream_t stream) {
    VL_CHECK(act && cseq && dc && dz, "vl_lstm_step_bwd: null argument");
    VL_CHECK(batch > 0 && H > 0 && T > 0 && t >= 0 && t < T, "vl_lstm_step_bwd: bad shape");
    hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3(ceil_div((int64_t)batch * H, 256)), dim3(256), 0, (hipStream_t)stream, dout,
                       dh_next, act, cseq, dc, dz, batch, T, t, H);
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- persistent per-clip LSTM recurrence ------------------------------------------------------------
// Clips are independent, so the whole T-step recurrence of a clip runs inside ONE workgroup with no
// inter-workgroup traffic: lane u owns hidden unit u (all four gates), h_{t-1} / dz_t are exchanged through
// LDS, and the recurrent weights (kh = kernel[D:], [H][4H]) are streamed from L2 every step (1 MB for H = 256).
// Loads are staged through a register array so a whole batch is in flight (hipcc otherwise waits after every
// load), and every clip starts at a different row so the CUs do not hit one L2 channel in lockstep.
// Replaces T x {vl_gemm(M = clips), vl_lstm_step_*} launches.
template <bool BWD>
__global__ void lstm_seq_kernel(const float* __restrict__ gx, const float* __restrict__ kmat, float* __restrict__ act,
                                float* __restrict__ cseq, float* __restrict__ hseq, float* __restrict__ hprev,
                                const float* __restrict__ dout, float* __restrict__ dz, int T, int H, float forget_bias, int KQ,
                                const float* __restrict__ h0, const float* __restrict__ c0, float* __restrict__ dh0,
                                float* __restrict__ dc0) {
    // blockDim = KQ * HP threads (HP = H rounded up to the wave): thread (u, kq) accumulates the kq-th slice of the recurrent
    // reduction for hidden unit u; the KQ partial sums meet in LDS and the kq = 0 threads do the gate math.  The recurrence is a
    // chain of T dependent steps whose length is set by load latency, not bandwidth (1 MB of weights per step from L2): splitting
    // the reduction four ways cuts the dependent load batches per step from 16 to 4.
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int HP = blockDim.x / KQ;
    const int b = blockIdx.x, u = threadIdx.x % HP, kq = threadIdx.x / HP;
    const bool live = u < H;
    const int uc = live ? u : 0;                       // clamped: idle lanes load valid addresses, results unused
    const int H4 = 4 * H;
    constexpr int KB = 16;
    if (!BWD) {
        float* hs = sm;                                // [H] h_{t-1}
        float* part = sm + H;                          // [KQ][4][H] partial gate pre-activations
        float c = (c0 && live) ? c0[(int64_t)b * H + u] : 0.f;
        for (int i = threadIdx.x; i < H; i += blockDim.x) hs[i] = h0 ? h0[(int64_t)b * H + i] : 0.f;
        __syncthreads();
        const bool rec0 = h0 != nullptr;               // an initial state makes step 0 a full recurrent step
        const int klen = (H + KQ - 1) / KQ, k0 = kq * klen, k1 = min(H, k0 + klen);
        const int nb = (k1 - k0) / KB, rot = (b * 5) % (nb > 0 ? nb : 1);
        for (int t = 0; t < T; ++t) {
            const int64_t r = (int64_t)b * T + t;
            float z[4] = {0.f, 0.f, 0.f, 0.f};
            if (t > 0 || rec0) {
                for (int ib = 0; ib < nb; ++ib) {
                    const int kb = k0 + ((ib + rot) % nb) * KB;
                    float w[KB][4];
#pragma unroll
                    for (int kk = 0; kk < KB; ++kk)
#pragma unroll
                        for (int q = 0; q < 4; ++q) w[kk][q] = kmat[(int64_t)(kb + kk) * H4 + q * H + uc];
#pragma unroll
                    for (int kk = 0; kk < KB; ++kk) {
                        const float hk = hs[kb + kk];
#pragma unroll
                        for (int q = 0; q < 4; ++q) z[q] += hk * w[kk][q];
                    }
                }
                for (int k = k0 + nb * KB; k < k1; ++k) {
                    const float hk = hs[k];
#pragma unroll
                    for (int q = 0; q < 4; ++q) z[q] += hk * kmat[(int64_t)k * H4 + q * H + uc];
                }
                if (live)
#pragma unroll
                    for (int q = 0; q < 4; ++q) part[(kq * 4 + q) * H + u] = z[q];
            }
            const float hp = hs[uc];
            __syncthreads();                           // partial sums complete; everyone has consumed h_{t-1}
            if (live && kq == 0) {
                float zz[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    zz[q] = gx[r * H4 + q * H + u];
                    if (t > 0 || rec0)
                        for (int j = 0; j < KQ; ++j) zz[q] += part[(j * 4 + q) * H + u];    // fixed order: reproducible
                }
                const float gi = sigmoidf_(zz[0]), gj = tanhf(zz[1]), gf = sigmoidf_(zz[2] + forget_bias), go = sigmoidf_(zz[3]);
                c = c * gf + gi * gj;
                const float h = tanhf(c) * go;
                hs[u] = h;
                float* a = act + r * H4 + u;
                a[0] = gi; a[H] = gj; a[2 * H] = gf; a[3 * H] = go;
                cseq[r * H + u] = c;
                hseq[r * H + u] = h;
                hprev[r * H + u] = hp;
            }
            __syncthreads();                           // h_t visible before the next step
        }
    } else {
        // kmat = khT [4H][H]: dh_prev[u] = sum_g dz[g] * khT[g][u]
        float* zs = sm;                                // [4H] dz_t
        float* part = sm + H4;                         // [KQ][H] partial dh
        float dc = 0.f, dh = 0.f;
        const int glen = (H4 + KQ - 1) / KQ, g0 = kq * glen, g1 = min(H4, g0 + glen);
        const int nb = (g1 - g0) / KB, rot = (b * 5) % (nb > 0 ? nb : 1);
        for (int t = T - 1; t >= 0; --t) {
            const int64_t r = (int64_t)b * T + t;
            if (live && kq == 0) {
                const float din = (dout ? dout[r * H + u] : 0.f) + dh;
                const float* a = act + r * H4 + u;
                const float gi = a[0], gj = a[H], gf = a[2 * H], go = a[3 * H];
                const float cc = cseq[r * H + u];
                const float cp = t > 0 ? cseq[(r - 1) * H + u] : (c0 ? c0[(int64_t)b * H + u] : 0.f);
                const float tc = tanhf(cc);
                const float d_o = din * tc;
                const float dcv = dc + din * go * (1.f - tc * tc);
                const float zi = dcv * gj * gi * (1.f - gi), zj = dcv * gi * (1.f - gj * gj);
                const float zf = dcv * cp * gf * (1.f - gf), zo = d_o * go * (1.f - go);
                dc = dcv * gf;
                float* zp = dz + r * H4 + u;
                zp[0] = zi; zp[H] = zj; zp[2 * H] = zf; zp[3 * H] = zo;
                zs[u] = zi; zs[H + u] = zj; zs[2 * H + u] = zf; zs[3 * H + u] = zo;
            }
            __syncthreads();                           // dz_t complete in LDS
            if (t > 0 || dh0) {
                float acc = 0.f;
                for (int ib = 0; ib < nb; ++ib) {
                    const int g = g0 + ((ib + rot) % nb) * KB;
                    float w[KB];
#pragma unroll
                    for (int gg = 0; gg < KB; ++gg) w[gg] = kmat[(int64_t)(g + gg) * H + uc];
#pragma unroll
                    for (int gg = 0; gg < KB; ++gg) acc += zs[g + gg] * w[gg];
                }
                for (int g = g0 + nb * KB; g < g1; ++g) acc += zs[g] * kmat[(int64_t)g * H + uc];
                if (live) part[kq * H + u] = acc;
            }
            __syncthreads();                           // partial dh complete; dz_t consumed before it is overwritten
            if ((t > 0 || dh0) && live && kq == 0) {
                float acc = 0.f;
                for (int j = 0; j < KQ; ++j) acc += part[j * H + u];
                dh = acc;
            }
            // (the next iteration's first barrier separates these reads of `part` from its next writes)
        }
        if (live && kq == 0) {
            if (dh0) dh0[(int64_t)b * H + u] = dh;
            if (dc0) dc0[(int64_t)b * H + u] = dc;
        }
    }
}

// threads per workgroup: H rounded up to the wave, times the reduction split (4 when that fits a 1024-thread workgroup)
static int lstm_seq_hp(int H) { return ((H + 63) / 64) * 64; }
static int lstm_seq_kq(int H) { return lstm_seq_hp(H) * 4 <= 1024 ? 4 : (lstm_seq_hp(H) * 2 <= 1024 ? 2 : 1); }


// Per-clip form: the fallback of vl_lstm_seq_fwd / _bwd (lstm_cluster.hip) for hidden sizes its LDS-resident weight slices do
// not hold (H > 512), and the A/B reference of the cluster form (VL_LSTM_PERCLIP=1).
int vl_lstm_perclip_fwd(const float* gx, const float* kh, const float* h0, const float* c0, float* act, float* cseq, float* hseq,
                        float* hprev, int batch, int T, int H, float forget_bias, hipStream_t stream) {
    const int kq = lstm_seq_kq(H);
    hipLaunchKernelGGL((lstm_seq_kernel<false>), dim3(batch), dim3(lstm_seq_hp(H) * kq), (size_t)(1 + 4 * kq) * H * sizeof(float),
                       stream, gx, kh, act, cseq, hseq, hprev, (const float*)nullptr, (float*)nullptr, T, H, forget_bias, kq, h0, c0,
                       (float*)nullptr, (float*)nullptr);
    VL_LAUNCH_CHECK();
    return 0;
}

int vl_lstm_perclip_bwd(const float* dout, const float* kh_t, const float* act, const float* cseq, const float* c0, float* dz,
                        float* dh0, float* dc0, int batch, int T, int H, hipStream_t stream) {
    const int kq = lstm_seq_kq(H);
    hipLaunchKernelGGL((lstm_seq_kernel<true>), dim3(batch), dim3(lstm_seq_hp(H) * kq), (size_t)(4 + kq) * H * sizeof(float),
                       stream, (const float*)nullptr, kh_t, const_cast<float*>(act), const_cast<float*>(cseq),
                       (float*)nullptr, (float*)nullptr, dout, dz, T, H, 0.f, kq, (const float*)nullptr, c0, dh0, dc0);
    VL_LAUNCH_CHECK();
    return 0;
}

__global__ void transpose2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols, int64_t ld) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int r = by + i, c = bx + threadIdx.x;
        tile[i][threadIdx.x] = (r < rows && c < cols) ? src[(int64_t)r * ld + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int r = bx + i, c = by + threadIdx.x;     // dst is [cols][rows]
        if (r < cols && c < rows) dst[(int64_t)r * rows + c] = tile[threadIdx.x][i];
    }
}

extern "C" int vl_transpose(const float* src, int64_t ld, float* dst, int rows, int cols, vl_stream_t stream) {
    VL_CHECK(src && dst && rows > 0 && cols > 0 && ld >= cols, "vl_transpose: bad argument");
    hipLaunchKernelGGL(transpose2d_kernel, dim3(ceil_div(cols, 32), ceil_div(rows, 32)), dim3(32, 8), 0, (hipStream_t)stream, src,
                       dst, rows, cols, ld);
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- temporal fusion (tf_util.py:4-30) --------------------------------------------------------
__global__ void temporal_fusion_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int batch, int T, int H,
                                           int method) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= batch * H) return;
    const int b = e / H, u = e - b * H;
    const float* p = x + (int64_t)b * T * H + u;
    if (method == 1) {
        y[e] = p[(int64_t)(T - 1) * H];
    } else {
        float a = 0.f;
        for (int t = 0; t < T; ++t) a += p[(int64_t)t * H];
        y[e] = a / (float)T;
    }
}

__global__ void temporal_fusion_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int batch, int T, int H,
                                           int method) {
    const int64_t total = (int64_t)batch * T * H;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int u = (int)(e % H);
        const int t = (int)((e / H) % T);
        const int b = (int)(e / ((int64_t)H * T));
        const float g = dy[(int64_t)b * H + u];
        dx[e] = method == 1 ? (t == T - 1 ? g : 0.f) : g / (float)T;
    }
}

extern "C" int vl_temporal_fusion_fwd(const float* x, float* y, int batch, int T, int H, int method, vl_stream_t stream) {
    VL_CHECK(x && y && batch > 0 && T > 0 && H > 0 && (method == 0 || method == 1), "vl_temporal_fusion_fwd: bad argument");
    hipLaunchKernelGGL(temporal_fusion_fwd_kernel, dim3(ceil_div((int64_t)batch * H, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       y, batch, T, H, method);
    VL_LAUNCH_CHECK();
    return 0;
}

extern "C" int vl_temporal_fusion_bwd(const float* dy, float* dx, int batch, int T, int H, int method, vl_stream_t stream) {
    VL_CHECK(dy && dx && batch > 0 && T > 0 && H > 0 && (method == 0 || method == 1), "vl_temporal_fusion_bwd: bad argument");
    hipLaunchKernelGGL(temporal_fusion_bwd_kernel, dim3(grid_for((int64_t)batch * T * H, 256, 4096)), dim3(256), 0,
                       (hipStream_t)stream, dy, dx, batch, T, H, method);
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- dropout (lstm.py:50-56) ------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ mask, int64_t count,
                                   float keep, uint64_t seed) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = splitmix64(seed ^ splitmix64((uint64_t)e));
        const float uni = (float)(h >> 40) * (1.0f / 16777216.0f);
        const uint8_t m = uni < keep ? 1 : 0;
        mask[e] = m;
        y[e] = m ? x[e] / keep : 0.f;
    }
}

__global__ void dropout_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ mask, float* __restrict__ dx,
                                   int64_t count, float keep) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x)
        dx[e] = mask[e] ? dy[e] / keep : 0.f;
}

extern "C" int vl_dropout_fwd(const float* x, float* y, uint8_t* mask, int64_t count, float keep, uint64_t seed,
                              vl_stream_t stream) {
    VL_CHECK(x && y && mask && count > 0 && keep > 0.f && keep <= 1.f, "vl_dropout_fwd: bad argument");
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(grid_for(count, 256, 4096)), dim3(256), 0, (hipStream_t)stream, x, y, mask, count,
                       keep, seed);
    VL_LAUNCH_CHECK();
    return 0;
}

extern "C" int vl_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t count, float keep, vl_stream_t stream) {
    VL_CHECK(dy && dx && mask && count > 0 && keep > 0.f && keep <= 1.f, "vl_dropout_bwd: bad argument");
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(grid_for(count, 256, 4096)), dim3(256), 0, (hipStream_t)stream, dy, mask, dx, count,
                       keep);
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- softmax cross-entropy, mean over the batch (train.py:120-123) + accuracy (142-149) --------
// One wave per row.  Row losses / hits go to a per-row workspace and are summed in a fixed order by a second one-workgroup
// kernel -> bitwise reproducible at any batch; without a workspace one workgroup walks all rows (small batches only).
__device__ __forceinline__ void softmax_xent_row(const float* __restrict__ z, const int32_t* __restrict__ y, float* __restrict__ dz,
                                                 int C, float gscale, int lane, float& loss, float& hit) {
    float mx = -INFINITY;
    int am = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float v = z[c];
        if (v > mx) {
            mx = v;
            am = c;
        }
    }
    const float gmx = wave_max(mx);
    // first index attaining the maximum
    int cand = (mx == gmx) ? am : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(z[c] - gmx);
    se = wave_sum(se);
    const float lse = logf(se) + gmx;
    float l = 0.f;
    int ymax = -2147483647 - 1, yarg = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const int yv = y[c];
        l += (float)yv * (lse - z[c]);
        if (yv > ymax) {
            ymax = yv;
            yarg = c;
        }
        if (dz) dz[c] = (expf(z[c] - lse) - (float)yv) * gscale;
    }
    l = wave_sum(l);
    int gy = ymax;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) gy = max(gy, __shfl_xor(gy, o, 64));
    int ycand = (ymax == gy) ? yarg : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ycand = min(ycand, __shfl_xor(ycand, o, 64));
    loss = l;
    hit = (cand == ycand) ? 1.f : 0.f;
}

__global__ void softmax_xent_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                    float* __restrict__ dlogits, float* __restrict__ stats, int batch, int C, float gscale) {
    __shared__ float sl[4], sc[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float loss_acc = 0.f, corr_acc = 0.f;
    for (int b = wv; b < batch; b += 4) {
        float l, h;
        softmax_xent_row(logits + (int64_t)b * C, labels + (int64_t)b * C, dlogits ? dlogits + (int64_t)b * C : nullptr, C, gscale,
                         lane, l, h);
        loss_acc += l;
        corr_acc += h;
    }
    if (lane == 0) {
        sl[wv] = loss_acc;
        sc[wv] = corr_acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        stats[0] += sl[0] + sl[1] + sl[2] + sl[3];
        stats[1] += sc[0] + sc[1] + sc[2] + sc[3];
    }
}

__global__ void softmax_xent_rows_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                         float* __restrict__ dlogits, float* __restrict__ rows, int batch, int C, float gscale) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= batch) return;
    float l, h;
    softmax_xent_row(logits + (int64_t)b * C, labels + (int64_t)b * C, dlogits ? dlogits + (int64_t)b * C : nullptr, C, gscale, lane,
                     l, h);
    if (lane == 0) {
        rows[b] = l;
        rows[batch + b] = h;
    }
}

// rows[0..batch) losses, rows[batch..2 batch) hits -> stats += their sums; thread t adds rows t, t+256, ... then a fixed tree.
__global__ void softmax_xent_sum_kernel(const float* __restrict__ rows, float* __restrict__ stats, int batch) {
    __shared__ float sl[256], sc[256];
    float l = 0.f, h = 0.f;
    for (int b = threadIdx.x; b < batch; b += 256) {
        l += rows[b];
        h += rows[batch + b];
    }
    sl[threadIdx.x] = l;
    sc[threadIdx.x] = h;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sl[threadIdx.x] += sl[threadIdx.x + s];
            sc[threadIdx.x] += sc[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        stats[0] += sl[0];
        stats[1] += sc[0];
    }
}

extern "C" int vl_softmax_xent(const float* logits, const int32_t* labels, float* dlogits, float* stats, float* rows, int batch,
                               int classes, float grad_scale, vl_stream_t stream) {
    VL_CHECK(logits && labels && stats && batch > 0 && classes > 0, "vl_softmax_xent: bad argument");
    if (!rows) {
        hipLaunchKernelGGL(softmax_xent_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, dlogits, stats, batch,
                           classes, grad_scale);
        VL_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(softmax_xent_rows_kernel, dim3((batch + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, labels, dlogits,
                       rows, batch, classes, grad_scale);
    VL_LAUNCH_CHECK();
    hipLaunchKernelGGL(softmax_xent_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rows, stats, batch);
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- global norm + SGD / Adam (train.py:199-222) ----------------------------------------------
__global__ void sumsq_stage1(const float* __restrict__ g, int64_t count, float* __restrict__ ws) {
    __shared__ float sm[4];
    float acc = 0.f;
    const int64_t nvec = count / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const bool aligned = ((uintptr_t)g & 15) == 0;
    if (aligned) {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
            const float4 v = g4[i];
            acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int64_t i = nvec * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
            acc += g[i] * g[i];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
            acc += g[i] * g[i];
    }
    acc = block_sum_256(acc, sm);
    if (threadIdx.x == 0) ws[blockIdx.x] = acc;
}

__global__ void sumsq_stage2(const float* __restrict__ ws, int nblocks, float* __restrict__ out, int accumulate) {
    __shared__ float sm[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += ws[i];
    acc = block_sum_256(acc, sm);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + acc : acc;
}

extern "C" int vl_sumsq(const float* g, int64_t count, float* out, float* ws, int accumulate, vl_stream_t stream) {
    VL_CHECK(g && out && ws && count > 0, "vl_sumsq: bad argument");
    const int blocks = grid_for(count / 4 + 1, 256, 1024);
    hipLaunchKernelGGL(sumsq_stage1, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, count, ws);
    VL_LAUNCH_CHECK();
    hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(256), 0, (hipStream_t)stream, ws, blocks, out, accumulate);
    VL_LAUNCH_CHECK();
    return 0;
}

__device__ __forceinline__ float clip_scale(float clip_norm, const float* sumsq, float gscale) {
    if (clip_norm <= 0.f || !sumsq) return gscale;
    const float norm = gscale * sqrtf(sumsq[0]);
    return gscale * clip_norm / fmaxf(norm, clip_norm);
}

__global__ void sgd_apply_kernel(float* __restrict__ w, const float* __restrict__ g, int64_t count, float lr, float clip_norm,
                                 const float* __restrict__ sumsq, float gscale, const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;                                        // the step's results are invalid (vl_status_or): no update
    const float a = lr * clip_scale(clip_norm, sumsq, gscale);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        w[i] -= a * g[i];
}

__global__ void adam_apply_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                  int64_t count, float lr_t, float clip_norm, const float* __restrict__ sumsq, float gscale,
                                  const uint32_t* __restrict__ skip) {
    if (skip && *skip) return;
    const float sc = clip_scale(clip_norm, sumsq, gscale);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * sc;
        const float mi = 0.9f * m[i] + 0.1f * gi;
        const float vi = 0.999f * v[i] + 0.001f * gi * gi;
        m[i] = mi;
        v[i] = vi;
        w[i] -= lr_t * mi / (sqrtf(vi) + 1e-8f);
    }
}

// *dst |= first word of an LSTM cluster workspace (its sticky time-out word, lstm_cluster.hip): the optimizer's `skip` word of a step
__global__ void status_or_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, int init) {
    const uint32_t old = init ? 0u : *dst;
    *dst = (*src != 0u) ? 1u : old;
}
extern "C" int vl_status_or(uint32_t* dst, const void* lstm_ws, int init, vl_stream_t stream) {
    VL_CHECK(dst && lstm_ws, "vl_status_or: null argument");
    hipLaunchKernelGGL(status_or_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst, (const uint32_t*)lstm_ws, init);
    VL_LAUNCH_CHECK();
    return 0;
}

extern "C" int vl_sgd_apply(float* w, const float* g, int64_t count, float lr, float clip_norm, const float* sumsq, float gscale,
                            const uint32_t* skip, vl_stream_t stream) {
    VL_CHECK(w && g && count > 0, "vl_sgd_apply: bad argument");
    hipLaunchKernelGGL(sgd_apply_kernel, dim3(grid_for(count, 256, 4096)), dim3(256), 0, (hipStream_t)stream, w, g, count, lr,
                       clip_norm, sumsq, gscale, skip);
    VL_LAUNCH_CHECK();
    return 0;
}

extern "C" int vl_adam_apply(float* w, const float* g, float* m, float* v, int64_t count, float lr, float clip_norm,
                             const float* sumsq, float gscale, int step, const uint32_t* skip, vl_stream_t stream) {
    VL_CHECK(w && g && m && v && count > 0 && step >= 1, "vl_adam_apply: bad argument");
    const double b1t = 1.0 - pow(0.9, (double)step), b2t = 1.0 - pow(0.999, (double)step);
    const float lr_t = (float)(lr * sqrt(b2t) / b1t);
    hipLaunchKernelGGL(adam_apply_kernel, dim3(grid_for(count, 256, 4096)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, count,
                       lr_t, clip_norm, sumsq, gscale, skip);
    VL_LAUNCH_CHECK();
    return 0;
}

__global__ void fill_kernel(float* __restrict__ p, int64_t count, float value) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) p[i] = value;
}

// ReluGrad in place (tf.nn.relu's gradient: 0 where the forward output is <= 0): d[i] = y[i] > 0 ? d[i] : 0
__global__ void relu_grad_kernel(float* __restrict__ d, const float* __restrict__ y, int64_t count) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = y[i] > 0.f ? d[i] : 0.f;
}

extern "C" int vl_relu_grad(float* d, const float* y, int64_t count, vl_stream_t stream) {
    VL_CHECK(d && y && count > 0, "vl_relu_grad: bad argument");
    hipLaunchKernelGGL(relu_grad_kernel, dim3(grid_for(count, 256, 4096)), dim3(256), 0, (hipStream_t)stream, d, y, count);
    VL_LAUNCH_CHECK();
    return 0;
}

extern "C" int vl_fill(float* p, int64_t count, float value, vl_stream_t stream) {
    VL_CHECK(p && count > 0, "vl_fill: bad argument");
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(count, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, count, value);
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- tensor-list plumbing of multi-input pipelines (tf_util.py:99-192) ----------------------------------------------------------
// concat / vec_seq_concat / ibias / replicate_auxilliary_tensor are all strided block copies; avg / maximum are elementwise.
__global__ void copy2d_kernel(const float* __restrict__ src, int64_t src_ld, float* __restrict__ dst, int64_t dst_ld, int rows,
                              int cols) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / cols, c = e - r * cols;
        dst[r * dst_ld + c] = src[r * src_ld + c];
    }
}

extern "C" int vl_copy2d(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int rows, int cols, vl_stream_t stream) {
    VL_CHECK(src && dst && rows > 0 && cols > 0 && src_ld >= 0 && dst_ld >= cols, "vl_copy2d: bad argument");
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for((int64_t)rows * cols, 256, 4096)), dim3(256), 0, (hipStream_t)stream, src, src_ld, dst,
                       dst_ld, rows, cols);
    VL_LAUNCH_CHECK();
    return 0;
}

// op 0: out = a + b;  1: out = (a + b) / 2 (tf.reduce_mean over two inputs);  2: out = max(a, b) (tf.reduce_max)
__global__ void eltwise2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int64_t count, int op) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x) {
        const float x = a[e], y = b[e];
        out[e] = op == 0 ? x + y : (op == 1 ? (x + y) * 0.5f : fmaxf(x, y));
    }
}

extern "C" int vl_eltwise2(const float* a, const float* b, float* out, int64_t count, int op, vl_stream_t stream) {
    VL_CHECK(a && b && out && count > 0 && op >= 0 && op <= 2, "vl_eltwise2: bad argument");
    hipLaunchKernelGGL(eltwise2_kernel, dim3(grid_for(count, 256, 4096)), dim3(256), 0, (hipStream_t)stream, a, b, out, count, op);
    VL_LAUNCH_CHECK();
    return 0;
}

// gradient of max(a, b) as tf.reduce_max registers it (_MinOrMaxGrad): the inputs equal to the maximum share the gradient evenly
// (two ReLU outputs that are both 0 tie all the time)
__global__ void max2_grad_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ d,
                                 float* __restrict__ da, float* __restrict__ db, int64_t count) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x) {
        const float x = a[e], y = b[e], g = d[e];
        da[e] = x > y ? g : (x == y ? 0.5f * g : 0.f);
        db[e] = y > x ? g : (x == y ? 0.5f * g : 0.f);
    }
}

extern "C" int vl_max2_grad(const float* a, const float* b, const float* d, float* da, float* db, int64_t count, vl_stream_t stream) {
    VL_CHECK(a && b && d && da && db && count > 0, "vl_max2_grad: bad argument");
    hipLaunchKernelGGL(max2_grad_kernel, dim3(grid_for(count, 256, 4096)), dim3(256), 0, (hipStream_t)stream, a, b, d, da, db, count);
    VL_LAUNCH_CHECK();
    return 0;
}

// apply_tensor_list_fusion avg | maximum over a LIST of equally shaped tensors (tf.reduce_mean / tf.reduce_max over axis 0 of the
// stacked list, tf_util.py:142-145) and its gradient.  The pointer lists are host arrays, passed to the kernel by value.
static constexpr int FUSE_MAX = 8;
struct FuseList { const float* in[FUSE_MAX]; float* din[FUSE_MAX]; int n; };

__global__ void fuse_n_kernel(const FuseList l, float* __restrict__ out, int64_t count, int op) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x) {
        float acc = l.in[0][e];
        for (int i = 1; i < l.n; ++i) acc = op == 0 ? acc + l.in[i][e] : fmaxf(acc, l.in[i][e]);
        out[e] = op == 0 ? acc / (float)l.n : acc;
    }
}

// avg: every input gets d / n; maximum: the inputs equal to the maximum share d evenly (tf _MinOrMaxGrad); din[i] == null is skipped
__global__ void fuse_n_grad_kernel(const FuseList l, const float* __restrict__ d, int64_t count, int op) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x) {
        const float g = d[e];
        if (op == 0) {
            for (int i = 0; i < l.n; ++i)
                if (l.din[i]) l.din[i][e] = g / (float)l.n;
        } else {
            float m = l.in[0][e];
            for (int i = 1; i < l.n; ++i) m = fmaxf(m, l.in[i][e]);
            int hits = 0;
            for (int i = 0; i < l.n; ++i) hits += l.in[i][e] == m;
            for (int i = 0; i < l.n; ++i)
                if (l.din[i]) l.din[i][e] = l.in[i][e] == m ? g / (float)hits : 0.f;
        }
    }
}

extern "C" int vl_fuse_n(const float* const* ins, int n, float* out, int64_t count, int op, vl_stream_t stream) {
    VL_CHECK(ins && out && n >= 1 && n <= FUSE_MAX && count > 0 && (op == 0 || op == 1), "vl_fuse_n: bad argument (1..8 inputs, op 0 avg | 1 maximum)");
    FuseList l = {};
    l.n = n;
    for (int i = 0; i < n; ++i) {
        VL_CHECK(ins[i], "vl_fuse_n: null input");
        l.in[i] = ins[i];
    }
    hipLaunchKernelGGL(fuse_n_kernel, dim3(grid_for(count, 256, 4096)), dim3(256), 0, (hipStream_t)stream, l, out, count, op);
    VL_LAUNCH_CHECK();
    return 0;
}

extern "C" int vl_fuse_n_grad(const float* const* ins, int n, const float* d, float* const* dins, int64_t count, int op, vl_stream_t stream) {
    VL_CHECK(d && dins && n >= 1 && n <= FUSE_MAX && count > 0 && (op == 0 || op == 1), "vl_fuse_n_grad: bad argument");
    VL_CHECK(op == 0 || ins, "vl_fuse_n_grad: the maximum needs the inputs");
    FuseList l = {};
    l.n = n;
    for (int i = 0; i < n; ++i) {
        if (op == 1) {
            VL_CHECK(ins[i], "vl_fuse_n_grad: null input");
            l.in[i] = ins[i];
        }
        l.din[i] = dins[i];
    }
    hipLaunchKernelGGL(fuse_n_grad_kernel, dim3(grid_for(count, 256, 4096)), dim3(256), 0, (hipStream_t)stream, l, d, count, op);
    VL_LAUNCH_CHECK();
    return 0;
}

// Packed-bf16 convolutions for gfx950: the "bf16" conv arithmetic (vl_set_conv_math mode 1: operands rounded to bf16, fp32
// accumulation) as a bf16 PATH -- operands are bf16 in HBM and in LDS, fetched 16 bytes per lane, and reach the matrix pipe
// (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate) without a single vector-ALU instruction in the loop.
//
// Layout "c8": an activation tensor [n][c][h][w] is stored [n][cb = ceil(c / 8)][h + 2 halo][w + 2 halo][8] bf16 -- 8 consecutive
// channels of one pixel are one 16-byte chunk, zero halo as in the fp32 padded layout (vltf.h, vl_conv_set_halo), zero beyond c.
// A chunk IS the MFMA operand of its pixel for 8 reduction positions, so:
//   * forward / dgrad: reduction order (channel block, kh, kw, channel in block) = "taps" of 8 positions.  One
//     `buffer_load_dwordx4 ... lds` moves one tap of 64 pixels (per-lane pixel offset decoded once per tile, the tap's offset is a
//     scalar) to 1 KB of LDS, lane l at M0 + 16 l = the [tap][pixel][16 B] image one ds_read_b128 per lane reads back as an operand
//     (conflict free: each 16-lane service group of ds_read_b128 covers 16 different chunks of a 512-byte run).  Weights are packed
//     once per step into [group][tap][cout][8] and fetched the same way.
//   * wgrad reduces over PIXELS, the strided index of both operands; the LDS image [pixel][chunk] is read column-wise with
//     ds_read_b64_tr_b16 (the hardware transpose read of gfx950), see wgrad_c8_kernel.
//   * conv1 (11 x 11 / 4 over 3 channels) runs the same two kernels as the equivalent 3 x 3 stride-1 layer over its space-to-depth
//     input (vl_s2d_*), dense products (fc6) run the wgrad kernel on reduction-major operands (vl_pack_kc8 / vl_gemm_kc8).
// Producers write the c8 copies (vl_pack_c8 as a stand-alone producer; the fused forms live with the producers: the conv epilogues
// here, vl_lrn_pool_fwd_c8 / vl_pool_lrn_bwd_c8 in pointwise.hip, vl_input_prep_u8_s2d below).  DESIGN.md 4.7 has the measurements.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "conv_desc.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) int* const_int_ptr;
__device__ __forceinline__ const_int_ptr as_const(const int* p) { return (const_int_ptr)(uintptr_t)p; }

static constexpr int64_t MAX_BUF_BYTES = 0xE0000000ll;

__device__ __forceinline__ i32x4 rsrc_words(const void* base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    return i32x4{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)(uint32_t)bytes, 0x00020000};
}
// LDS-DMA, 16 bytes per lane: lane l lands at M0 + 16 l (inline asm on purpose, see mfma_gemm.hip lds_dma_row)
__device__ __forceinline__ void lds_dma16(i32x4 rs, uint32_t lds_byte_addr, uint32_t voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_byte_addr), "v"(voff), "s"(rs), "s"(soff)
                 : "m0");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
}
__device__ __forceinline__ f32x16 mfma_bf16(const i32x4& a, const i32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {   // round to nearest even (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
template <int I> struct IntC { static constexpr int value = I; };
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {     // f(IntC<B>) ... f(IntC<E-1>): indices that must be compile-time constants
    if constexpr (B < E) {
        f(IntC<B>{});
        static_for<B + 1, E>(f);
    }
}
__device__ __forceinline__ int xcd_swizzle(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7, k = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// ---- stand-alone producers -----------------------------------------------------------------------------------------------
// fp32 NCHW (halo hin) -> bf16 c8 (halo hout), interiors only; one thread per (n, cb, h, w)
__global__ void pack_c8_kernel(const float* __restrict__ x, uint4* __restrict__ xb, int C, int H, int W, int hin, int hout, int CB,
                               int64_t total, FastDiv dW, FastDiv dH, FastDiv dCB) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const uint32_t i = (uint32_t)idx;
    const uint32_t r1 = fd_div(i, dW), w = i - r1 * dW.d;
    const uint32_t r2 = fd_div(r1, dH), h = r1 - r2 * dH.d;
    const uint32_t n = fd_div(r2, dCB), cb = r2 - n * dCB.d;
    const int Wi = W + 2 * hin, Hi = H + 2 * hin, Wo = W + 2 * hout, Ho = H + 2 * hout;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cb * 8 + j;
        v[j] = c < C ? x[(((int64_t)n * C + c) * Hi + h + hin) * Wi + w + hin] : 0.f;
    }
    uint4 o;
    o.x = pack_bf16(v[0], v[1]);
    o.y = pack_bf16(v[2], v[3]);
    o.z = pack_bf16(v[4], v[5]);
    o.w = pack_bf16(v[6], v[7]);
    xb[(((int64_t)n * CB + cb) * Ho + h + hout) * Wo + w + hout] = o;
}

extern "C" size_t vl_c8_bytes(int n, int c, int h, int w, int halo) {
    return (size_t)n * ((c + 7) / 8) * (h + 2 * halo) * (w + 2 * halo) * 16;
}

extern "C" int vl_pack_c8(const float* x, void* xb, int n, int c, int h, int w, int x_halo, int xb_halo, vl_stream_t stream) {
    VL_CHECK(x && xb && n > 0 && c > 0 && h > 0 && w > 0 && x_halo >= 0 && xb_halo >= 0, "vl_pack_c8: bad argument");
    const int CB = (c + 7) / 8;
    const int64_t total = (int64_t)n * CB * h * w;
    VL_CHECK(total < (1ll << 31), "vl_pack_c8: tensor too large");
    hipLaunchKernelGGL(pack_c8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, (uint4*)xb, c, h, w, x_halo,
                       xb_halo, CB, total, make_fastdiv(w), make_fastdiv(h), make_fastdiv(CB));
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- reduction taps and packed weights -------------------------------------------------------------------------------------
#ifndef C8_KT
#define C8_KT 4
#endif
static constexpr int KT = C8_KT;   // taps (8 reduction positions each; 4 taps = two MFMA steps) per pipeline stage of conv_c8_kernel
// (round 3: -DC8_KT=8 -DC8_NBUF=2 with 128-pixel tiles -- 16 MFMAs per wave between barriers, two workgroups per CU at 64 KB -- was
// measured: the conv stack of the benchmark step 5.24 ms against 4.60 with 4-tap stages, three of them in the ring, 256-pixel tiles;
// the barrier count is not what holds these kernels back, the fetch depth and the tile's operand reuse are)
#ifndef C8_EXP
#define C8_EXP 0      // limit experiments on conv_c8_kernel (tools/c8_limits.sh; results are garbage): 1 no fetches inside the loop,
#endif                // 2 operands read from LDS once, 4 no MFMAs, 8 no epilogue -- never set in a product build
#ifndef C8_NBUF
#define C8_NBUF 0     // 0: by tile (C8Cfg)
#endif
#ifndef C8_WG_NBUF
#define C8_WG_NBUF 3
#endif

// taps of one group: t = (cb * kh + ky) * kw + kx over the group's cbg channel blocks, padded to a multiple of KT
static int c8_taps(int cg, int kh, int kw) { return ((cg + 7) / 8) * kh * kw; }
static int c8_taps_padded(int cg, int kh, int kw) { return (c8_taps(cg, kh, kw) + KT - 1) / KT * KT; }
static int c8_cop(int cog) { return (cog + 127) / 128 * 128; }   // channel pitch of the packed weights: a 96-wide tile fetches 128 channels

// out[g][t][co < CoP][8]: forward  (bwd = 0): W[ky][kx][cb * 8 + j][g * cog + co]              (reduce over input channels)
//                         dgrad    (bwd = 1): W[kh-1-ky][kw-1-kx][co][g * cog + cb * 8 + j]     (reduce over output channels; "co" = ci)
__global__ void pack_w_c8_kernel(const float* __restrict__ w, uint4* __restrict__ out, int kh, int kw, int cig, int cog, int cout,
                                 int bwd, int ntaps, int ntaps_p, int CoP) {
    const int g = blockIdx.y;
    const int rows = bwd ? cig : cog, red = bwd ? cog : cig;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < ntaps_p * CoP; idx += gridDim.x * blockDim.x) {
        const int co = idx % CoP, t = idx / CoP;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (t < ntaps && co < rows) {
            const int kx = t % kw, ky = (t / kw) % kh, cb = t / (kw * kh);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = cb * 8 + j;
                if (c < red)
                    v[j] = bwd ? w[((int64_t)((kh - 1 - ky) * kw + (kw - 1 - kx)) * cig + co) * cout + g * cog + c]
                               : w[((int64_t)(ky * kw + kx) * cig + c) * cout + g * cog + co];
            }
        }
        uint4 o;
        o.x = pack_bf16(v[0], v[1]);
        o.y = pack_bf16(v[2], v[3]);
        o.z = pack_bf16(v[4], v[5]);
        o.w = pack_bf16(v[6], v[7]);
        out[((int64_t)g * ntaps_p + t) * CoP + co] = o;
    }
}

extern "C" size_t vl_conv_c8_w_bytes(const vl_conv_desc* d, int bwd) {
    if (!d) return 0;
    const int red = bwd ? d->cog : d->cig, rows = bwd ? d->cig : d->cog;
    return (size_t)d->groups * c8_taps_padded(red, d->kh, d->kw) * c8_cop(rows) * 16;
}

extern "C" int vl_conv_c8_pack_w(const vl_conv_desc* d, const float* w_hwio, void* wb, int bwd, vl_stream_t stream) {
    VL_CHECK(d && w_hwio && wb, "vl_conv_c8_pack_w: bad argument");
    const int red = bwd ? d->cog : d->cig, rows = bwd ? d->cig : d->cog;
    const int nt = c8_taps(red, d->kh, d->kw), ntp = c8_taps_padded(red, d->kh, d->kw), CoP = c8_cop(rows);
    hipLaunchKernelGGL(pack_w_c8_kernel, dim3((ntp * CoP + 255) / 256, d->groups), dim3(256), 0, (hipStream_t)stream, w_hwio, (uint4*)wb,
                       d->kh, d->kw, d->cig, d->cog, d->cout, bwd, nt, ntp, CoP);
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- forward / dgrad ---------------------------------------------------------------------------------------------------------
// out[co][pixel] = sum over taps W[tap][co][8] . X[tap][pixel][8]; MFMA A = weights (rows = co), B = pixels (columns = lanes), so an
// accumulator register q of lane l is (co = 8 (q >> 2) + 4 (l >> 5) + (q & 3), pixel = l & 31).
// Workgroup = WP x WQ waves, each TP x TQ blocks of 32 pixels x 32 channels: P = 32 WP TP pixels x Q = 32 WQ TQ channels.
// Pipeline: ring of NBUF stages of KT taps; per stage every wave issues the same F fetches (ids wrap: surplus fetches repeat a
// piece, same bytes to the same place), one barrier per stage, `s_waitcnt vmcnt` counted so that NBUF - 2 stages stay in flight.
struct C8ConvArgs {
    const char* x;          // gathered operand, c8
    int64_t x_img;          // bytes per image
    int64_t x_total;        // bytes in the tensor
    int64_t x_grp;          // bytes between the first blocks of consecutive groups
    int x_row;              // bytes per row of a plane (Wp * 16)
    int stride;
    int OW, OHW, M;         // output pixels per row / image / in all
    FastDiv dOHW, dOW;
    const int* toff;        // byte offset of tap t within an image's group (padded taps: 0)
    int nstages;
    const char* w;          // packed weights [g][nstages * KT][CoP][8]
    int CoP, Cog, Ctot;     // channel pitch of w; output channels per group / in all
    const float* bias;      // [Ctot] or null
    int relu;
    float* y;               // fp32 NCHW output with halo (null: not written)
    const float* mask;      // fp32, y's layout: out = mask > 0 ? out : 0 (fused ReluGrad), or null
    int y_halo, y_w;        // y: plane width (OW + 2 halo)
    int y_wide;             // y is dense (no halo) and unmasked: staged through LDS and stored 16 bytes per lane
    int64_t y_plane;        // floats per plane
    char* yb;               // bf16 c8 output with halo (null: not written)
    int yb_halo, yb_row;    // bytes per row
    int64_t yb_plane;       // bytes per plane
    int yb_cb;              // channel blocks per image
    const char* maskb;      // bf16 c8 ReluGrad mask (the producing layer's packed output), or null
    int mb_halo, mb_row;
    int64_t mb_plane;
    int sched;              // 1: the second half of the waves issues its stage fetches after its first MFMA group (stagger)
};

template <int WP, int WQ, int TP, int TQ>
struct C8Cfg {
    static constexpr int NW = WP * WQ, NT = 64 * NW, P = 32 * WP * TP, Q = 32 * WQ * TQ;
    static constexpr int QF = (Q + 63) / 64 * 64;                          // channels fetched per tap: whole 64-lane fetches (96 -> 128)
    static constexpr int PIX_BYTES = KT * P * 16, W_BYTES = KT * QF * 16, SLOT = PIX_BYTES + W_BYTES;
    // ring depth: 3 stages where that lets TWO workgroups share a CU's 160 KB (measured: conv3 forward 0.42 -> 0.32 ms, one
    // workgroup's prologue / epilogue behind the other's loop), else 4
    static constexpr int NBUF = C8_NBUF ? C8_NBUF : (3 * SLOT <= 80 * 1024 ? 3 : 4);
    static constexpr int NPI = KT * P / 64, NWI = KT * QF / 64;            // 1 KB fetches per stage: pixels, weights
    static constexpr int FP = (NPI + NW - 1) / NW, FW = (NWI + NW - 1) / NW, F = FP + FW;
    static constexpr int STG = 32 * TQ * 36;                               // floats per wave of the wide-store staging ([channel][32 pixels + 4])
    static constexpr int RING_BYTES = NBUF * SLOT > NW * STG * 4 ? NBUF * SLOT : NW * STG * 4;   // ring, later the staging
    static constexpr size_t LDS_BYTES = (size_t)RING_BYTES + QF * 4;                       // + the tile's bias vector
    static constexpr int WPS = (2 * LDS_BYTES <= 160 * 1024 ? 2 : 1) * NW / 4;             // waves per SIMD the registers must allow
    static_assert(P % 64 == 0, "whole 64-lane fetches");
};

// EPI: what the epilogue writes, at COMPILE time -- with run-time flags every one of a lane's 64-96 outputs paid five scalar branches
// (conv1 forward: 12 us of epilogue per 10 us tile).  0 = by the arguments (any combination; tests), 1 = dense fp32 y, 16-byte
// stores, 2 = packed yb, 3 = packed yb under the packed ReluGrad mask, 4 = fp32 y with a halo.  ReLU is a floor (0 or -inf): no branch.
template <int WP, int WQ, int TP, int TQ, int EPI>
__global__ __launch_bounds__(64 * WP * WQ, (C8Cfg<WP, WQ, TP, TQ>::WPS)) void conv_c8_kernel(const C8ConvArgs a, int tiles_q) {
    using C = C8Cfg<WP, WQ, TP, TQ>;
    constexpr int NW = C::NW, P = C::P, Q = C::Q, QF = C::QF, NBUF = C::NBUF, SLOT = C::SLOT, FP = C::FP, FW = C::FW, F = C::F;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tq = bid % tiles_q, tp = bid / tiles_q, g = blockIdx.y;
    const int p0 = tp * P, q0 = tq * Q;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wp = wave % WP, wq = wave / WP;
    const int nstages = a.nstages;

    // ---- fetch plan ----
    const int n0 = (int)fd_div((uint32_t)p0, a.dOHW);
    const int64_t xoff = (int64_t)n0 * a.x_img + (int64_t)g * a.x_grp;
    int64_t xbytes = a.x_total - xoff;
    if (xbytes > MAX_BUF_BYTES) xbytes = MAX_BUF_BYTES;
    const i32x4 rs_x = rsrc_words(a.x + xoff, xbytes);
    const int64_t wgrp = (int64_t)nstages * KT * a.CoP * 16;
    const i32x4 rs_w = rsrc_words(a.w + (int64_t)g * wgrp, wgrp);
    uint32_t voff_p[FP], lds_p[FP], voff_w[FW], lds_w[FW];
    int tap_p[FP], tap_w[FW];
#pragma unroll
    for (int j = 0; j < FP; ++j) {
        const int id = (wave + j * NW) % C::NPI, tap = id / (P / 64), pc = id % (P / 64);
        int m = p0 + pc * 64 + lane;
        m = m < a.M ? m : a.M - 1;
        const uint32_t n = fd_div((uint32_t)m, a.dOHW), r = (uint32_t)m - n * a.dOHW.d;
        const uint32_t oh = fd_div(r, a.dOW), ow = r - oh * a.dOW.d;
        voff_p[j] = (uint32_t)((int64_t)(n - n0) * a.x_img) + oh * a.stride * a.x_row + ow * a.stride * 16;
        tap_p[j] = tap;
        lds_p[j] = tap * P * 16 + pc * 1024;
    }
#pragma unroll
    for (int j = 0; j < FW; ++j) {
        const int id = (wave + j * NW) % C::NWI, tap = id / (QF / 64), qc = id % (QF / 64);
        voff_w[j] = (uint32_t)((q0 + qc * 64 + lane) * 16);
        tap_w[j] = tap;
        lds_w[j] = C::PIX_BYTES + tap * QF * 16 + qc * 1024;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;   // LDS byte address of the ring
    // the tile's bias vector -> LDS (read back in the epilogue: a global load between two stores would make the wave wait for every
    // store in flight, one counter orders both); visible after the loop's first barrier
    float* bias_l = reinterpret_cast<float*>(lds + C::RING_BYTES);
    if ((int)threadIdx.x < QF) {
        const int cg = q0 + threadIdx.x;
        bias_l[threadIdx.x] = (a.bias && cg < a.Cog) ? a.bias[g * a.Cog + cg] : 0.f;
    }
    // the tap offsets of the NEXT stage to be issued are fetched (scalar loads) one stage ahead: read at the point of use, each
    // would put a scalar-memory round trip between the barrier and the fetch it feeds (the table has KT spare entries)
    int toff_next[FP];
#pragma unroll
    for (int j = 0; j < FP; ++j) toff_next[j] = as_const(a.toff)[tap_p[j]];
    auto issue = [&](int st) {
        const uint32_t slot = lds0 + (uint32_t)(st % NBUF) * SLOT;
#pragma unroll
        for (int j = 0; j < FP; ++j) lds_dma16(rs_x, slot + lds_p[j], voff_p[j], toff_next[j]);
#pragma unroll
        for (int j = 0; j < FW; ++j) lds_dma16(rs_w, slot + lds_w[j], voff_w[j], (st * KT + tap_w[j]) * a.CoP * 16);
#pragma unroll
        for (int j = 0; j < FP; ++j) toff_next[j] = as_const(a.toff)[(st + 1) * KT + tap_p[j]];
    };

    f32x16 acc[TQ][TP];
#pragma unroll
    for (int j = 0; j < TQ; ++j)
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][i][q] = 0.f;

    for (int st = 0; st < NBUF - 1 && st < nstages; ++st) issue(st);
#if C8_EXP & 2
    i32x4 exp_bp[KT / 2][TP], exp_aq[KT / 2][TQ];
#endif
    // Stagger (MI355X guide, "two waves per SIMD", item 9): all waves run the same program with one barrier per stage, so they reach
    // their fetch issue (3 LDS-DMA pieces, ~100 cycles each inside such a phase), their LDS read burst and their MFMAs together.  The
    // second-dispatched half (the SIMD partners of waves 0 .. NW/2 - 1) issues its fetches after its first MFMA group instead: the
    // partners' MFMAs run beside them.  Measured (tools/c8_probe.py, 1024 frames): conv2 fwd 0.529 -> 0.503 ms, conv3 fwd 0.279 ->
    // 0.273, conv3 dgrad 0.297 -> 0.284; the 4-wave tiles (conv4) unchanged.  A static s_setprio 1 for that half on top was mixed
    // (conv2 fwd 0.497 but conv4 +3..9 %) and is not used; the FIRST half issuing at the end of its stage instead of its start: no
    // difference; a stage's fetches issued ONE AT A TIME between the wave's MFMAs (piece p behind MFMA (p + 1) NM / (F + 1), pinned
    // by sched_barriers; 102 instead of 88 VGPRs): within +-1 % on every launch.  VL_C8_SCHED=0 runs the lockstep order (A/B).
    // What the loop's time is made of (C8_EXP builds, tools/c8_limits.sh; conv3 forward, 1024 frames, 0.267 ms): MFMAs + barriers
    // alone 0.183 (1675 TFLOP/s: the ceiling of this tiling at the clock the chip holds), + the LDS operand reads 0.027, + the fetches
    // 0.034, + the epilogue 0.023 -- the parts ADD; fetches + reads without MFMAs take 0.125 (12.8 GB through the LDS arrays = 0.098 ms
    // at 256 B/clk/CU), so the LDS array is busy for a third of the kernel, in bursts the MFMAs of all waves wait behind.
    const bool late = NW >= 8 && a.sched != 0 && wave >= NW / 2;       // (uniform; 4-wave tiles: no effect measured, not applied)
    const uint32_t rd_p = (uint32_t)((lane >> 5) * P * 16 + (wp * TP * 32 + (lane & 31)) * 16);
    const uint32_t rd_w = (uint32_t)(C::PIX_BYTES + (lane >> 5) * QF * 16 + (wq * TQ * 32 + (lane & 31)) * 16);
    for (int st = 0; st < nstages; ++st) {
        // stage st has landed when at most the fetches of the later stages already issued are outstanding
        const int later = (nstages - 1 - st) < (NBUF - 2) ? (nstages - 1 - st) : (NBUF - 2);
        if (later >= 2)
            wait_vm<2 * F>();
        else if (later == 1)
            wait_vm<F>();
        else
            wait_vm<0>();
        __syncthreads();   // every wave's pieces of stage st are in LDS; every wave is done reading stage st - 1
        if (!(C8_EXP & 1) && !late && st + NBUF - 1 < nstages) issue(st + NBUF - 1);
        const char* slot = lds + (((C8_EXP & 2) ? 0 : st) % NBUF) * SLOT;
#pragma unroll
        for (int kk = 0; kk < KT / 2; ++kk) {
            if (!(C8_EXP & 1) && kk == 1 && late && st + NBUF - 1 < nstages) issue(st + NBUF - 1);
            i32x4 bp[TP], aq[TQ];
            if (!(C8_EXP & 2) || st == 0) {
#pragma unroll
                for (int i = 0; i < TP; ++i) bp[i] = *reinterpret_cast<const i32x4*>(slot + rd_p + kk * 2 * P * 16 + i * 512);
#pragma unroll
                for (int j = 0; j < TQ; ++j) aq[j] = *reinterpret_cast<const i32x4*>(slot + rd_w + kk * 2 * QF * 16 + j * 512);
#if C8_EXP & 2
#pragma unroll
                for (int i = 0; i < TP; ++i) exp_bp[kk][i] = bp[i];
#pragma unroll
                for (int j = 0; j < TQ; ++j) exp_aq[kk][j] = aq[j];
#endif
            }
#if C8_EXP & 2
#pragma unroll
            for (int i = 0; i < TP; ++i) bp[i] = exp_bp[kk][i];
#pragma unroll
            for (int j = 0; j < TQ; ++j) aq[j] = exp_aq[kk][j];
#endif
#if C8_EXP & 4
#pragma unroll
            for (int i = 0; i < TP; ++i) asm volatile("" ::"v"(bp[i]));
#pragma unroll
            for (int j = 0; j < TQ; ++j) asm volatile("" ::"v"(aq[j]));
#else
#pragma unroll
            for (int j = 0; j < TQ; ++j)
#pragma unroll
                for (int i = 0; i < TP; ++i) acc[j][i] = mfma_bf16(aq[j], bp[i], acc[j][i]);
#endif
        }
    }
#if (C8_EXP & 8) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int j = 0; j < TQ; ++j)
#pragma unroll
        for (int i = 0; i < TP; ++i) asm volatile("" ::"v"(acc[j][i]));
#endif

    // ---- epilogue ----
    typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at 4-byte alignment
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const bool wide = EPI == 0 ? a.y_wide != 0 : EPI == 1;   // dense unmasked fp32 output: staged through LDS, stored 16 bytes per lane
    const bool has_y = EPI == 0 ? a.y != nullptr : (EPI == 1 || EPI == 4);
    const bool has_yb = EPI == 0 ? a.yb != nullptr : (EPI == 2 || EPI == 3);
    const bool has_maskb = EPI == 0 ? a.maskb != nullptr : EPI == 3;
    const bool has_mask = EPI == 0 && a.mask != nullptr;
    const float floor_v = a.relu ? 0.f : -INFINITY;
    float* stg = reinterpret_cast<float*>(lds) + wave * C::STG;             // this wave's [32 TQ channels][32 pixels + 4]
    if (wide) __syncthreads();                                              // every wave is done with the ring
    // Outputs and masks go through buffer instructions: per-lane 32-bit offset (pixel within the tile's first image n0, + this lane
    // half's 4-channel step) + a scalar channel offset -- no 64-bit address arithmetic per store; pixels past the end carry an
    // offset that fails the range check (loads answer 0, stores are dropped), so the epilogue has no per-lane branch.
    constexpr uint32_t OOB = 0xF0000000u;
    const int nimg = a.M / a.OHW;
    auto out_rsrc = [&](const void* base, int64_t img_bytes) {
        int64_t bytes = (int64_t)(nimg - n0) * img_bytes;
        if (bytes > MAX_BUF_BYTES) bytes = MAX_BUF_BYTES;
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + (int64_t)n0 * img_bytes, 0,
                                                 (int)(uint32_t)bytes, 0x00020000);
    };
    const int64_t y_img = (int64_t)a.Ctot * a.y_plane * 4, yb_img = (int64_t)a.yb_cb * a.yb_plane, mb_img = (int64_t)a.yb_cb * a.mb_plane;
    const __amdgpu_buffer_rsrc_t rs_y = out_rsrc(a.y ? (const void*)a.y : (const void*)a.x, y_img);
    const __amdgpu_buffer_rsrc_t rs_yb = out_rsrc(a.yb ? (const void*)a.yb : (const void*)a.x, yb_img);
    const __amdgpu_buffer_rsrc_t rs_mb = out_rsrc(a.maskb ? (const void*)a.maskb : (const void*)a.x, mb_img);
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < ((C8_EXP & 8) ? 0 : TP); ++i) {
        const int mb = p0 + (wp * TP + i) * 32;                             // first pixel of this 32-pixel block (uniform)
        const int m = mb + (lane & 31);
        const bool valid = m < a.M;
        const int mc = valid ? m : a.M - 1;
        const uint32_t n = fd_div((uint32_t)mc, a.dOHW), r = (uint32_t)mc - n * a.dOHW.d;
        const uint32_t oh = fd_div(r, a.dOW), ow = r - oh * a.dOW.d;
        const int64_t ypix = (int64_t)(oh + a.y_halo) * a.y_w + ow + a.y_halo;
        const uint32_t dn = n - (uint32_t)n0;
        const uint32_t voff_y = valid ? (uint32_t)(dn * y_img) + (uint32_t)((ypix + (int64_t)4 * (lane >> 5) * a.y_plane) * 4) : OOB;
        const uint32_t voff_yb = valid ? (uint32_t)(dn * yb_img) + (uint32_t)((oh + a.yb_halo) * a.yb_row + (ow + a.yb_halo) * 16 + 8 * (lane >> 5)) : OOB;
        const uint32_t voff_mb = valid ? (uint32_t)(dn * mb_img) + (uint32_t)((oh + a.mb_halo) * a.mb_row + (ow + a.mb_halo) * 16 + 8 * (lane >> 5)) : OOB;
        // packed ReluGrad masks: the 4 words of channel block j + 1 are requested before the stores of block j go out
        u32x2 mkn[4];
        auto load_masks = [&](int j) {
#pragma unroll
            for (int qg = 0; qg < 4; ++qg) {
                const int cog0 = q0 + (wq * TQ + j) * 32 + 8 * qg;
                mkn[qg] = u32x2{0x3f803f80u, 0x3f803f80u};
                if (has_maskb && cog0 < a.Cog)
                    mkn[qg] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_mb, (int)voff_mb, (int)(((g * a.Cog + cog0) >> 3) * a.mb_plane), 0));
            }
        };
        load_masks(0);
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            u32x2 mkc[4];
#pragma unroll
            for (int qg = 0; qg < 4; ++qg) mkc[qg] = mkn[qg];
            if (j + 1 < TQ) load_masks(j + 1);
#pragma unroll
            for (int qg = 0; qg < 4; ++qg) {
                const int cog0 = q0 + (wq * TQ + j) * 32 + 8 * qg;   // first channel (within the group) of this 8-channel block
                if (cog0 >= a.Cog) continue;                         // (uniform)
                const int cu = g * a.Cog + cog0;                     // uniform part of the channel; this lane's 4: cu + 4 (lane >> 5) + e
                float v[4];
                const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_l + (wq * TQ + j) * 32 + 8 * qg + 4 * (lane >> 5));
                const u32x2 mk = mkc[qg];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaxf(acc[j][i][qg * 4 + e] + bq[e], floor_v);
                    if (has_mask && valid) v[e] = a.mask[((int64_t)n * a.Ctot + cu + 4 * (lane >> 5) + e) * a.y_plane + ypix] > 0.f ? v[e] : 0.f;
                    const uint32_t mw = e < 2 ? mk[0] : mk[1];
                    if (has_maskb) v[e] = (int16_t)((e & 1) ? (mw >> 16) : (mw & 0xffffu)) > 0 ? v[e] : 0.f;   // bf16 > 0
                    if (has_y && !wide) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v[e]), rs_y, (int)voff_y, (int)((cu + e) * a.y_plane * 4), 0);
                    if (wide) stg[(j * 32 + qg * 8 + 4 * (lane >> 5) + e) * 36 + (lane & 31)] = v[e];
                }
                if (has_yb) __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])}, rs_yb, (int)voff_yb, (int)((cu >> 3) * a.yb_plane), 0);
            }
        }
        if (wide) {
            // the block's 32 pixels x 32 TQ channels leave as 16-byte stores: lane = (pixel quad lane & 7, channel lane >> 3 of 8 per pass).
            // y is dense, so consecutive pixels of an image are consecutive floats; a quad that straddles two images (or the end) goes
            // pixel by pixel.
            const int m4 = mb + (lane & 7) * 4;
            const uint32_t n4 = fd_div((uint32_t)(m4 < a.M ? m4 : 0), a.dOHW), r4 = (uint32_t)(m4 < a.M ? m4 : 0) - n4 * a.dOHW.d;
            const bool whole = m4 + 3 < a.M && (int)r4 + 3 < a.OHW;
#pragma unroll
            for (int it = 0; it < TQ * 4; ++it) {
                const int ch = it * 8 + (lane >> 3), cog = q0 + wq * TQ * 32 + ch;
                if (cog >= a.Cog || m4 >= a.M) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + ch * 36 + (lane & 7) * 4);
                float* dst = a.y + ((int64_t)n4 * a.Ctot + g * a.Cog + cog) * a.y_plane + r4;
                if (whole) {
                    *reinterpret_cast<f32x4u*>(dst) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (m4 + e >= a.M) break;
                        const bool next = (int)r4 + e >= a.OHW;        // this pixel is in the next image
                        dst[(next ? (int64_t)a.Ctot * a.y_plane - a.OHW : 0) + e] = v[e];
                    }
                }
            }
        }
    }
}

static int* upload_taps(int cg, int kh, int kw, int Hp, int Wp, int row_shift, int col_shift) {
    const int nt = c8_taps(cg, kh, kw), ntp = c8_taps_padded(cg, kh, kw);
    int* h = (int*)calloc(ntp + KT, sizeof(int));     // + KT: conv_c8_kernel reads one stage past the last (prefetch)
    if (!h) return nullptr;
    for (int t = 0; t < nt; ++t) {
        const int kx = t % kw, ky = (t / kw) % kh, cb = t / (kw * kh);
        h[t] = (((cb * Hp) + ky + row_shift) * Wp + kx + col_shift) * 16;
    }
    int* dev = nullptr;
    if (hipMalloc((void**)&dev, sizeof(int) * (ntp + KT)) != hipSuccess ||
        hipMemcpy(dev, h, sizeof(int) * (ntp + KT), hipMemcpyHostToDevice) != hipSuccess) {
        free(h);
        return nullptr;
    }
    free(h);
    return dev;
}

// the tap tables depend on the halos in force: built lazily, rebuilt by vl_conv_set_halo (which frees them through c8_free_tables)
void conv_c8_free_tables(vl_conv_desc* d) {
    if (!d) return;
    if (d->c8_toff_fwd) (void)hipFree(d->c8_toff_fwd);
    if (d->c8_toff_bwd) (void)hipFree(d->c8_toff_bwd);
    d->c8_toff_fwd = d->c8_toff_bwd = nullptr;
}

// Built at vl_conv_create / vl_conv_set_halo (rebuild_tables in mfma_gemm.hip): nothing is allocated inside the compute calls.
// A layer the packed kernels cannot run (no SAME halo, phase-split input, channels not in whole 8-blocks) simply has no tables.
int conv_c8_build_tables(vl_conv_desc* d) {
    conv_c8_free_tables(d);
    if (d->cig % 8 != 0 || d->cog % 8 != 0) return 0;
    if (d->fwd_padded && d->x_phase <= 1) {
        d->c8_toff_fwd = upload_taps(d->cig, d->kh, d->kw, d->h + 2 * d->x_halo, d->w + 2 * d->x_halo, d->x_halo - d->pt, d->x_halo - d->pl);
        if (!d->c8_toff_fwd) return 2;
    }
    if (d->stride == 1 && d->bwd_padded) {
        // dgrad: dx[h][w] = sum dy[h + ky' - (kh-1-pt)][..] Wflip[ky'][kx'] over dy with halo dy_halo
        d->c8_toff_bwd = upload_taps(d->cog, d->kh, d->kw, d->oh + 2 * d->dy_halo, d->ow + 2 * d->dy_halo, d->dy_halo - (d->kh - 1 - d->pt),
                                     d->dy_halo - (d->kw - 1 - d->pl));
        if (!d->c8_toff_bwd) return 2;
    }
    return 0;
}

static int c8_tables(vl_conv_desc* d) {
    VL_CHECK(d->fwd_padded && d->x_phase <= 1, "conv c8: x needs the padded layout (halo >= SAME padding), not phase split");
    VL_CHECK(d->c8_toff_fwd, "conv c8: no tap tables for this layer (vl_conv_set_halo builds them)");
    return 0;
}

template <int WP, int WQ, int TP, int TQ, int EPI>
static int launch_c8e(const C8ConvArgs& a, int groups, hipStream_t stream) {
    using C = C8Cfg<WP, WQ, TP, TQ>;
    auto kern = conv_c8_kernel<WP, WQ, TP, TQ, EPI>;
    static bool attr = false;
    if (!attr) {
        VL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr = true;
    }
    const int tiles_q = (a.Cog + C::Q - 1) / C::Q, tiles_p = (a.M + C::P - 1) / C::P;
    VL_CHECK((tiles_q - 1) * C::Q + C::QF <= a.CoP, "conv c8: packed weights narrower than the channel tiling");
    hipLaunchKernelGGL(kern, dim3(tiles_q * tiles_p, groups), dim3(C::NT), C::LDS_BYTES, stream, a, tiles_q);
    VL_LAUNCH_CHECK();
    return 0;
}

template <int WP, int WQ, int TP, int TQ>
static int launch_c8(const C8ConvArgs& a, int groups, hipStream_t stream) {
    static const bool generic = vl_exp_env("VL_C8_GENERIC_EPILOGUE") != nullptr;      // A/B: the run-time-flag epilogue everywhere
    const bool y = a.y != nullptr, yb = a.yb != nullptr, mb = a.maskb != nullptr, mf = a.mask != nullptr;
    if (!generic && !mf) {
        if (y && !yb && !mb && a.y_wide) return launch_c8e<WP, WQ, TP, TQ, 1>(a, groups, stream);
        if (!y && yb && !mb) return launch_c8e<WP, WQ, TP, TQ, 2>(a, groups, stream);
        if (!y && yb && mb) return launch_c8e<WP, WQ, TP, TQ, 3>(a, groups, stream);
        if (y && !yb && !mb && !a.y_wide) return launch_c8e<WP, WQ, TP, TQ, 4>(a, groups, stream);
    }
    return launch_c8e<WP, WQ, TP, TQ, 0>(a, groups, stream);
}

// channel tile by the group's channel count: 128-wide tiles, 192 as one tile of 192 (conv4 / conv5 dgrad), 64 for narrow groups
static int dispatch_c8(C8ConvArgs& a, int groups, hipStream_t stream) {
    static const int sched = vl_exp_env("VL_C8_SCHED") ? atoi(vl_exp_env("VL_C8_SCHED")) : 1;      // 0: lockstep fetch issue (A/B)
    a.sched = sched;
    if (a.Cog <= 64) return launch_c8<4, 1, 2, 2>(a, groups, stream);          // 256 pixels x 64 channels, 4 waves
    if (a.Cog <= 96) return launch_c8<4, 1, 2, 3>(a, groups, stream);          // 256 x 96 (conv1 as a 3x3 conv over 48 channels)
    // 192-channel groups (conv4 forward, conv4 / conv5 dgrad): two 96-wide tiles of 4 waves, two workgroups per CU (72 KB each), rather
    // than one 192-wide tile of 8 waves that owns the CU (112 KB): 3-6 % faster (VL_C8_Q192=1 runs the wide tile)
    static const bool q192 = vl_exp_env("VL_C8_Q192") != nullptr;
    if (a.Cog % 128 != 0 && a.Cog % 192 == 0 && !q192) return launch_c8<4, 1, 2, 3>(a, groups, stream);
    if (a.Cog % 128 != 0 && a.Cog % 192 == 0) return launch_c8<4, 2, 2, 3>(a, groups, stream);   // 256 x 192
    // 256 x 128 on 8 waves of 64 x 64.  Measured alternatives: 128 x 128 tiles of 4 waves, three per CU: 1-4 % slower; 256 x 128 on FOUR
    // waves of 64 pixels x 128 channels (16 MFMAs per wave between barriers instead of 8): within +-4 % on every launch (round 3)
    return launch_c8<4, 2, 2, 2>(a, groups, stream);
}

/* y = conv(x) + bias (ReLU) from the c8 operand xb and vl_conv_c8_pack_w(bwd = 0)'s weights: y (fp32 NCHW, y_halo) and / or yb
 * (bf16 c8, y_halo) are written. */
extern "C" int vl_conv_c8_fwd(vl_conv_desc* d, const void* xb, const void* wb, const float* bias, float* y, void* yb, int n, int relu,
                              vl_stream_t stream) {
    VL_CHECK(d && xb && wb && (y || yb) && n > 0, "vl_conv_c8_fwd: bad argument");
    VL_CHECK(d->cig % 8 == 0 && d->cog % 8 == 0, "vl_conv_c8_fwd: channels per group must be a multiple of 8");
    if (int rc = c8_tables(d)) return rc;
    const int Hp = d->h + 2 * d->x_halo, Wp = d->w + 2 * d->x_halo, CB = d->cin / 8;
    VL_CHECK((int64_t)n * d->oh * d->ow < (1ll << 31), "vl_conv_c8_fwd: too many output pixels");
    C8ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const char*)xb;
    a.x_img = (int64_t)CB * Hp * Wp * 16;
    a.x_total = a.x_img * n;
    a.x_grp = (int64_t)(d->cig / 8) * Hp * Wp * 16;
    a.x_row = Wp * 16;
    a.stride = d->stride;
    a.OW = d->ow;
    a.OHW = d->oh * d->ow;
    a.M = n * a.OHW;
    a.dOHW = make_fastdiv(a.OHW);
    a.dOW = make_fastdiv(a.OW);
    a.toff = d->c8_toff_fwd;
    a.nstages = c8_taps_padded(d->cig, d->kh, d->kw) / KT;
    a.w = (const char*)wb;
    a.CoP = c8_cop(d->cog);
    a.Cog = d->cog;
    a.Ctot = d->cout;
    a.bias = bias;
    a.relu = relu;
    a.y = y;
    a.y_wide = (y != nullptr && d->y_halo == 0 && d->oh * d->ow >= 4) ? 1 : 0;
    a.y_halo = d->y_halo;
    a.y_w = d->ow + 2 * d->y_halo;
    a.y_plane = (int64_t)(d->oh + 2 * d->y_halo) * a.y_w;
    a.yb = (char*)yb;
    a.yb_halo = d->y_halo;
    a.yb_row = a.y_w * 16;
    a.yb_plane = a.y_plane * 16;
    a.yb_cb = d->cout / 8;
    return dispatch_c8(a, d->groups, (hipStream_t)stream);
}

/* dx = d(loss)/dx from the c8 operand dyb (dy_halo) and vl_conv_c8_pack_w(bwd = 1)'s weights (stride-1 layers): dx (fp32 NCHW,
 * dx_halo) and / or dxb (bf16 c8, dx_halo); relu_mask (fp32, dx's layout) or relu_mask_c8 (the layer's own packed input xb, x_halo)
 * fuses the ReluGrad of the producing layer. */
extern "C" int vl_conv_c8_dgrad(vl_conv_desc* d, const void* dyb, const void* wbt, float* dx, void* dxb, const float* relu_mask,
                                const void* relu_mask_c8, int n, vl_stream_t stream) {
    VL_CHECK(d && dyb && wbt && (dx || dxb) && n > 0, "vl_conv_c8_dgrad: bad argument");
    VL_CHECK(d->stride == 1 && d->bwd_padded && d->c8_toff_bwd, "vl_conv_c8_dgrad: stride-1 layers in the padded layout only");
    VL_CHECK(d->cig % 8 == 0 && d->cog % 8 == 0, "vl_conv_c8_dgrad: channels per group must be a multiple of 8");
    if (int rc = c8_tables(d)) return rc;
    const int Hp = d->oh + 2 * d->dy_halo, Wp = d->ow + 2 * d->dy_halo, CB = d->cout / 8;
    C8ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const char*)dyb;
    a.x_img = (int64_t)CB * Hp * Wp * 16;
    a.x_total = a.x_img * n;
    a.x_grp = (int64_t)(d->cog / 8) * Hp * Wp * 16;
    a.x_row = Wp * 16;
    a.stride = 1;
    a.OW = d->w;
    a.OHW = d->h * d->w;
    a.M = n * a.OHW;
    a.dOHW = make_fastdiv(a.OHW);
    a.dOW = make_fastdiv(a.OW);
    a.toff = d->c8_toff_bwd;
    a.nstages = c8_taps_padded(d->cog, d->kh, d->kw) / KT;
    a.w = (const char*)wbt;
    a.CoP = c8_cop(d->cig);
    a.Cog = d->cig;
    a.Ctot = d->cin;
    a.mask = relu_mask;
    a.maskb = (const char*)relu_mask_c8;
    a.mb_halo = d->x_halo;
    a.mb_row = (d->w + 2 * d->x_halo) * 16;
    a.mb_plane = (int64_t)(d->h + 2 * d->x_halo) * a.mb_row;
    a.y = dx;
    a.y_halo = d->dx_halo;
    a.y_w = d->w + 2 * d->dx_halo;
    a.y_plane = (int64_t)(d->h + 2 * d->dx_halo) * a.y_w;
    a.yb = (char*)dxb;
    a.yb_halo = d->dx_halo;
    a.yb_row = a.y_w * 16;
    a.yb_plane = a.y_plane * 16;
    a.yb_cb = d->cin / 8;
    return dispatch_c8(a, d->groups, (hipStream_t)stream);
}

// ---- wgrad ---------------------------------------------------------------------------------------------------------------------
// dW[tap = (cb, ky, kx)][ci][co] = sum over pixels x[pixel + tap][ci] * dy[pixel][co] for a stride-1 layer whose x and dy share one
// padded plane geometry (x_halo == dy_halo: every SAME layer with an odd kernel).  Then a pixel of dy and its tap of x differ by a
// CONSTANT number of chunks, so the reduction sweeps each image's plane linearly from the first to the last valid pixel (L = (OH - 1)
// Wp + OW positions; the halo columns in between hold zeros in dy and cost 2 hh / Wp of the MFMAs) and no fetch needs a per-pixel
// address: lane = (position in a group of 4, chunk), per-lane offset constant, the stage's position a scalar.  The only per-lane
// work is a compare + select where a stage's 32 positions straddle two images or the end of the slab.
// Both operands reduce over their STRIDED index, so the LDS image is [4-position group][position q][chunk ^ 4q] (16-byte chunks, 1 KB
// per fetch = per group) and an MFMA operand is two ds_read_b64_tr_b16: each returns 4 positions x 1 column per lane out of a 4 x 16
// block, i.e. the transpose.  The xor keeps the four rows of a block on different banks (conflict free per 32-lane half).
// MFMA A = x taps (rows = (tap, ci)), B = dy (columns = co = lanes): the accumulators store coalesced along co into per-slab images
// [slab][group][tap][ci][co], summed in slab order (deterministic) into HWIO by wgrad_c8_reduce_kernel.
struct C8WgradArgs {
    const char* x;        // c8 forward input
    const char* dy;       // c8 output gradient
    int64_t x_img, dy_img, x_grp, dy_grp, x_total, dy_total;   // bytes
    int64_t dy_plane;     // bytes per channel-block plane of dy
    const int* toff;      // forward tap offsets (c8_toff_fwd); null: tap t at t * tap_stride (dense products, vl_gemm_kc8)
    int64_t tap_stride;
    int ntaps;            // real taps per group
    int Cog;              // output channels per group
    int L;                // positions swept per image
    int qstart;           // first valid position of a plane (hh * Wp + hh)
    int nimg;
    int64_t Rtot;         // nimg * L
    int slab_len;         // positions per slab (multiple of 32)
    float* ws;            // [slab][group][rowsP][CoP]
    int rowsP, CoP;       // rows (taps * 8) / columns of a slab image, padded to the tiling
    FastDiv dL;
    int sched;            // 1: the second half of the waves issues its stage fetches after its first MFMA group (conv_c8_kernel's stagger)
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char* lds_ptr;
// ds_read_b64_tr_b16 at p + IMM: pointer arithmetic on the LDS array (not an integer cast), so that IMM lands in the instruction's
// offset field instead of costing a vector add per read
template <int IMM>
__device__ __forceinline__ i32x2 lds_read_tr(lds_ptr p) {
    return __builtin_bit_cast(i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + IMM)));
}

template <int WA, int WB, int TB>
struct C8WgCfg {
    static constexpr int NW = WA * WB, NT = 64 * NW, ROWS = 64 * WA, COLS = 32 * TB * WB, TAPS = ROWS / 8, NBUF = C8_WG_NBUF, KP = 32;
    static constexpr int SUBA = (TAPS + 15) / 16, SUBB = (COLS / 8 + 15) / 16;      // 16-chunk sub-images per operand
    static constexpr int A_BYTES = SUBA * KP * 256, B_BYTES = SUBB * KP * 256, SLOT = A_BYTES + B_BYTES;
    static constexpr int NAI = SUBA * KP / 4, NBI = SUBB * KP / 4;                  // 1 KB fetches per stage
    static constexpr int FA = (NAI + NW - 1) / NW, FB = (NBI + NW - 1) / NW, F = FA + FB;
    static constexpr size_t LDS_BYTES = (size_t)NBUF * SLOT;
};

template <int WA, int WB, int TB>
__global__ __launch_bounds__(64 * WA * WB) void wgrad_c8_kernel(const C8WgradArgs a, int tiles_a, int tiles_b) {
    using C = C8WgCfg<WA, WB, TB>;
    constexpr int NW = C::NW, NBUF = C::NBUF, SLOT = C::SLOT, KP = C::KP, FA = C::FA, FB = C::FB, F = C::F;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int bid = blockIdx.x;
    const int ta = bid % tiles_a, tb = bid / tiles_a, g = blockIdx.y, zs = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wa = wave % WA, wb = wave / WA;
    const int tap0 = ta * C::TAPS, co0 = tb * C::COLS;

    const int64_t gp_begin = (int64_t)zs * a.slab_len;
    int64_t gp_end = gp_begin + a.slab_len;
    if (gp_end > a.Rtot) gp_end = a.Rtot;
    const int nstages = gp_end > gp_begin ? (int)((gp_end - gp_begin + KP - 1) / KP) : 0;
    const int n_first = (int)fd_div((uint32_t)gp_begin, a.dL);

    // ---- fetch plan: lane = (position q = lane >> 4 of a 4-position group, chunk slot lane & 15 holding chunk (lane & 15) ^ 4q) ----
    const int q4 = lane >> 4, ch = (lane & 15) ^ (q4 << 2);
    const int64_t xo = (int64_t)n_first * a.x_img + (int64_t)g * a.x_grp, dyo = (int64_t)n_first * a.dy_img + (int64_t)g * a.dy_grp;
    int64_t xb = a.x_total - xo, db = a.dy_total - dyo;
    if (xb > MAX_BUF_BYTES) xb = MAX_BUF_BYTES;
    if (db > MAX_BUF_BYTES) db = MAX_BUF_BYTES;
    const i32x4 rs_x = rsrc_words(a.x + xo, xb), rs_dy = rsrc_words(a.dy + dyo, db);
    uint32_t voff_a[FA], voff_b[FB], lds_a[FA], lds_b[FB];
    int grp_a[FA], grp_b[FB];          // 4-position group (0 .. KP / 4 - 1) the fetch carries
#pragma unroll
    for (int j = 0; j < FA; ++j) {
        const int id = (wave + j * NW) % C::NAI, sub = id / (KP / 4), pg = id % (KP / 4);
        int t = tap0 + sub * 16 + ch;
        t = t < a.ntaps ? t : 0;                                   // rows past the last tap: any valid address, discarded by the epilogue
        voff_a[j] = (a.toff ? (uint32_t)a.toff[t] : (uint32_t)(t * a.tap_stride)) + (uint32_t)(q4 * 16);
        grp_a[j] = pg;
        lds_a[j] = sub * KP * 256 + pg * 1024;
    }
#pragma unroll
    for (int j = 0; j < FB; ++j) {
        const int id = (wave + j * NW) % C::NBI, sub = id / (KP / 4), pg = id % (KP / 4);
        int cb = co0 / 8 + sub * 16 + ch;
        cb = cb < a.Cog / 8 ? cb : 0;
        voff_b[j] = (uint32_t)((int64_t)cb * a.dy_plane) + (uint32_t)(q4 * 16);
        grp_b[j] = pg;
        lds_b[j] = C::A_BYTES + sub * KP * 256 + pg * 1024;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const uint32_t dx_delta = (uint32_t)(a.x_img - (int64_t)a.L * 16), dy_delta = (uint32_t)(a.dy_img - (int64_t)a.L * 16);
    constexpr uint32_t OOB = 0xF0000000u;
    auto issue = [&](int st) {
        const uint32_t slot = lds0 + (uint32_t)(st % NBUF) * SLOT;
        int gp0 = (int)(gp_begin + (int64_t)st * KP);                          // first position of the stage (Rtot < 2^31)
        // opaque to the optimiser: otherwise the per-lane offsets of the rarely taken branches below become loop-carried vector
        // induction variables, a dozen vector adds in EVERY stage
        asm volatile("" : "+s"(gp0));
        const int n_s = (int)fd_div((uint32_t)gp0, a.dL);                      // its image; at most one image boundary inside (L >= KP)
        const int r0 = gp0 - n_s * a.L;                                        // position within the image's sweep
        const int to_next = a.L - r0;                                          // positions of the stage before the next image starts
        const int to_end = (int)((gp_end - gp0) < KP ? (gp_end - gp0) : KP);   // positions before the slab ends
        if (to_next >= KP && to_end >= KP) {                                   // the whole stage inside one image: no per-lane work at all
            const uint32_t sx = (uint32_t)((int64_t)(n_s - n_first) * a.x_img) + (uint32_t)r0 * 16;
            const uint32_t sdy = (uint32_t)((int64_t)(n_s - n_first) * a.dy_img) + (uint32_t)(a.qstart + r0) * 16;
            // (readfirstlane: the offsets ARE uniform; it keeps hipcc from folding them into a per-stage vector add on the lane offsets)
#pragma unroll
            for (int j = 0; j < FA; ++j)
                lds_dma16(rs_x, slot + lds_a[j], voff_a[j], __builtin_amdgcn_readfirstlane((int)(sx + (uint32_t)grp_a[j] * 64)));
#pragma unroll
            for (int j = 0; j < FB; ++j)
                lds_dma16(rs_dy, slot + lds_b[j], voff_b[j], __builtin_amdgcn_readfirstlane((int)(sdy + (uint32_t)grp_b[j] * 64)));
        } else if (a.L >= KP) {                                                // at most one image boundary inside the stage
            const uint32_t sx = (uint32_t)((int64_t)(n_s - n_first) * a.x_img) + (uint32_t)r0 * 16;
            const uint32_t sdy = (uint32_t)((int64_t)(n_s - n_first) * a.dy_img) + (uint32_t)(a.qstart + r0) * 16;
#pragma unroll
            for (int j = 0; j < FA; ++j) {
                const int p = grp_a[j] * 4 + q4;                               // this lane's position within the stage
                uint32_t v = voff_a[j] + (p >= to_next ? dx_delta : 0u);
                v = p >= to_end ? OOB : v;
                lds_dma16(rs_x, slot + lds_a[j], v, (int)(sx + (uint32_t)grp_a[j] * 64));
            }
#pragma unroll
            for (int j = 0; j < FB; ++j) {
                const int p = grp_b[j] * 4 + q4;
                uint32_t v = voff_b[j] + (p >= to_next ? dy_delta : 0u);
                v = p >= to_end ? OOB : v;
                lds_dma16(rs_dy, slot + lds_b[j], v, (int)(sdy + (uint32_t)grp_b[j] * 64));
            }
        } else {                                                               // tiny planes: every lane decodes its own image
#pragma unroll
            for (int j = 0; j < FA; ++j) {
                const int p = grp_a[j] * 4 + q4;
                const uint32_t gp = (uint32_t)gp0 + (uint32_t)p, n = fd_div(gp, a.dL), r = gp - n * (uint32_t)a.L;
                uint32_t v = voff_a[j] - (uint32_t)(q4 * 16) + (uint32_t)((int64_t)((int)n - n_first) * a.x_img) + r * 16;   // (voff holds q4's 16 bytes)
                v = p >= to_end ? OOB : v;
                lds_dma16(rs_x, slot + lds_a[j], v, 0);
            }
#pragma unroll
            for (int j = 0; j < FB; ++j) {
                const int p = grp_b[j] * 4 + q4;
                const uint32_t gp = (uint32_t)gp0 + (uint32_t)p, n = fd_div(gp, a.dL), r = gp - n * (uint32_t)a.L;
                uint32_t v = voff_b[j] - (uint32_t)(q4 * 16) + (uint32_t)((int64_t)((int)n - n_first) * a.dy_img) + (r + (uint32_t)a.qstart) * 16;
                v = p >= to_end ? OOB : v;
                lds_dma16(rs_dy, slot + lds_b[j], v, 0);
            }
        }
    };

    f32x16 acc[2][TB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    for (int st = 0; st < NBUF - 1 && st < nstages; ++st) issue(st);
    // stagger, see conv_c8_kernel; workgroups of 8 / 12 waves only (conv4 wgrad 0.336 -> 0.299 ms, conv3 0.319 -> 0.307, conv2 0.506 ->
    // 0.484 in the benchmark step; the 4-wave tile of conv1 ran 8 % SLOWER staggered: its SIMDs hold one wave of the workgroup each)
    const bool late = NW >= 8 && a.sched != 0 && wave >= NW / 2;       // (uniform)
    // transposed operand reads: 16-lane group g16 -> (columns 16 (g16 & 1) .. + 15 of the 32-wide block, positions 8 (g16 >> 1) .. + 7);
    // lane 4 q + p of the group addresses row q, columns 4 p .. 4 p + 3 = chunk 2 (g16 & 1) + (p >> 1), byte 8 (p & 1)
    const int g16 = lane >> 4, rq = (lane & 15) >> 2, rp = lane & 3;
    const int cl = 2 * (g16 & 1) + (rp >> 1);                        // chunk within the block's 4 chunks
    uint32_t rd_a[2], rd_b[TB];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ca = (wa * 2 + i) * 4 + cl;                                   // chunk within the tile: sub-image chunk >> 4, slot chunk & 15
        rd_a[i] = (uint32_t)((ca >> 4) * KP * 256 + (g16 >> 1) * 2048 + rq * 256 + (((ca & 15) ^ (rq << 2)) * 16) + 8 * (rp & 1));
    }
#pragma unroll
    for (int i = 0; i < TB; ++i) {
        const int cbk = (wb * TB + i) * 4 + cl;
        rd_b[i] = (uint32_t)(C::A_BYTES + (cbk >> 4) * KP * 256 + (g16 >> 1) * 2048 + rq * 256 + (((cbk & 15) ^ (rq << 2)) * 16) + 8 * (rp & 1));
    }
    for (int st = 0; st < nstages; ++st) {
        const int later = (nstages - 1 - st) < (NBUF - 2) ? (nstages - 1 - st) : (NBUF - 2);
        if (later >= 2)
            wait_vm<2 * F>();
        else if (later == 1)
            wait_vm<F>();
        else
            wait_vm<0>();
        __syncthreads();
        if (!late && st + NBUF - 1 < nstages) issue(st + NBUF - 1);
        const uint32_t so = lds0 + (uint32_t)(st % NBUF) * SLOT;
        lds_ptr pa[2], pb[TB];
        // one vector add per operand block and stage; the asm pins the sum in a register so that every read below is base + immediate
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint32_t t = so + rd_a[i];
            asm volatile("" : "+v"(t));
            pa[i] = (lds_ptr)(uintptr_t)t;
        }
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            uint32_t t = so + rd_b[i];
            asm volatile("" : "+v"(t));
            pb[i] = (lds_ptr)(uintptr_t)t;
        }
        static_for<0, KP / 16>([&](auto kc) {
            constexpr int kk = decltype(kc)::value;
            if constexpr (kk == 1) {
                if (late && st + NBUF - 1 < nstages) issue(st + NBUF - 1);
            }
            i32x4 av[2], bv[TB];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const i32x2 a0 = lds_read_tr<kk * 4096>(pa[i]), a1 = lds_read_tr<kk * 4096 + 1024>(pa[i]);
                av[i] = i32x4{a0[0], a0[1], a1[0], a1[1]};
            }
#pragma unroll
            for (int i = 0; i < TB; ++i) {
                const i32x2 b0 = lds_read_tr<kk * 4096>(pb[i]), b1 = lds_read_tr<kk * 4096 + 1024>(pb[i]);
                bv[i] = i32x4{b0[0], b0[1], b1[0], b1[1]};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TB; ++j) acc[i][j] = mfma_bf16(av[i], bv[j], acc[i][j]);
        });
    }

    // ---- epilogue: slab image [zs][g][row = tap * 8 + ci][co] ----
    float* out = a.ws + ((int64_t)zs * gridDim.y + g) * a.rowsP * a.CoP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const int col = co0 + (wb * TB + j) * 32 + (lane & 31);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = tap0 * 8 + (wa * 2 + i) * 32 + 8 * (q >> 2) + 4 * (lane >> 5) + (q & 3);
                if (row < a.rowsP && col < a.CoP) out[(int64_t)row * a.CoP + col] = acc[i][j][q];
            }
        }
}

// sum of p[z * stride], z = 0 .. slabs - 1, added to s in slab order (deterministic), eight loads in flight per pass: with one load per
// pass these reductions were `slabs` dependent round trips per thread (round 4)
__device__ __forceinline__ float sum_slabs_c8(const float* __restrict__ p, int64_t stride, int slabs, float s) {
    int z = 0;
    for (; z + 8 <= slabs; z += 8) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p[(int64_t)(z + e) * stride];
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[e];
    }
    for (; z < slabs; ++z) s += p[(int64_t)z * stride];
    return s;
}

// dw[ky][kx][c][g * cog + co] = sum over slabs of ws[slab][g][((c / 8 * kh + ky) * kw + kx) * 8 + c % 8][co], in slab order
__global__ void wgrad_c8_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int kh, int kw, int cig, int cog, int groups,
                                       int rowsP, int CoP, int slabs, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int cout = cog * groups;
    const int cf = (int)(idx % cout);
    const int64_t r = idx / cout;
    const int c = (int)(r % cig), kx = (int)((r / cig) % kw), ky = (int)(r / ((int64_t)cig * kw));
    const int g = cf / cog, co = cf % cog;
    const int row = (((c >> 3) * kh + ky) * kw + kx) * 8 + (c & 7);
    const float* p = ws + ((int64_t)g * rowsP + row) * CoP + co;
    const int64_t slab_stride = (int64_t)groups * rowsP * CoP;
    dw[idx] = sum_slabs_c8(p, slab_stride, slabs, 0.f);
}

struct C8WgPlan {
    int wa, wb, tb, tiles_a, tiles_b, slabs, slab_len, rowsP, CoP;
    int64_t Rtot;
    int L;
};

static int c8_wgrad_plan(const vl_conv_desc* d, int n, C8WgPlan* p) {
    const int ntaps = c8_taps(d->cig, d->kh, d->kw);
    // waves (taps x channels) and 32-channel blocks per wave: 32 taps x 128 channels; 192-channel groups: 32 taps x 192 on 12 waves;
    // up to 96 channels (conv1 as a 3x3 conv over 48 channels): 32 taps x 96 with every wave on all 96
    p->tb = 2;
    p->wb = d->cog % 128 == 0 ? 2 : d->cog % 192 == 0 ? 3 : 2;
    static const bool narrow192 = vl_exp_env("VL_C8_WG192_NARROW") != nullptr;   // A/B: 16 taps x 192 channels on 6 waves (8 % slower)
    p->wa = p->wb == 3 ? (narrow192 ? 2 : 4) : 4;
    if (d->cog <= 96 && d->cog > 64) p->wa = 4, p->wb = 1, p->tb = 3;
    // 192-channel groups (conv4): two 96-wide 4-wave tiles of 72 KB -- TWO workgroups per CU -- instead of one 12-wave, 96 KB tile that
    // owns the CU with all its waves in lockstep: 0.349 -> 0.303 ms at 1024 frames (round 3; VL_C8_WG192_WIDE=1 runs the wide tile)
    static const bool wide192 = vl_exp_env("VL_C8_WG192_WIDE") != nullptr;
    if (!wide192 && d->cog % 192 == 0) p->wa = 4, p->wb = 1, p->tb = 3;
    const int taps_tile = p->wa * 8, cols_tile = p->wb * p->tb * 32;
    p->tiles_a = (ntaps + taps_tile - 1) / taps_tile;
    p->tiles_b = (d->cog + cols_tile - 1) / cols_tile;
    p->rowsP = p->tiles_a * taps_tile * 8;
    p->CoP = p->tiles_b * cols_tile;
    const int Wp = d->w + 2 * d->x_halo;
    p->L = (d->oh - 1) * Wp + d->ow;
    p->Rtot = (int64_t)n * p->L;
    const int tiles = p->tiles_a * p->tiles_b * d->groups;
    int slabs = (2 * vl_device_cus()) / tiles;                       // ONE round of two workgroups per CU: one more would run alone
    const int64_t stages = (p->Rtot + 31) / 32;
    if (slabs > stages / 8) slabs = (int)(stages / 8);               // at least 8 stages per slab
    if (slabs < 1) slabs = 1;
    int64_t per = (stages + slabs - 1) / slabs;
    p->slab_len = (int)(per * 32);
    p->slabs = (int)((p->Rtot + p->slab_len - 1) / p->slab_len);
    return 0;
}

extern "C" size_t vl_conv_c8_wgrad_ws_bytes(const vl_conv_desc* d, int n) {
    if (!d || n <= 0) return 0;
    C8WgPlan p;
    c8_wgrad_plan(d, n, &p);
    return (size_t)p.slabs * d->groups * p.rowsP * p.CoP * sizeof(float);
}

static int c8_wgrad_sched() {
    static const int v = vl_exp_env("VL_C8_WG_SCHED") ? atoi(vl_exp_env("VL_C8_WG_SCHED")) : 1;      // 0: lockstep fetch issue (A/B)
    return v;
}

template <int WA, int WB, int TB>
static int launch_c8_wgrad(const C8WgradArgs& a, const C8WgPlan& p, int groups, hipStream_t stream) {
    using C = C8WgCfg<WA, WB, TB>;
    auto kern = wgrad_c8_kernel<WA, WB, TB>;
    static bool attr = false;
    if (!attr) {
        VL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_a * p.tiles_b, groups, p.slabs), dim3(C::NT), C::LDS_BYTES, stream, a, p.tiles_a, p.tiles_b);
    VL_LAUNCH_CHECK();
    return 0;
}

/* dw (HWIO fp32) = d(loss)/dw from the c8 operands xb (x_halo) and dyb (dy_halo == x_halo), stride-1 layers; deterministic slab
 * reduction through ws (>= vl_conv_c8_wgrad_ws_bytes). */
extern "C" int vl_conv_c8_wgrad(vl_conv_desc* d, const void* xb, const void* dyb, float* dw, void* ws, size_t ws_bytes, int n,
                                vl_stream_t stream) {
    VL_CHECK(d && xb && dyb && dw && ws && n > 0, "vl_conv_c8_wgrad: bad argument");
    VL_CHECK(d->stride == 1 && d->x_halo == d->dy_halo && d->oh == d->h && d->ow == d->w, "vl_conv_c8_wgrad: stride-1 SAME layers with x_halo == dy_halo");
    VL_CHECK(d->cig % 8 == 0 && d->cog % 8 == 0, "vl_conv_c8_wgrad: channels per group must be a multiple of 8");
    if (int rc = c8_tables(d)) return rc;
    C8WgPlan p;
    c8_wgrad_plan(d, n, &p);
    VL_CHECK(ws_bytes >= (size_t)p.slabs * d->groups * p.rowsP * p.CoP * sizeof(float), "vl_conv_c8_wgrad: workspace too small");
    VL_CHECK(p.Rtot < (1ll << 31), "vl_conv_c8_wgrad: too many positions");
    const int Hp = d->h + 2 * d->x_halo, Wp = d->w + 2 * d->x_halo;
    C8WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.sched = c8_wgrad_sched();
    a.x = (const char*)xb;
    a.dy = (const char*)dyb;
    a.dy_plane = (int64_t)Hp * Wp * 16;
    a.x_img = (int64_t)(d->cin / 8) * a.dy_plane;
    a.dy_img = (int64_t)(d->cout / 8) * a.dy_plane;
    a.x_grp = (int64_t)(d->cig / 8) * a.dy_plane;
    a.dy_grp = (int64_t)(d->cog / 8) * a.dy_plane;
    a.x_total = a.x_img * n;
    a.dy_total = a.dy_img * n;
    a.toff = d->c8_toff_fwd;
    a.ntaps = c8_taps(d->cig, d->kh, d->kw);
    a.Cog = d->cog;
    a.L = p.L;
    a.qstart = d->x_halo * Wp + d->x_halo;
    a.nimg = n;
    a.Rtot = p.Rtot;
    a.slab_len = p.slab_len;
    a.ws = (float*)ws;
    a.rowsP = p.rowsP;
    a.CoP = p.CoP;
    a.dL = make_fastdiv(p.L);
    int rc = p.tb == 3   ? launch_c8_wgrad<4, 1, 3>(a, p, d->groups, (hipStream_t)stream)
             : p.wb == 3 ? (p.wa == 4 ? launch_c8_wgrad<4, 3, 2>(a, p, d->groups, (hipStream_t)stream)
                                      : launch_c8_wgrad<2, 3, 2>(a, p, d->groups, (hipStream_t)stream))
                         : launch_c8_wgrad<4, 2, 2>(a, p, d->groups, (hipStream_t)stream);
    if (rc) return rc;
    const int64_t total = (int64_t)d->kh * d->kw * d->cig * d->cout;
    hipLaunchKernelGGL(wgrad_c8_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, dw,
                       d->kh, d->kw, d->cig, d->cog, d->groups, p.rowsP, p.CoP, p.slabs, total);
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- bias gradient from the packed gradient: db[c] = sum over images and pixels of dyb (halo zeros add nothing) ----------------
// grid (CB, S): block (cb, s) sums the 8 channels of block cb over images [s * per, (s + 1) * per): wave w takes images w, w + 4, ..,
// lanes take 16-byte chunks; fixed order -> bitwise reproducible.  ws: float[S][8 CB].
__device__ __forceinline__ float bf16_lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }

__global__ void bias_grad_c8_stage1(const uint4* __restrict__ dyb, float* __restrict__ ws, int n, int CB, int plane, int per) {
    __shared__ float sm[4][8];
    const int cb = blockIdx.x, s = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = s * per, n1 = min(n, n0 + per);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int img = n0 + wave; img < n1; img += 4) {
        const uint4* p = dyb + ((int64_t)img * CB + cb) * plane;
        for (int i = lane; i < plane; i += 64) {
            const uint4 v = p[i];
            acc[0] += bf16_lo(v.x);
            acc[1] += bf16_hi(v.x);
            acc[2] += bf16_lo(v.y);
            acc[3] += bf16_hi(v.y);
            acc[4] += bf16_lo(v.z);
            acc[5] += bf16_hi(v.z);
            acc[6] += bf16_lo(v.w);
            acc[7] += bf16_hi(v.w);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float t = wave_sum(acc[j]);
        if (lane == 0) sm[wave][j] = t;
    }
    __syncthreads();
    if (threadIdx.x < 8) ws[((int64_t)s * CB + cb) * 8 + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

__global__ void bias_grad_c8_stage2(const float* __restrict__ ws, float* __restrict__ db, int C, int CP, int S) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    db[c] = sum_slabs_c8(ws + c, CP, S, 0.f);                         // (one load per pass: 64 dependent round trips, 16 us per layer)
}

/* db[c] = sum_{n,h,w} dy[n][c][h][w] from the packed gradient dyb (c8, halo); ws: float[64 * 8 * ceil(c / 8)]. */
extern "C" int vl_bias_grad_c8(const void* dyb, float* db, float* ws, int n, int c, int h, int w, int halo, vl_stream_t stream) {
    VL_CHECK(dyb && db && ws && n > 0 && c > 0 && h > 0 && w > 0 && halo >= 0, "vl_bias_grad_c8: bad argument");
    const int CB = (c + 7) / 8, plane = (h + 2 * halo) * (w + 2 * halo);
    const int S = n < 64 ? n : 64, per = ceil_div(n, S), S2 = ceil_div(n, per);
    hipLaunchKernelGGL(bias_grad_c8_stage1, dim3(CB, S2), dim3(256), 0, (hipStream_t)stream, (const uint4*)dyb, ws, n, CB, plane, per);
    VL_LAUNCH_CHECK();
    hipLaunchKernelGGL(bias_grad_c8_stage2, dim3(ceil_div(c, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, db, c, CB * 8, S2);
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- a strided first layer as a stride-1 layer: space to depth ---------------------------------------------------------------------
// conv1 (11 x 11, stride s = 4, 3 input channels) has neither 8 channels for a chunk nor unit stride.  With kh = s a + py, kw = s b +
// px it IS a ka x ka stride-1 convolution (ka = ceil(k / s) = 3) over the C s^2 = 48 "channels" (c, py, px) of the space-to-depth input
//   x'[n][(c, py, px)][R][S] = xpad[c][s R + py][s S + px]      (xpad: x behind its pt / pl leading SAME zeros; R = oh + a, S = ow + b)
//   W'[a][b][(c, py, px)][co] = W[s a + py][s b + px][c][co]    (zero where s a + py or s b + px >= k)
// -- same products, same sums, 19 % more multiplies by zero -- so forward and wgrad run the c8 kernels above on a 48-channel layer
// whose packed input is written in its "SAME halo 1" form (R, S run over oh + 2, ow + 2: the halo positions hold data here).
__global__ void s2d_c8_kernel(const float* __restrict__ x0, uint4* __restrict__ xb, int C, int s, int H, int W, int halo, int phase, int pt,
                              int pl, int OHp, int OWp, int CB, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int S = (int)(idx % OWp), R = (int)((idx / OWp) % OHp), cb = (int)((idx / ((int64_t)OWp * OHp)) % CB);
    const int n = (int)(idx / ((int64_t)OWp * OHp * CB));
    const int Hp = H + 2 * halo, Wfull = W + 2 * halo, Wq = phase > 1 ? (Wfull + phase - 1) / phase : Wfull;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cp = cb * 8 + j, px = cp % s, py = (cp / s) % s, c = cp / (s * s);
        const int row = s * R + py - pt + halo, col = s * S + px - pl + halo;       // physical position in x0's padded plane
        float t = 0.f;
        if (c < C && row >= 0 && row < Hp && col >= 0 && col < Wfull)
            t = phase > 1 ? x0[((((int64_t)n * C + c) * phase + col % phase) * Hp + row) * Wq + col / phase]
                          : x0[(((int64_t)n * C + c) * Hp + row) * Wq + col];
        v[j] = t;
    }
    uint4 o;
    o.x = pack_bf16(v[0], v[1]);
    o.y = pack_bf16(v[2], v[3]);
    o.z = pack_bf16(v[4], v[5]);
    o.w = pack_bf16(v[6], v[7]);
    xb[idx] = o;
}

// fwd (grad = 0): ws[a][b][(c, py, px)][co] = W[s a + py][s b + px][c][co] or 0;  grad = 1: dW[kh][kw][c][co] = dws[kh / s][kw / s][(c, kh % s, kw % s)][co]
__global__ void s2d_weights_kernel(const float* __restrict__ src, float* __restrict__ dst, int k, int s, int C, int cout, int ka, int grad,
                                   int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int co = (int)(idx % cout);
    if (!grad) {
        const int cp = (int)((idx / cout) % (C * s * s)), b = (int)((idx / ((int64_t)cout * C * s * s)) % ka), a = (int)(idx / ((int64_t)cout * C * s * s * ka));
        const int px = cp % s, py = (cp / s) % s, c = cp / (s * s), kh = s * a + py, kw = s * b + px;
        dst[idx] = (kh < k && kw < k) ? src[((int64_t)(kh * k + kw) * C + c) * cout + co] : 0.f;
    } else {
        const int c = (int)((idx / cout) % C), kw = (int)((idx / ((int64_t)cout * C)) % k), kh = (int)(idx / ((int64_t)cout * C * k));
        const int cp = (c * s + kh % s) * s + kw % s;
        dst[idx] = src[((int64_t)((kh / s) * ka + kw / s) * (C * s * s) + cp) * cout + co];
    }
}

static int s2d_check(const vl_conv_desc* d, const char* who) {
    VL_CHECK(d && d->stride > 1 && d->groups == 1 && d->kh == d->kw && d->fwd_padded, "%s: a strided, ungrouped, square layer in the padded layout", who);
    VL_CHECK((((d->kh - 1) / d->stride + 1) & 1) == 1, "%s: ceil(k / stride) must be odd", who);
    return 0;
}

/* x0 (fp32, the strided layer's input as vl_conv_fwd takes it: x_halo, plain or phase split) -> the packed space-to-depth input of
 * the equivalent stride-1 layer: c8 [n][cin s^2 / 8][oh + ka - 1][ow + ka - 1][8], ka = ceil(k / stride). */
extern "C" int vl_s2d_c8_from_x0(const vl_conv_desc* d, const float* x0, void* xb, int n, vl_stream_t stream) {
    VL_CHECK(x0 && xb && n > 0, "vl_s2d_c8_from_x0: bad argument");
    if (int rc = s2d_check(d, "vl_s2d_c8_from_x0")) return rc;
    const int s = d->stride, ka = (d->kh - 1) / s + 1, OHp = d->oh + ka - 1, OWp = d->ow + ka - 1, CB = (d->cin * s * s + 7) / 8;
    const int64_t total = (int64_t)n * CB * OHp * OWp;
    hipLaunchKernelGGL(s2d_c8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x0, (uint4*)xb, d->cin, s, d->h,
                       d->w, d->x_halo, d->x_phase > 1 ? d->x_phase : 1, d->pt, d->pl, OHp, OWp, CB, total);
    VL_LAUNCH_CHECK();
    return 0;
}

/* grad = 0: w (HWIO [k][k][cin][cout]) -> w_s2d ([ka][ka][cin s^2][cout]);  grad = 1: dw_s2d -> dw (the inverse gather). */
extern "C" int vl_s2d_weights(const vl_conv_desc* d, const float* src, float* dst, int grad, vl_stream_t stream) {
    VL_CHECK(src && dst, "vl_s2d_weights: bad argument");
    if (int rc = s2d_check(d, "vl_s2d_weights")) return rc;
    const int s = d->stride, ka = (d->kh - 1) / s + 1;
    const int64_t total = grad ? (int64_t)d->kh * d->kw * d->cin * d->cout : (int64_t)ka * ka * d->cin * s * s * d->cout;
    hipLaunchKernelGGL(s2d_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, d->kh, s, d->cin,
                       d->cout, ka, grad, total);
    VL_LAUNCH_CHECK();
    return 0;
}

// The TFRecord frames straight into that packed input: Dataset.process_image (dataset_.py:481-501: crop, mirror, mean) as
// vl_input_prep_u8 does it, the fp32 difference rounded to bf16 -- identical values to vl_input_prep_u8 + vl_s2d_c8_from_x0.
__global__ void input_prep_u8_s2d_kernel(const uint8_t* __restrict__ src, uint4* __restrict__ xb, int C, int s, int H, int W, int rh, int rw,
                                         const int32_t* __restrict__ cy, const int32_t* __restrict__ cx, const uint8_t* __restrict__ mir,
                                         const float* __restrict__ mean, int pt, int pl, int OHp, int OWp, int CB, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int S = (int)(idx % OWp), R = (int)((idx / OWp) % OHp), cb = (int)((idx / ((int64_t)OWp * OHp)) % CB);
    const int n = (int)(idx / ((int64_t)OWp * OHp * CB));
    const int oy = cy ? cy[n] : 0, ox = cx ? cx[n] : 0;
    const bool flip = mir && mir[n];
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cp = cb * 8 + j, px = cp % s, py = (cp / s) % s, c = cp / (s * s);
        const int ih = s * R + py - pt, iw = s * S + px - pl;
        float t = 0.f;
        if (c < C && ih >= 0 && ih < H && iw >= 0 && iw < W)
            t = (float)src[(((int64_t)n * rh + ih + oy) * rw + (flip ? W - 1 - iw : iw) + ox) * C + c] - (mean ? mean[c] : 0.f);
        v[j] = t;
    }
    uint4 o;
    o.x = pack_bf16(v[0], v[1]);
    o.y = pack_bf16(v[2], v[3]);
    o.z = pack_bf16(v[4], v[5]);
    o.w = pack_bf16(v[6], v[7]);
    xb[idx] = o;
}

// The AlexNet case (3 channels, stride 4): one thread = 2 source rows x 4 pixels x 3 channels = 2 x 12 contiguous source bytes (three
// unaligned dword loads per row away from the borders; the generic kernel issues 8 byte loads per chunk and reads every source byte
// from three threads) -> the three chunks (one per channel) of its (row pair, column quad): 0.38 -> 0.2 ms per 1024 frames.
__global__ void input_prep_u8_s2d_c3s4_kernel(const uint8_t* __restrict__ src, uint4* __restrict__ xb, int H, int W, int rh, int rw,
                                              const int32_t* __restrict__ cy, const int32_t* __restrict__ cx, const uint8_t* __restrict__ mir,
                                              const float* __restrict__ mean, int pt, int pl, int OHp, int OWp, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int S = (int)(idx % OWp), hp = (int)((idx / OWp) & 1), R = (int)((idx / (2 * OWp)) % OHp);
    const int n = (int)(idx / ((int64_t)2 * OWp * OHp));
    const int oy = cy ? cy[n] : 0, ox = cx ? cx[n] : 0;
    const bool flip = mir && mir[n];
    const float m0 = mean ? mean[0] : 0.f, m1 = mean ? mean[1] : 0.f, m2 = mean ? mean[2] : 0.f;
    const int iw0 = 4 * S - pl;
    const bool interior = iw0 >= 0 && iw0 + 3 < W;
    float v[3][8];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int ih = 4 * R + 2 * hp + r - pt;
        const bool rowok = ih >= 0 && ih < H;
        const uint8_t* row = src + ((int64_t)n * rh + (rowok ? ih : 0) + oy) * rw * 3;
        uint8_t b[12];
        if (rowok && interior) {
            const uint8_t* q = row + (int64_t)((flip ? W - 1 - (iw0 + 3) : iw0) + ox) * 3;
            uint32_t w3[3];
            __builtin_memcpy(w3, q, 12);
#pragma unroll
            for (int k = 0; k < 12; ++k) b[k] = (uint8_t)(w3[k >> 2] >> (8 * (k & 3)));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sk = flip ? 3 - k : k;
                v[0][4 * r + k] = (float)b[3 * sk] - m0;
                v[1][4 * r + k] = (float)b[3 * sk + 1] - m1;
                v[2][4 * r + k] = (float)b[3 * sk + 2] - m2;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int iw = iw0 + k;
                const bool ok = rowok && iw >= 0 && iw < W;
                const uint8_t* q = row + (int64_t)((flip ? W - 1 - iw : iw) + ox) * 3;
                v[0][4 * r + k] = ok ? (float)q[0] - m0 : 0.f;
                v[1][4 * r + k] = ok ? (float)q[1] - m1 : 0.f;
                v[2][4 * r + k] = ok ? (float)q[2] - m2 : 0.f;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        uint4 o;
        o.x = pack_bf16(v[c][0], v[c][1]);
        o.y = pack_bf16(v[c][2], v[c][3]);
        o.z = pack_bf16(v[c][4], v[c][5]);
        o.w = pack_bf16(v[c][6], v[c][7]);
        xb[(((int64_t)n * 6 + 2 * c + hp) * OHp + R) * OWp + S] = o;
    }
}

extern "C" int vl_input_prep_u8_s2d(const vl_conv_desc* d, const uint8_t* src, void* xb, int n, int raw_h, int raw_w, const int32_t* crop_y,
                                    const int32_t* crop_x, const uint8_t* mirror, const float* mean_bgr, vl_stream_t stream) {
    VL_CHECK(src && xb && n > 0, "vl_input_prep_u8_s2d: bad argument");
    if (int rc = s2d_check(d, "vl_input_prep_u8_s2d")) return rc;
    VL_CHECK(d->h <= raw_h && d->w <= raw_w, "vl_input_prep_u8_s2d: bad shape");
    const int s = d->stride, ka = (d->kh - 1) / s + 1, OHp = d->oh + ka - 1, OWp = d->ow + ka - 1, CB = (d->cin * s * s + 7) / 8;
    if (d->cin == 3 && s == 4) {
        const int64_t t2 = (int64_t)n * 2 * OHp * OWp;
        hipLaunchKernelGGL(input_prep_u8_s2d_c3s4_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (uint4*)xb, d->h,
                           d->w, raw_h, raw_w, crop_y, crop_x, mirror, mean_bgr, d->pt, d->pl, OHp, OWp, t2);
        VL_LAUNCH_CHECK();
        return 0;
    }
    const int64_t total = (int64_t)n * CB * OHp * OWp;
    hipLaunchKernelGGL(input_prep_u8_s2d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (uint4*)xb, d->cin,
                       s, d->h, d->w, raw_h, raw_w, crop_y, crop_x, mirror, mean_bgr, d->pt, d->pl, OHp, OWp, CB, total);
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- dense products on the wgrad kernel --------------------------------------------------------------------------------------------
// C[m][n] = sum_k A[k][m] B[k][n] is what wgrad_c8_kernel computes when "positions" are k and both operands are stored position-major
// in 8-channel blocks: "kc8" = [channel block][k][8] bf16 (one image, no halo, one tap per block of A).  fc6's three products are
// this with (A, B) = (pool5^T, W), (dfc6^T, W^T), (pool5, dfc6): vl_pack_kc8 writes the operands from fp32 matrices with any strides.
__global__ void pack_kc8_kernel(const float* __restrict__ src, uint4* __restrict__ dst, int64_t P, int C, int64_t ps, int64_t cs, int CB,
                                int cb_fastest, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    // thread order follows the SOURCE's contiguous index: positions when cs != 1 (8 strided reads, coalesced over threads; one
    // position per thread), channel blocks when cs == 1 (32 contiguous bytes per read, coalesced over threads; FOUR positions per
    // thread so that its stores are 64 contiguous bytes: the chunks of one position are a whole plane apart)
    const int PP = cb_fastest ? 4 : 1;
    const int64_t pos0 = cb_fastest ? (idx / CB) * 4 : idx % P;
    const int cb = cb_fastest ? (int)(idx % CB) : (int)(idx / P);
    for (int q = 0; q < PP; ++q) {
        const int64_t pos = pos0 + q;
        if (pos >= P) break;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cb * 8 + j;
            v[j] = c < C ? src[pos * ps + (int64_t)c * cs] : 0.f;
        }
        uint4 o;
        o.x = pack_bf16(v[0], v[1]);
        o.y = pack_bf16(v[2], v[3]);
        o.z = pack_bf16(v[4], v[5]);
        o.w = pack_bf16(v[6], v[7]);
        dst[(int64_t)cb * P + pos] = o;
    }
}

/* dst (kc8: [ceil(channels / 8)][positions][8] bf16) = src[position * pos_stride + channel * ch_stride] rounded to bf16. */
extern "C" int vl_pack_kc8(const float* src, void* dst, int64_t positions, int channels, int64_t pos_stride, int64_t ch_stride, vl_stream_t stream) {
    VL_CHECK(src && dst && positions > 0 && channels > 0, "vl_pack_kc8: bad argument");
    const int CB = (channels + 7) / 8;
    const int64_t total = ch_stride == 1 ? ((positions + 3) / 4) * CB : positions * CB;
    hipLaunchKernelGGL(pack_kc8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (uint4*)dst, positions, channels,
                       pos_stride, ch_stride, CB, ch_stride == 1 ? 1 : 0, total);
    VL_LAUNCH_CHECK();
    return 0;
}

__global__ void gemm_kc8_reduce_kernel(const float* __restrict__ ws, float* __restrict__ c, int M, int N, int rowsP, int CoP, int slabs,
                                       const float* __restrict__ bias, int relu) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)M * N) return;
    const int col = (int)(idx % N), row = (int)(idx / N);
    const float* p = ws + (int64_t)row * CoP + col;
    const float s = sum_slabs_c8(p, (int64_t)rowsP * CoP, slabs, bias ? bias[col] : 0.f);
    c[idx] = relu ? fmaxf(s, 0.f) : s;
}

static void gemm_kc8_plan(int m, int n, int k, int* tiles_a, int* tiles_b, int* slabs, int* slab_len) {
    *tiles_a = (m / 8 + 31) / 32;            // 32 taps = 256 rows of C per tile
    *tiles_b = (n + 127) / 128;
    const int tiles = *tiles_a * *tiles_b;
    const int64_t stages = ((int64_t)k + 31) / 32;
    int s = (2 * vl_device_cus()) / tiles;
    if (s > stages / 8) s = (int)(stages / 8);
    if (s < 1) s = 1;
    const int64_t per = (stages + s - 1) / s;
    *slab_len = (int)(per * 32);
    *slabs = (int)(((int64_t)k + *slab_len - 1) / *slab_len);
}

extern "C" size_t vl_gemm_kc8_ws_bytes(int m, int n, int k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    int ta, tb, slabs, sl;
    gemm_kc8_plan(m, n, k, &ta, &tb, &slabs, &sl);
    return (size_t)slabs * ta * 256 * tb * 128 * sizeof(float);
}

/* c[m][n] (fp32, row-major) = sum_k a[k][m] * b[k][n] (+ bias[n]) (ReLU), a and b in the kc8 layout of vl_pack_kc8 (bf16 products,
 * fp32 accumulation); m, n multiples of 8.  ws: vl_gemm_kc8_ws_bytes (split-k slabs, summed in slab order). */
extern "C" int vl_gemm_kc8(const void* a_kc8, const void* b_kc8, float* c, int m, int n, int k, const float* bias, int relu, void* ws,
                           size_t ws_bytes, vl_stream_t stream) {
    VL_CHECK(a_kc8 && b_kc8 && c && m > 0 && n > 0 && k > 0 && m % 8 == 0 && n % 8 == 0, "vl_gemm_kc8: bad argument (m, n multiples of 8)");
    VL_CHECK((int64_t)k * 16 * ((m > n ? m : n) / 8) < MAX_BUF_BYTES, "vl_gemm_kc8: operand too large for 32-bit offsets");
    C8WgPlan p;
    memset(&p, 0, sizeof(p));
    p.wa = 4; p.wb = 2; p.tb = 2;
    gemm_kc8_plan(m, n, k, &p.tiles_a, &p.tiles_b, &p.slabs, &p.slab_len);
    const bool direct = p.slabs == 1 && !bias && !relu && m % 256 == 0 && n % 128 == 0;   // one slab, nothing to add: straight into c
    C8WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.sched = c8_wgrad_sched();
    a.x = (const char*)a_kc8;
    a.dy = (const char*)b_kc8;
    a.dy_plane = (int64_t)k * 16;
    a.tap_stride = a.dy_plane;
    a.x_img = a.x_total = (int64_t)(m / 8) * a.dy_plane;
    a.dy_img = a.dy_total = (int64_t)(n / 8) * a.dy_plane;
    a.ntaps = m / 8;
    a.Cog = n;
    a.L = k;
    a.qstart = 0;
    a.nimg = 1;
    a.Rtot = k;
    a.slab_len = p.slab_len;
    a.rowsP = direct ? m : p.tiles_a * 256;
    a.CoP = direct ? n : p.tiles_b * 128;
    a.ws = direct ? c : (float*)ws;
    a.dL = make_fastdiv(k);
    VL_CHECK(direct || (ws && ws_bytes >= (size_t)p.slabs * a.rowsP * a.CoP * sizeof(float)), "vl_gemm_kc8: workspace too small");
    if (int rc = launch_c8_wgrad<4, 2, 2>(a, p, 1, (hipStream_t)stream)) return rc;
    if (!direct) {
        const int64_t total = (int64_t)m * n;
        hipLaunchKernelGGL(gemm_kc8_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, c, m, n,
                           a.rowsP, a.CoP, p.slabs, bias, relu);
        VL_LAUNCH_CHECK();
    }
    return 0;
}

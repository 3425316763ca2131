// fp32-MFMA tiled contraction engine for gfx950 + the conv / GEMM entry points built on it.
//
// One kernel template, D[i][j] = sum_r A(i, r) * B(r, j), computed with v_mfma_f32_32x32x2_f32
// (exact fp32, 64 FLOP/clk/SIMD).  256 threads = 4 wavefronts per workgroup; each wave owns a
// (BM/WM) x (BN/WN) block of 32x32 accumulator tiles.
//
// Measured on MI355X (tools/ubench/mfma_issue.hip): the fp32 MFMA sustains 156 TFLOP/s, LDS reads are
// almost free beside it, but EVERY other vector-ALU instruction of a co-resident wave costs ~3-4 of
// the MFMA's 64 issue cycles (+4 v_add per MFMA: 130 TF, +8: 112 TF) at any occupancy.  The steady
// state of this kernel therefore contains (almost) no VALU address arithmetic:
//   * global -> register staging uses raw buffer loads whose per-lane byte offset is a per-thread
//     CONSTANT and whose per-tile part is a scalar (soffset, or a buffer resource rebuilt per tile
//     with scalar ops); out-of-range lanes get an offset past num_records and read 0 in hardware;
//   * im2col taps need no bounds tests when activations carry a zero halo (padded layout);
//   * LDS is addressed with per-thread constant bases + immediates: the double buffer is unrolled
//     by two so that buffer offsets are compile-time;
//   * operand tiles live in LDS as [x][BR+2]: a lane's reduction elements are adjacent, so one
//     ds_read_b64 feeds two MFMA steps (the reduction order is permuted identically for A and B),
//     and both the b64 fragment reads and the b32 staging writes are bank-conflict free.
//
//   conv fwd / dgrad : A = HWIO weights [r=(kh,kw,ci)][i=co]               (DenseKX)
//                      B = implicit im2col gather [r][j=pixel], NCHW input  (ConvGather)
//                      epilogue writes NCHW (+bias, ReLU | ReluGrad mask)
//   conv wgrad       : A = implicit im2col gather [i=(kh,kw,ci)][r=pixel]   (WgradGather)
//                      B = dy [j=co][r=pixel]                               (DyRows)
//                      epilogue writes per-split HWIO slabs, reduced deterministically
//   dense GEMM       : A, B = DenseKX / DenseXK by transpose flag (fc6/7/8, LSTM, output fc)
//
// MFMA operand maps (cdna_hip_programming.md section 3): lane l holds A[i = l&31][r = l>>5] and
// B[r = l>>5][j = l&31]; D register q of lane l is D[i = (q&3) + 8*(q>>2) + 4*(l>>5)][j = l&31].
// j is therefore the coalesced (lane) dimension of every store.
#include <stdlib.h>
#include <string.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
static constexpr int NT = 256;  // threads per workgroup

// Byte offset that fails the buffer range check of every resource we build (num_records < 0xE0000000);
// adding any per-element offset < 2^28 to it still fails and does not wrap.
static constexpr uint32_t OOB_OFF = 0xF0000000u;
static constexpr int64_t MAX_BUF_BYTES = 0xE0000000ll;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(uint32_t)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, soff, 0));
}

// LDS operand tile: [x][BR + 2] floats (x = non-reduction index, r = reduction index within the tile)
template <int BX, int BR>
struct LdsTile {
    static constexpr int STR = BR + 2;
    static constexpr int SIZE = BX * STR;
};

// ---- loaders ----------------------------------------------------------------------------------
// Interface: init(P, x0, zg); prefetch(rt); load(rt, v[NLD]); store(tile_base, v[NLD]).

// Dense operand stored [r][x] (x contiguous in memory); lanes walk x.
template <int BX, int BR>
struct DenseKX {
    using L = LdsTile<BX, BR>;
    static constexpr int NLD = BX * BR / NT;
    static_assert(BX * BR % NT == 0, "tile must divide over 256 threads");
    struct Params {
        const float* p;
        int64_t ld;
        int X, R;
        int64_t zg_stride;
    };
    const float* base;
    int64_t step;            // elements per reduction tile
    int64_t total;           // elements from base to the end of row R-1 (rows are ld apart)
    uint32_t voff[NLD];
    int loff[NLD];
    __device__ __forceinline__ void init(const Params& P, int x0, int zg) {
        base = P.p + (int64_t)zg * P.zg_stride;
        step = (int64_t)BR * P.ld;
        total = (int64_t)P.R * P.ld;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int e = threadIdx.x + NT * j;
            const int xl = e % BX, rl = e / BX;
            voff[j] = (x0 + xl < P.X) ? (uint32_t)(((int64_t)rl * P.ld + x0 + xl) * 4) : OOB_OFF;
            loff[j] = xl * L::STR + rl;
        }
    }
    __device__ __forceinline__ void prefetch(int) {}
    __device__ __forceinline__ void load(int rt, float (&v)[NLD]) const {
        // resource rebuilt per tile with scalar ops: rows past R fail the range check and read 0
        const int64_t done = (int64_t)rt * step;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(base + done, (total - done) * 4);
#pragma unroll
        for (int j = 0; j < NLD; ++j) v[j] = buf_load(rs, voff[j], 0);
    }
    __device__ __forceinline__ void store(float* tile, const float (&v)[NLD]) const {
#pragma unroll
        for (int j = 0; j < NLD; ++j) tile[loff[j]] = v[j];
    }
};

// Dense operand stored [x][r] (r contiguous in memory); lanes walk r.
template <int BX, int BR>
struct DenseXK {
    using L = LdsTile<BX, BR>;
    static constexpr int NLD = BX * BR / NT;
    static_assert(BX * BR % NT == 0, "tile must divide over 256 threads");
    struct Params {
        const float* p;
        int64_t ld;
        int X, R;
        int64_t zg_stride;
    };
    const float* base;
    int64_t total;
    int R, rl;
    uint32_t voff[NLD];
    int loff[NLD];
    __device__ __forceinline__ void init(const Params& P, int x0, int zg) {
        base = P.p + (int64_t)zg * P.zg_stride;
        total = (int64_t)(P.X - 1) * P.ld + P.R;   // last valid element + 1
        R = P.R;
        rl = threadIdx.x % BR;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int xl = (int)(threadIdx.x + NT * j) / BR;
            voff[j] = (x0 + xl < P.X) ? (uint32_t)(((int64_t)(x0 + xl) * P.ld + rl) * 4) : OOB_OFF;
            loff[j] = xl * L::STR + rl;
        }
    }
    __device__ __forceinline__ void prefetch(int) {}
    __device__ __forceinline__ void load(int rt, float (&v)[NLD]) const {
        const int r0 = rt * BR;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(base + r0, (total - r0) * 4);
        if (r0 + BR <= R) {   // uniform: whole tile inside the reduction range
#pragma unroll
            for (int j = 0; j < NLD; ++j) v[j] = buf_load(rs, voff[j], 0);
        } else {              // last partial tile: columns past R belong to the next row, mask them
            const bool ok = r0 + rl < R;
#pragma unroll
            for (int j = 0; j < NLD; ++j) v[j] = buf_load(rs, ok ? voff[j] : OOB_OFF, 0);
        }
    }
    __device__ __forceinline__ void store(float* tile, const float (&v)[NLD]) const {
#pragma unroll
        for (int j = 0; j < NLD; ++j) tile[loff[j]] = v[j];
    }
};

// Geometry shared by the im2col loaders.  The gathered tensor is NCHW with a zero halo of `halo`
// pixels on every side of each plane (physical plane (H + 2 halo) x (W + 2 halo)).
//   padded mode  (halo >= SAME padding): ktab[k] = byte offset (ci*Pp + kh*Wp + kw)*4, no tests.
//   checked mode (halo too small, e.g. 0): ktab2[k] = {that offset, kh << 16 | kw}; taps outside the
//                 logical plane are replaced by OOB_OFF with two compares per element.
struct ConvGeom {
    const float* x;
    const int* ktab;     // padded mode
    const int2* ktab2;   // checked mode
    int K;               // im2col rows per group
    int M;               // n * OH * OW output pixels
    int H, W;            // logical input plane
    int halo, Wp;        // physical row pitch
    int stride, pt, pl;
    int OHW, OW;
    FastDiv dOHW, dOW;
    int64_t img_stride;  // elements between images   (Cin_total * Pp)
    int64_t grp_stride;  // elements between groups   (Cin_g * Pp)
    int64_t total;       // elements in x
    int zero;            // always 0: lets a loader make an address depend on the tile index without changing it
};

// im2col gather for forward / dgrad: x = output pixel across lanes (coalesced along ow), r = im2col
// row.  A wave owns NLD consecutive rows of the tile: its table entries are one aligned block read
// with one wide scalar load, fetched a tile ahead, and used as the load's scalar offset.
template <int BX, int BR, bool PADDED>
struct ConvGather {
    using L = LdsTile<BX, BR>;
    static constexpr int NLD = BX * BR / NT;   // rows per wave
    static_assert(BX % 64 == 0, "pixel tile must be a multiple of the wave size");
    static_assert(NLD * (NT / BX) == BR, "rows must split evenly over the wave groups");
    using Params = ConvGeom;
    __amdgpu_buffer_rsrc_t rsrc;
    const int* tab;
    const int2* tab2;
    int ent[NLD], ent_hw[NLD];
    uint32_t voff;
    int ih0, iw0, H, W, lbase;
    __device__ __forceinline__ void init(const Params& P, int x0, int zg) {
        const int m = x0 + threadIdx.x % BX;
        const bool vm = m < P.M;
        const uint32_t mm = vm ? m : 0;
        const uint32_t n = fd_div(mm, P.dOHW);
        const uint32_t p = mm - n * P.OHW;
        const uint32_t oh = fd_div(p, P.dOW);
        const uint32_t ow = p - oh * P.OW;
        ih0 = (int)oh * P.stride - P.pt;
        iw0 = (int)ow * P.stride - P.pl;
        H = P.H;
        W = P.W;
        // checked mode may point up to (pt, pl) before the plane origin: bias the base so offsets stay >= 0
        const int bias = PADDED ? 0 : P.pt * P.Wp + P.pl;
        rsrc = make_rsrc(P.x + (int64_t)zg * P.grp_stride - bias, (P.total - (int64_t)zg * P.grp_stride + bias) * 4);
        const int64_t o = (int64_t)n * P.img_stride + (int64_t)(ih0 + P.halo) * P.Wp + (iw0 + P.halo) + bias;
        voff = vm ? (uint32_t)o * 4u : OOB_OFF;
        if (!vm) ih0 = 1 << 28;   // checked mode: every row test fails
        const int row0 = __builtin_amdgcn_readfirstlane((int)threadIdx.x / BX) * NLD;
        tab = P.ktab + row0;
        tab2 = P.ktab2 + row0;
        lbase = (threadIdx.x % BX) * L::STR + row0;
    }
    __device__ __forceinline__ void prefetch(int rt) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            if (PADDED) {
                ent[j] = tab[rt * BR + j];
            } else {
                const int2 e = tab2[rt * BR + j];
                ent[j] = e.x;
                ent_hw[j] = e.y;
            }
        }
    }
    __device__ __forceinline__ void load(int rt, float (&v)[NLD]) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            if (PADDED) {
                v[j] = buf_load(rsrc, voff, ent[j]);
            } else {
                const int ih = ih0 + (ent_hw[j] >> 16), iw = iw0 + (ent_hw[j] & 0xffff);
                const bool ok = ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
                v[j] = buf_load(rsrc, ok ? voff : OOB_OFF, ent[j]);
            }
        }
        prefetch(rt + 1);   // tables are padded by more than a tile: reading one past the end is safe
    }
    __device__ __forceinline__ void store(float* tile, const float (&v)[NLD]) const {
#pragma unroll
        for (int j = 0; j < NLD; ++j) tile[lbase + j] = v[j];
    }
};

// im2col gather for wgrad: x = im2col row (fixed per thread: table entries live in registers),
// reduction index r = output pixel across lanes.
template <int BX, int BR, bool PADDED>
struct WgradGather {
    using L = LdsTile<BX, BR>;
    static constexpr int NLD = BX * BR / NT;
    using Params = ConvGeom;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t off[NLD];
    int khkw[NLD];
    int loff[NLD];
    int H, W, M, OHW, OW, stride, pt, pl, bias, halo, Wp;
    FastDiv dOHW, dOW;
    int64_t img_stride;
    __device__ __forceinline__ void init(const Params& P, int x0, int zg) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int xl = (int)(threadIdx.x + NT * j) / BR;
            if (PADDED) {
                off[j] = (uint32_t)P.ktab[x0 + xl];
            } else {
                const int2 e = P.ktab2[x0 + xl];
                off[j] = (uint32_t)e.x;
                khkw[j] = e.y;
            }
            loff[j] = xl * L::STR + threadIdx.x % BR;
        }
        bias = PADDED ? 0 : P.pt * P.Wp + P.pl;
        rsrc = make_rsrc(P.x + (int64_t)zg * P.grp_stride - bias, (P.total - (int64_t)zg * P.grp_stride + bias) * 4);
        H = P.H; W = P.W; M = P.M; OHW = P.OHW; OW = P.OW;
        stride = P.stride; pt = P.pt; pl = P.pl; halo = P.halo; Wp = P.Wp;
        dOHW = P.dOHW; dOW = P.dOW;
        img_stride = P.img_stride;
    }
    __device__ __forceinline__ void prefetch(int) {}
    __device__ __forceinline__ void load(int rt, float (&v)[NLD]) const {
        const int m = rt * BR + threadIdx.x % BR;
        const bool vm = m < M;
        const uint32_t mm = vm ? m : 0;
        const uint32_t n = fd_div(mm, dOHW);
        const uint32_t p = mm - n * OHW;
        const uint32_t oh = fd_div(p, dOW);
        const uint32_t ow = p - oh * OW;
        const int ih0r = (int)oh * stride - pt, iw0 = (int)ow * stride - pl;
        const uint32_t vo = (uint32_t)((int64_t)n * img_stride + (int64_t)(ih0r + halo) * Wp + (iw0 + halo) + bias) * 4u;
        const uint32_t voff = vm ? vo : OOB_OFF;
        if (PADDED) {
#pragma unroll
            for (int j = 0; j < NLD; ++j) v[j] = buf_load(rsrc, voff + off[j], 0);
        } else {
            const int ih0 = vm ? ih0r : (1 << 28);
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int ih = ih0 + (khkw[j] >> 16), iw = iw0 + (khkw[j] & 0xffff);
                const bool ok = ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
                v[j] = buf_load(rsrc, ok ? voff + off[j] : OOB_OFF, 0);
            }
        }
    }
    __device__ __forceinline__ void store(float* tile, const float (&v)[NLD]) const {
#pragma unroll
        for (int j = 0; j < NLD; ++j) tile[loff[j]] = v[j];
    }
};

// dy rows for wgrad: x = output channel (within the group), r = output pixel across lanes.
// dy is NCHW [n][Cout_total][OH + 2 halo][OW + 2 halo].
struct DyParams {
    const float* dy;
    int M, OHW, OW, Cog, Cout_total, halo, OWp;
    FastDiv dOHW, dOW;
    int64_t plane;   // (OH + 2 halo) * OWp
    int64_t total;   // elements in dy
    int zero;        // always 0 (see ConvGeom::zero)
};

template <int BX, int BR>
struct DyRows {
    using L = LdsTile<BX, BR>;
    static constexpr int NLD = BX * BR / NT;
    using Params = DyParams;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t co_off[NLD];
    bool co_ok[NLD];
    int loff[NLD];
    bool all_ok;
    int M, OHW, OW, halo, OWp;
    int64_t img_stride;
    FastDiv dOHW, dOW;
    __device__ __forceinline__ void init(const Params& P, int x0, int zg) {
        const int64_t goff = (int64_t)zg * P.Cog * P.plane;
        rsrc = make_rsrc(P.dy + goff, (P.total - goff) * 4);
        M = P.M; OHW = P.OHW; OW = P.OW; halo = P.halo; OWp = P.OWp;
        img_stride = (int64_t)P.Cout_total * P.plane;
        dOHW = P.dOHW; dOW = P.dOW;
        all_ok = x0 + BX <= P.Cog;   // uniform
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int xl = (int)(threadIdx.x + NT * j) / BR;
            co_ok[j] = x0 + xl < P.Cog;
            co_off[j] = (uint32_t)((int64_t)(x0 + xl) * P.plane * 4);
            loff[j] = xl * L::STR + threadIdx.x % BR;
        }
    }
    __device__ __forceinline__ void prefetch(int) {}
    __device__ __forceinline__ void load(int rt, float (&v)[NLD]) const {
        const int m = rt * BR + threadIdx.x % BR;
        const bool vm = m < M;
        const uint32_t mm = vm ? m : 0;
        const uint32_t n = fd_div(mm, dOHW);
        const uint32_t p = mm - n * OHW;
        const uint32_t oh = fd_div(p, dOW);
        const uint32_t ow = p - oh * OW;
        const uint32_t vo = (uint32_t)((int64_t)n * img_stride + (int64_t)(oh + halo) * OWp + ow + halo) * 4u;
        const uint32_t voff = vm ? vo : OOB_OFF;
        if (all_ok) {
#pragma unroll
            for (int j = 0; j < NLD; ++j) v[j] = buf_load(rsrc, voff + co_off[j], 0);
        } else {
#pragma unroll
            for (int j = 0; j < NLD; ++j) v[j] = buf_load(rsrc, co_ok[j] ? voff + co_off[j] : OOB_OFF, 0);
        }
    }
    __device__ __forceinline__ void store(float* tile, const float (&v)[NLD]) const {
#pragma unroll
        for (int j = 0; j < NLD; ++j) tile[loff[j]] = v[j];
    }
};

// ---- wgrad, VALU-free form (padded layout only): the reduction tile is 64 pixels = the 64 lanes of a wave,
// so one load instruction fetches ONE operand row for those pixels: the row's byte offset is a scalar
// (soffset, from the table / from co * plane) and the per-lane pixel offset is decoded once per tile.
template <int BX, int BR>
struct WgradGatherU {
    using L = LdsTile<BX, BR>;
    static constexpr int NLD = BX * BR / NT;   // rows per wave (each wave owns NLD consecutive rows)
    static_assert(BR == 64 && NLD * 4 == BX, "lanes are the reduction pixels");
    using Params = ConvGeom;
    __amdgpu_buffer_rsrc_t rsrc;
    const int* tab;
    int lbase, M, OHW, OW, stride, pt, pl, halo, Wp, zero;
    FastDiv dOHW, dOW;
    int64_t img_stride;
    __device__ __forceinline__ void init(const Params& P, int x0, int zg) {
        const int row0 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) * NLD;
        zero = P.zero;
        tab = P.ktab + x0 + row0;
        lbase = row0 * L::STR + (threadIdx.x & 63);
        rsrc = make_rsrc(P.x + (int64_t)zg * P.grp_stride, (P.total - (int64_t)zg * P.grp_stride) * 4);
        M = P.M; OHW = P.OHW; OW = P.OW; stride = P.stride; pt = P.pt; pl = P.pl; halo = P.halo; Wp = P.Wp;
        dOHW = P.dOHW; dOW = P.dOW;
        img_stride = P.img_stride;
    }
    __device__ __forceinline__ void prefetch(int) {}
    __device__ __forceinline__ void load(int rt, float (&v)[NLD]) const {
        const int m = rt * BR + (threadIdx.x & 63);
        const bool vm = m < M;
        const uint32_t mm = vm ? m : 0;
        const uint32_t n = fd_div(mm, dOHW);
        const uint32_t p = mm - n * OHW;
        const uint32_t oh = fd_div(p, dOW);
        const uint32_t ow = p - oh * OW;
        const int ih0 = (int)oh * stride - pt + halo, iw0 = (int)ow * stride - pl + halo;
        const uint32_t vo = (uint32_t)((int64_t)n * img_stride + (int64_t)ih0 * Wp + iw0) * 4u;
        const uint32_t voff = vm ? vo : OOB_OFF;
        // opaque per tile: keeps hipcc from hoisting NLD scalar offsets out of the tile loop (it spilled ~90 SGPRs)
        const int* t = tab + rt * zero;   // zero is a kernel argument that is always 0 (uniform, but not foldable)
#pragma unroll
        for (int j = 0; j < NLD; ++j) v[j] = buf_load(rsrc, voff, t[j]);
    }
    __device__ __forceinline__ void store(float* tile, const float (&v)[NLD]) const {
#pragma unroll
        for (int j = 0; j < NLD; ++j) tile[lbase + j * L::STR] = v[j];
    }
};

template <int BX, int BR>
struct DyRowsU {
    using L = LdsTile<BX, BR>;
    static constexpr int NLD = BX * BR / NT;
    static_assert(BR == 64 && NLD * 4 == BX, "lanes are the reduction pixels");
    using Params = DyParams;
    __amdgpu_buffer_rsrc_t rsrc;
    int lbase, M, OHW, OW, halo, OWp, row_bytes, first_bytes, nvalid, zero, co0;
    int64_t img_stride;
    FastDiv dOHW, dOW;
    float bsum[NLD];   // per-lane partial of sum_pixels dy[co][pixel] (bias gradient), only used by the i-tile-0 workgroups
    __device__ __forceinline__ void init(const Params& P, int x0, int zg) {
        const int row0 = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) * NLD;
        co0 = zg * P.Cog + x0 + row0;
#pragma unroll
        for (int j = 0; j < NLD; ++j) bsum[j] = 0.f;
        zero = P.zero;
        const int64_t goff = (int64_t)zg * P.Cog * P.plane;
        rsrc = make_rsrc(P.dy + goff, (P.total - goff) * 4);
        lbase = row0 * L::STR + (threadIdx.x & 63);
        M = P.M; OHW = P.OHW; OW = P.OW; halo = P.halo; OWp = P.OWp;
        img_stride = (int64_t)P.Cout_total * P.plane;
        dOHW = P.dOHW; dOW = P.dOW;
        row_bytes = (int)(P.plane * 4);
        first_bytes = (x0 + row0) * row_bytes;
        nvalid = P.Cog - (x0 + row0);   // rows of this wave that are real channels (uniform)
    }
    __device__ __forceinline__ void prefetch(int) {}
    __device__ __forceinline__ void load(int rt, float (&v)[NLD]) const {
        const int m = rt * BR + (threadIdx.x & 63);
        const bool vm = m < M;
        const uint32_t mm = vm ? m : 0;
        const uint32_t n = fd_div(mm, dOHW);
        const uint32_t p = mm - n * OHW;
        const uint32_t oh = fd_div(p, dOW);
        const uint32_t ow = p - oh * OW;
        const uint32_t vo = (uint32_t)((int64_t)n * img_stride + (int64_t)(oh + halo) * OWp + ow + halo) * 4u;
        const uint32_t voff = vm ? vo : OOB_OFF;
        const int rb = row_bytes + rt * zero;   // opaque per tile (see WgradGatherU::load)
        int so = first_bytes;
        if (nvalid >= NLD) {
#pragma unroll
            for (int j = 0; j < NLD; ++j, so += rb) v[j] = buf_load(rsrc, voff, so);
        } else {   // channel tail of the group (uniform per wave)
#pragma unroll
            for (int j = 0; j < NLD; ++j, so += rb) v[j] = j < nvalid ? buf_load(rsrc, voff, so) : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* tile, const float (&v)[NLD]) const {
#pragma unroll
        for (int j = 0; j < NLD; ++j) tile[lbase + j * L::STR] = v[j];
    }
    // bias gradient (tf.nn.bias_add, alexnet.py:31): every dy element passes through v[] exactly once per i-tile
    __device__ __forceinline__ void accumulate(const float (&v)[NLD]) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) bsum[j] += v[j];
    }
    __device__ __forceinline__ void flush(float* db_slab) const {   // db_slab: [Cout_total] partial of this split
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const float t = wave_sum(bsum[j]);
            if ((threadIdx.x & 63) == 0 && j < nvalid) db_slab[co0 + j] = t;
        }
    }
};

// ---- epilogues --------------------------------------------------------------------------------
// Row-major C[i][j] (ldc).  With zs_stride != 0 the raw partial goes to slab zs (split reduction).
struct EpiRowMajor {
    struct Params {
        float* c;
        int64_t ldc;
        int M, N;
        const float* bias;  // per column j
        const float* mask;  // same layout as c: c = mask > 0 ? c : 0
        int relu;
        int64_t zg_stride, zs_stride;
    };
    template <int TM, int TN>
    static __device__ __forceinline__ void apply(const Params& P, int zg, int zs, int i0, int j0, f32x16 (&acc)[TM][TN]) {
        const int lane = threadIdx.x & 63;
        float* c = P.c + (int64_t)zg * P.zg_stride + (int64_t)zs * P.zs_stride;
        const float* mask = P.mask ? P.mask + (int64_t)zg * P.zg_stride : nullptr;
#pragma unroll
        for (int tj = 0; tj < TN; ++tj) {
            const int j = j0 + 32 * tj + (lane & 31);
            if (j >= P.N) continue;
            const float bj = P.bias ? P.bias[j] : 0.f;
#pragma unroll
            for (int ti = 0; ti < TM; ++ti) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int i = i0 + 32 * ti + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                    if (i < P.M) {
                        float v = acc[ti][tj][q] + bj;
                        if (P.relu) v = fmaxf(v, 0.f);
                        const int64_t o = (int64_t)i * P.ldc + j;
                        if (mask) v = mask[o] > 0.f ? v : 0.f;
                        c[o] = v;
                    }
                }
            }
        }
    }
};

// NCHW conv output with halo: i = output channel within the group, j = output pixel.
struct EpiConvNCHW {
    struct Params {
        float* y;
        const float* bias;  // [Cout_total] or null
        const float* mask;  // NCHW (its own halo) or null
        int relu;
        int Cog, Cout_total, OHW, OW, M;
        FastDiv dOHW, dOW;
        int y_halo, y_wp;
        int64_t y_plane;
        int m_halo, m_wp;
        int64_t m_plane;
    };
    template <int TM, int TN>
    static __device__ __forceinline__ void apply(const Params& P, int zg, int zs, int i0, int j0, f32x16 (&acc)[TM][TN]) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int tj = 0; tj < TN; ++tj) {
            const int m = j0 + 32 * tj + (lane & 31);
            if (m >= P.M) continue;
            const uint32_t n = fd_div((uint32_t)m, P.dOHW);
            const uint32_t p = m - n * P.OHW;
            const uint32_t oh = fd_div(p, P.dOW);
            const uint32_t ow = p - oh * P.OW;
            const int64_t c0 = (int64_t)n * P.Cout_total + (int64_t)zg * P.Cog;
            const int64_t ybase = c0 * P.y_plane + (int64_t)(oh + P.y_halo) * P.y_wp + ow + P.y_halo;
            const int64_t mbase = c0 * P.m_plane + (int64_t)(oh + P.m_halo) * P.m_wp + ow + P.m_halo;
#pragma unroll
            for (int ti = 0; ti < TM; ++ti) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = i0 + 32 * ti + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                    if (co < P.Cog) {
                        float v = acc[ti][tj][q];
                        if (P.bias) v += P.bias[zg * P.Cog + co];
                        if (P.relu) v = fmaxf(v, 0.f);
                        if (P.mask) v = P.mask[mbase + (int64_t)co * P.m_plane] > 0.f ? v : 0.f;
                        P.y[ybase + (int64_t)co * P.y_plane] = v;
                    }
                }
            }
        }
    }
};

// XCD-aware tile order (cdna_hip_programming.md T1): workgroup ids are dealt round-robin over the 8 XCDs, each
// with a private L2.  Remapping id -> (id % 8) * chunk + id / 8 gives every XCD a contiguous chunk of the tile
// grid, so tiles that share operand panels (the co-tiles of one pixel tile, neighbouring pixel tiles) hit the
// same L2.  Bijective for any grid size.  Placement only affects speed, never results.
__device__ __forceinline__ int xcd_swizzle(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7, k = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// ---- the kernel -------------------------------------------------------------------------------
// grid.x = tiles_i * tiles_j (i fastest), grid.y = groups (zg), grid.z = reduction splits (zs).
template <int BM, int BN, int BR, int WM, int WN, class LA, class LB, class EP>
__global__ __launch_bounds__(NT) void mfma_contract(const typename LA::Params pa, const typename LB::Params pb,
                                                    const typename EP::Params pe, int tiles_i, int rtiles,
                                                    int rt_per_split) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BR % 4 == 0, "two MFMA steps per 8-byte fragment read");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TM * WM * 32 == BM && TN * WN * 32 == BN, "tile shape");
    constexpr int STR = BR + 2;
    constexpr int SA = BM * STR, SB = BN * STR;
    __shared__ __attribute__((aligned(16))) float lds[2 * (SA + SB)];

    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ti_blk = bid % tiles_i, tj_blk = bid / tiles_i;
    const int zg = blockIdx.y, zs = blockIdx.z;
    const int i0 = ti_blk * BM, j0 = tj_blk * BN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi0 = (wave / WN) * (BM / WM), wj0 = (wave % WN) * (BN / WN);

    LA la;
    LB lb;
    la.init(pa, i0, zg);
    lb.init(pb, j0, zg);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

    // Fragment addresses: lane half h = lane>>5 consumes, at MFMA step s, the tile's reduction element
    // 4*(s>>1) + 2*h + (s&1) (the same permutation for A and B), i.e. 8 consecutive bytes per step pair.
    const float* fa = lds + (wi0 + (lane & 31)) * STR + 2 * (lane >> 5);
    const float* fb = lds + SA + (wj0 + (lane & 31)) * STR + 2 * (lane >> 5);

    auto compute = [&](const int cur) {   // cur is a literal at both call sites
#pragma unroll
        for (int t = 0; t < BR / 4; ++t) {
            float2 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const float2*>(fa + cur * (SA + SB) + a * 32 * STR + 4 * t);
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const float2*>(fb + cur * (SA + SB) + b * 32 * STR + 4 * t);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
        }
    };

    const int rt0 = zs * rt_per_split;
    const int rt1 = min(rtiles, rt0 + rt_per_split);

    float ra[LA::NLD], rb[LB::NLD];
    if (rt0 < rt1) {
        la.prefetch(rt0);
        lb.prefetch(rt0);
        la.load(rt0, ra);
        lb.load(rt0, rb);
        la.store(lds, ra);
        lb.store(lds + SA, rb);
    }
    __syncthreads();
    // Tile rt0 + i lives in LDS buffer i & 1.  The loop is unrolled by two so that every LDS address is a
    // per-thread constant plus an immediate; an odd trailing tile is computed after the loop.
    const int ntiles = rt1 - rt0;
    int rt = rt0;
    for (int i = 0; i + 1 < ntiles; i += 2, rt += 2) {
        // sched_barrier pins the phases: issue the next tile's loads FIRST (a whole tile of MFMAs hides
        // their latency), then the MFMAs, then the LDS stores; hipcc otherwise sinks the loads to their use.
        la.load(rt + 1, ra);
        lb.load(rt + 1, rb);
        __builtin_amdgcn_sched_barrier(0);
        compute(0);
        __builtin_amdgcn_sched_barrier(0);
        la.store(lds + (SA + SB), ra);
        lb.store(lds + (SA + SB) + SA, rb);
        __syncthreads();
        const bool more = i + 2 < ntiles;
        if (more) {
            la.load(rt + 2, ra);
            lb.load(rt + 2, rb);
        }
        __builtin_amdgcn_sched_barrier(0);
        compute(1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            la.store(lds, ra);
            lb.store(lds + SA, rb);
        }
        __syncthreads();
    }
    if (ntiles > 0 && (ntiles & 1)) compute(0);   // ntiles <= 0 for a trailing, empty reduction split
    EP::template apply<TM, TN>(pe, zg, zs, i0 + wi0, j0 + wj0, acc);
}

// Same contraction with ONE LDS buffer (large reduction tiles: wgrad's 64-pixel tiles need 68 KB): the next
// tile is prefetched into registers during the MFMAs and written after a barrier; two workgroups per CU
// cover each other's barrier/store phases.
template <int BM, int BN, int BR, int WM, int WN, class LA, class LB, class EP>
__global__ __launch_bounds__(NT) void mfma_contract_1buf(const typename LA::Params pa, const typename LB::Params pb,
                                                         const typename EP::Params pe, int tiles_i, int rtiles,
                                                         int rt_per_split, float* db_slabs, int db_stride) {
    static_assert(WM * WN == 4 && BR % 4 == 0, "shape");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int STR = BR + 2;
    constexpr int SA = BM * STR;
    extern __shared__ __attribute__((aligned(16))) float lds1[];
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ti_blk = bid % tiles_i, tj_blk = bid / tiles_i;
    const int zg = blockIdx.y, zs = blockIdx.z;
    const int i0 = ti_blk * BM, j0 = tj_blk * BN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi0 = (wave / WN) * (BM / WM), wj0 = (wave % WN) * (BN / WN);
    LA la;
    LB lb;
    la.init(pa, i0, zg);
    lb.init(pb, j0, zg);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
    const float* fa = lds1 + (wi0 + (lane & 31)) * STR + 2 * (lane >> 5);
    const float* fb = lds1 + SA + (wj0 + (lane & 31)) * STR + 2 * (lane >> 5);
    const int rt0 = zs * rt_per_split;
    const int rt1 = min(rtiles, rt0 + rt_per_split);
    const bool do_bias = db_slabs != nullptr && ti_blk == 0;   // uniform: one i-tile per (j-tile, group, split)
    float ra[LA::NLD], rb[LB::NLD];
    if (rt0 < rt1) {
        la.load(rt0, ra);
        lb.load(rt0, rb);
        la.store(lds1, ra);
        lb.store(lds1 + SA, rb);
        if (do_bias) lb.accumulate(rb);
    }
    __syncthreads();
    for (int rt = rt0; rt < rt1; ++rt) {
        const bool more = rt + 1 < rt1;
        if (more) {
            la.load(rt + 1, ra);
            lb.load(rt + 1, rb);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < BR / 4; ++t) {
            float2 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const float2*>(fa + a * 32 * STR + 4 * t);
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const float2*>(fb + b * 32 * STR + 4 * t);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].x, bf[b].x, acc[a][b], 0, 0, 0);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a].y, bf[b].y, acc[a][b], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (more) {
            la.store(lds1, ra);
            lb.store(lds1 + SA, rb);
            if (do_bias) lb.accumulate(rb);
        }
        __syncthreads();
    }
    if (do_bias) lb.flush(db_slabs + (int64_t)zs * db_stride);
    EP::template apply<TM, TN>(pe, zg, zs, i0 + wi0, j0 + wj0, acc);
}

// out[e] = sum_s slab[s][e] (+bias[e % n_cols]) (relu) (mask) : deterministic split reduction.
__global__ void reduce_slabs_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t count, int splits,
                                    int64_t slab_stride, const float* __restrict__ bias, int ncols, int64_t ldc,
                                    int relu, const float* __restrict__ mask) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += ws[(int64_t)z * slab_stride + e];
        int64_t o = e;
        if (ncols > 0) {
            const int64_t row = e / ncols;
            const int col = (int)(e - row * ncols);
            if (bias) s += bias[col];
            o = row * ldc + col;
        }
        if (relu) s = fmaxf(s, 0.f);
        if (mask) s = mask[o] > 0.f ? s : 0.f;
        out[o] = s;
    }
}

// ---- convolution descriptor -------------------------------------------------------------------
struct vl_conv_desc {
    int cin, h, w, cout, kh, kw, stride, groups;
    int oh, ow, pt, pl, pb, pr;   // SAME padding before / after
    int cig, cog;
    int K;    // kh*kw*cig
    int Kd;   // kh*kw*cog (dgrad reduction length)
    int x_halo, y_halo, dy_halo, dx_halo;
    int2* ktab2_fwd;  // checked-mode tables (device), halo-aware offsets
    int2* ktab2_bwd;
    int* ktab_fwd;    // padded-mode tables
    int* ktab_bwd;
    int fwd_padded, bwd_padded;
};

static void tf_same_pad(int in, int k, int s, int* out, int* before, int* after) {
    *out = (in + s - 1) / s;
    int total = (*out - 1) * s + k - in;
    if (total < 0) total = 0;
    *before = total / 2;
    *after = total - *before;
}

// tables over planes of physical size (H + 2 halo) x (W + 2 halo); rows ordered (kh, kw, c) like HWIO
static int build_ktabs(int** dev1, int2** dev2, int kh, int kw, int cg, int H, int W, int halo) {
    const int K = kh * kw * cg;
    const int pad = ((K + 127) / 128) * 128 + 256;
    const int Wp = W + 2 * halo;
    const int64_t Pp = (int64_t)(H + 2 * halo) * Wp;
    int* h1 = (int*)malloc(sizeof(int) * pad);
    int2* h2 = (int2*)malloc(sizeof(int2) * pad);
    if (!h1 || !h2) {
        free(h1);
        free(h2);
        return 1;
    }
    for (int k = 0; k < pad; ++k) {
        if (k < K) {
            const int c = k % cg, kx = (k / cg) % kw, ky = k / (cg * kw);
            const int64_t off = ((int64_t)c * Pp + (int64_t)ky * Wp + kx) * 4;
            h1[k] = (int)off;
            h2[k].x = (int)off;
            h2[k].y = (ky << 16) | kx;
        } else {
            h1[k] = 0;               // padded mode: a valid address; the weight row is zero (range check)
            h2[k].x = 0;
            h2[k].y = 0x4000 << 16;  // checked mode: fails the row test
        }
    }
    if (*dev1) (void)hipFree(*dev1);
    if (*dev2) (void)hipFree(*dev2);
    *dev1 = nullptr;
    *dev2 = nullptr;
    hipError_t e = hipMalloc((void**)dev1, sizeof(int) * pad);
    if (e == hipSuccess) e = hipMalloc((void**)dev2, sizeof(int2) * pad);
    if (e == hipSuccess) e = hipMemcpy(*dev1, h1, sizeof(int) * pad, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(*dev2, h2, sizeof(int2) * pad, hipMemcpyHostToDevice);
    free(h1);
    free(h2);
    return e == hipSuccess ? 0 : 2;
}

static int rebuild_tables(vl_conv_desc* d) {
    int rc = build_ktabs(&d->ktab_fwd, &d->ktab2_fwd, d->kh, d->kw, d->cig, d->h, d->w, d->x_halo);
    d->fwd_padded = d->x_halo >= d->pt && d->x_halo >= d->pb && d->x_halo >= d->pl && d->x_halo >= d->pr;
    if (rc == 0 && d->stride == 1) {
        rc = build_ktabs(&d->ktab_bwd, &d->ktab2_bwd, d->kh, d->kw, d->cog, d->oh, d->ow, d->dy_halo);
        // dgrad pads dy by K-1-pad before and by the forward pad-before after
        const int need = (d->kh - 1 - d->pt > d->pt ? d->kh - 1 - d->pt : d->pt);
        const int needw = (d->kw - 1 - d->pl > d->pl ? d->kw - 1 - d->pl : d->pl);
        d->bwd_padded = d->dy_halo >= need && d->dy_halo >= needw;
    }
    return rc;
}

extern "C" int vl_conv_create(vl_conv_desc** out, int cin, int h, int w, int cout, int kh, int kw, int stride, int groups) {
    VL_CHECK(out, "vl_conv_create: null out");
    VL_CHECK(cin > 0 && cout > 0 && h > 0 && w > 0 && kh > 0 && kw > 0 && stride > 0 && groups > 0, "vl_conv_create: bad geometry");
    VL_CHECK(cin % groups == 0 && cout % groups == 0, "vl_conv_create: channels not divisible by groups");
    VL_CHECK(h < 0x4000 && w < 0x4000 && kh < 256 && kw < 256, "vl_conv_create: plane too large");
    vl_conv_desc* d = (vl_conv_desc*)calloc(1, sizeof(vl_conv_desc));
    VL_CHECK(d, "vl_conv_create: out of host memory");
    d->cin = cin; d->h = h; d->w = w; d->cout = cout; d->kh = kh; d->kw = kw; d->stride = stride; d->groups = groups;
    tf_same_pad(h, kh, stride, &d->oh, &d->pt, &d->pb);
    tf_same_pad(w, kw, stride, &d->ow, &d->pl, &d->pr);
    d->cig = cin / groups;
    d->cog = cout / groups;
    d->K = kh * kw * d->cig;
    d->Kd = kh * kw * d->cog;
    if (rebuild_tables(d)) {
        vl_conv_destroy(d);
        vl_set_error("vl_conv_create: device table allocation failed");
        return 2;
    }
    *out = d;
    return 0;
}

extern "C" int vl_conv_set_halo(vl_conv_desc* d, int x_halo, int y_halo, int dy_halo, int dx_halo) {
    VL_CHECK(d, "vl_conv_set_halo: null descriptor");
    VL_CHECK(x_halo >= 0 && y_halo >= 0 && dy_halo >= 0 && dx_halo >= 0 && x_halo < 64 && y_halo < 64 && dy_halo < 64 && dx_halo < 64,
             "vl_conv_set_halo: bad halo");
    d->x_halo = x_halo; d->y_halo = y_halo; d->dy_halo = dy_halo; d->dx_halo = dx_halo;
    VL_CHECK(rebuild_tables(d) == 0, "vl_conv_set_halo: device table allocation failed");
    return 0;
}

extern "C" void vl_conv_destroy(vl_conv_desc* d) {
    if (!d) return;
    if (d->ktab_fwd) (void)hipFree(d->ktab_fwd);
    if (d->ktab_bwd) (void)hipFree(d->ktab_bwd);
    if (d->ktab2_fwd) (void)hipFree(d->ktab2_fwd);
    if (d->ktab2_bwd) (void)hipFree(d->ktab2_bwd);
    free(d);
}

extern "C" int vl_conv_out_hw(const vl_conv_desc* d, int* oh, int* ow) {
    VL_CHECK(d, "vl_conv_out_hw: null descriptor");
    if (oh) *oh = d->oh;
    if (ow) *ow = d->ow;
    return 0;
}

// ---- conv forward / dgrad launch ---------------------------------------------------------------
struct ConvOut {
    float* y;
    const float* bias;
    const float* mask;
    int relu;
    int y_halo, m_halo, OH, OW;
};

template <int BM, int WM, int WN, bool PADDED>
static int launch_conv(const ConvGeom& g, const float* w, int64_t w_ld, int w_grp_stride, int Cog, int Cout_total,
                       const ConvOut& o, hipStream_t s) {
    constexpr int BN = 128, BR = 16;
    using LA = DenseKX<BM, BR>;
    using LB = ConvGather<BN, BR, PADDED>;
    typename LA::Params pa{w, w_ld, Cog, g.K, (int64_t)w_grp_stride};
    EpiConvNCHW::Params pe;
    pe.y = o.y; pe.bias = o.bias; pe.mask = o.mask; pe.relu = o.relu;
    pe.Cog = Cog; pe.Cout_total = Cout_total; pe.OHW = g.OHW; pe.OW = g.OW; pe.M = g.M;
    pe.dOHW = g.dOHW; pe.dOW = g.dOW;
    pe.y_halo = o.y_halo; pe.y_wp = o.OW + 2 * o.y_halo; pe.y_plane = (int64_t)(o.OH + 2 * o.y_halo) * pe.y_wp;
    pe.m_halo = o.m_halo; pe.m_wp = o.OW + 2 * o.m_halo; pe.m_plane = (int64_t)(o.OH + 2 * o.m_halo) * pe.m_wp;
    const int tiles_i = ceil_div(Cog, BM), tiles_j = ceil_div(g.M, BN);
    const int rtiles = ceil_div(g.K, BR);
    dim3 grid(tiles_i * tiles_j, (unsigned)(Cout_total / Cog), 1);
    hipLaunchKernelGGL((mfma_contract<BM, BN, BR, WM, WN, LA, LB, EpiConvNCHW>), grid, dim3(NT), 0, s, pa, g, pe, tiles_i, rtiles,
                       rtiles);
    VL_LAUNCH_CHECK();
    return 0;
}

template <bool PADDED>
static int dispatch_conv(const ConvGeom& g, const float* w, int64_t w_ld, int w_grp_stride, int Cog, int Cout_total,
                         const ConvOut& o, hipStream_t s) {
    // output-channel tile: 128 when it divides well, else 96 (conv1: 96, conv4: 192) or 64 (conv2 dgrad: 48)
    const int w128 = ceil_div(Cog, 128) * 128, w96 = ceil_div(Cog, 96) * 96, w64 = ceil_div(Cog, 64) * 64;
    if (w128 <= w96 && w128 <= w64) return launch_conv<128, 2, 2, PADDED>(g, w, w_ld, w_grp_stride, Cog, Cout_total, o, s);
    if (w96 <= w64) return launch_conv<96, 1, 4, PADDED>(g, w, w_ld, w_grp_stride, Cog, Cout_total, o, s);
    return launch_conv<64, 1, 4, PADDED>(g, w, w_ld, w_grp_stride, Cog, Cout_total, o, s);
}

static void fill_geom(ConvGeom& g, const float* x, int n, int cin_total, int cig, int H, int W, int halo, int OH, int OW,
                      int stride, int pt, int pl, int K, const int* tab, const int2* tab2) {
    g.x = x; g.ktab = tab; g.ktab2 = tab2; g.K = K; g.M = n * OH * OW; g.H = H; g.W = W;
    g.halo = halo; g.Wp = W + 2 * halo;
    g.stride = stride; g.pt = pt; g.pl = pl; g.OHW = OH * OW; g.OW = OW;
    g.dOHW = make_fastdiv(g.OHW); g.dOW = make_fastdiv(g.OW);
    const int64_t Pp = (int64_t)(H + 2 * halo) * g.Wp;
    g.img_stride = (int64_t)cin_total * Pp;
    g.grp_stride = (int64_t)cig * Pp;
    g.total = g.img_stride * n;
    g.zero = 0;
}

extern "C" int vl_conv_fwd(const vl_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int n,
                           int relu, vl_stream_t stream) {
    VL_CHECK(d && x && w && y, "vl_conv_fwd: null argument");
    VL_CHECK(n > 0 && (int64_t)n * d->oh * d->ow < (1ll << 31), "vl_conv_fwd: bad batch %d", n);
    ConvGeom g;
    fill_geom(g, x, n, d->cin, d->cig, d->h, d->w, d->x_halo, d->oh, d->ow, d->stride, d->pt, d->pl, d->K, d->ktab_fwd, d->ktab2_fwd);
    VL_CHECK(g.total * 4 < MAX_BUF_BYTES, "vl_conv_fwd: input of %lld elements exceeds the buffer-offset range", (long long)g.total);
    ConvOut o{y, bias, nullptr, relu, d->y_halo, 0, d->oh, d->ow};
    // HWIO weights are the [K][Cout_total] GEMM operand as they stand; group g = column block g*cog.
    if (d->fwd_padded) return dispatch_conv<true>(g, w, d->cout, d->cog, d->cog, d->cout, o, (hipStream_t)stream);
    return dispatch_conv<false>(g, w, d->cout, d->cog, d->cog, d->cout, o, (hipStream_t)stream);
}

__global__ void conv_wt_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int KH, int KW, int cig,
                                         int cog, int groups) {
    const int cin = cig * groups, cout = cog * groups;
    const int64_t total = (int64_t)KH * KW * cog * cin;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        // e indexes wt[kh'][kw'][co][g*cig + ci]
        const int cc = (int)(e % cin);
        const int co = (int)((e / cin) % cog);
        const int kx = (int)((e / ((int64_t)cin * cog)) % KW);
        const int ky = (int)(e / ((int64_t)cin * cog * KW));
        const int g = cc / cig, ci = cc % cig;
        wt[e] = w[(((int64_t)(KH - 1 - ky) * KW + (KW - 1 - kx)) * cig + ci) * cout + g * cog + co];
    }
}

extern "C" int vl_conv_wt_transpose(const vl_conv_desc* d, const float* w, float* wt, vl_stream_t stream) {
    VL_CHECK(d && w && wt, "vl_conv_wt_transpose: null argument");
    const int64_t total = (int64_t)d->kh * d->kw * d->cog * d->cin;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(conv_wt_transpose_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wt, d->kh, d->kw, d->cig,
                       d->cog, d->groups);
    VL_LAUNCH_CHECK();
    return 0;
}

extern "C" int vl_conv_dgrad(const vl_conv_desc* d, const float* dy, const float* wt, float* dx, const float* relu_mask,
                             int n, vl_stream_t stream) {
    VL_CHECK(d && dy && wt && dx, "vl_conv_dgrad: null argument");
    VL_CHECK(d->stride == 1, "vl_conv_dgrad: only stride-1 layers (conv1 needs no input gradient)");
    VL_CHECK(n > 0 && (int64_t)n * d->h * d->w < (1ll << 31), "vl_conv_dgrad: bad batch %d", n);
    // dx = conv(dy, wt) with pad' = K-1-pad: the same kernel with the roles of the channel sets swapped.
    ConvGeom g;
    fill_geom(g, dy, n, d->cout, d->cog, d->oh, d->ow, d->dy_halo, d->h, d->w, 1, d->kh - 1 - d->pt, d->kw - 1 - d->pl, d->Kd,
              d->ktab_bwd, d->ktab2_bwd);
    VL_CHECK(g.total * 4 < MAX_BUF_BYTES, "vl_conv_dgrad: dy of %lld elements exceeds the buffer-offset range", (long long)g.total);
    ConvOut o{dx, nullptr, relu_mask, 0, d->dx_halo, d->x_halo, d->h, d->w};
    if (d->bwd_padded) return dispatch_conv<true>(g, wt, d->cin, d->cig, d->cig, d->cin, o, (hipStream_t)stream);
    return dispatch_conv<false>(g, wt, d->cin, d->cig, d->cig, d->cin, o, (hipStream_t)stream);
}

// ---- conv wgrad -------------------------------------------------------------------------------
static void dy_params(const vl_conv_desc* d, const ConvGeom& g, const float* dy, DyParams& pb) {
    pb.dy = dy; pb.M = g.M; pb.OHW = g.OHW; pb.OW = g.OW; pb.Cog = d->cog; pb.Cout_total = d->cout; pb.halo = d->dy_halo;
    pb.OWp = d->ow + 2 * d->dy_halo; pb.dOHW = g.dOHW; pb.dOW = g.dOW;
    pb.plane = (int64_t)(d->oh + 2 * d->dy_halo) * pb.OWp;
    pb.total = (int64_t)d->cout * pb.plane * (g.M / g.OHW);
    pb.zero = 0;
}

static int reduce_wgrad(const vl_conv_desc* d, float* dw, const float* ws, int splits, hipStream_t s) {
    const int64_t slab = (int64_t)d->K * d->cout;
    if (splits > 1) {
        const int blocks = (int)((slab + 255) / 256 < 4096 ? (slab + 255) / 256 : 4096);
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, s, ws, dw, slab, splits, slab, nullptr, 0, 0, 0, nullptr);
        VL_LAUNCH_CHECK();
    }
    return 0;
}

// checked-mode (dense NCHW input) wgrad: 32-pixel tiles, per-element bounds tests
template <int BN, int WM, int WN>
static int launch_wgrad(const vl_conv_desc* d, const ConvGeom& g, const float* dy, float* dw, float* ws, int splits,
                        hipStream_t s) {
    constexpr int BM = 128, BR = 32;
    using LA = WgradGather<BM, BR, false>;
    using LB = DyRows<BN, BR>;
    DyParams pb;
    dy_params(d, g, dy, pb);
    const int64_t slab = (int64_t)d->K * d->cout;
    EpiRowMajor::Params pe{splits > 1 ? ws : dw, d->cout, d->K, d->cog, nullptr, nullptr, 0, d->cog, splits > 1 ? slab : 0};
    const int tiles_i = ceil_div(d->K, BM), tiles_j = ceil_div(d->cog, BN);
    const int rtiles = ceil_div(g.M, BR);
    dim3 grid(tiles_i * tiles_j, d->groups, splits);
    hipLaunchKernelGGL((mfma_contract<BM, BN, BR, WM, WN, LA, LB, EpiRowMajor>), grid, dim3(NT), 0, s, g, pb, pe, tiles_i, rtiles,
                       ceil_div(rtiles, splits));
    VL_LAUNCH_CHECK();
    return reduce_wgrad(d, dw, ws, splits, s);
}

// padded-mode wgrad: 64-pixel tiles, wave-uniform rows, no per-element VALU
template <int BN, int WM, int WN>
static int launch_wgrad_u(const vl_conv_desc* d, const ConvGeom& g, const float* dy, float* dw, float* db, float* ws, int splits,
                          hipStream_t s) {
    constexpr int BM = 128, BR = 64;
    using LA = WgradGatherU<BM, BR>;
    using LB = DyRowsU<BN, BR>;
    DyParams pb;
    dy_params(d, g, dy, pb);
    const int64_t slab = (int64_t)d->K * d->cout;
    EpiRowMajor::Params pe{splits > 1 ? ws : dw, d->cout, d->K, d->cog, nullptr, nullptr, 0, d->cog, splits > 1 ? slab : 0};
    const int tiles_i = ceil_div(d->K, BM), tiles_j = ceil_div(d->cog, BN);
    const int rtiles = ceil_div(g.M, BR);
    constexpr size_t lds = (size_t)(BM + BN) * (BR + 2) * sizeof(float);
    static bool attr_set = false;
    auto kern = mfma_contract_1buf<BM, BN, BR, WM, WN, LA, LB, EpiRowMajor>;
    if (!attr_set) {
        VL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    // bias-gradient partials live behind the weight slabs: [splits][Cout_total]
    float* db_slabs = db ? (splits > 1 ? ws + (int64_t)splits * slab : db) : nullptr;
    dim3 grid(tiles_i * tiles_j, d->groups, splits);
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds, s, g, pb, pe, tiles_i, rtiles, ceil_div(rtiles, splits), db_slabs, d->cout);
    VL_LAUNCH_CHECK();
    if (db && splits > 1) {
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ceil_div(d->cout, 256)), dim3(256), 0, s, db_slabs, db, (int64_t)d->cout, splits,
                           (int64_t)d->cout, nullptr, 0, 0, 0, nullptr);
        VL_LAUNCH_CHECK();
    }
    return reduce_wgrad(d, dw, ws, splits, s);
}

static int wgrad_splits(const vl_conv_desc* d, int n) {
    // the reduction runs over 32-pixel (checked) or 64-pixel (padded) tiles; aim at ~2048 workgroups
    const int64_t M = (int64_t)n * d->oh * d->ow;
    const int rtiles = ceil_div(M, d->fwd_padded ? 64 : 32);
    const int tiles = ceil_div(d->K, 128) * ceil_div(d->cog, d->cog % 128 == 0 ? 128 : 96) * d->groups;
    int splits = ceil_div(2048, tiles);
    if (splits > rtiles) splits = rtiles;
    if (splits < 1) splits = 1;
    return splits;
}

extern "C" size_t vl_conv_wgrad_ws_bytes(const vl_conv_desc* d, int n) {
    if (!d || n <= 0) return 0;
    return (size_t)wgrad_splits(d, n) * ((size_t)d->K + 1) * d->cout * sizeof(float);   // weight slabs + bias partials
}

extern "C" int vl_conv_wgrad_fuses_bias(const vl_conv_desc* d) { return d && d->fwd_padded ? 1 : 0; }

extern "C" int vl_conv_wgrad(const vl_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* ws,
                             size_t ws_bytes, int n, vl_stream_t stream) {
    VL_CHECK(d && x && dy && dw, "vl_conv_wgrad: null argument");
    VL_CHECK(!db || d->fwd_padded, "vl_conv_wgrad: the fused bias gradient needs the padded layout (see vl_conv_wgrad_fuses_bias)");
    VL_CHECK(n > 0 && (int64_t)n * d->oh * d->ow < (1ll << 31), "vl_conv_wgrad: bad batch %d", n);
    const int splits = wgrad_splits(d, n);
    VL_CHECK(splits == 1 || (ws && ws_bytes >= vl_conv_wgrad_ws_bytes(d, n)), "vl_conv_wgrad: workspace too small (%zu < %zu)",
             ws_bytes, vl_conv_wgrad_ws_bytes(d, n));
    ConvGeom g;
    fill_geom(g, x, n, d->cin, d->cig, d->h, d->w, d->x_halo, d->oh, d->ow, d->stride, d->pt, d->pl, d->K, d->ktab_fwd, d->ktab2_fwd);
    const int64_t dy_total = (int64_t)n * d->cout * (d->oh + 2 * d->dy_halo) * (d->ow + 2 * d->dy_halo);
    VL_CHECK(g.total * 4 < MAX_BUF_BYTES && dy_total * 4 < MAX_BUF_BYTES, "vl_conv_wgrad: operand exceeds the buffer-offset range");
    hipStream_t s = (hipStream_t)stream;
    if (d->cog % 128 == 0) {
        if (d->fwd_padded) return launch_wgrad_u<128, 2, 2>(d, g, dy, dw, db, (float*)ws, splits, s);
        return launch_wgrad<128, 2, 2>(d, g, dy, dw, (float*)ws, splits, s);
    }
    if (d->fwd_padded) return launch_wgrad_u<96, 4, 1>(d, g, dy, dw, db, (float*)ws, splits, s);
    return launch_wgrad<96, 4, 1>(d, g, dy, dw, (float*)ws, splits, s);
}

// ---- dense GEMM -------------------------------------------------------------------------------
template <int BM, int WM, int WN, class LA, class LB>
static int launch_gemm(int m, int n, int k, const float* a, int64_t lda, const float* b, int64_t ldb, float* c, int64_t ldc,
                       const float* bias, int relu, const float* mask, float* ws, int splits, hipStream_t s) {
    constexpr int BN = 128, BR = 16;
    typename LA::Params pa{a, lda, m, k, 0};
    typename LB::Params pb{b, ldb, n, k, 0};
    const int tiles_i = ceil_div(m, BM), tiles_j = ceil_div(n, BN);
    const int rtiles = ceil_div(k, BR);
    const int per = ceil_div(rtiles, splits);
    EpiRowMajor::Params pe;
    if (splits > 1)
        pe = EpiRowMajor::Params{ws, n, m, n, nullptr, nullptr, 0, 0, (int64_t)m * n};
    else
        pe = EpiRowMajor::Params{c, ldc, m, n, bias, mask, relu, 0, 0};
    dim3 grid(tiles_i * tiles_j, 1, splits);
    hipLaunchKernelGGL((mfma_contract<BM, BN, BR, WM, WN, LA, LB, EpiRowMajor>), grid, dim3(NT), 0, s, pa, pb, pe, tiles_i, rtiles, per);
    VL_LAUNCH_CHECK();
    if (splits > 1) {
        const int64_t cnt = (int64_t)m * n;
        const int blocks = (int)((cnt + 255) / 256 < 4096 ? (cnt + 255) / 256 : 4096);
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, s, ws, c, cnt, splits, cnt, bias, n, ldc, relu, mask);
        VL_LAUNCH_CHECK();
    }
    return 0;
}

template <int BM, int WM, int WN>
static int dispatch_gemm_t(int ta, int tb, int m, int n, int k, const float* a, int64_t lda, const float* b, int64_t ldb,
                           float* c, int64_t ldc, const float* bias, int relu, const float* mask, float* ws, int splits,
                           hipStream_t s) {
    constexpr int BN = 128, BR = 16;
    if (!ta && !tb) return launch_gemm<BM, WM, WN, DenseXK<BM, BR>, DenseKX<BN, BR>>(m, n, k, a, lda, b, ldb, c, ldc, bias, relu, mask, ws, splits, s);
    if (!ta && tb) return launch_gemm<BM, WM, WN, DenseXK<BM, BR>, DenseXK<BN, BR>>(m, n, k, a, lda, b, ldb, c, ldc, bias, relu, mask, ws, splits, s);
    if (ta && !tb) return launch_gemm<BM, WM, WN, DenseKX<BM, BR>, DenseKX<BN, BR>>(m, n, k, a, lda, b, ldb, c, ldc, bias, relu, mask, ws, splits, s);
    return launch_gemm<BM, WM, WN, DenseKX<BM, BR>, DenseXK<BN, BR>>(m, n, k, a, lda, b, ldb, c, ldc, bias, relu, mask, ws, splits, s);
}

extern "C" int vl_gemm(int transa, int transb, int m, int n, int k, const float* a, int64_t lda, const float* b, int64_t ldb,
                       float* c, int64_t ldc, const float* bias, int relu, const float* relu_mask, void* ws, size_t ws_bytes,
                       vl_stream_t stream) {
    VL_CHECK(a && b && c, "vl_gemm: null argument");
    VL_CHECK(m > 0 && n > 0 && k > 0, "vl_gemm: bad shape %d x %d x %d", m, n, k);
    VL_CHECK(lda >= (transa ? m : k) && ldb >= (transb ? k : n) && ldc >= n, "vl_gemm: leading dimension too small");
    const int64_t ea = transa ? (int64_t)(k - 1) * lda + m : (int64_t)(m - 1) * lda + k;
    const int64_t eb = transb ? (int64_t)(n - 1) * ldb + k : (int64_t)(k - 1) * ldb + n;
    VL_CHECK(ea * 4 < MAX_BUF_BYTES && eb * 4 < MAX_BUF_BYTES, "vl_gemm: operand exceeds the buffer-offset range");
    const int bm = m <= 64 ? 64 : 128;
    const int tiles = ceil_div(m, bm) * ceil_div(n, 128);
    // split the reduction when the output alone cannot fill 256 CUs and a workspace was provided
    int splits = 1;
    if (ws && tiles < 192) {
        splits = ceil_div(512, tiles);
        const int maxs = k / 256 > 0 ? k / 256 : 1;
        if (splits > maxs) splits = maxs;
        const size_t cap = ws_bytes / ((size_t)m * n * sizeof(float));
        if ((size_t)splits > cap) splits = (int)cap;
        if (splits < 1) splits = 1;
    }
    hipStream_t s = (hipStream_t)stream;
    if (bm == 64) return dispatch_gemm_t<64, 1, 4>(transa, transb, m, n, k, a, lda, b, ldb, c, ldc, bias, relu, relu_mask, (float*)ws, splits, s);
    return dispatch_gemm_t<128, 2, 2>(transa, transb, m, n, k, a, lda, b, ldb, c, ldc, bias, relu, relu_mask, (float*)ws, splits, s);
}

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
//   conv wgrad       : A = implicit im2col gather [i=(kh,kw,ci)][r=pixel], B = dy [j=co][r=pixel];
//                      epilogue writes per-split HWIO slabs, reduced deterministically.
//                      padded layout: wgrad_dma_kernel (LDS-DMA operand rows, one workgroup per CU);
//                      dense layout:  the template with WgradGather / DyRows (bounds-tested)
//   dense GEMM       : A, B = DenseKX / DenseXK by transpose flag (fc6/7/8, LSTM, output fc)
//
// MFMA operand maps (cdna_hip_programming.md section 3): lane l holds A[i = l&31][r = l>>5] and
// B[r = l>>5][j = l&31]; D register q of lane l is D[i = (q&3) + 8*(q>>2) + 4*(l>>5)][j = l&31].
// j is therefore the coalesced (lane) dimension of every store.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "conv_desc.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
// Read-only tables are read through the constant address space: a uniform read from it is always a scalar load (s_load).
// Through a plain pointer hipcc falls back to a per-lane load + readfirstlane loop whenever it cannot prove that no store
// in the kernel clobbers the table.
typedef const __attribute__((address_space(4))) int* const_int_ptr;
__device__ __forceinline__ const_int_ptr as_const(const int* p) { return (const_int_ptr)(uintptr_t)p; }
static constexpr int NT = 256;  // threads per workgroup
static constexpr int KBLK = 16;  // channel block of the conv reduction order (build_ktabs) = the reduction tile of launch_conv

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

// Conv weights as the A operand: stored [r][x] like DenseKX, but reduction tile rt covers the BR consecutive rows that
// start at row_tab[rt] (one scalar table read per tile, prefetched a tile ahead): the reduction runs over a PERMUTED row
// order (channel block outermost, taps inside it -- see build_ktabs) while the weights keep their HWIO layout.
template <int BX, int BR>
struct ConvWeightKX {
    using L = LdsTile<BX, BR>;
    static constexpr int NLD = BX * BR / NT;
    static_assert(BX * BR % NT == 0, "tile must divide over 256 threads");
    struct Params {
        const float* p;
        int64_t ld;
        int X, R;
        int64_t zg_stride;
        const int* row_tab;   // [ceil(R / BR) + 1] first row of each reduction tile
    };
    const float* base;
    const int* row_tab;
    int64_t ld;
    int R, row0;
    uint32_t voff[NLD];
    int loff[NLD];
    __device__ __forceinline__ void init(const Params& P, int x0, int zg) {
        base = P.p + (int64_t)zg * P.zg_stride;
        ld = P.ld;
        R = P.R;
        row_tab = P.row_tab;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int e = threadIdx.x + NT * j;
            const int xl = e % BX, rl = e / BX;
            voff[j] = (x0 + xl < P.X) ? (uint32_t)(((int64_t)rl * P.ld + x0 + xl) * 4) : OOB_OFF;
            loff[j] = xl * L::STR + rl;
        }
    }
    __device__ __forceinline__ void prefetch(int rt) { row0 = as_const(row_tab)[rt]; }
    __device__ __forceinline__ void load(int rt, float (&v)[NLD]) {
        // rows of this tile that exist (a natural-order reduction may end in a partial tile): the others fail the range check
        // (rt is always a real tile here, so valid >= 1; a min AND a max would become a VALU v_med3 and drag the whole
        // resource descriptor into VGPRs -> a readfirstlane loop around every load)
        const int valid = min(BR, R - rt * BR);
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(base + (int64_t)row0 * ld, (int64_t)valid * ld * 4);
#pragma unroll
        for (int j = 0; j < NLD; ++j) v[j] = buf_load(rs, voff[j], 0);
        prefetch(rt + 1);
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
    int col_mul, col_add; // padded mode: physical column of output column ow, tap 0 = ow * col_mul + col_add
                         // (stride, halo - pad_left; or 1, 0 when x is stored column-phase-split, see vl_conv_set_x_phase_split)
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
        const int colp = PADDED ? (int)ow * P.col_mul + P.col_add : iw0 + P.halo;
        const int64_t o = (int64_t)n * P.img_stride + (int64_t)(ih0 + P.halo) * P.Wp + colp + bias;
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
// OCC = workgroups per CU the register allocation must allow (__launch_bounds__' second argument = waves per SIMD).
// Measured: the conv instances use 144 registers (3 waves per SIMD); with OCC = 4 hipcc fits the same loop in 126 without
// a spill, and nothing gets faster (conv2..5 fwd / dgrad within +-2 %) -- occupancy is not what holds these kernels back.
template <int BM, int BN, int BR, int WM, int WN, class LA, class LB, class EP, int OCC = 1>
__global__ __launch_bounds__(NT, OCC) void mfma_contract(const typename LA::Params pa, const typename LB::Params pb,
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

// ---- wgrad, LDS-DMA form (padded layout) -------------------------------------------------------
// dW[i = im2col row][j = co] = sum over pixels.  128 x BN x 64-pixel tiles, ONE workgroup per CU, two LDS
// buffers (2 x (128 + BN) x 66 floats = 135 KB of the CU's 160 KB).  Operand rows go global -> LDS with
// `buffer_load_dword ... lds` (LDS-DMA): a wave instruction fetches ONE operand row for the tile's 64 pixels
// (lane = pixel, per-lane byte offset decoded once per tile; the row's offset is a scalar) and the hardware
// writes the 64 dwords to consecutive LDS addresses = one [x][66] row.  No staging registers, no ds_write,
// no VALU per element.  The 64 row fetches of tile rt+1 are issued one per MFMA during the FIRST half of
// tile rt's 128 MFMAs (a burst of 64 loads ahead of the MFMAs held the wave at the issue port for ~2000
// cycles: the earlier 2-workgroup form ran its matrix pipe 67 % busy), the second half covers their latency,
// then `s_waitcnt vmcnt(0)` + one barrier per tile.
// Bias gradient: db[co] = sum_pixels 1 * dy[co][pixel] is one more row of the same product.  When K is not a
// multiple of 128 the last i-tile has spare rows: row K of the A tile is overwritten with 1.0f in LDS after the
// DMA lands, and the epilogue writes accumulator row K to the db slab (exact: fma(1, dy, acc)).
template <int BN>
struct WgradDmaCfg {
    static constexpr int BM = 128, BR = 64, STR = BR + 2;
    static constexpr int BUF = (BM + BN) * STR;           // floats per LDS buffer
    static constexpr size_t LDS_BYTES = 2 * BUF * sizeof(float);
};

// LDS-DMA row fetch, in inline asm on purpose: behind the `__builtin_amdgcn_raw_ptr_buffer_load_lds` form hipcc (ROCm 7.2)
// puts `s_waitcnt vmcnt(0)` in front of every later ds_read (it cannot tell the DMA's LDS buffer from the one being read)
// and turns every later table read into a vector load + readfirstlane loop (the builtin counts as a store to any memory).
// The asm form is invisible to both: the kernel orders DMA -> ds_read itself (vmcnt(0) + barrier at the end of a tile).
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 rsrc_words(const float* base, int64_t bytes) {
    const uint64_t a = (uint64_t)base;
    return i32x4{(int)(uint32_t)a, (int)((uint32_t)(a >> 32) & 0xffffu), (int)(uint32_t)bytes, 0x00020000};
}
__device__ __forceinline__ void lds_dma_row(i32x4 rs, uint32_t lds_byte_addr, uint32_t voff, int soff) {
    // M0 = LDS destination of lane 0; lane l lands at M0 + 4 l.  s_nop: SALU write of M0 -> LDS-DMA needs one wait state.
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(lds_byte_addr), "v"(voff), "s"(rs), "s"(soff)
                 : "m0");
}

// 16 bytes per lane: lane l lands at M0 + 16 l (tools/ubench/lds_dma_x4.hip)
__device__ __forceinline__ void lds_dma_row4(i32x4 rs, uint32_t lds_byte_addr, uint32_t voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_byte_addr), "v"(voff), "s"(rs), "s"(soff)
                 : "m0");
}

// Helpers of the split-product convolutions (vl_set_conv_math; wgrad_dma_kernel's tile_split, conv_ring_kernel and
// conv_ring4_kernel below): an fp32 operand x is split into bf16 pieces, x = p0 + p1 (+ p2) + O(2^-17 (2^-26) |x|), and a product
// is the sum of the piece products above a threshold (kProdA / kProdB) on the bf16 matrix pipe (16x the fp32 MFMA rate) with
// fp32 accumulators.  Operands stay fp32 in HBM, so every other kernel is unchanged.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int I> struct IntC { static constexpr int value = I; };
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {     // f(IntC<B>) ... f(IntC<E-1>): indices that must be compile-time constants
    if constexpr (B < E) {
        f(IntC<B>{});
        static_for<B + 1, E>(f);
    }
}
struct SplitBf16 {        // 8 reduction positions of one operand row/column as MFMA operand tuples (4 dwords of bf16 pairs) per piece:
    i32x4 p[3];           // p[0] head, p[1] = bf16(x - head), p[2] = bf16(x - head - p[1]) (bf16x6 only)
};
// NPC pieces of the pair (x0, x1): each piece is the bf16 rounding (to nearest even) of what the previous pieces left over.
template <int NPC>
__device__ __forceinline__ void split_pieces(float x0, float x1, int (&out)[3]) {
    f32x2 r = {x0, x1};
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        const uint32_t h = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
        out[k] = (int)h;
        // (the subtraction is left to hipcc, which emits one v_pk_add_f32: two scalar v_sub_f32 measured 3-10 % slower)
        if (k + 1 < NPC) r = f32x2{r[0] - __uint_as_float(h << 16), r[1] - __uint_as_float(h & 0xffff0000u)};
    }
}
// products of a split contraction, in issue order: piece of the A operand x piece of the B operand.  The first NP entries are
// the mode: NP = 1 plain bf16, 3 bf16x3 (drops tail*tail, ~2^-18), 6 bf16x6 (drops terms below ~2^-23: fp32 rounding level)
__device__ constexpr int kProdA[6] = {0, 0, 1, 1, 0, 2};
__device__ constexpr int kProdB[6] = {0, 1, 0, 1, 2, 0};
constexpr int split_products(int math) { return math == 6 ? 6 : math == 3 ? 3 : 1; }
constexpr int split_pieces_of(int math) { return math == 6 ? 3 : math == 3 ? 2 : 1; }
__device__ __forceinline__ f32x16 mfma_bf16(const i32x4& a, const i32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// X4 (split-product modes, BN = 128, output rows a multiple of 4 pixels wide, unit column stride): 16-byte fetches.  One
// instruction moves FOUR rows x 64 pixels: lane l = (row slot l >> 4, pixel quad l & 15), i.e. 4 x 256 contiguous bytes of source
// (about ten cache lines: the texture path's cost follows the lines an instruction touches) landing as 1 KB of LDS.  The four rows
// of an instruction are g, g + 32, g + 64, g + 96, and group g sits at float 258 g: an MFMA block's 32 lanes (rows g = 0..31 of one
// slot) then read at a stride of 258 floats, two banks apart -- conflict-free for the float2 operand reads.  16 fetches per wave and
// tile instead of 64 (the dword form is bound by its fetch instructions at bf16 rates).
template <int BN, int WM, int WN, int MATH = 0, bool X4 = false>   // MATH != 0: split bf16 products (vl_set_conv_math), see tile_split
__global__ __launch_bounds__(NT, 1) void wgrad_dma_kernel(const ConvGeom g, const DyParams d, const EpiRowMajor::Params pe,
                                                          int tiles_i, int tiles_j, int groups, int units, int rtiles,
                                                          int rt_per_split, float* db_slabs, int db_stride) {
    using C = WgradDmaCfg<BN>;
    constexpr int BM = C::BM, STR = C::STR, SA = BM * STR, BUF = C::BUF;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RA = BM / 4, RB = BN / 4;               // rows each wave fetches per tile (A: 32, B: 32 | 24)
    static_assert(WM * WN == 4 && TM * WM * 32 == BM && TN * WN * 32 == BN, "tile shape");
    static_assert(RA + RB <= 10 * 2 * TM * TN, "row fetches must leave the last third of a tile's MFMAs to cover their latency");
    static_assert(!X4 || (MATH != 0 && BN == 128 && BM == 128), "the 16-byte fetch form: split-product modes, 128 x 128 tiles");
    constexpr int GSTR = 258, SA4 = 32 * GSTR, BUF4 = 64 * GSTR;     // X4: floats per row group, per operand region, per buffer
    extern __shared__ __attribute__((aligned(16))) float ldsw[];

    // Workgroup -> (unit = (group, split), tile) so that ALL tiles of a unit run on ONE XCD, back to back: workgroup ids are
    // dealt round-robin over the 8 XCDs (id % 8 labels the XCD), so XCD x takes units x, x + 8, ... and walks each unit's
    // tiles (i fastest) in dispatch order.  The tiles of a unit read the same pixels of x and dy at the same pace, so one
    // workgroup's fetch is the others' L2 hit.  Placement only affects speed.
    const int tiles = tiles_i * tiles_j;
    const int k8 = blockIdx.x >> 3;
    const int unit = (k8 / tiles) * 8 + (blockIdx.x & 7);
    if (unit >= units) return;                            // grid padded to 8 * ceil(units / 8) * tiles
    const int tile_id = k8 % tiles;
    const int ti_blk = tile_id % tiles_i, tj_blk = tile_id / tiles_i;
    const int zg = unit % groups, zs = unit / groups;
    const int i0 = ti_blk * BM, j0 = tj_blk * BN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi0 = (wave / WN) * (BM / WM), wj0 = (wave % WN) * (BN / WN);

    // operand A: im2col rows of x (table of byte offsets); operand B: dy rows (co * plane)
    const i32x4 rs_x = rsrc_words(g.x + (int64_t)zg * g.grp_stride, (g.total - (int64_t)zg * g.grp_stride) * 4);
    const int64_t goff = (int64_t)zg * d.Cog * d.plane;
    const i32x4 rs_dy = rsrc_words(d.dy + goff, (d.total - goff) * 4);
    const int* tab = g.ktab + i0 + wave * RA;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)ldsw;   // LDS byte address of the array
    const int row_bytes = (int)(d.plane * 4);
    const int b_row0 = j0 + wave * RB;                    // first dy channel (within the group) this wave fetches
    const int b_last = d.Cog - 1;                         // rows past the group's channels re-read the last one (outputs dropped)
    const int64_t dy_img = (int64_t)d.Cout_total * d.plane;

    uint32_t voff_x, voff_dy;
    auto decode = [&](int rt) __attribute__((always_inline)) {                           // per-lane byte offsets of the tile's 64 pixels
        const int m = X4 ? rt * 64 + 4 * (lane & 15) : rt * 64 + lane;       // X4: lane (mod 16) = pixel quad, its first pixel
        const bool vm = m < g.M;
        const uint32_t mm = vm ? m : 0;
        const uint32_t n = fd_div(mm, g.dOHW);
        const uint32_t p = mm - n * g.OHW;
        const uint32_t oh = fd_div(p, g.dOW);
        const uint32_t ow = p - oh * g.OW;
        const int ih0 = (int)oh * g.stride - g.pt + g.halo, iw0 = (int)ow * g.col_mul + g.col_add;
        const uint32_t vx = (uint32_t)((int64_t)n * g.img_stride + (int64_t)ih0 * g.Wp + iw0) * 4u;
        const uint32_t vd = (uint32_t)((int64_t)n * dy_img + (int64_t)(oh + d.halo) * d.OWp + ow + d.halo) * 4u;
        voff_x = vm ? vx : OOB_OFF;                       // pixels past M: range check fails, the DMA writes 0
        voff_dy = vm ? vd : OOB_OFF;
        asm volatile("" : "+v"(voff_x), "+v"(voff_dy));   // computed HERE (hipcc otherwise sinks the arithmetic to the first use)
    };
    // one t-step's share of the next tile's row fetches, into LDS buffer `nb` (a literal).  Every scalar that depends only
    // on the row (LDS destination, table entry, dy row offset) is derived from a base made opaque per tile (`+ rt * zero`,
    // zero is a kernel argument that is always 0): left loop-invariant, hipcc hoists all ~200 of them out of the tile loop
    // and spills SGPRs to VGPR lanes.
    // The wave's RA table entries (byte offsets of its im2col rows) sit in SGPRs.  They never change, but they are re-read
    // in the quiet tail of every tile and made opaque: left to itself hipcc re-materialises the s_loads right in front of the
    // first fetch of each tile and waits for them there.
    int tabv[RA];
    auto load_table = [&](int rt) __attribute__((always_inline)) {                        // issue the wide s_loads ...
        const const_int_ptr tt = (const_int_ptr)(uintptr_t)(tab + rt * g.zero);
#pragma unroll
        for (int f = 0; f < RA; ++f) tabv[f] = tt[f];
    };
    auto pin_table = [&]() __attribute__((always_inline)) {                               // ... and, a few MFMAs later, wait for them and fix them in SGPRs
#pragma unroll
        for (int f = 0; f < RA; ++f) asm volatile("" : "+s"(tabv[f]));
    };
    // fetch number f (0 .. RA + RB - 1) of the next tile: A rows first, then B rows
    auto dma = [&](const int nb, const int f, int rt_next) __attribute__((always_inline)) {
        const int z = rt_next * g.zero;
        if (f < RA) {
            const uint32_t la = lds0 + (uint32_t)(wave * (RA * STR) + z) * 4u;
            lds_dma_row(rs_x, la + (uint32_t)(nb * BUF + f * STR) * 4u, voff_x, tabv[f]);
        } else {
            const int r = f - RA;
            const uint32_t lb = lds0 + (uint32_t)(SA + wave * (RB * STR) + z) * 4u;
            lds_dma_row(rs_dy, lb + (uint32_t)(nb * BUF + r * STR) * 4u, voff_dy, min(b_row0 + z + r, b_last) * row_bytes);
        }
    };
    // X4: this wave fetches row groups 8 wave .. 8 wave + 7 of both operands; lane's row of group g is g + 32 (lane >> 4).  The row
    // part of the address is fixed for the kernel: table entry of the im2col row (rows >= K: out of range -> zeros), plane of the
    // dy channel (channels >= Cog likewise)
    uint32_t vrow_a[8], vrow_b[8];
    if constexpr (X4) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int rl = wave * 8 + j + 32 * (lane >> 4);
            vrow_a[j] = i0 + rl < g.K ? (uint32_t)g.ktab[i0 + rl] : OOB_OFF;
            vrow_b[j] = j0 + rl < d.Cog ? (uint32_t)(j0 + rl) * (uint32_t)row_bytes : OOB_OFF;
        }
    }
    // X4 fetch f (0..15) of the next tile: f < 8 = im2col group 8 wave + f, else dy group 8 wave + f - 8
    auto dma_x4 = [&](const int nb, const int f, int rt_next) __attribute__((always_inline)) {
        const int z = rt_next * g.zero;
        if (f < 8) {
            const uint32_t vo = (vrow_a[f] == OOB_OFF || voff_x == OOB_OFF) ? OOB_OFF : vrow_a[f] + voff_x;
            lds_dma_row4(rs_x, lds0 + (uint32_t)(nb * BUF4 + (wave * 8 + f) * GSTR + z) * 4u, vo, 0);
        } else {
            const uint32_t vo = (vrow_b[f - 8] == OOB_OFF || voff_dy == OOB_OFF) ? OOB_OFF : vrow_b[f - 8] + voff_dy;
            lds_dma_row4(rs_dy, lds0 + (uint32_t)(nb * BUF4 + SA4 + (wave * 8 + f - 8) * GSTR + z) * 4u, vo, 0);
        }
    };
    // bias row: local row K - i0 of the LAST i-tile (uniform)
    const int ones_row = (db_slabs != nullptr && g.K >= i0 && g.K < i0 + BM) ? g.K - i0 : -1;
    auto finish_tile = [&](const int nb) __attribute__((always_inline)) {                // DMAs of buffer nb landed -> visible to every wave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (X4) {         // row r lives in group r % 32 (fetched by wave (r % 32) / 8), slot r / 32
            if (ones_row >= 0 && ((ones_row & 31) >> 3) == wave) ldsw[nb * BUF4 + (ones_row & 31) * GSTR + (ones_row >> 5) * 64 + lane] = 1.0f;
        } else {
            if (ones_row >= 0 && (ones_row / RA) == wave) ldsw[nb * BUF + ones_row * STR + lane] = 1.0f;
        }
        __syncthreads();
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

    const int rt0 = zs * rt_per_split;
    const int rt1 = min(rtiles, rt0 + rt_per_split);
    const int ntiles = rt1 - rt0;
    if (ntiles > 0) {
        decode(rt0);
        load_table(rt0);
        pin_table();
        if constexpr (X4) {
#pragma unroll
            for (int f = 0; f < 16; ++f) dma_x4(0, f, rt0);
        } else {
#pragma unroll
            for (int f = 0; f < RA + RB; ++f) dma(0, f, rt0);
        }
        decode(rt0 + 1);
        finish_tile(0);
    }
    const float* fa = ldsw + (wi0 + (lane & 31)) * STR + 2 * (lane >> 5);
    const float* fb = ldsw + SA + (wj0 + (lane & 31)) * STR + 2 * (lane >> 5);

    // one tile: MFMAs on buffer `cur` while tile rt_next streams into the other buffer.  The fetch for a tile past
    // this split's range is harmless (valid or range-checked addresses, never read) and keeps the body branch-free.
    auto tile = [&](const int cur, int rt_next) __attribute__((always_inline)) {
        float2 af[2][TM], bf[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) af[0][a] = *reinterpret_cast<const float2*>(fa + cur * BUF + a * 32 * STR);
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[0][b] = *reinterpret_cast<const float2*>(fb + cur * BUF + b * 32 * STR);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int c = t & 1, nx = c ^ 1;
            if (t + 1 < 16) {                             // fragments of the next step: their latency hides behind this step's MFMAs
#pragma unroll
                for (int a = 0; a < TM; ++a) af[nx][a] = *reinterpret_cast<const float2*>(fa + cur * BUF + a * 32 * STR + 4 * (t + 1));
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[nx][b] = *reinterpret_cast<const float2*>(fb + cur * BUF + b * 32 * STR + 4 * (t + 1));
            }
            // one row fetch in the shadow of each MFMA (64 cycles of matrix pipe): its ~7 scalar instructions and the DMA
            // issue while the pipe works.  sched_barrier pins the pairing; left alone hipcc bunches 8 fetches (~250 cycles
            // of issue) between two MFMAs and the pipe idles.
#pragma unroll
            for (int m = 0; m < 2 * TM * TN; ++m) {
                const int a = (m % (TM * TN)) / TN, b = m % TN;
                if (m < TM * TN) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][a].x, bf[c][b].x, acc[a][b], 0, 0, 0);
                else acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][a].y, bf[c][b].y, acc[a][b], 0, 0, 0);
                const int f = t * (2 * TM * TN) + m;
                if (f < RA + RB) {
                    dma(cur ^ 1, f, rt_next);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // pixel offsets of the tile after next, computed in the quiet tail of this tile (the fetches above have read
            // their address registers at issue), so that a tile starts with MFMAs + fetches, not with ~25 VALU instructions
            if (t == 10) {
                load_table(rt_next + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (t == 13) {
                int rtn = rt_next + 1;
                asm volatile("" : "+s"(rtn));             // the decode's input becomes known HERE, not at the top of the tile
                decode(rtn);
                pin_table();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        finish_tile(cur ^ 1);
    };
    // The same tile in split-product arithmetic: the reduction index is the pixel, so one 32x32x16 MFMA consumes 16 of the
    // tile's 64 pixels; lane half h owns pixels 16 s + 8 h .. + 7 of step s, which sit next to each other in the LDS row, so
    // a float2 read IS a bf16 pair after the head / tail split (5 VALU per pair, both operands: they are activations).
    // (TM + TN) x 4 pairs and 3 TM TN MFMAs per step; the row fetches ride in the shadow of the first MFMAs as above, FPM
    // of them per MFMA because a tile has 4x fewer MFMAs than in fp32.
    const float* fa3 = ldsw + (wi0 + (lane & 31)) * STR + 8 * (lane >> 5);
    const float* fb3 = ldsw + SA + (wj0 + (lane & 31)) * STR + 8 * (lane >> 5);
    auto tile_split = [&](const int cur, int rt_next) __attribute__((always_inline)) {
        constexpr int NP = split_products(MATH), NPC = split_pieces_of(MATH);   // MATH 1: heads only (plain bf16 products)
        constexpr int NMS = NP * TM * TN, NTOT = 4 * NMS, NF = X4 ? 16 : RA + RB;
        constexpr int FSPAN = NTOT >= 28 ? NTOT - 14 : NTOT / 2;      // the fetches ride on the first FSPAN MFMAs of the tile
        constexpr int FPM = (NF + FSPAN - 1) / FSPAN, FEND = (NF + FPM - 1) / FPM;
        constexpr int LT = FEND > NTOT - 12 ? FEND : NTOT - 12;       // table of the tile after next: once the fetches are out
        constexpr int PIN = LT + 2 > NTOT - 6 ? LT + 2 : NTOT - 6;
        static_assert(PIN < NTOT, "the tile must leave room to load and pin the next table");
        SplitBf16 As[2][TM], Bs[2][TN];
        auto load_split = [&](const int st, SplitBf16 (&A)[TM], SplitBf16 (&B)[TN]) __attribute__((always_inline)) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // X4: row wi0 + 32 a + i = group i, slot (wi0 + 32 a) / 32
                    const float2 v = X4 ? *reinterpret_cast<const float2*>(ldsw + cur * BUF4 + (lane & 31) * GSTR + (wi0 / 32 + a) * 64 + 16 * st +
                                                                            8 * (lane >> 5) + 2 * q)
                                        : *reinterpret_cast<const float2*>(fa3 + cur * BUF + a * 32 * STR + 16 * st + 2 * q);
                    int pc[3] = {0, 0, 0};
                    split_pieces<NPC>(v.x, v.y, pc);
#pragma unroll
                    for (int k = 0; k < NPC; ++k) A[a].p[k][q] = pc[k];
                }
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float2 v = X4 ? *reinterpret_cast<const float2*>(ldsw + cur * BUF4 + SA4 + (lane & 31) * GSTR + (wj0 / 32 + b) * 64 +
                                                                            16 * st + 8 * (lane >> 5) + 2 * q)
                                        : *reinterpret_cast<const float2*>(fb3 + cur * BUF + b * 32 * STR + 16 * st + 2 * q);
                    int pc[3] = {0, 0, 0};
                    split_pieces<NPC>(v.x, v.y, pc);
#pragma unroll
                    for (int k = 0; k < NPC; ++k) B[b].p[k][q] = pc[k];
                }
        };
        load_split(0, As[0], Bs[0]);
        static_for<0, 4>([&](auto SI) __attribute__((always_inline)) {
            constexpr int st = decltype(SI)::value, c = st & 1;
            if constexpr (st + 1 < 4) load_split(st + 1, As[c ^ 1], Bs[c ^ 1]);
            static_for<0, NMS>([&](auto MI) __attribute__((always_inline)) {
                constexpr int m = decltype(MI)::value;
                constexpr int prod = m / (TM * TN), a = (m % (TM * TN)) / TN, b = m % TN;
                acc[a][b] = mfma_bf16(As[c][a].p[kProdA[prod]], Bs[c][b].p[kProdB[prod]], acc[a][b]);
                constexpr int idx = st * NMS + m;
                static_for<0, FPM>([&](auto EI) __attribute__((always_inline)) {
                    constexpr int f = idx * FPM + decltype(EI)::value;
                    if constexpr (f < NF) {
                        if constexpr (X4) dma_x4(cur ^ 1, f, rt_next);
                        else dma(cur ^ 1, f, rt_next);
                    }
                });
                __builtin_amdgcn_sched_barrier(0x106);        // VALU / SALU / LDS reads may move across, MFMAs and fetches stay put
                if constexpr (idx == LT) load_table(rt_next + 1);
                if constexpr (idx == PIN) {
                    int rtn = rt_next + 1;
                    asm volatile("" : "+s"(rtn));
                    decode(rtn);
                    pin_table();
                }
            });
        });
        finish_tile(cur ^ 1);
    };
    auto run_tile = [&](const int cur, int rt_next) __attribute__((always_inline)) {
        if constexpr (MATH != 0) tile_split(cur, rt_next);
        else tile(cur, rt_next);
    };
    int rt = rt0;
    for (int i = 0; i + 1 < ntiles; i += 2, rt += 2) {
        run_tile(0, rt + 1);
        run_tile(1, rt + 2);
    }
    if (ntiles > 0 && (ntiles & 1)) run_tile(0, rt + 1);

    EpiRowMajor::template apply<TM, TN>(pe, zg, zs, i0 + wi0, j0 + wj0, acc);
    if (ones_row >= 0) {                                  // accumulator row K -> this split's db partial
        float* dbs = db_slabs + (int64_t)zs * db_stride + zg * d.Cog;
        const int lr = ones_row - wi0;                    // row within this wave's block of BM / WM rows
        if (lr >= 0 && lr < BM / WM) {
            const int a = lr >> 5, rr = lr & 31;          // D register q of lane l holds row (q&3) + 8*(q>>2) + 4*(l>>5)
            const int q = (rr & 3) + 4 * (rr >> 3);
            if (((rr >> 2) & 1) == (lane >> 5)) {
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    const int j = j0 + wj0 + 32 * b + (lane & 31);
                    float v = 0.f;
#pragma unroll
                    for (int aa = 0; aa < TM; ++aa)
#pragma unroll
                        for (int qq = 0; qq < 16; ++qq)
                            if (aa == a && qq == q) v = acc[aa][b][qq];
                    if (j < d.Cog) dbs[j] = v;
                }
            }
        }
    }
}

// ---- conv forward / dgrad, LDS-DMA form (padded layout, 128 x 128 tiles) -----------------------------------------
// The same pipeline as wgrad_dma_kernel applied to y[co][pixel] = sum_r W[r][co] * im2col[r][pixel]: one workgroup per
// CU, 64-deep reduction stages (= four 16-row blocks of the permuted order), two LDS buffers of 64 KB, operand rows
// fetched global -> LDS by `buffer_load_dword ... lds`, one fetch in the shadow of each MFMA of a stage's first half,
// vmcnt(0) + one barrier per stage (128 MFMAs per wave) instead of one per 16-row tile (32 MFMAs).
// Both operands land in LDS as [r][x] rows (the DMA writes a wave's 64 lanes to consecutive dwords): a weight row is 64
// consecutive output channels, an im2col row is the tile's 64 consecutive pixels of one (tap, channel).  Along the channel
// axis a wave's two 32-wide MFMA tiles are the EVEN and the ODD channels of its 64-wide block, so lane l reads the adjacent
// pair co = 2 (l & 31), +1 with one ds_read_b64 (conflict free: the two lane halves read different rows r).  Along the
// pixel axis the tiles are the two contiguous 32-pixel halves (two b32 reads): an even/odd split there made every output
// store a stride-2 half-filled line and doubled the kernel's HBM write traffic (measured 747 vs 360 MB algorithmic).
// Accumulator (a, b, q) of lane l is y[co = 64 wm + 2 i + a][pixel = 64 wn + 32 b + j], i = (q&3) + 8 (q>>2) + 4 (l>>5), j = l & 31.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at 4-byte alignment

struct ConvDmaParams {
    const float* w;          // [K][w_ld] weights (HWIO forward, flipped/transposed for dgrad)
    int64_t w_ld;
    int64_t w_grp_stride;    // elements between groups (column block)
    int64_t w_bytes;         // bytes from w to the end of the weight tensor
    const int* row_tab;      // first weight row of every 16-row block of the reduction order
    int nblk;                // ceil(K / 16) blocks
};

template <int SR>   // reduction rows per stage: 64 (one workgroup per CU) or 32 (two: one's epilogue under the other's MFMAs)
__global__ __launch_bounds__(NT, 64 / SR) void conv_dma_kernel(const ConvDmaParams pa, const ConvGeom g, const EpiConvNCHW::Params pe,
                                                               int tiles_i, int wide) {
    constexpr int BM = 128, BN = 128;
    constexpr int ABUF = SR * BM, BUF = SR * (BM + BN);               // floats: A tile, whole buffer
    constexpr int RW = SR / 4;                                        // rows of a stage each wave fetches (within ONE 16-row block)
    constexpr int NF = RW * 4;                                        // row fetches per wave per stage: RW rows x (2 A + 2 B halves)
    constexpr int NT2 = SR / 4;                                       // t-steps (two MFMA steps = four reduction rows each)
    constexpr int FPM = 1;                                            // fetches per MFMA shadow (2 measured equal: latency is not the limit)
    static_assert(SR == 64 || SR == 32, "stage depth");
    extern __shared__ __attribute__((aligned(16))) float ldsc[];
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ti_blk = bid % tiles_i, tj_blk = bid / tiles_i;
    const int zg = blockIdx.y;
    const int i0 = ti_blk * BM, j0 = tj_blk * BN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- per-lane source offsets, fixed for the workgroup: weight column (co) and im2col pixel of each 64-wide half
    const i32x4 rs_w = rsrc_words(pa.w + (int64_t)zg * pa.w_grp_stride, pa.w_bytes - (int64_t)zg * pa.w_grp_stride * 4);
    const i32x4 rs_x = rsrc_words(g.x + (int64_t)zg * g.grp_stride, (g.total - (int64_t)zg * g.grp_stride) * 4);
    uint32_t voff_a[2], voff_b[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int co = i0 + hf * 64 + lane;
        voff_a[hf] = co < pe.Cog ? (uint32_t)co * 4u : OOB_OFF;
        const int m = j0 + hf * 64 + lane;
        const bool vm = m < g.M;
        const uint32_t mm = vm ? m : 0;
        const uint32_t n = fd_div(mm, g.dOHW);
        const uint32_t p = mm - n * g.OHW;
        const uint32_t oh = fd_div(p, g.dOW);
        const uint32_t ow = p - oh * g.OW;
        const int ih0 = (int)oh * g.stride - g.pt + g.halo, iw0 = (int)ow * g.col_mul + g.col_add;
        voff_b[hf] = vm ? (uint32_t)((int64_t)n * g.img_stride + (int64_t)ih0 * g.Wp + iw0) * 4u : OOB_OFF;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)ldsc;
    const int ld_bytes = (int)(pa.w_ld * 4);

    // this wave fetches the stage's block `wave` (16 rows): their im2col offsets sit in SGPRs (see wgrad_dma_kernel)
    int tabv[RW];
    int row0 = 0;
    bool blk_ok = false;
    const int wrow = wave * RW;                                       // first stage row of this wave
    auto load_table = [&](int st) {
        const int k0 = st * SR + wrow;                                // position in the reduction order
        const const_int_ptr tt = as_const(g.ktab) + k0 + st * g.zero;
#pragma unroll
        for (int f = 0; f < RW; ++f) tabv[f] = tt[f];
        row0 = as_const(pa.row_tab)[k0 / KBLK + st * g.zero] + k0 % KBLK;
        blk_ok = k0 / KBLK < pa.nblk;                                 // blocks past K: both operands fetch zeros
    };
    auto pin_table = [&]() {
#pragma unroll
        for (int f = 0; f < RW; ++f) asm volatile("" : "+s"(tabv[f]));
        asm volatile("" : "+s"(row0));
    };
    // fetch f (0..63) of a stage into buffer nb: row rr = f >> 2 of the wave's block; f & 3 = {A half 0, A half 1, B half 0, B half 1}
    auto dma = [&](const int nb, const int f, int st) {
        const int rr = f >> 2, kind = f & 3, hf = kind & 1;
        const int z = st * g.zero;
        const uint32_t row_lds = lds0 + (uint32_t)(nb * BUF + (wrow + rr) * 128 + hf * 64 + z) * 4u;
        // reduction positions past K (the tail of a last, partial 16-block; whole blocks past the end): the im2col side
        // fetches zeros (range check) and the weight side re-reads row K - 1, so the product is exactly 0 and stays in bounds
        const bool live = blk_ok && st * SR + wrow + rr < g.K;
        if (kind < 2) lds_dma_row(rs_w, row_lds, voff_a[hf], min(row0 + rr, g.K - 1) * ld_bytes);
        else lds_dma_row(rs_x, row_lds + ABUF * 4u, live ? voff_b[hf] : OOB_OFF, tabv[rr]);
    };
    // Steady-state form of the same fetch: every reduction position of the target stage lies below K, so there is no liveness
    // select (a VALU instruction per fetch, and each VALU instruction costs MFMA issue cycles) and the weight row's offset is a
    // running scalar -- 4 scalar instructions per fetch, inside the shadow of one MFMA.  The general form serves the prologue
    // and the last stage(s).
    int a_soff = 0;
    auto dma_fast = [&](const int nb, const int f, int st) {
        const int rr = f >> 2, kind = f & 3, hf = kind & 1;
        const uint32_t row_lds = lds0 + (uint32_t)(nb * BUF + (wrow + rr) * 128 + hf * 64 + st * g.zero) * 4u;
        if (f == 0) a_soff = row0 * ld_bytes;
        if (kind < 2) lds_dma_row(rs_w, row_lds, voff_a[hf], a_soff);
        else lds_dma_row(rs_x, row_lds + ABUF * 4u, voff_b[hf], tabv[rr]);
        if (kind == 1) a_soff += ld_bytes;
    };
    auto finish_stage = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

    const int nstages = (pa.nblk * KBLK + SR - 1) / SR;
    load_table(0);
    pin_table();
#pragma unroll
    for (int f = 0; f < NF; ++f) dma(0, f, 0);
    load_table(1);
    pin_table();
    finish_stage();

    // this lane's 32 bias values, loaded ahead of the main loop (inside the epilogue every one was a dependent load in front of a store)
    float bias_r[2][16];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int co = i0 + wm * 64 + 2 * ((q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)) + a;
            bias_r[a][q] = (pe.bias && co < pe.Cog) ? pe.bias[zg * pe.Cog + co] : 0.f;
        }

    // lane half h = lane >> 5 consumes row 2 s + h at MFMA step s
    const float* fa = ldsc + (lane >> 5) * 128 + wm * 64 + 2 * (lane & 31);
    const float* fb = ldsc + ABUF + (lane >> 5) * 128 + wn * 64 + (lane & 31);

    auto stage = [&](const int cur, const bool fast, int st_next) {   // MFMAs on buffer cur while stage st_next streams into the other
        float2 af[2], bf[2], an[2], bn[2];
        auto read_b = [&](int step) {                                 // pixel tiles: [0, 32) and [32, 64) of the wave's block
            const float* q = fb + cur * BUF + step * 256;
            return float2{q[0], q[32]};
        };
        af[0] = *reinterpret_cast<const float2*>(fa + cur * BUF);
        bf[0] = read_b(0);
        af[1] = *reinterpret_cast<const float2*>(fa + cur * BUF + 256);
        bf[1] = read_b(1);
#pragma unroll
        for (int t = 0; t < NT2; ++t) {                               // t = two MFMA steps = four reduction rows
            if (t + 1 < NT2) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    an[u] = *reinterpret_cast<const float2*>(fa + cur * BUF + (2 * (t + 1) + u) * 256);
                    bn[u] = read_b(2 * (t + 1) + u);
                }
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int u = m >> 2, a = (m >> 1) & 1, b = m & 1;
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a ? af[u].y : af[u].x, b ? bf[u].y : bf[u].x, acc[a][b], 0, 0, 0);
                const int f = (t * 8 + m) * FPM;                     // FPM fetches in the shadow of each of the first NF / FPM MFMAs
                if (f < NF) {
#pragma unroll
                    for (int e = 0; e < FPM; ++e) {
                        if (fast) dma_fast(cur ^ 1, f + e, st_next);
                        else dma(cur ^ 1, f + e, st_next);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the table of the stage after next: read once this stage's fetches (t < NF / 8) no longer need the current one
            static_assert(NF / FPM / 8 <= NT2 - 4, "fetches must end before the table is replaced");
            if (t == NT2 - 4) {
                load_table(st_next + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (t == NT2 - 2) {
                pin_table();
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                af[u] = an[u];
                bf[u] = bn[u];
            }
        }
        finish_stage();
    };
    // stages 0 .. nfull-1 lie entirely below K: a stage whose prefetch target is one of them runs the fast fetch form
    const int nfull = g.K / SR;
    int st = 0;
    for (; st + 2 < nfull; st += 2) {
        stage(0, true, st + 1);
        stage(1, true, st + 2);
    }
    for (; st < nstages; ++st) {
        if (st & 1) stage(1, false, st + 1);
        else stage(0, false, st + 1);
    }

    // ---- epilogue: NCHW (+ halo), bias, ReLU, ReluGrad mask
    // dense output planes without a mask (conv2 / conv5 forward): through LDS as 16-byte stores of 4 consecutive pixels, see
    // conv_dma16_kernel (a run with the stores removed was 4-6 % faster; the launcher sizes the LDS for the [128][132] staging tile)
    if (wide) {
        constexpr int SP = BN + 4;                                     // lanes 32..63 write channel + 8: 8 * 132 = 32 mod 64 banks
        float* T = ldsc;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int i = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                    float v = acc[a][b][q] + bias_r[a][q];
                    if (pe.relu) v = fmaxf(v, 0.f);
                    T[(wm * 64 + 2 * i + a) * SP + wn * 64 + 32 * b + (lane & 31)] = v;
                }
        __syncthreads();
        const int px = 4 * (lane & 31);
        const int m = j0 + px;
        const uint32_t n = fd_div((uint32_t)min(m, pe.M - 1), pe.dOHW);
        const int p = m - (int)n * pe.OHW;
        const bool run4 = m + 3 < pe.M && p + 3 < pe.OHW;
        const int64_t cbase = ((int64_t)n * pe.Cout_total + (int64_t)zg * pe.Cog) * pe.y_plane + p;
#pragma unroll
        for (int i = 0; i < BM / 8; ++i) {                             // 2 channel rows per wave instruction, 32 rows per wave
            const int cl = wave * (BM / 4) + 2 * i + (lane >> 5);
            const int co = i0 + cl;
            const f32x4 v = *reinterpret_cast<const f32x4*>(T + cl * SP + px);
            if (co >= pe.Cog) continue;
            float* dst = pe.y + cbase + (int64_t)co * pe.y_plane;
            if (run4) {
                *reinterpret_cast<f32x4u*>(dst) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int me = m + e;
                    if (me < pe.M) {
                        const uint32_t ne = fd_div((uint32_t)me, pe.dOHW);
                        const int pe_ = me - (int)ne * pe.OHW;
                        pe.y[((int64_t)ne * pe.Cout_total + (int64_t)zg * pe.Cog + co) * pe.y_plane + pe_] = v[e];
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int m = j0 + wn * 64 + 32 * b + (lane & 31);
        if (m >= pe.M) continue;
        const uint32_t n = fd_div((uint32_t)m, pe.dOHW);
        const uint32_t p = m - n * pe.OHW;
        const uint32_t oh = fd_div(p, pe.dOW);
        const uint32_t ow = p - oh * pe.OW;
        const int64_t c0 = (int64_t)n * pe.Cout_total + (int64_t)zg * pe.Cog;
        const int64_t ybase = c0 * pe.y_plane + (int64_t)(oh + pe.y_halo) * pe.y_wp + ow + pe.y_halo;
        const int64_t mbase = c0 * pe.m_plane + (int64_t)(oh + pe.m_halo) * pe.m_wp + ow + pe.m_halo;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float mk[16];                                          // mask loads of these 16 channels first, all in flight (conv_dma16_kernel)
            if (pe.mask) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = i0 + wm * 64 + 2 * ((q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)) + a;
                    mk[q] = co < pe.Cog ? pe.mask[mbase + (int64_t)co * pe.m_plane] : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                const int co = i0 + wm * 64 + 2 * i + a;
                if (co < pe.Cog) {
                    float v = acc[a][b][q] + bias_r[a][q];
                    if (pe.relu) v = fmaxf(v, 0.f);
                    if (pe.mask) v = mk[q] > 0.f ? v : 0.f;
                    pe.y[ybase + (int64_t)co * pe.y_plane] = v;
                }
            }
        }
    }
}

// ---- conv forward / dgrad with split-bf16 products: deep LDS ring, 16-byte fetches (padded layout, 128 x 256 tiles) ---
// What bounds the split-product form is not the matrix pipe: a v_mfma_f32_32x32x16_bf16 occupies it for 32 cycles but only
// ~24 cycles of other vector issue (6 VALU / LDS instructions, shared by the SIMD's two waves) hide beside it, and an
// in-register head/tail split costs 5 VALU per operand pair.  With both operands split in the kernel (conv_dma_kernel<32, 3>)
// that is 11 VALU per MFMA and bf16x3 stops at 1.65x the fp32 kernel, whatever the fetch pipeline does (measured: removing the
// fetches changes nothing, removing the split -25 %).  So:
//  * the WEIGHT operand is split once per launch by conv_wsplit_kernel into the exact image a stage's LDS slot holds
//    ([stage][head | tail][lane half][channel][4 bf16 pairs]): the conv kernel fetches it with linear 16-byte LDS-DMA and
//    reads an MFMA operand with one ds_read_b128, no VALU;
//  * the im2col operand is fetched 16 B per lane (`buffer_load_dwordx4 ... lds`, lane l lands at M0 + 16 l): ONE reduction
//    row x 256 pixels per instruction, lane l = the 4 consecutive output pixels 4 l .. 4 l + 3.  Four consecutive output
//    pixels are four consecutive input floats only inside an image row, so the tile enumerates pixels over rows padded to
//    OWp = roundup(OW, 4) (13 -> 16, 57 -> 60): the dead pixels read whatever follows the row (halo or the next row: finite,
//    inside the buffer's range check), cost MFMA cycles and are dropped by the epilogue.  Requires unit column stride in
//    memory (stride-1 layers, or the phase-split x0 of conv1).  It is split in registers (5 VALU per pair);
//  * the LDS is spent on latency: ONE workgroup of 8 waves per CU, stages of 16 reduction rows (24 KB), a ring of NBUF = 5.
// Stage st: MFMAs on the operands of stage st (registers) while the VALU splits the im2col operands of stage st + 1 (read
// from LDS one stage ago), the LDS reads of stage st + 1 (weights) and st + 2 (im2col) are in flight, and stage st + NBUF
// streams into the slot stage st occupied.  The barrier at the end of stage st waits (vmcnt) only for stage st + 3.  Every
// wave issues exactly FW fetches per stage, in stage order, so "all but the last (NBUF - 3) FW" means stage st + 3 has landed.
// Reduction position p = 16 st + 8 h + 2 q + e of a stage belongs to lane half h, bf16 pair q, element e of the MFMA operand.

// out[group][stage][plane < planes][h][co < CogP][q] = bf16 pair of weight rows row_tab[stage] + 8 h + 2 q, + 1 (zero past K / Cog);
// plane 0 = heads, 1 = tails, 2 = what heads + tails leave over (bf16x6)
__global__ void conv_wsplit_kernel(const float* __restrict__ w, int64_t w_ld, int64_t w_grp_stride, const int* __restrict__ row_tab,
                                   int K, int Cog, int CogP, int nstages, int planes, uint32_t* __restrict__ out) {
    const int st = blockIdx.x, zg = blockIdx.y;
    const int row0 = row_tab[st];
    uint32_t* o = out + ((int64_t)zg * nstages + st) * planes * 8 * CogP;
    for (int idx = threadIdx.x; idx < 8 * CogP; idx += blockDim.x) {
        const int q = idx & 3, co = (idx >> 2) % CogP, h = idx / (4 * CogP);
        const int r = 8 * h + 2 * q, p = st * 16 + r;
        float x0 = 0.f, x1 = 0.f;
        if (co < Cog) {
            const float* src = w + (int64_t)(row0 + r) * w_ld + zg * w_grp_stride + co;
            if (p < K) x0 = src[0];
            if (p + 1 < K) x1 = src[w_ld];
        }
        int pc[3] = {0, 0, 0};
        split_pieces<3>(x0, x1, pc);
        for (int k = 0; k < planes; ++k) o[(int64_t)((2 * k + h) * CogP + co) * 4 + q] = (uint32_t)pc[k];
    }
}

struct ConvRingParams {
    const uint32_t* wsplit;  // conv_wsplit_kernel's image
    int CogP;                // its channel pitch (Cog rounded up to the 128-channel tile)
    int nstages;             // ceil(K / 16)
};

// BM = 128: 2 x 4 waves, 128 channels x 256 pixels, ring of 5 x 24 KB.  BM = 64 (48-channel groups: conv2 dgrad): 1 x 8 waves,
// 64 channels x 512 pixels, ring of 4 x 36 KB; waves 4-7 re-fetch the weight pieces of waves 0-3 (same bytes to the same place)
// so that every wave issues the same number of fetches per stage.
template <int MATH, int BM>
__global__ __launch_bounds__(512, 1) void conv_ring_kernel(const ConvRingParams pa, const ConvGeom g, const EpiConvNCHW::Params pe,
                                                           int tiles_i, int OWp, int Mp, FastDiv dOHWp, FastDiv dOWp) {
    constexpr int BN = BM == 128 ? 256 : 512, SR = 16, NBUF = BM == 128 ? 5 : 4;
    constexpr int NH = BN / 256;                                      // 256-pixel fetches per im2col row
    static_assert(BM == 128 || BM == 64, "channel tile");
    constexpr int NPL = MATH == 6 ? 3 : 2;                            // planes of the weight image (heads, tails[, leftovers])
    constexpr int ABUF = NPL * 8 * BM, BUF = ABUF + SR * BN;          // dwords
    constexpr int NAF = (NPL * 2 * (BM / 64) + 7) / 8;                // weight pieces (1 KB) each wave fetches per stage: 1, or 2 (bf16x6, BM = 128)
    constexpr int FW = NAF + 2 * NH;                                  // fetches per wave per stage: weight pieces, im2col rows 2w, 2w + 1
    constexpr int VMW = (NBUF - 3) * FW;                              // fetches that may stay in flight at a stage's end: NBUF - 3 stages
    constexpr int NP = split_products(MATH), NPC = split_pieces_of(MATH);
    constexpr int NM = (BM / 32) * (BN / 256) * NP;                   // MFMAs per wave per stage (TA x TB blocks x products)
    static_assert(SR == KBLK, "one stage = one block of the reduction order");
    extern __shared__ __attribute__((aligned(16))) float ldsr[];
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ti_blk = bid % tiles_i, tj_blk = bid / tiles_i;
    const int zg = blockIdx.y;
    const int i0 = ti_blk * BM, j0 = tj_blk * BN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // A wave's tile is TA x TB blocks of 32 x 32: all BM channels x 32 pixels (BM = 128: 4 x 1) or x 64 pixels (BM = 64: 2 x 2).
    // Every im2col element is then split by ONE wave only (in a 2 x 4 wave grid two waves split the same pixels, and the
    // split, not the MFMA, is what a stage's time is made of).
    constexpr int TA = BM / 32, TB = BN / 8 / 32, WPX = 32 * TB;     // blocks along channels / pixels; pixels per wave
    const int OHWp = g.OHW / g.OW * OWp;
    const int nstages = pa.nstages;

    // weights: wave w fetches one 1 KB piece of a stage's image, 16 B per lane.  BM = 128: piece w = (plane w >> 2, lane half
    // (w >> 1) & 1, channels 64 (w & 1) ..+63); BM = 64: piece w & 3 = (plane, lane half), 64 channels.  It lands at byte
    // 1024 piece of the slot, i.e. the slot holds [plane][h][BM channels][4 pairs]
    const int64_t stage_dw = (int64_t)NPL * 8 * pa.CogP;              // dwords per stage of the image
    const uint32_t* wbase = pa.wsplit + (int64_t)zg * nstages * stage_dw;
    const i32x4 rs_w = rsrc_words(reinterpret_cast<const float*>(wbase), nstages * stage_dw * 4);
    // a stage's image is NPL * 2 * (BM / 64) pieces of 1 KB; wave w takes piece w (mod the count: surplus waves re-fetch a piece,
    // same bytes to the same place) and, with three planes of 128 channels (12 pieces), also piece 8 + (w & 3)
    constexpr int NPIECE = NPL * 2 * (BM / 64);
    int piece[NAF];
    uint32_t voff_a[NAF];
#pragma unroll
    for (int e = 0; e < NAF; ++e) {
        piece[e] = e == 0 ? wave % NPIECE : 8 + (wave & 3);
        voff_a[e] = BM == 128 ? (uint32_t)(((piece[e] >> 1) * pa.CogP + i0 + (piece[e] & 1) * 64 + lane) * 16)
                              : (uint32_t)((piece[e] * pa.CogP + i0 + lane) * 16);
    }
    const int stage_bytes = (int)(stage_dw * 4);
    const i32x4 rs_x = rsrc_words(g.x + (int64_t)zg * g.grp_stride, (g.total - (int64_t)zg * g.grp_stride) * 4);
    // pixels 256 hf + 4 lane .. + 3 of the tile, in the padded-row enumeration
    uint32_t voff_b[NH];
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
        const int m = j0 + 256 * hf + 4 * lane;
        const bool vm = m < Mp;
        const uint32_t mm = vm ? m : 0;
        const uint32_t n = fd_div(mm, dOHWp);
        const uint32_t p = mm - n * OHWp;
        const uint32_t oh = fd_div(p, dOWp);
        const uint32_t ow = p - oh * OWp;
        const int ih0 = (int)oh * g.stride - g.pt + g.halo, iw0 = (int)ow * g.col_mul + g.col_add;
        voff_b[hf] = vm ? (uint32_t)((int64_t)n * g.img_stride + (int64_t)ih0 * g.Wp + iw0) * 4u : OOB_OFF;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)ldsr;

    // this wave fetches im2col rows 2 wave, 2 wave + 1 of every stage; their offsets sit in SGPRs
    int tab0 = 0, tab1 = 0;
    const int wrow = wave * 2;
    auto load_table = [&](int st) __attribute__((always_inline)) {
        const int sc = min(st, nstages - 1) + st * g.zero;            // stages past the end are never fetched: any valid entry
        const const_int_ptr tt = as_const(g.ktab) + sc * SR + wrow;
        tab0 = tt[0];
        tab1 = tt[1];
    };
    auto pin_table = [&]() __attribute__((always_inline)) {
        asm volatile("" : "+s"(tab0));
        asm volatile("" : "+s"(tab1));
    };
    // fetch f of stage st into the ring slot at byte offset wr: f < NAF = a weight piece; NAF + NH r + hf = pixels 256 hf ..+255
    // of im2col row 2w + r.
    // Reduction positions past K (tail of the last stage): the weight image holds zeros there and the im2col side fetches
    // zeros through the per-lane range check (an offset of OOB_OFF).
    auto dma = [&](const uint32_t wr, const int f, int st) __attribute__((always_inline)) {
        if (f < NAF) {
            lds_dma_row4(rs_w, lds0 + wr + (uint32_t)piece[f] * 1024u, voff_a[f], st * stage_bytes);
        } else {
            const int r = (f - NAF) / NH, hf = (f - NAF) % NH;
            const bool live = st * SR + wrow + r < g.K;
            lds_dma_row4(rs_x, lds0 + wr + (uint32_t)(ABUF + (wrow + r) * BN + 256 * hf) * 4u, live ? voff_b[hf] : OOB_OFF,
                         r == 0 ? tab0 : tab1);
        }
    };
    auto dma_fast = [&](const uint32_t wr, const int f, int st) __attribute__((always_inline)) {   // every position of the stage lies below K
        if (f < NAF) {
            lds_dma_row4(rs_w, lds0 + wr + (uint32_t)piece[f] * 1024u, voff_a[f], st * stage_bytes);
        } else {
            const int r = (f - NAF) / NH, hf = (f - NAF) % NH;
            lds_dma_row4(rs_x, lds0 + wr + (uint32_t)(ABUF + (wrow + r) * BN + 256 * hf) * 4u, voff_b[hf], r == 0 ? tab0 : tab1);
        }
    };

    f32x16 acc[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

    // lane half h = lane >> 5 owns reduction rows 8 h .. 8 h + 7 of a stage.  Weights: block a = channels 32 a + (lane & 31); its
    // operand is the 16 B at [plane][h][channel].  im2col: rows 8 h + j, pixel WPX wave + 32 b + (lane & 31).
    const float* fa = ldsr + ((lane >> 5) * BM + (lane & 31)) * 4;
    const float* fb = ldsr + ABUF + (lane >> 5) * 8 * BN + wave * WPX + (lane & 31);
    struct Raw {                                                      // a stage's im2col operands of this lane as read from LDS
        float b[TB][8];
    };
    auto load_raw = [&](int rd, Raw& r) __attribute__((always_inline)) {                  // rd: ring slot offset in dwords
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* q = fb + rd + j * BN;
#pragma unroll
            for (int b = 0; b < TB; ++b) r.b[b][j] = q[32 * b];
        }
    };
    auto load_a = [&](int rd, SplitBf16 (&A)[TA]) __attribute__((always_inline)) {       // the weight operands come split: no VALU
#pragma unroll
        for (int a = 0; a < TA; ++a) {
#pragma unroll
            for (int k = 0; k < NPC; ++k) A[a].p[k] = *reinterpret_cast<const i32x4*>(fa + rd + k * 2 * BM * 4 + a * 32 * 4);
        }
    };
    auto split = [&](const Raw& r, SplitBf16 (&B)[TB]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                int pc[3] = {0, 0, 0};
                split_pieces<NPC>(r.b[b][2 * j], r.b[b][2 * j + 1], pc);
#pragma unroll
                for (int k = 0; k < NPC; ++k) B[b].p[k][j] = pc[k];
            }
    };

    // ---- prologue: stages 0 .. NBUF-1 in flight; stages 0, 1, 2 landed -> stage 0's operands in registers (weights as
    // fetched, im2col split), stage 1's im2col operands raw in registers
    for (int s0 = 0; s0 < NBUF; ++s0) {
        load_table(s0);
        pin_table();
        if (s0 < nstages) {
#pragma unroll
            for (int f = 0; f < FW; ++f) dma((uint32_t)(s0 * BUF * 4), f, s0);
        }
    }
    load_table(NBUF);
    pin_table();
    if (nstages >= NBUF) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");      // groups 0, 1, 2 of NBUF done
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    SplitBf16 A[TA] = {}, B[TB] = {}, An[TA] = {}, Bn[TB] = {};
    Raw R0 = {}, R1 = {};
    load_a(0, A);
    load_raw(0, R0);
    load_raw(BUF, R1);
    split(R0, B);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // Stage st: MFMAs on the operands of stage st (Ac / Bc) while the weight operands of stage st + 1 are read from LDS into
    // An, the VALU splits the im2col operands of stage st + 1 (Rc, read one stage ago) into Bn, the LDS reads of the im2col
    // operands of stage st + 2 fill Rn, and stage st + NBUF streams into the slot of stage st (its weights were read one stage
    // ago, its im2col rows two).  The barrier at the end of stage st waits (vmcnt) for stage st + 3 only.  The register sets
    // swap roles every stage.
    int rda = BUF, rdb = 2 * BUF;                                     // ring slots (dwords) of stages st + 1, st + 2
    uint32_t wr = 0;                                                  // ring slot (bytes) of stage st + NBUF = the one of stage st
    constexpr int D = NBUF;
    auto stage = [&](const bool fast, int st, SplitBf16 (&Ac)[TA], SplitBf16 (&Bc)[TB], SplitBf16 (&An)[TA], SplitBf16 (&Bn)[TB],
                     const Raw& Rc, Raw& Rn) __attribute__((always_inline)) {
        load_a(rda, An);
        load_raw(rdb, Rn);                                            // garbage past the end: never multiplied
        static_for<0, (NM > FW ? NM : FW)>([&](auto MI) __attribute__((always_inline)) {
            constexpr int m = decltype(MI)::value;
            if constexpr (m < NM) {
                constexpr int prod = m / (TA * TB), a = (m % (TA * TB)) / TB, b = m % TB;
                acc[a][b] = mfma_bf16(Ac[a].p[kProdA[prod]], Bc[b].p[kProdB[prod]], acc[a][b]);
            }
            if constexpr (m < FW) {                                   // one fetch in the shadow of each of the first FW MFMAs
                if (fast) dma_fast(wr, m, st + D);
                else if (st + D < nstages) dma(wr, m, st + D);
            }
            __builtin_amdgcn_sched_barrier(0x106);                    // VALU / SALU / LDS reads may move across, MFMAs and fetches stay put
        });
        split(Rc, Bn);
        load_table(st + D + 1);
        pin_table();
        rda = rda + BUF == NBUF * BUF ? 0 : rda + BUF;
        rdb = rdb + BUF == NBUF * BUF ? 0 : rdb + BUF;
        wr = wr + BUF * 4 == NBUF * BUF * 4 ? 0u : wr + BUF * 4;
    };
    auto finish = [&](int st) __attribute__((always_inline)) {
        // groups issued so far end with stage min(st + D, nstages - 1); stage st + 3 must have landed
        const int after = min(st + D, nstages - 1) - (st + 3);
        if (after >= 2 && NBUF - 3 >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * FW) : "memory");
        else if (after >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(FW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // stage st + D lies entirely below K (and exists) for st + D < nfull: the fast fetch form
    const int nfull = g.K / SR;
    int st = 0;
    for (; st + 1 + D < nfull; st += 2) {
        stage(true, st, A, B, An, Bn, R1, R0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VMW) : "memory");
        __syncthreads();
        stage(true, st + 1, An, Bn, A, B, R0, R1);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(VMW) : "memory");
        __syncthreads();
    }
    for (; st < nstages; st += 2) {
        stage(false, st, A, B, An, Bn, R1, R0);
        finish(st);
        if (st + 1 < nstages) {
            stage(false, st + 1, An, Bn, A, B, R0, R1);
            finish(st + 1);
        }
    }

    // ---- epilogue: NCHW (+ halo), bias, ReLU, ReluGrad mask.  Accumulator (a, b, q) of lane l: channel 32 a + i,
    // i = (q & 3) + 8 (q >> 2) + 4 (l >> 5); pixel WPX wave + 32 b + (l & 31) of the padded-row enumeration
#pragma unroll
    for (int b = 0; b < TB; ++b) {
        const int m = j0 + wave * WPX + 32 * b + (lane & 31);
        if (m >= Mp) continue;
        const uint32_t n = fd_div((uint32_t)m, dOHWp);
        const uint32_t p = m - n * OHWp;
        const uint32_t oh = fd_div(p, dOWp);
        const uint32_t ow = p - oh * OWp;
        if ((int)ow >= pe.OW) continue;                                // dead pixel of the row padding
        const int64_t c0 = (int64_t)n * pe.Cout_total + (int64_t)zg * pe.Cog;
        const int64_t ybase = c0 * pe.y_plane + (int64_t)(oh + pe.y_halo) * pe.y_wp + ow + pe.y_halo;
        const int64_t mbase = c0 * pe.m_plane + (int64_t)(oh + pe.m_halo) * pe.m_wp + ow + pe.m_halo;
#pragma unroll
        for (int a = 0; a < TA; ++a) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                const int co = i0 + 32 * a + i;
                if (co < pe.Cog) {
                    float v = acc[a][b][q];
                    if (pe.bias) v += pe.bias[zg * pe.Cog + co];
                    if (pe.relu) v = fmaxf(v, 0.f);
                    if (pe.mask) v = pe.mask[mbase + (int64_t)co * pe.m_plane] > 0.f ? v : 0.f;
                    pe.y[ybase + (int64_t)co * pe.y_plane] = v;
                }
            }
        }
    }
}

// ---- the 128-channel split-product kernel with TWO workgroups per CU ---------------------------------------------------
// conv_ring_kernel<3, 128> reads the whole weight slab in every one of its 8 waves (a wave tile is 128 channels x 32 pixels):
// 72 KB of LDS reads per 96 MFMAs, and with one workgroup per CU nothing runs while its waves sit at a barrier.  Here a
// workgroup is 4 waves, each 128 channels x 64 pixels (4 x 2 blocks, 128 accumulator registers): 48 KB of LDS reads per 96
// MFMAs, 8 split pairs per 24 MFMAs, and two workgroups share a CU (3 slots of 24 KB each), one's barrier under the other's
// MFMAs.  Two-level pipeline: during stage st the operands of stage st + 1 are read (weights: straight into MFMA tuples) and
// split, and stage st + 3 streams into the slot of stage st; the barrier at the end of stage st waits for stage st + 2.
// Same weight image, same im2col fetch (16 B per lane, padded pixel rows) as conv_ring_kernel.
template <int MATH>
__global__ __launch_bounds__(256, 2) void conv_ring4_kernel(const ConvRingParams pa, const ConvGeom g, const EpiConvNCHW::Params pe,
                                                            int tiles_i, int OWp, int Mp, FastDiv dOHWp, FastDiv dOWp) {
    constexpr int BM = 128, BN = 256, SR = 16, NBUF = 3, D = 3;
    constexpr int ABUF = SR * BM, BUF = SR * (BM + BN);               // dwords
    constexpr int TA = 4, TB = 2, WPX = 64;
    constexpr int FW = 6;                                             // fetches per wave per stage: 2 weight pieces, im2col rows 4w .. 4w + 3
    constexpr int NP = split_products(MATH), NPC = split_pieces_of(MATH);
    static_assert(NPC <= 2, "the two-workgroup form has LDS for two weight planes");
    constexpr int NM = TA * TB * NP;
    static_assert(SR == KBLK, "one stage = one block of the reduction order");
    extern __shared__ __attribute__((aligned(16))) float ldsq[];
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ti_blk = bid % tiles_i, tj_blk = bid / tiles_i;
    const int zg = blockIdx.y;
    const int i0 = ti_blk * BM, j0 = tj_blk * BN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int OHWp = g.OHW / g.OW * OWp;
    const int nstages = pa.nstages;

    const int64_t stage_dw = (int64_t)16 * pa.CogP;
    const uint32_t* wbase = pa.wsplit + (int64_t)zg * nstages * stage_dw;
    const i32x4 rs_w = rsrc_words(reinterpret_cast<const float*>(wbase), nstages * stage_dw * 4);
    // wave w fetches pieces 2w, 2w + 1 of a stage's weight image = (plane w >> 1, lane half w & 1), channels 0-63 / 64-127 of the tile
    uint32_t voff_a[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) voff_a[hf] = (uint32_t)((wave * pa.CogP + i0 + hf * 64 + lane) * 16);
    const int stage_bytes = (int)(stage_dw * 4);
    const i32x4 rs_x = rsrc_words(g.x + (int64_t)zg * g.grp_stride, (g.total - (int64_t)zg * g.grp_stride) * 4);
    uint32_t voff_b;
    {
        const int m = j0 + 4 * lane;
        const bool vm = m < Mp;
        const uint32_t mm = vm ? m : 0;
        const uint32_t n = fd_div(mm, dOHWp);
        const uint32_t p = mm - n * OHWp;
        const uint32_t oh = fd_div(p, dOWp);
        const uint32_t ow = p - oh * OWp;
        const int ih0 = (int)oh * g.stride - g.pt + g.halo, iw0 = (int)ow * g.col_mul + g.col_add;
        voff_b = vm ? (uint32_t)((int64_t)n * g.img_stride + (int64_t)ih0 * g.Wp + iw0) * 4u : OOB_OFF;
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)ldsq;

    int tab[4] = {0, 0, 0, 0};
    const int wrow = wave * 4;
    auto load_table = [&](int st) __attribute__((always_inline)) {
        const int sc = min(st, nstages - 1) + st * g.zero;
        const const_int_ptr tt = as_const(g.ktab) + sc * SR + wrow;
#pragma unroll
        for (int r = 0; r < 4; ++r) tab[r] = tt[r];
    };
    auto pin_table = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+s"(tab[r]));
    };
    // fetch f of stage st into the slot at byte offset wr: 0, 1 = weight pieces 2w, 2w + 1; 2 + r = im2col row 4w + r
    auto dma = [&](const uint32_t wr, const int f, int st, const bool fast) __attribute__((always_inline)) {
        if (f < 2) {
            lds_dma_row4(rs_w, lds0 + wr + (uint32_t)(2 * wave + f) * 1024u, voff_a[f], st * stage_bytes);
        } else {
            const int r = f - 2;
            const bool live = fast || st * SR + wrow + r < g.K;
            lds_dma_row4(rs_x, lds0 + wr + (uint32_t)(ABUF + (wrow + r) * BN) * 4u, live ? voff_b : OOB_OFF, tab[r]);
        }
    };

    f32x16 acc[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

    const float* fa = ldsq + ((lane >> 5) * BM + (lane & 31)) * 4;
    const float* fb = ldsq + ABUF + (lane >> 5) * 8 * BN + wave * WPX + (lane & 31);
    auto load_ops = [&](int rd, SplitBf16 (&A)[TA], float (&rb)[TB][8]) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < TA; ++a) {
#pragma unroll
            for (int k = 0; k < NPC; ++k) A[a].p[k] = *reinterpret_cast<const i32x4*>(fa + rd + k * 2 * BM * 4 + a * 32 * 4);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int b = 0; b < TB; ++b) rb[b][j] = fb[rd + j * BN + 32 * b];
    };
    auto split = [&](const float (&rb)[TB][8], SplitBf16 (&B)[TB]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                int pc[3] = {0, 0, 0};
                split_pieces<NPC>(rb[b][2 * j], rb[b][2 * j + 1], pc);
#pragma unroll
                for (int k = 0; k < NPC; ++k) B[b].p[k][j] = pc[k];
            }
    };

    // ---- prologue: stages 0, 1, 2 in flight; stage 0 landed -> its operands in registers; stage 1 landed
    for (int s0 = 0; s0 < NBUF; ++s0) {
        load_table(s0);
        pin_table();
        if (s0 < nstages) {
#pragma unroll
            for (int f = 0; f < FW; ++f) dma((uint32_t)(s0 * BUF * 4), f, s0, false);
        }
    }
    load_table(NBUF);
    pin_table();
    if (nstages >= NBUF) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * FW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    SplitBf16 A[TA] = {}, B[TB] = {}, An[TA] = {}, Bn[TB] = {};
    {
        float rb[TB][8];
        load_ops(0, A, rb);
        split(rb, B);
    }
    if (nstages >= NBUF) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(FW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    int rd = BUF;                                                     // slot (dwords) of stage st + 1
    uint32_t wr = 0;                                                  // slot (bytes) of stage st + 3 = the one of stage st
    auto stage = [&](const bool fast, int st, SplitBf16 (&Ac)[TA], SplitBf16 (&Bc)[TB], SplitBf16 (&An)[TA], SplitBf16 (&Bn)[TB])
                     __attribute__((always_inline)) {
        float rb[TB][8];
        load_ops(rd, An, rb);                                         // garbage past the end: never multiplied
        static_assert(NM >= FW, "a fetch rides on each of the first FW MFMAs");
        static_for<0, NM>([&](auto MI) __attribute__((always_inline)) {
            constexpr int m = decltype(MI)::value;
            constexpr int prod = m / (TA * TB), a = (m % (TA * TB)) / TB, b = m % TB;
            acc[a][b] = mfma_bf16(Ac[a].p[kProdA[prod]], Bc[b].p[kProdB[prod]], acc[a][b]);
            if constexpr (m < FW) {
                if (fast || st + D < nstages) dma(wr, m, st + D, fast);
            }
            __builtin_amdgcn_sched_barrier(0x106);
        });
        split(rb, Bn);
        load_table(st + D + 1);
        pin_table();
        rd = rd + BUF == NBUF * BUF ? 0 : rd + BUF;
        wr = wr + BUF * 4 == NBUF * BUF * 4 ? 0u : wr + BUF * 4;
    };
    auto finish = [&](int st) __attribute__((always_inline)) {
        // groups issued so far end with stage min(st + D, nstages - 1); stage st + 2 must have landed
        if (min(st + D, nstages - 1) - (st + 2) >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(FW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    };
    const int nfull = g.K / SR;
    int st = 0;
    for (; st + 1 + D < nfull; st += 2) {
        stage(true, st, A, B, An, Bn);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(FW) : "memory");
        __syncthreads();
        stage(true, st + 1, An, Bn, A, B);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(FW) : "memory");
        __syncthreads();
    }
    for (; st < nstages; st += 2) {
        stage(false, st, A, B, An, Bn);
        finish(st);
        if (st + 1 < nstages) {
            stage(false, st + 1, An, Bn, A, B);
            finish(st + 1);
        }
    }

    // ---- epilogue (as conv_ring_kernel): channel 32 a + i, pixel 64 wave + 32 b + (l & 31) of the padded-row enumeration
#pragma unroll
    for (int b = 0; b < TB; ++b) {
        const int m = j0 + wave * WPX + 32 * b + (lane & 31);
        if (m >= Mp) continue;
        const uint32_t n = fd_div((uint32_t)m, dOHWp);
        const uint32_t p = m - n * OHWp;
        const uint32_t oh = fd_div(p, dOWp);
        const uint32_t ow = p - oh * OWp;
        if ((int)ow >= pe.OW) continue;
        const int64_t c0 = (int64_t)n * pe.Cout_total + (int64_t)zg * pe.Cog;
        const int64_t ybase = c0 * pe.y_plane + (int64_t)(oh + pe.y_halo) * pe.y_wp + ow + pe.y_halo;
        const int64_t mbase = c0 * pe.m_plane + (int64_t)(oh + pe.m_halo) * pe.m_wp + ow + pe.m_halo;
#pragma unroll
        for (int a = 0; a < TA; ++a) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                const int co = i0 + 32 * a + i;
                if (co < pe.Cog) {
                    float v = acc[a][b][q];
                    if (pe.bias) v += pe.bias[zg * pe.Cog + co];
                    if (pe.relu) v = fmaxf(v, 0.f);
                    if (pe.mask) v = pe.mask[mbase + (int64_t)co * pe.m_plane] > 0.f ? v : 0.f;
                    pe.y[ybase + (int64_t)co * pe.y_plane] = v;
                }
            }
        }
    }
}

// out[e] = sum_s slab[s][e] (+bias[e % n_cols]) (relu) (mask) : deterministic split reduction.
// WAYS = 1: one thread per element over all slabs -- few slabs, many elements (a conv layer's weight gradient, split-k products).
// WAYS = 4: a workgroup owns 64 consecutive elements; its four waves sum a quarter of the slabs each (whole-line loads, four
// independent chains) and wave 0 adds the four partial sums in a fixed order -- many slabs, few elements (conv1's 35 k-element,
// ~85-slab reduction took 40 us with one thread per element, 6 us this way).
template <int WAYS>
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t count, int splits,
                                                          int64_t slab_stride, const float* __restrict__ bias, int ncols, int64_t ldc,
                                                          int relu, const float* __restrict__ mask) {
    __shared__ float part[3][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int EPB = WAYS == 4 ? 64 : 256;                      // elements per workgroup
    const int per = WAYS == 4 ? (splits + 3) >> 2 : splits;
    const int z0 = WAYS == 4 ? w * per : 0, z1 = min(splits, z0 + per);
    for (int64_t e0 = (int64_t)blockIdx.x * EPB; e0 < count; e0 += (int64_t)gridDim.x * EPB) {
        const int64_t e = e0 + (WAYS == 4 ? lane : (int)threadIdx.x);
        float s = 0.f;
        if (e < count) {
            const float* p = ws + (int64_t)z0 * slab_stride + e;
            int z = z0;
            for (; z + 4 <= z1; z += 4, p += 4 * slab_stride) {
                const float a = p[0], b = p[slab_stride], c = p[2 * slab_stride], d = p[3 * slab_stride];
                s += a; s += b; s += c; s += d;
            }
            for (; z < z1; ++z, p += slab_stride) s += p[0];
        }
        if constexpr (WAYS == 4) {
            if (w > 0) part[w - 1][lane] = s;
            __syncthreads();
            if (w == 0) s = (s + part[0][lane]) + (part[1][lane] + part[2][lane]);
        }
        if ((WAYS == 1 || w == 0) && e < count) {
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
        if constexpr (WAYS == 4) __syncthreads();
    }
}

static void launch_reduce_slabs(hipStream_t s, const float* ws, float* out, int64_t count, int splits, int64_t slab_stride, const float* bias,
                                int ncols, int64_t ldc, int relu, const float* mask) {
    static const int force = vl_exp_env("VL_REDUCE_WAYS") ? atoi(vl_exp_env("VL_REDUCE_WAYS")) : 0;      // A/B: 1 or 4
    const bool four = force ? force == 4 : (splits >= 16 && count * 4 <= (int64_t)splits * 65536);
    if (four) {
        const int blocks = (int)((count + 63) / 64 < 16384 ? (count + 63) / 64 : 16384);
        hipLaunchKernelGGL(reduce_slabs_kernel<4>, dim3(blocks), dim3(256), 0, s, ws, out, count, splits, slab_stride, bias, ncols, ldc, relu, mask);
    } else {
        const int blocks = (int)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
        hipLaunchKernelGGL(reduce_slabs_kernel<1>, dim3(blocks), dim3(256), 0, s, ws, out, count, splits, slab_stride, bias, ncols, ldc, relu, mask);
    }
}

// A/B switches for measurements, read once: VL_CONV_STAGED=1 runs conv forward / dgrad on the register-staged mfma_contract
// template instead of the LDS-DMA kernels, VL_GEMM_NOSPLIT=1 disables the split-K of small dense GEMMs.
static const bool kConvStaged = vl_exp_env("VL_CONV_STAGED") != nullptr;
static const bool kGemmNoSplit = vl_exp_env("VL_GEMM_NOSPLIT") != nullptr;
static const bool kConvNoWideStore = vl_exp_env("VL_CONV_NO_WIDE_STORE") != nullptr;   // A/B: per-accumulator stores in conv_dma16_kernel's epilogue
static const bool kConvNoTailSplit = vl_exp_env("VL_CONV_NO_TAIL_SPLIT") != nullptr;   // A/B: never split a few-frame launch's last round off
static const bool kConvNoLoadPick = vl_exp_env("VL_CONV_NO_LOAD_PICK") != nullptr;   // A/B: tile width by padded rows only (round-1 rule)
static const bool kWgradDword = vl_exp_env("VL_WGRAD_DWORD") != nullptr;  // split-product wgrad: keep the dword fetches where 16-byte ones apply
static const bool kRing8 = vl_exp_env("VL_CONV_RING8") != nullptr;      // bf16x3, 128-channel layers: the 8-wave conv_ring_kernel instead of conv_ring4_kernel

// contraction arithmetic of the three conv kernels: 0 = fp32 MFMA (default, the parity path), 3 = bf16x3 and 6 = bf16x6 split
// products, 1 = plain bf16 products (heads only) (vl_set_conv_math; VL_CONV_MATH=bf16x3 | bf16x6 | bf16 presets it)
static int g_conv_math = [] {
    const char* e = vl_exp_env("VL_CONV_MATH");
    return e == nullptr ? 0 : strcmp(e, "bf16x3") == 0 ? 3 : strcmp(e, "bf16x6") == 0 ? 6 : strcmp(e, "bf16") == 0 ? 1 : 0;
}();

extern "C" int vl_set_conv_math(int math) {
    VL_CHECK(math == 0 || math == 1 || math == 3 || math == 6, "vl_set_conv_math: 0 (fp32), 1 (bf16), 3 (bf16x3) or 6 (bf16x6)");
    g_conv_math = math;
    return 0;
}

extern "C" int vl_conv_math(void) { return g_conv_math; }

// ---- convolution descriptor -------------------------------------------------------------------
// struct vl_conv_desc: conv_desc.h (shared with conv_c8.hip)

static void tf_same_pad(int in, int k, int s, int* out, int* before, int* after) {
    *out = (in + s - 1) / s;
    int total = (*out - 1) * s + k - in;
    if (total < 0) total = 0;
    *before = total / 2;
    *after = total - *before;
}

// Reduction-order position k of a conv contraction -> weight / im2col row (ky, kx, c) (the HWIO row index is
// (ky * kw + kx) * cg + c).
//   natural  : k = (ky, kx, c), c fastest -- the HWIO row order; wgrad uses it (its rows are the output rows).
//   permuted : k = (c / 16, ky, kx, c % 16) when cg % 16 == 0 -- forward and dgrad.  A 16-row reduction tile is then ONE
//              tap of ONE 16-channel block, and the kh*kw tiles that follow it are the other taps of the same block: they
//              re-touch the same ~(16 channels x pixel-tile footprint) = 10-20 KB of the input, which stays in L1/L2.
//              In natural order a tile sweeps ALL cg channels before the next tap returns to them; with 128 workgroups
//              per XCD that is a 7-19 MB working set against a 4 MB L2 (measured: conv2 dgrad fetched 19 GB per launch,
//              18x its input, at 4 TB/s -- HBM-bound).  Each 16-row tile is still 16 CONSECUTIVE weight rows.

static void k_to_row(int k, int kh, int kw, int cg, bool permuted, int* ky, int* kx, int* c) {
    if (permuted) {
        const int cl = k % KBLK, t = k / KBLK;
        *kx = t % kw;
        *ky = (t / kw) % kh;
        *c = (t / (kw * kh)) * KBLK + cl;
    } else {
        *c = k % cg;
        *kx = (k / cg) % kw;
        *ky = k / (cg * kw);
    }
}

static void free_dev(void* p) {
    if (p) (void)hipFree(p);
}

template <class T>
static int upload(T** dev, const T* host, size_t count) {
    free_dev(*dev);
    *dev = nullptr;
    hipError_t e = hipMalloc((void**)dev, sizeof(T) * count);
    if (e == hipSuccess) e = hipMemcpy(*dev, host, sizeof(T) * count, hipMemcpyHostToDevice);
    return e == hipSuccess ? 0 : 2;
}

// Gather tables over planes of physical size (H + 2 halo) x (W + 2 halo) in the given reduction order:
//   dev1[k] = byte offset of tap k                        (padded mode)
//   dev2[k] = {byte offset, ky << 16 | kx}                (checked mode)
//   rowtab[t] = first HWIO weight row of reduction tile t (16 consecutive rows), when rowtab != null
//   phase > 1 (padded mode only): the tensor is stored column-phase-split, [c][phase][H + 2 halo][ceil((W + 2 halo) / phase)]
//   with physical column iw at [iw % phase][iw / phase]; tap kx of output column ow is physical column ow * phase + kx +
//   col_shift, so its phase and its offset within the phase plane depend on the tap only -> still "table entry + pixel".
static int build_ktabs(int** dev1, int2** dev2, int** rowtab, int kh, int kw, int cg, int H, int W, int halo, bool permuted,
                       int phase = 1, int col_shift = 0) {
    const int K = kh * kw * cg;
    const int pad = ((K + 127) / 128) * 128 + 256;
    const int Wp = phase > 1 ? (W + 2 * halo + phase - 1) / phase : W + 2 * halo;
    const int64_t Pp = (int64_t)(H + 2 * halo) * Wp;
    const int ntiles = pad / KBLK;
    int* h1 = (int*)malloc(sizeof(int) * pad);
    int2* h2 = (int2*)malloc(sizeof(int2) * pad);
    int* h3 = (int*)malloc(sizeof(int) * ntiles);
    if (!h1 || !h2 || !h3) {
        free(h1);
        free(h2);
        free(h3);
        return 1;
    }
    for (int k = 0; k < pad; ++k) {
        if (k < K) {
            int ky, kx, c;
            k_to_row(k, kh, kw, cg, permuted, &ky, &kx, &c);
            int64_t off = ((int64_t)c * Pp + (int64_t)ky * Wp + kx) * 4;
            if (phase > 1) off = (((int64_t)c * phase + (kx + col_shift) % phase) * Pp + (int64_t)ky * Wp + (kx + col_shift) / phase) * 4;
            h1[k] = (int)off;
            h2[k].x = (int)off;
            h2[k].y = (ky << 16) | kx;
            if (k % KBLK == 0) h3[k / KBLK] = (ky * kw + kx) * cg + c;
        } else {
            h1[k] = 0;               // padded mode: a valid address; the weight row is zero (range check)
            h2[k].x = 0;
            h2[k].y = 0x4000 << 16;  // checked mode: fails the row test
            if (k % KBLK == 0) h3[k / KBLK] = 0;
        }
    }
    int rc = upload(dev1, h1, pad);
    if (rc == 0) rc = upload(dev2, h2, pad);
    if (rc == 0 && rowtab) rc = upload(rowtab, h3, ntiles);
    free(h1);
    free(h2);
    free(h3);
    return rc;
}

static int rebuild_tables(vl_conv_desc* d) {
    d->fwd_padded = d->x_halo >= d->pt && d->x_halo >= d->pb && d->x_halo >= d->pl && d->x_halo >= d->pr;
    if (!d->fwd_padded) d->x_phase = 1;   // the phase-split layout exists in the padded layout only
    const int ph = d->x_phase > 1 ? d->x_phase : 1, shift = d->x_halo - d->pl;
    int rc = build_ktabs(&d->ktab_fwd, &d->ktab2_fwd, nullptr, d->kh, d->kw, d->cig, d->h, d->w, d->x_halo, false, ph, shift);
    if (rc == 0)
        rc = build_ktabs(&d->ptab_fwd, &d->ptab2_fwd, &d->rowtab_fwd, d->kh, d->kw, d->cig, d->h, d->w, d->x_halo, d->cig % KBLK == 0, ph,
                         shift);
    if (rc == 0 && d->stride == 1) {
        rc = build_ktabs(&d->ptab_bwd, &d->ptab2_bwd, &d->rowtab_bwd, d->kh, d->kw, d->cog, d->oh, d->ow, d->dy_halo, d->cog % KBLK == 0);
        // dgrad pads dy by K-1-pad before and by the forward pad-before after
        const int need = (d->kh - 1 - d->pt > d->pt ? d->kh - 1 - d->pt : d->pt);
        const int needw = (d->kw - 1 - d->pl > d->pl ? d->kw - 1 - d->pl : d->pl);
        d->bwd_padded = d->dy_halo >= need && d->dy_halo >= needw;
    }
    if (rc == 0) rc = conv_c8_build_tables(d);     // the packed-bf16 path's tap tables (conv_c8.hip) follow the same halos
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
    // weight images of the split-product kernels: [groups][ceil(K / 16)][8 x planes][channels rounded up to 128] dwords each
    const size_t sf = (size_t)groups * ceil_div(d->K, KBLK) * 24 * (ceil_div(d->cog, 128) * 128) * 4;     // up to three planes (bf16x6)
    const size_t sb = (size_t)groups * ceil_div(d->Kd, KBLK) * 24 * (ceil_div(d->cig, 128) * 128) * 4;
    // (a descriptor of a dense layer -- fc6 as a 1 x 1 layer of the bf16 path, csrc/conv_c8.hip -- would hold 2 x 226 MB here: none
    // above 64 MB; such a layer keeps fp32 products in vl_conv_fwd / vl_conv_dgrad whatever vl_set_conv_math says)
    const bool images = sf <= (64u << 20) && sb <= (64u << 20);
    if (rebuild_tables(d) || (images && (hipMalloc((void**)&d->wsplit_fwd, sf) != hipSuccess || hipMalloc((void**)&d->wsplit_bwd, sb) != hipSuccess))) {
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

extern "C" int vl_conv_set_x_phase_split(vl_conv_desc* d, int on) {
    VL_CHECK(d, "vl_conv_set_x_phase_split: null descriptor");
    VL_CHECK(!on || d->fwd_padded, "vl_conv_set_x_phase_split: needs the padded x layout (set the halo first)");
    d->x_phase = (on && d->stride > 1) ? d->stride : 1;
    VL_CHECK(rebuild_tables(d) == 0, "vl_conv_set_x_phase_split: device table allocation failed");
    return 0;
}

extern "C" int vl_conv_x_phase(const vl_conv_desc* d) { return d && d->x_phase > 1 ? d->x_phase : 1; }

extern "C" void vl_conv_destroy(vl_conv_desc* d) {
    if (!d) return;
    free_dev(d->ktab_fwd);
    free_dev(d->ktab2_fwd);
    free_dev(d->ptab_fwd);
    free_dev(d->ptab2_fwd);
    free_dev(d->ptab_bwd);
    free_dev(d->ptab2_bwd);
    free_dev(d->rowtab_fwd);
    free_dev(d->rowtab_bwd);
    free_dev(d->wsplit_fwd);
    free_dev(d->wsplit_bwd);
    conv_c8_free_tables(d);
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
static int launch_conv(const ConvGeom& g, const float* w, int64_t w_ld, int w_grp_stride, const int* row_tab, int Cog,
                       int Cout_total, const ConvOut& o, hipStream_t s) {
    constexpr int BN = 128, BR = KBLK;
    using LA = ConvWeightKX<BM, BR>;
    using LB = ConvGather<BN, BR, PADDED>;
    typename LA::Params pa{w, w_ld, Cog, g.K, (int64_t)w_grp_stride, row_tab};
    EpiConvNCHW::Params pe;
    pe.y = o.y; pe.bias = o.bias; pe.mask = o.mask; pe.relu = o.relu;
    pe.Cog = Cog; pe.Cout_total = Cout_total; pe.OHW = g.OHW; pe.OW = g.OW; pe.M = g.M;
    pe.dOHW = g.dOHW; pe.dOW = g.dOW;
    pe.y_halo = o.y_halo; pe.y_wp = o.OW + 2 * o.y_halo; pe.y_plane = (int64_t)(o.OH + 2 * o.y_halo) * pe.y_wp;
    pe.m_halo = o.m_halo; pe.m_wp = o.OW + 2 * o.m_halo; pe.m_plane = (int64_t)(o.OH + 2 * o.m_halo) * pe.m_wp;
    const int tiles_i = ceil_div(Cog, BM), tiles_j = ceil_div(g.M, BN);
    const int rtiles = ceil_div(g.K, BR);
    dim3 grid(tiles_i * tiles_j, (unsigned)(Cout_total / Cog), 1);
    hipLaunchKernelGGL((mfma_contract<BM, BN, BR, WM, WN, LA, LB, EpiConvNCHW>), grid, dim3(NT), 0, s, pa, g, pe,
                       tiles_i, rtiles, rtiles);
    VL_LAUNCH_CHECK();
    return 0;
}

// ---- the same for 48- and 96-channel output tiles, on v_mfma_f32_16x16x4_f32 -----------------------------------------
// conv2's dgrad produces 48 input channels per group: on 32x32 MFMA tiles that is a 64-row tile with 25 % of the matrix
// work wasted (its pipe was 93 % busy and still only 0.69 of peak).  The 16x16x4 MFMA has the same rate (64 FLOP/clk/SIMD)
// and tiles 48 = 3 x 16 exactly.  Four waves side by side, each 48 channels x 32 pixels = 3 x 2 tiles of 16 x 16 (6 independent
// accumulators of 4 registers); TA = 6 gives the 96-channel tile of the 192-channel layers (conv4 fwd/dgrad, conv5 dgrad).  Operand rows arrive by LDS-DMA exactly as above (a weight row = 48 of 64 lanes); LDS row
// strides 80 / 144 floats (= 16 mod 32) keep the four 16-lane groups of a fragment read -- lane group g reads row 4 s + g --
// on disjoint banks.  Lane l holds A[co = l & 15][r = l >> 4], B[r = l >> 4][pixel = l & 15]; D register q is
// y[co = 4 (l >> 4) + q][pixel = l & 15].

template <int TA>   // channel tiles of 16 per workgroup: 3 (48 channels: conv2 dgrad) or 6 (96: the 192-channel layers)
__global__ __launch_bounds__(NT, 2) void conv_dma16_kernel(const ConvDmaParams pa, const ConvGeom g, const EpiConvNCHW::Params pe,
                                                           int tiles_i, int wide) {
    constexpr int BM = 16 * TA, BN = 128, SR = 32;
    constexpr int NA = (BM + 63) / 64;                                // 64-lane fetches per weight row
    constexpr int SA = NA * 64 + 16, SB = 144;                        // LDS row strides (floats), both = 16 mod 32
    constexpr int ABUF = SR * SA, BUF = SR * (SA + SB);
    constexpr int RW = SR / 4, FPR = NA + 2, NF = RW * FPR, NSTEP = SR / 4;   // rows per wave, fetches per row / per wave, k4-steps
    constexpr int NM = 2 * TA;                                        // MFMAs per k4-step
    extern __shared__ __attribute__((aligned(16))) float ldsc[];
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int ti_blk = bid % tiles_i, tj_blk = bid / tiles_i;
    const int zg = blockIdx.y;
    const int i0 = ti_blk * BM, j0 = tj_blk * BN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    const i32x4 rs_w = rsrc_words(pa.w + (int64_t)zg * pa.w_grp_stride, pa.w_bytes - (int64_t)zg * pa.w_grp_stride * 4);
    const i32x4 rs_x = rsrc_words(g.x + (int64_t)zg * g.grp_stride, (g.total - (int64_t)zg * g.grp_stride) * 4);
    uint32_t voff_a[NA];
#pragma unroll
    for (int hf = 0; hf < NA; ++hf) {
        const int cl = hf * 64 + lane;
        voff_a[hf] = (cl < BM && i0 + cl < pe.Cog) ? (uint32_t)(i0 + cl) * 4u : OOB_OFF;
    }
    uint32_t voff_b[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int m = j0 + hf * 64 + lane;
        const bool vm = m < g.M;
        const uint32_t mm = vm ? m : 0;
        const uint32_t n = fd_div(mm, g.dOHW);
        const uint32_t p = mm - n * g.OHW;
        const uint32_t oh = fd_div(p, g.dOW);
        const uint32_t ow = p - oh * g.OW;
        const int ih0 = (int)oh * g.stride - g.pt + g.halo, iw0 = (int)ow * g.col_mul + g.col_add;
        voff_b[hf] = vm ? (uint32_t)((int64_t)n * g.img_stride + (int64_t)ih0 * g.Wp + iw0) * 4u : OOB_OFF;
    }
    // this lane's bias values, loaded before the main loop: fetched inside the epilogue (one dependent load in front of each of
    // the 48 stores) they cost conv1's forward 0.5 of its 2.4 ms (a run with the stores removed: 1.9 ms)
    float bias_r[TA][4];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = i0 + 16 * a + 4 * (lane >> 4) + q;
            bias_r[a][q] = (pe.bias && co < pe.Cog) ? pe.bias[zg * pe.Cog + co] : 0.f;
        }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)ldsc;
    const int ld_bytes = (int)(pa.w_ld * 4);

    int tabv[RW];
    int row0 = 0;
    bool blk_ok = false;
    const int wrow = wave * RW;
    auto load_table = [&](int st) {
        const int k0 = st * SR + wrow;
        const const_int_ptr tt = as_const(g.ktab) + k0 + st * g.zero;
#pragma unroll
        for (int f = 0; f < RW; ++f) tabv[f] = tt[f];
        row0 = as_const(pa.row_tab)[k0 / KBLK + st * g.zero] + k0 % KBLK;
        blk_ok = k0 / KBLK < pa.nblk;
    };
    auto pin_table = [&]() {
#pragma unroll
        for (int f = 0; f < RW; ++f) asm volatile("" : "+s"(tabv[f]));
        asm volatile("" : "+s"(row0));
    };
    auto dma = [&](const int nb, const int f, int st) {                // f = FPR rr + {weight row pieces, then the two im2col halves}
        const int rr = f / FPR, kind = f % FPR;
        const int z = st * g.zero;
        const bool live = blk_ok && st * SR + wrow + rr < g.K;         // positions past K: see conv_dma_kernel
        if (kind < NA) {
            lds_dma_row(rs_w, lds0 + (uint32_t)(nb * BUF + (wrow + rr) * SA + kind * 64 + z) * 4u, voff_a[kind],
                        min(row0 + rr, g.K - 1) * ld_bytes);
        } else {
            const int hf = kind - NA;
            lds_dma_row(rs_x, lds0 + (uint32_t)(nb * BUF + ABUF + (wrow + rr) * SB + hf * 64 + z) * 4u, live ? voff_b[hf] : OOB_OFF,
                        tabv[rr]);
        }
    };
    int a_soff = 0;
    auto dma_fast = [&](const int nb, const int f, int st) {           // steady state: no liveness select, running weight-row offset
        const int rr = f / FPR, kind = f % FPR;
        const int z = st * g.zero;
        if (f == 0) a_soff = row0 * ld_bytes;
        if (kind < NA) {
            lds_dma_row(rs_w, lds0 + (uint32_t)(nb * BUF + (wrow + rr) * SA + kind * 64 + z) * 4u, voff_a[kind], a_soff);
        } else {
            const int hf = kind - NA;
            lds_dma_row(rs_x, lds0 + (uint32_t)(nb * BUF + ABUF + (wrow + rr) * SB + hf * 64 + z) * 4u, voff_b[hf], tabv[rr]);
        }
        if (kind == NA - 1) a_soff += ld_bytes;
    };
    auto finish_stage = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    f32x4 acc[TA][2];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[a][b][q] = 0.f;

    const int nstages = (pa.nblk * KBLK + SR - 1) / SR;
    load_table(0);
    pin_table();
#pragma unroll
    for (int f = 0; f < NF; ++f) dma(0, f, 0);
    load_table(1);
    pin_table();
    finish_stage();

    const float* fa = ldsc + (lane >> 4) * SA + (lane & 15);
    const float* fb = ldsc + ABUF + (lane >> 4) * SB + wave * 32 + (lane & 15);

    // nsteps: k4-steps of this stage that hold reduction rows below K (a tile's last stage: K = 363 needs 3 of 8)
    auto stage = [&](const int cur, const bool fast, int st_next, const int nsteps) {
        float af[2][TA], bf[2][2];
        auto read = [&](int step, float (&a)[TA], float (&b)[2]) {
            const float* pa_ = fa + cur * BUF + step * 4 * SA;
            const float* pb_ = fb + cur * BUF + step * 4 * SB;
#pragma unroll
            for (int i = 0; i < TA; ++i) a[i] = pa_[16 * i];
#pragma unroll
            for (int i = 0; i < 2; ++i) b[i] = pb_[16 * i];
        };
        read(0, af[0], bf[0]);
#pragma unroll
        for (int t = 0; t < NSTEP; ++t) {
            const int c = t & 1;
            if (t + 1 < NSTEP) read(t + 1, af[c ^ 1], bf[c ^ 1]);
            if (t < nsteps) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int a = m >> 1, b = m & 1;
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c][a], bf[c][b], acc[a][b], 0, 0, 0);
                    const int f = t * NM + m;
                    if (f < NF) {
                        if (fast) dma_fast(cur ^ 1, f, st_next);
                        else dma(cur ^ 1, f, st_next);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {                                               // rows past K: no products; the (zero) fetches of the stage after the last keep the counters in step
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int f = t * NM + m;
                    if (f < NF) dma(cur ^ 1, f, st_next);
                }
            }
            static_assert(NF <= NM * (NSTEP - 3), "fetches must end before the table is replaced");
            if (t == NSTEP - 3) {
                load_table(st_next + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (t == NSTEP - 1) {
                pin_table();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        finish_stage();
    };
    const int nfull = g.K / SR;                                        // stages entirely below K (see conv_dma_kernel)
    const int last_steps = (g.K - (nstages - 1) * SR + 3) / 4;         // live k4-steps of the last stage
    int st = 0;
    for (; st + 2 < nfull; st += 2) {
        stage(0, true, st + 1, NSTEP);
        stage(1, true, st + 2, NSTEP);
    }
    for (; st < nstages; ++st) {
        const int ns = st == nstages - 1 ? last_steps : NSTEP;
        if (st & 1) stage(1, false, st + 1, ns);
        else stage(0, false, st + 1, ns);
    }

    // ---- epilogue.  Dense output planes (y_halo == 0: the layers followed by LRN / pool) without a ReluGrad mask go out through LDS
    // as 16-byte stores of 4 consecutive pixels: a channel's 128 pixels of the tile are 512 contiguous bytes in NCHW, but the
    // accumulator layout hands a store instruction 16 pixels x 4 channels = four 64-byte pieces at arbitrary 4-byte alignment, which
    // the memory side splits again (measured on conv1 forward: 3.4e7 write requests for 1.28 GB, 1.7x the algorithmic bytes written,
    // and 0.4 of the launch's 2.3 ms gone with the stores removed).
    if (wide) {
        constexpr int SP = BN + 4;                                     // row stride 132: the four 16-lane groups of a write hit disjoint bank quarters
        float* T = ldsc;                                               // both stage buffers are free after the last barrier
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = acc[a][b][q] + bias_r[a][q];
                    if (pe.relu) v = fmaxf(v, 0.f);
                    T[(16 * a + 4 * (lane >> 4) + q) * SP + wave * 32 + 16 * b + (lane & 15)] = v;
                }
        __syncthreads();
        const int px = 4 * (lane & 31);
        const int m = j0 + px;
        const uint32_t n = fd_div((uint32_t)min(m, pe.M - 1), pe.dOHW);
        const int p = m - (int)n * pe.OHW;
        const bool run4 = m + 3 < pe.M && p + 3 < pe.OHW;              // four pixels of one image: contiguous
        const int64_t cbase = ((int64_t)n * pe.Cout_total + (int64_t)zg * pe.Cog) * pe.y_plane + p;
#pragma unroll
        for (int i = 0; i < BM / 8; ++i) {                             // 2 channel rows per wave instruction, BM / 4 rows per wave
            const int cl = wave * (BM / 4) + 2 * i + (lane >> 5);
            const int co = i0 + cl;
            const f32x4 v = *reinterpret_cast<const f32x4*>(T + cl * SP + px);
            if (co >= pe.Cog) continue;
            float* dst = pe.y + cbase + (int64_t)co * pe.y_plane;
            if (run4) {
                *reinterpret_cast<f32x4u*>(dst) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int me = m + e;
                    if (me < pe.M) {
                        const uint32_t ne = fd_div((uint32_t)me, pe.dOHW);
                        const int pe_ = me - (int)ne * pe.OHW;
                        pe.y[((int64_t)ne * pe.Cout_total + (int64_t)zg * pe.Cog + co) * pe.y_plane + pe_] = v[e];
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int m = j0 + wave * 32 + 16 * b + (lane & 15);
        if (m >= pe.M) continue;
        const uint32_t n = fd_div((uint32_t)m, pe.dOHW);
        const uint32_t p = m - n * pe.OHW;
        const uint32_t oh = fd_div(p, pe.dOW);
        const uint32_t ow = p - oh * pe.OW;
        const int64_t c0 = (int64_t)n * pe.Cout_total + (int64_t)zg * pe.Cog;
        const int64_t ybase = c0 * pe.y_plane + (int64_t)(oh + pe.y_halo) * pe.y_wp + ow + pe.y_halo;
        const int64_t mbase = c0 * pe.m_plane + (int64_t)(oh + pe.m_halo) * pe.m_wp + ow + pe.m_halo;
        float mk[TA][4];                                          // mask loads first, all in flight (see conv_dma16p_kernel)
        if (pe.mask) {
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = i0 + 16 * a + 4 * (lane >> 4) + q;
                    mk[a][q] = co < pe.Cog ? pe.mask[mbase + (int64_t)co * pe.m_plane] : 0.f;
                }
        }
#pragma unroll
        for (int a = 0; a < TA; ++a) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = i0 + 16 * a + 4 * (lane >> 4) + q;
                if (co < pe.Cog) {
                    float v = acc[a][b][q] + bias_r[a][q];
                    if (pe.relu) v = fmaxf(v, 0.f);
                    if (pe.mask) v = mk[a][q] > 0.f ? v : 0.f;
                    pe.y[ybase + (int64_t)co * pe.y_plane] = v;
                }
            }
        }
    }
}


static int device_cus();

template <int BM>   // 128: conv_dma_kernel (32x32 MFMA tiles); 48 / 96: conv_dma16_kernel (16x16 tiles)
static int launch_conv_dma(const ConvGeom& g, const float* w, int64_t w_ld, int w_grp_stride, const int* row_tab, int Cog,
                           int Cout_total, const ConvOut& o, hipStream_t s) {
    constexpr int BN = 128, SR = 32;
    // 128-wide: 64 KB of stage buffers, 66 KB with the wide-store staging tile [128][132] (two workgroups per CU either way)
    constexpr size_t lds = BM == 128 ? (size_t)BM * (BN + 4) * sizeof(float)
                                     : (size_t)2 * SR * ((BM + 63) / 64 * 64 + 16 + 144) * sizeof(float);   // 56 / 72 KB
    static_assert(BM != 128 || (size_t)2 * SR * (BM + BN) * sizeof(float) <= (size_t)BM * (BN + 4) * sizeof(float), "stage buffers fit");
    ConvDmaParams pa{w, w_ld, (int64_t)w_grp_stride, (int64_t)g.K * w_ld * 4, row_tab, ceil_div(g.K, KBLK)};
    EpiConvNCHW::Params pe;
    pe.y = o.y; pe.bias = o.bias; pe.mask = o.mask; pe.relu = o.relu;
    pe.Cog = Cog; pe.Cout_total = Cout_total; pe.OHW = g.OHW; pe.OW = g.OW; pe.M = g.M;
    pe.dOHW = g.dOHW; pe.dOW = g.dOW;
    pe.y_halo = o.y_halo; pe.y_wp = o.OW + 2 * o.y_halo; pe.y_plane = (int64_t)(o.OH + 2 * o.y_halo) * pe.y_wp;
    pe.m_halo = o.m_halo; pe.m_wp = o.OW + 2 * o.m_halo; pe.m_plane = (int64_t)(o.OH + 2 * o.m_halo) * pe.m_wp;
    static bool attr_set = false;
    if (!attr_set) {
        const void* kern;
        if constexpr (BM == 128) kern = reinterpret_cast<const void*>(conv_dma_kernel<SR>);
        else kern = reinterpret_cast<const void*>(conv_dma16_kernel<BM / 16>);
        VL_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int tiles_i = ceil_div(Cog, BM), tiles_j = ceil_div(g.M, BN);
    dim3 grid(tiles_i * tiles_j, (unsigned)(Cout_total / Cog), 1);
    // wide-store epilogue: dense output planes, no mask, and the [BM][132] staging tile must fit the LDS allocation
    const int wide = (o.y_halo == 0 && !o.mask && !kConvNoWideStore && (size_t)BM * (BN + 4) * sizeof(float) <= lds) ? 1 : 0;
    if constexpr (BM == 128) {
        hipLaunchKernelGGL(conv_dma_kernel<SR>, grid, dim3(NT), lds, s, pa, g, pe, tiles_i, wide);
    } else {
        hipLaunchKernelGGL((conv_dma16_kernel<BM / 16>), grid, dim3(NT), lds, s, pa, g, pe, tiles_i, wide);
    }
    VL_LAUNCH_CHECK();
    return 0;
}

template <int BM>
static int launch_conv_ring(const ConvGeom& g, const float* w, int64_t w_ld, int w_grp_stride, const int* row_tab, int Cog,
                            int Cout_total, const ConvOut& o, uint32_t* wsplit, hipStream_t s) {
    constexpr int BN = BM == 128 ? 256 : 512;
    const int math = g_conv_math, planes = math == 6 ? 3 : 2;
    const int groups = Cout_total / Cog, nstages = ceil_div(g.K, KBLK), CogP = ceil_div(Cog, BM) * BM;
    hipLaunchKernelGGL(conv_wsplit_kernel, dim3(nstages, groups), dim3(256), 0, s, w, w_ld, (int64_t)w_grp_stride, row_tab, g.K, Cog,
                       CogP, nstages, planes, wsplit);
    VL_LAUNCH_CHECK();
    ConvRingParams pa{wsplit, CogP, nstages};
    EpiConvNCHW::Params pe;
    pe.y = o.y; pe.bias = o.bias; pe.mask = o.mask; pe.relu = o.relu;
    pe.Cog = Cog; pe.Cout_total = Cout_total; pe.OHW = g.OHW; pe.OW = g.OW; pe.M = g.M;
    pe.dOHW = g.dOHW; pe.dOW = g.dOW;
    pe.y_halo = o.y_halo; pe.y_wp = o.OW + 2 * o.y_halo; pe.y_plane = (int64_t)(o.OH + 2 * o.y_halo) * pe.y_wp;
    pe.m_halo = o.m_halo; pe.m_wp = o.OW + 2 * o.m_halo; pe.m_plane = (int64_t)(o.OH + 2 * o.m_halo) * pe.m_wp;
    const int OH = g.OHW / g.OW, OWp = (g.OW + 3) / 4 * 4, Mp = g.M / g.OW * OWp;
    const int tiles_i = ceil_div(Cog, BM), tiles_j = ceil_div(Mp, BN);
    dim3 grid(tiles_i * tiles_j, (unsigned)groups, 1);
    if (BM == 128 && math != 6 && !kRing8) {                            // 4-wave workgroups, two per CU (VL_CONV_RING8=1: the 8-wave form)
        constexpr size_t lds4 = (size_t)3 * 16 * (128 + 256) * sizeof(float);    // 72 KB
        static bool attr4[2] = {false, false};
        const int v4 = math == 3;
        auto k4 = v4 ? conv_ring4_kernel<3> : conv_ring4_kernel<1>;
        if (!attr4[v4]) {
            VL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
            attr4[v4] = true;
        }
        hipLaunchKernelGGL(k4, grid, dim3(256), lds4, s, pa, g, pe, tiles_i, OWp, Mp, make_fastdiv(OH * OWp), make_fastdiv(OWp));
        VL_LAUNCH_CHECK();
        return 0;
    }
    // the 8-wave form, one workgroup per CU: ring slots x (weight planes + 16 im2col rows): 120 / 144 KB, bf16x6 140 / 152 KB
    const size_t lds = (size_t)(BM == 128 ? 5 : 4) * (planes * 8 * BM + 16 * BN) * sizeof(float);
    static bool attr_set[3] = {false, false, false};
    const int v8 = math == 6 ? 2 : math == 3 ? 1 : 0;
    auto kern = v8 == 2 ? conv_ring_kernel<6, BM> : v8 == 1 ? conv_ring_kernel<3, BM> : conv_ring_kernel<1, BM>;
    if (!attr_set[v8]) {
        VL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[v8] = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, pa, g, pe, tiles_i, OWp, Mp, make_fastdiv(OH * OWp), make_fastdiv(OWp));
    VL_LAUNCH_CHECK();
    return 0;
}

template <bool PADDED>
static int dispatch_conv(const ConvGeom& g, const float* w, int64_t w_ld, int w_grp_stride, const int* row_tab, int Cog,
                         int Cout_total, const ConvOut& o, uint32_t* wsplit, hipStream_t s) {
    // output-channel tile: 128 when it divides well, else 96 (conv1: 96, conv4: 192) or 64 (conv2 dgrad: 48)
    const int w128 = ceil_div(Cog, 128) * 128, w96 = ceil_div(Cog, 96) * 96, w64 = ceil_div(Cog, 64) * 64;
    // split products: the ring kernel (128-channel tiles; its 16-byte im2col fetches need unit column stride in memory)
    if (g_conv_math != 0 && PADDED && g.col_mul == 1 && wsplit != nullptr) {
        if (Cog >= 96) return launch_conv_ring<128>(g, w, w_ld, w_grp_stride, row_tab, Cog, Cout_total, o, wsplit, s);
        if (Cog >= 40 && Cog <= 64) return launch_conv_ring<64>(g, w, w_ld, w_grp_stride, row_tab, Cog, Cout_total, o, wsplit, s);
    }
    // Few frames (one rank's shard of a multi-GPU job): the tile with the fewest padded rows can leave CUs idle or half of them
    // with two workgroups and the rest with one.  Per-CU load of a tile width = ceil(workgroups / CUs) x its rows; take the
    // LDS-DMA kernel (128 / 96 / 48 rows) with the smallest load, ties to the wider tile.  With many frames the widths agree
    // with the padded-row rule below (loads differ by the padding only), which then decides alone.
    const int64_t px = ceil_div(g.M, 128), cus = device_cus(), ngrp = Cout_total / Cog;
    const bool few = px * ceil_div(Cog, 128) * ngrp < 8 * cus;        // under 8 workgroups per CU at the widest tile: quantisation matters
    if (PADDED && few && (int64_t)g.K * w_ld * 4 < MAX_BUF_BYTES && !kConvStaged && !kConvNoLoadPick) {
        const int widths[3] = {128, 96, 48};
        auto pick = [&](int64_t pxt, int64_t& load) {                 // width with the smallest per-CU load for pxt pixel tiles
            int b = 0;
            for (int i = 0; i < 3; ++i) {
                const int64_t l = ceil_div(pxt * ceil_div(Cog, widths[i]) * ngrp, cus) * widths[i];
                if (i == 0 || l < load) { b = i; load = l; }
            }
            return b;
        };
        auto run = [&](int wi, const ConvGeom& gg, const ConvOut& oo) {
            if (wi == 0) return launch_conv_dma<128>(gg, w, w_ld, w_grp_stride, row_tab, Cog, Cout_total, oo, s);
            if (wi == 1) return launch_conv_dma<96>(gg, w, w_ld, w_grp_stride, row_tab, Cog, Cout_total, oo, s);
            return launch_conv_dma<48>(gg, w, w_ld, w_grp_stride, row_tab, Cog, Cout_total, oo, s);
        };
        int64_t best_load = 0;
        const int best = pick(px, best_load);
        // A last round that is mostly empty (conv2 forward at 128 frames: 1568 workgroups = 6.1 rounds of 256 CUs, paid as 7): the
        // leading frames fill whole rounds at that width, the remaining few frames run as a second launch at the width that suits THEM
        // (conv2: 125 frames in 6 rounds of 128-wide tiles + 3 frames as 114 48-wide tiles: load 816 + a launch instead of 896).
        // The second launch runs one narrow tile per CU, whose reduction loop is then fetch LATENCY per stage (measured: 1.3 us per
        // stage for conv2's 5 x 5 over 28 x 28, 2.9 for conv5's 3 x 3 over 13 x 13) against the 1.9 us per stage and 128 rows the saved
        // round would have cost: conv2 forward at 128 frames 0.511 -> 0.490 ms (8-clip step 5.16 -> 5.135), conv5 forward at 512 frames
        // 0.612 -> 0.666.  So only where the model gains >= 5 % with the second launch charged half a 128-row tile AND at least six
        // whole rounds remain in the first (the second launch is then < 1/6 of a round's worth of frames): conv2's case, not conv5's.
        const int nimg = g.M / g.OHW;
        const int64_t wgs_per_px = (int64_t)ceil_div(Cog, widths[best]) * ngrp;
        const int64_t rounds = px * wgs_per_px / cus;                  // whole rounds
        if (rounds >= 6 && !kConvNoTailSplit) {
            const int64_t px_a = rounds * cus / wgs_per_px;            // pixel tiles that fit them
            const int n_a = (int)(px_a * 128 / g.OHW);                 // whole frames within those tiles
            const int n_b = nimg - n_a;
            if (n_a > 0 && n_b > 0) {
                int64_t load_b = 0;
                const int64_t px_b = ceil_div((int64_t)n_b * g.OHW, 128);
                const int wb = pick(px_b, load_b);
                const int64_t load_a = ceil_div(ceil_div((int64_t)n_a * g.OHW, 128) * wgs_per_px, cus) * widths[best];
                if ((load_a + (load_b > 64 ? load_b : 64) + 16) * 100 <= best_load * 95) {
                    ConvGeom ga = g, gb = g;
                    ga.M = n_a * g.OHW; ga.total = g.img_stride * n_a;
                    gb.x = g.x + (int64_t)n_a * g.img_stride; gb.M = n_b * g.OHW; gb.total = g.img_stride * n_b;
                    ConvOut ob = o;
                    ob.y = o.y + (int64_t)n_a * Cout_total * (o.OH + 2 * o.y_halo) * (o.OW + 2 * o.y_halo);
                    if (o.mask) ob.mask = o.mask + (int64_t)n_a * Cout_total * (o.OH + 2 * o.m_halo) * (o.OW + 2 * o.m_halo);
                    if (int rc = run(best, ga, o)) return rc;
                    return run(wb, gb, ob);
                }
            }
        }
        return run(best, g, o);
    }
    if (w128 <= w96 && w128 <= w64) {
        // 128-wide tiles in the padded layout: the LDS-DMA kernel
        if (PADDED && (int64_t)g.K * w_ld * 4 < MAX_BUF_BYTES && !kConvStaged) {
            return launch_conv_dma<128>(g, w, w_ld, w_grp_stride, row_tab, Cog, Cout_total, o, s);
        }
        return launch_conv<128, 2, 2, PADDED>(g, w, w_ld, w_grp_stride, row_tab, Cog, Cout_total, o, s);
    }
    // 48-wide tiles beat the 64-wide ones whenever they waste fewer rows (conv2 dgrad: 48 channels per group)
    if (PADDED && ceil_div(Cog, 48) * 48 < w64 && ceil_div(Cog, 48) * 48 < w96 &&
        (int64_t)g.K * w_ld * 4 < MAX_BUF_BYTES && !kConvStaged)
        return launch_conv_dma<48>(g, w, w_ld, w_grp_stride, row_tab, Cog, Cout_total, o, s);
    if (w96 <= w64) {
        if (PADDED && (int64_t)g.K * w_ld * 4 < MAX_BUF_BYTES && !kConvStaged)
            return launch_conv_dma<96>(g, w, w_ld, w_grp_stride, row_tab, Cog, Cout_total, o, s);
        return launch_conv<96, 1, 4, PADDED>(g, w, w_ld, w_grp_stride, row_tab, Cog, Cout_total, o, s);
    }
    return launch_conv<64, 1, 4, PADDED>(g, w, w_ld, w_grp_stride, row_tab, Cog, Cout_total, o, s);
}

static void fill_geom(ConvGeom& g, const float* x, int n, int cin_total, int cig, int H, int W, int halo, int OH, int OW,
                      int stride, int pt, int pl, int K, const int* tab, const int2* tab2, int phase = 1) {
    g.x = x; g.ktab = tab; g.ktab2 = tab2; g.K = K; g.M = n * OH * OW; g.H = H; g.W = W;
    g.halo = halo;
    g.Wp = phase > 1 ? (W + 2 * halo + phase - 1) / phase : W + 2 * halo;
    g.col_mul = phase > 1 ? 1 : stride;
    g.col_add = phase > 1 ? 0 : halo - pl;
    g.stride = stride; g.pt = pt; g.pl = pl; g.OHW = OH * OW; g.OW = OW;
    g.dOHW = make_fastdiv(g.OHW); g.dOW = make_fastdiv(g.OW);
    const int64_t Pp = (int64_t)(H + 2 * halo) * g.Wp;
    g.img_stride = (int64_t)cin_total * phase * Pp;
    g.grp_stride = (int64_t)cig * phase * Pp;
    g.total = g.img_stride * n;
    g.zero = 0;
}

extern "C" int vl_conv_fwd(const vl_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int n,
                           int relu, vl_stream_t stream) {
    VL_CHECK(d && x && w && y, "vl_conv_fwd: null argument");
    VL_CHECK(n > 0 && (int64_t)n * d->oh * d->ow < (1ll << 31), "vl_conv_fwd: bad batch %d", n);
    ConvGeom g;
    fill_geom(g, x, n, d->cin, d->cig, d->h, d->w, d->x_halo, d->oh, d->ow, d->stride, d->pt, d->pl, d->K, d->ptab_fwd, d->ptab2_fwd,
              d->x_phase > 1 ? d->x_phase : 1);
    VL_CHECK(g.total * 4 < MAX_BUF_BYTES, "vl_conv_fwd: input of %lld elements exceeds the buffer-offset range", (long long)g.total);
    ConvOut o{y, bias, nullptr, relu, d->y_halo, 0, d->oh, d->ow};
    // HWIO weights are the [K][Cout_total] GEMM operand as they stand; group g = column block g*cog.
    if (d->fwd_padded) return dispatch_conv<true>(g, w, d->cout, d->cog, d->rowtab_fwd, d->cog, d->cout, o, d->wsplit_fwd, (hipStream_t)stream);
    return dispatch_conv<false>(g, w, d->cout, d->cog, d->rowtab_fwd, d->cog, d->cout, o, d->wsplit_fwd, (hipStream_t)stream);
}

// wt[KH-1-ky][KW-1-kx][co][g*cig + ci] = w[ky][kx][ci][g*cog + co]: per tap and group a [cig][cog] -> [cog][cig] transpose, 64 x 64
// tiles through LDS so that both the reads (along co) and the writes (along ci) are whole lines.  (Round 2's element-per-thread form
// read one line per lane: 64 us per layer, 0.26 ms of an 8-clip step.)
__global__ __launch_bounds__(256) void conv_wt_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int KH, int KW,
                                                                int cig, int cog, int groups) {
    __shared__ float tile[64][65];
    const int cin = cig * groups, cout = cog * groups;
    const int tci = (cig + 63) / 64, tco = (cog + 63) / 64;
    int b = blockIdx.x;
    const int co0 = (b % tco) * 64; b /= tco;
    const int ci0 = (b % tci) * 64; b /= tci;
    const int g = b % groups;       b /= groups;
    const int kx = b % KW, ky = b / KW;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const float* src = w + ((int64_t)(ky * KW + kx) * cig) * cout + g * cog;
    for (int r = ly; r < 64; r += 4) {
        const int ci = ci0 + r, co = co0 + lx;
        if (ci < cig && co < cog) tile[r][lx] = src[(int64_t)ci * cout + co];
    }
    __syncthreads();
    float* dst = wt + ((int64_t)((KH - 1 - ky) * KW + (KW - 1 - kx)) * cog) * cin + g * cig;
    for (int r = ly; r < 64; r += 4) {
        const int co = co0 + r, ci = ci0 + lx;
        if (co < cog && ci < cig) dst[(int64_t)co * cin + ci] = tile[lx][r];
    }
}

extern "C" int vl_conv_wt_transpose(const vl_conv_desc* d, const float* w, float* wt, vl_stream_t stream) {
    VL_CHECK(d && w && wt, "vl_conv_wt_transpose: null argument");
    const int blocks = d->kh * d->kw * d->groups * ((d->cig + 63) / 64) * ((d->cog + 63) / 64);
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
              d->ptab_bwd, d->ptab2_bwd);
    VL_CHECK(g.total * 4 < MAX_BUF_BYTES, "vl_conv_dgrad: dy of %lld elements exceeds the buffer-offset range", (long long)g.total);
    ConvOut o{dx, nullptr, relu_mask, 0, d->dx_halo, d->x_halo, d->h, d->w};
    if (d->bwd_padded) return dispatch_conv<true>(g, wt, d->cin, d->cig, d->rowtab_bwd, d->cig, d->cin, o, d->wsplit_bwd, (hipStream_t)stream);
    return dispatch_conv<false>(g, wt, d->cin, d->cig, d->rowtab_bwd, d->cig, d->cin, o, d->wsplit_bwd, (hipStream_t)stream);
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
        launch_reduce_slabs(s, ws, dw, slab, splits, slab, nullptr, 0, 0, 0, nullptr);
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

// padded-mode wgrad, LDS-DMA form: one workgroup per CU (135 KB of LDS)
template <int BN, int WM, int WN>
static int launch_wgrad_dma(const vl_conv_desc* d, const ConvGeom& g, const float* dy, float* dw, float* db, float* ws, int splits,
                            hipStream_t s) {
    using C = WgradDmaCfg<BN>;
    DyParams pb;
    dy_params(d, g, dy, pb);
    const int64_t slab = (int64_t)d->K * d->cout;
    EpiRowMajor::Params pe{splits > 1 ? ws : dw, d->cout, d->K, d->cog, nullptr, nullptr, 0, d->cog, splits > 1 ? slab : 0};
    const int tiles_i = ceil_div(d->K, C::BM), tiles_j = ceil_div(d->cog, BN);
    const int rtiles = ceil_div(g.M, C::BR);
    static bool attr_set[8] = {false, false, false, false, false, false, false, false};
    int v = g_conv_math == 6 ? 3 : g_conv_math == 3 ? 2 : g_conv_math == 1 ? 1 : 0;
    auto kern = v == 3 ? wgrad_dma_kernel<BN, WM, WN, 6> : v == 2 ? wgrad_dma_kernel<BN, WM, WN, 3>
              : v == 1 ? wgrad_dma_kernel<BN, WM, WN, 1> : wgrad_dma_kernel<BN, WM, WN, 0>;
    if constexpr (BN == 128) {
        // 16-byte fetches: four consecutive output pixels must be four consecutive floats of x and of dy, inside one image row
        if (v != 0 && g.OW % 4 == 0 && g.col_mul == 1 && !kWgradDword) {
            kern = v == 3 ? wgrad_dma_kernel<BN, WM, WN, 6, true> : v == 2 ? wgrad_dma_kernel<BN, WM, WN, 3, true> : wgrad_dma_kernel<BN, WM, WN, 1, true>;
            v += 4;
        }
    }
    if (!attr_set[v]) {
        VL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_set[v] = true;
    }
    float* db_slabs = db ? (splits > 1 ? ws + (int64_t)splits * slab : db) : nullptr;   // [splits][Cout_total] behind the weight slabs
    const int units = d->groups * splits;
    dim3 grid((unsigned)(ceil_div(units, 8) * 8 * tiles_i * tiles_j), 1, 1);
    hipLaunchKernelGGL(kern, grid, dim3(NT), C::LDS_BYTES, s, g, pb, pe, tiles_i, tiles_j, d->groups, units, rtiles,
                       ceil_div(rtiles, splits), db_slabs, d->cout);
    VL_LAUNCH_CHECK();
    if (db && splits > 1) {
        launch_reduce_slabs(s, db_slabs, db, (int64_t)d->cout, splits, (int64_t)d->cout, nullptr, 0, 0, 0, nullptr);
        VL_LAUNCH_CHECK();
    }
    return reduce_wgrad(d, dw, ws, splits, s);
}

static int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

static int wgrad_splits(const vl_conv_desc* d, int n) {
    const int64_t M = (int64_t)n * d->oh * d->ow;
    const int tiles = ceil_div(d->K, 128) * ceil_div(d->cog, d->cog % 128 == 0 ? 128 : 96) * d->groups;
    if (!d->fwd_padded) {   // checked mode: 32-pixel tiles, several workgroups per CU; aim at ~2048 workgroups
        const int rtiles = ceil_div(M, 32);
        int splits = ceil_div(2048, tiles);
        return splits > rtiles ? rtiles : (splits < 1 ? 1 : splits);
    }
    // LDS-DMA form: one workgroup per CU, all of equal length, units = groups * splits dealt over the 8 XCDs.  Minimise
    // (rounds of the chip) x (tiles per workgroup + ~2 tiles of prologue / epilogue) over split counts that give every XCD
    // the same number of units; slabs are capped at 512 MB.
    const int rtiles = ceil_div(M, 64);
    static const int forced = vl_exp_env("VL_WGRAD_SPLITS") ? atoi(vl_exp_env("VL_WGRAD_SPLITS")) : 0;   // experiments: force the split count
    if (forced >= 1 && forced <= rtiles) return forced;      // (this function IS the ws_bytes query's split count too: one source)
    const int per_xcd = device_cus() / 8 > 0 ? device_cus() / 8 : 1;
    const int64_t slab_bytes = ((int64_t)d->K + 1) * d->cout * 4;
    int best = 1;
    int64_t best_cost = -1;
    for (int s = 1; s <= rtiles && s <= 4096; ++s) {
        if (s > 1 && slab_bytes * s > (512ll << 20)) break;
        const bool balanced = ((int64_t)d->groups * s) % 8 == 0;
        const int64_t wgs_xcd = (int64_t)ceil_div((int64_t)d->groups * s, 8) * (tiles / d->groups);
        const int64_t cost = (int64_t)ceil_div(wgs_xcd, per_xcd) * (ceil_div(rtiles, s) + 2) + (balanced ? 0 : 1);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best = s;
        }
    }
    return best;
}

extern "C" size_t vl_conv_wgrad_ws_bytes(const vl_conv_desc* d, int n) {
    if (!d || n <= 0) return 0;
    return (size_t)wgrad_splits(d, n) * ((size_t)d->K + 1) * d->cout * sizeof(float);   // weight slabs + bias partials
}

// the bias row rides in a spare row of the last 128-row tile of K (wgrad_dma_kernel); K % 128 == 0 leaves none
extern "C" int vl_conv_wgrad_fuses_bias(const vl_conv_desc* d) { return d && d->fwd_padded && d->K % 128 != 0 ? 1 : 0; }

extern "C" int vl_conv_wgrad(const vl_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* ws,
                             size_t ws_bytes, int n, vl_stream_t stream) {
    VL_CHECK(d && x && dy && dw, "vl_conv_wgrad: null argument");
    VL_CHECK(!db || vl_conv_wgrad_fuses_bias(d), "vl_conv_wgrad: no fused bias gradient for this layer (see vl_conv_wgrad_fuses_bias)");
    VL_CHECK(n > 0 && (int64_t)n * d->oh * d->ow < (1ll << 31), "vl_conv_wgrad: bad batch %d", n);
    const int splits = wgrad_splits(d, n);
    VL_CHECK(splits == 1 || (ws && ws_bytes >= vl_conv_wgrad_ws_bytes(d, n)), "vl_conv_wgrad: workspace too small (%zu < %zu)",
             ws_bytes, vl_conv_wgrad_ws_bytes(d, n));
    ConvGeom g;
    fill_geom(g, x, n, d->cin, d->cig, d->h, d->w, d->x_halo, d->oh, d->ow, d->stride, d->pt, d->pl, d->K, d->ktab_fwd, d->ktab2_fwd,
              d->x_phase > 1 ? d->x_phase : 1);
    const int64_t dy_total = (int64_t)n * d->cout * (d->oh + 2 * d->dy_halo) * (d->ow + 2 * d->dy_halo);
    VL_CHECK(g.total * 4 < MAX_BUF_BYTES && dy_total * 4 < MAX_BUF_BYTES, "vl_conv_wgrad: operand exceeds the buffer-offset range");
    hipStream_t s = (hipStream_t)stream;
    if (d->cog % 128 == 0) {
        if (d->fwd_padded) return launch_wgrad_dma<128, 2, 2>(d, g, dy, dw, db, (float*)ws, splits, s);
        return launch_wgrad<128, 2, 2>(d, g, dy, dw, (float*)ws, splits, s);
    }
    if (d->fwd_padded) return launch_wgrad_dma<96, 4, 1>(d, g, dy, dw, db, (float*)ws, splits, s);
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
        launch_reduce_slabs(s, ws, c, cnt, splits, cnt, bias, n, ldc, relu, mask);
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

// ---- dense GEMM in split-bf16 products (vl_set_conv_math 3 / 6 / 1) -------------------------------------------------------
// Both operands are dense, so both are split ONCE into the image conv_wsplit_kernel makes of conv weights --
// [stage of 16 k][piece plane][lane half h][row or column][4 bf16 pairs], k = 16 stage + 8 h + 2 q + e -- and the GEMM kernel
// is conv_ring4_kernel without any VALU in its loop: linear 16-byte LDS-DMA of 1 KB pieces, ds_read_b128 straight into MFMA
// operand tuples, 128 x 256 tiles, 4 waves of 128 x 64, three slots of 24 KB (36 KB with three planes: then one workgroup
// per CU), two workgroups per CU, split-K over the stage range when the tiles alone do not fill the chip.
// X(k, r) = base[k * sk + r * sr]; one of sk, sr is 1.  Rows r >= R and positions k >= K are zero.
__global__ void gemm_split_image_kernel(const float* __restrict__ x, int64_t sk, int64_t sr, int K, int R, int Rp, int planes,
                                        uint32_t* __restrict__ out) {
    const int st = blockIdx.y;
    uint32_t* o = out + (int64_t)st * planes * 8 * Rp;
    const int r0 = blockIdx.x * 64;
    if (sk == 1) {
        // k is the contiguous axis: thread = (row, group of 4 consecutive k) reads 16 B
        const int rr = threadIdx.x >> 2, kq = threadIdx.x & 3, r = r0 + rr, k0 = st * 16 + 4 * kq;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < R) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k0 + e < K) v[e] = x[(int64_t)r * sr + k0 + e];
        }
        const int h = kq >> 1;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            int pc[3] = {0, 0, 0};
            split_pieces<3>(v[2 * e], v[2 * e + 1], pc);
            const int q = (kq & 1) * 2 + e;
            for (int pl = 0; pl < planes; ++pl) o[(int64_t)((2 * pl + h) * Rp + r) * 4 + q] = (uint32_t)pc[pl];
        }
    } else {
        // r is the contiguous axis: thread = (pair index, row): neighbouring threads read neighbouring rows
        for (int idx = threadIdx.x; idx < 8 * 64; idx += blockDim.x) {
            const int rr = idx & 63, hq = idx >> 6, h = hq >> 2, q = hq & 3, r = r0 + rr, k = st * 16 + 8 * h + 2 * q;
            float x0 = 0.f, x1 = 0.f;
            if (r < R) {
                if (k < K) x0 = x[(int64_t)k * sk + r];
                if (k + 1 < K) x1 = x[(int64_t)(k + 1) * sk + r];
            }
            int pc[3] = {0, 0, 0};
            split_pieces<3>(x0, x1, pc);
            for (int pl = 0; pl < planes; ++pl) o[(int64_t)((2 * pl + h) * Rp + r) * 4 + q] = (uint32_t)pc[pl];
        }
    }
}

struct GemmSplitParams {
    const uint32_t* ia;      // image of opA: rows = m (tile 128), pitch Mp
    const uint32_t* ib;      // image of opB: columns = n (tile 256), pitch Np
    int Mp, Np, nstages, stages_per_split;
    float* c;                // output (splits == 1) or slabs [split][m][n]
    int64_t ldc, slab_stride;
    const float* bias;
    const float* mask;
    int relu, m, n, splits;
};

template <int MATH>
__global__ __launch_bounds__(256, MATH == 6 ? 1 : 2) void gemm_split_kernel(const GemmSplitParams P) {
    constexpr int BM = 128, BN = 256, NBUF = 3, D = 3, TA = 4, TB = 2;
    constexpr int NP = split_products(MATH), NPC = split_pieces_of(MATH), NPL = MATH == 6 ? 3 : 2;
    constexpr int APC = NPL * 2 * 2, BPC = NPL * 2 * 4;               // 1 KB pieces of a stage: A (2 per plane and half), B (4)
    constexpr int ABUF = APC * 256, BUF = (APC + BPC) * 256;          // dwords
    constexpr int FA = APC / 4, FB = BPC / 4, FW = FA + FB;           // pieces per wave per stage
    constexpr int NM = TA * TB * NP;
    static_assert(NM >= FW, "a fetch rides on each of the first FW MFMAs");
    extern __shared__ __attribute__((aligned(16))) float ldsg[];
    const int tiles_i = P.Mp / BM;
    const int bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int i0 = (bid % tiles_i) * BM, j0 = (bid / tiles_i) * BN;
    const int zs = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int s_begin = zs * P.stages_per_split;
    const int nst = min(P.nstages, s_begin + P.stages_per_split) - s_begin;      // stages of this split (>= 1 by construction)

    const int64_t sdw_a = (int64_t)NPL * 8 * P.Mp, sdw_b = (int64_t)NPL * 8 * P.Np;
    const i32x4 rs_a = rsrc_words(reinterpret_cast<const float*>(P.ia), P.nstages * sdw_a * 4);
    const i32x4 rs_b = rsrc_words(reinterpret_cast<const float*>(P.ib), P.nstages * sdw_b * 4);
    // piece p of A = (plane-and-half p >> 1, rows 64 (p & 1) ..+63), of B = (p >> 2, columns 64 (p & 3) ..+63); lands at byte 1024 p
    uint32_t voff[FW];
#pragma unroll
    for (int f = 0; f < FA; ++f) {
        const int p = wave * FA + f;
        voff[f] = (uint32_t)(((p >> 1) * P.Mp + i0 + (p & 1) * 64 + lane) * 16);
    }
#pragma unroll
    for (int f = 0; f < FB; ++f) {
        const int p = wave * FB + f;
        voff[FA + f] = (uint32_t)(((p >> 2) * P.Np + j0 + (p & 3) * 64 + lane) * 16);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)ldsg;
    const int sb_a = (int)(sdw_a * 4), sb_b = (int)(sdw_b * 4);
    auto dma = [&](const uint32_t wr, const int f, int st) __attribute__((always_inline)) {        // st: absolute stage
        if (f < FA) lds_dma_row4(rs_a, lds0 + wr + (uint32_t)(wave * FA + f) * 1024u, voff[f], st * sb_a);
        else lds_dma_row4(rs_b, lds0 + wr + (uint32_t)(ABUF * 4 + (wave * FB + f - FA) * 1024), voff[f], st * sb_b);
    };

    f32x16 acc[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

    const float* fa = ldsg + ((lane >> 5) * BM + (lane & 31)) * 4;
    const float* fb = ldsg + ABUF + ((lane >> 5) * BN + wave * 64 + (lane & 31)) * 4;
    auto load_ops = [&](int rd, SplitBf16 (&A)[TA], SplitBf16 (&B)[TB]) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NPC; ++k) {
#pragma unroll
            for (int a = 0; a < TA; ++a) A[a].p[k] = *reinterpret_cast<const i32x4*>(fa + rd + k * 2 * BM * 4 + a * 32 * 4);
#pragma unroll
            for (int b = 0; b < TB; ++b) B[b].p[k] = *reinterpret_cast<const i32x4*>(fb + rd + k * 2 * BN * 4 + b * 32 * 4);
        }
    };

    // prologue: stages 0, 1, 2 (of this split) in flight; stage 0 -> registers; stage 1 landed
    for (int s0 = 0; s0 < NBUF; ++s0) {
        if (s0 < nst) {
#pragma unroll
            for (int f = 0; f < FW; ++f) dma((uint32_t)(s0 * BUF * 4), f, s_begin + s0);
        }
    }
    if (nst >= NBUF) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * FW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    SplitBf16 A[TA] = {}, B[TB] = {}, An[TA] = {}, Bn[TB] = {};
    load_ops(0, A, B);
    if (nst >= NBUF) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(FW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    int rd = BUF;
    uint32_t wr = 0;
    auto stage = [&](int st, SplitBf16 (&Ac)[TA], SplitBf16 (&Bc)[TB], SplitBf16 (&An)[TA], SplitBf16 (&Bn)[TB])
                     __attribute__((always_inline)) {
        load_ops(rd, An, Bn);                                         // stage st + 1 (garbage past the end: never multiplied)
        static_for<0, NM>([&](auto MI) __attribute__((always_inline)) {
            constexpr int m = decltype(MI)::value;
            constexpr int prod = m / (TA * TB), a = (m % (TA * TB)) / TB, b = m % TB;
            acc[a][b] = mfma_bf16(Ac[a].p[kProdA[prod]], Bc[b].p[kProdB[prod]], acc[a][b]);
            if constexpr (m < FW) {
                if (st + D < nst) dma(wr, m, s_begin + st + D);
            }
            __builtin_amdgcn_sched_barrier(0x106);
        });
        rd = rd + BUF == NBUF * BUF ? 0 : rd + BUF;
        wr = wr + BUF * 4 == NBUF * BUF * 4 ? 0u : wr + BUF * 4;
        // groups issued so far end with stage min(st + D, nst - 1); stage st + 2 must have landed
        if (min(st + D, nst - 1) - (st + 2) >= 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(FW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int st = 0; st < nst; st += 2) {
        stage(st, A, B, An, Bn);
        if (st + 1 < nst) stage(st + 1, An, Bn, A, B);
    }

    // epilogue: row 32 a + i of the tile, i = (q & 3) + 8 (q >> 2) + 4 (l >> 5); column 64 wave + 32 b + (l & 31)
    float* out = P.c + (P.splits > 1 ? (int64_t)zs * P.slab_stride : 0);
    const int64_t ld = P.splits > 1 ? P.n : P.ldc;
#pragma unroll
    for (int b = 0; b < TB; ++b) {
        const int col = j0 + wave * 64 + 32 * b + (lane & 31);
        if (col >= P.n) continue;
        const float bv = (P.splits == 1 && P.bias) ? P.bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < TA; ++a) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = i0 + 32 * a + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                if (row < P.m) {
                    float v = acc[a][b][q] + bv;
                    if (P.splits == 1) {
                        if (P.relu) v = fmaxf(v, 0.f);
                        if (P.mask) v = P.mask[(int64_t)row * P.ldc + col] > 0.f ? v : 0.f;
                    }
                    out[(int64_t)row * ld + col] = v;
                }
            }
        }
    }
}

// split-K of the split-product GEMM: as many splits as it takes to fill `slots` workgroup slots, each at least 8 stages deep
static int gemm_split_splits(int m, int n, int k, int slots) {
    const int nstages = ceil_div(k, 16), tiles = ceil_div(m, 128) * ceil_div(n, 256);
    int splits = 1;
    if (tiles < slots) {
        splits = ceil_div(slots, tiles);
        if (splits > 8) splits = 8;
        if (splits > nstages / 8) splits = nstages / 8 > 0 ? nstages / 8 : 1;
    }
    const int sps = ceil_div(nstages, splits);
    return ceil_div(nstages, sps);                                   // no empty split
}

// bytes of workspace the split-product GEMM needs in any mode: operand images of three planes + its split-K slabs
extern "C" size_t vl_gemm_split_ws_bytes(int m, int n, int k) {
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const size_t st = (size_t)ceil_div(k, 16), mp = (size_t)ceil_div(m, 128) * 128, np = (size_t)ceil_div(n, 256) * 256;
    const int splits = gemm_split_splits(m, n, k, 2 * device_cus());
    return st * 24 * (mp + np) * 4 + (splits > 1 ? (size_t)splits * m * n * 4 : 0) + 4096;
}

static int launch_gemm_split(int transa, int transb, int m, int n, int k, const float* a, int64_t lda, const float* b, int64_t ldb,
                             float* c, int64_t ldc, const float* bias, int relu, const float* relu_mask, void* ws, hipStream_t s) {
    const int math = g_conv_math, planes = math == 6 ? 3 : 2;
    const int nstages = ceil_div(k, 16), Mp = ceil_div(m, 128) * 128, Np = ceil_div(n, 256) * 256;
    uint32_t* ia = (uint32_t*)ws;
    uint32_t* ib = ia + (size_t)nstages * planes * 8 * Mp;
    float* slabs = (float*)(ib + (size_t)nstages * planes * 8 * Np);
    // X(k, r): opA rows = m, opB columns = n
    hipLaunchKernelGGL(gemm_split_image_kernel, dim3(Mp / 64, nstages), dim3(256), 0, s, a, transa ? lda : (int64_t)1,
                       transa ? (int64_t)1 : lda, k, m, Mp, planes, ia);
    hipLaunchKernelGGL(gemm_split_image_kernel, dim3(Np / 64, nstages), dim3(256), 0, s, b, transb ? (int64_t)1 : ldb,
                       transb ? ldb : (int64_t)1, k, n, Np, planes, ib);
    VL_LAUNCH_CHECK();
    const int tiles = (Mp / 128) * (Np / 256);
    const int splits = gemm_split_splits(m, n, k, device_cus() * (math == 6 ? 1 : 2));
    const int sps = ceil_div(nstages, splits);
    GemmSplitParams P{ia, ib, Mp, Np, nstages, sps, splits > 1 ? slabs : c, ldc, (int64_t)m * n, bias, relu_mask, relu, m, n, splits};
    const size_t lds = (size_t)3 * (planes * 2 * 2 + planes * 2 * 4) * 1024;      // 72 KB (108 KB with three planes)
    static bool attr_set[3] = {false, false, false};
    const int v = math == 6 ? 2 : math == 3 ? 1 : 0;
    auto kern = v == 2 ? gemm_split_kernel<6> : v == 1 ? gemm_split_kernel<3> : gemm_split_kernel<1>;
    if (!attr_set[v]) {
        VL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[v] = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(256), lds, s, P);
    VL_LAUNCH_CHECK();
    if (splits > 1) {
        const int64_t count = (int64_t)m * n;
        launch_reduce_slabs(s, slabs, c, count, splits, count, bias, n, ldc, relu, relu_mask);
        VL_LAUNCH_CHECK();
    }
    return 0;
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
    // split-bf16 products (vl_set_conv_math != 0) for the large GEMMs when the caller's workspace holds the operand images
    if (g_conv_math != 0 && m >= 128 && n >= 128 && k >= 128 && ws && ws_bytes >= vl_gemm_split_ws_bytes(m, n, k))
        return launch_gemm_split(transa, transb, m, n, k, a, lda, b, ldb, c, ldc, bias, relu, relu_mask, ws, (hipStream_t)stream);
    const int bm = m <= 64 ? 64 : 128;
    const int tiles = ceil_div(m, bm) * ceil_div(n, 128);
    // split the reduction when the output tiles alone leave fewer than ~3 workgroups per CU (this kernel hides its load and
    // barrier latency with co-resident workgroups: fc6 forward = 256 tiles = one per CU ran at 0.65 of peak) and a workspace
    // was provided
    int splits = 1;
    static const int kWantPerCu = vl_exp_env("VL_GEMM_WANT") ? atoi(vl_exp_env("VL_GEMM_WANT")) : 3;   // experiments: workgroups per CU aimed at
    const int want = kWantPerCu * device_cus();
    if (ws && tiles < want && !kGemmNoSplit) {
        splits = ceil_div(want, tiles);
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

// fp32-MFMA tiled contraction engine for gfx950 + the conv / GEMM entry points built on it.
//
// One kernel template, D[i][j] = sum_r A(i, r) * B(r, j), computed with v_mfma_f32_32x32x2_f32
// (exact fp32, 64 FLOP/clk/SIMD).  256 threads = 4 wavefronts per workgroup; each wave owns a
// (BM/WM) x (BN/WN) block of 32x32 accumulator tiles.  Operand tiles are staged
// HBM/L2 -> registers -> LDS (double buffered, next tile's loads issued before the MFMAs of the
// current one, written to LDS after them: one barrier per r-tile).  What differs between the
// uses is only the *loader* (how element (x, r) of an operand is found in memory) and the epilogue:
//
//   conv fwd / dgrad : A = HWIO weights [r=(kh,kw,ci)][i=co]        (DenseKX)
//                      B = implicit im2col gather [r][j=pixel]      (ConvGather), NCHW input
//                      epilogue writes NCHW (+bias, ReLU | ReluGrad mask)
//   conv wgrad       : A = implicit im2col gather [i=(kh,kw,ci)][r=pixel]   (WgradGather)
//                      B = dy [j=co][r=pixel]                               (DyRows)
//                      epilogue writes per-split HWIO slabs, reduced deterministically
//   dense GEMM       : A, B = DenseKX / DenseXK by transpose flag (fc6/7/8, LSTM, output fc)
//
// MFMA operand maps (cdna_hip_programming.md section 3): lane l holds A[i = l&31][r = l>>5] and
// B[r = l>>5][j = l&31]; D register q of lane l is D[i = (q&3) + 8*(q>>2) + 4*(l>>5)][j = l&31].
// j is therefore the coalesced (lane) dimension of every store.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
static constexpr int NT = 256;  // threads per workgroup

// ---- LDS tile layouts -------------------------------------------------------------------------
// KX: [r][x], x contiguous.  Fragment reads walk x across lanes 0..31 -> conflict free.
template <int BX, int BR>
struct LdsKX {
    static constexpr int SIZE = BX * BR;
    static __device__ __forceinline__ int idx(int x, int r) { return r * BX + x; }
};
// XK: [x][r] with an odd row stride so that fragment reads (x across lanes) are conflict free.
template <int BX, int BR>
struct LdsXK {
    static constexpr int SIZE = BX * (BR + 1);
    static __device__ __forceinline__ int idx(int x, int r) { return x * (BR + 1) + r; }
};

// ---- loaders ----------------------------------------------------------------------------------
// Dense operand stored [r][x] (x contiguous in memory).
template <int BX, int BR>
struct DenseKX {
    using L = LdsKX<BX, BR>;
    static constexpr int NLD = BX * BR / NT;
    static_assert(BX * BR % NT == 0, "tile must divide over 256 threads");
    struct Params {
        const float* p;
        int64_t ld;
        int X, R;
        int64_t zg_stride;
    };
    const float* base;
    int64_t ld;
    int X, R, x0;
    __device__ __forceinline__ void init(const Params& P, int x0_, int zg) {
        base = P.p + (int64_t)zg * P.zg_stride;
        ld = P.ld;
        X = P.X;
        R = P.R;
        x0 = x0_;
    }
    __device__ __forceinline__ void prefetch(int) {}
    __device__ __forceinline__ void load(int rt, float (&v)[NLD]) const {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int e = threadIdx.x + NT * j;
            const int r = rt * BR + e / BX, x = x0 + e % BX;
            v[j] = (r < R && x < X) ? base[(int64_t)r * ld + x] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* lds, const float (&v)[NLD]) const {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int e = threadIdx.x + NT * j;
            lds[L::idx(e % BX, e / BX)] = v[j];
        }
    }
};

// Dense operand stored [x][r] (r contiguous in memory).
template <int BX, int BR>
struct DenseXK {
    using L = LdsXK<BX, BR>;
    static constexpr int NLD = BX * BR / NT;
    static_assert(BX * BR % NT == 0, "tile must divide over 256 threads");
    struct Params {
        const float* p;
        int64_t ld;
        int X, R;
        int64_t zg_stride;
    };
    const float* base;
    int64_t ld;
    int X, R, x0;
    __device__ __forceinline__ void init(const Params& P, int x0_, int zg) {
        base = P.p + (int64_t)zg * P.zg_stride;
        ld = P.ld;
        X = P.X;
        R = P.R;
        x0 = x0_;
    }
    __device__ __forceinline__ void prefetch(int) {}
    __device__ __forceinline__ void load(int rt, float (&v)[NLD]) const {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int e = threadIdx.x + NT * j;
            const int r = rt * BR + e % BR, x = x0 + e / BR;
            v[j] = (r < R && x < X) ? base[(int64_t)x * ld + r] : 0.f;
        }
    }
    __device__ __forceinline__ void store(float* lds, const float (&v)[NLD]) const {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int e = threadIdx.x + NT * j;
            lds[L::idx(e / BR, e % BR)] = v[j];
        }
    }
};

// Geometry shared by the im2col loaders.  ktab[k] = {offset (ci*H + kh)*W + kw, kh << 16 | kw};
// entries past the real K carry kh = 0x4000 so that the bounds test fails (zero fill).
struct ConvGeom {
    const float* x;
    const int2* ktab;
    int K;              // im2col rows per group
    int M;              // n * OH * OW pixels
    int H, W;           // input plane
    int stride, pt, pl;
    int OHW, OW;
    FastDiv dOHW, dOW;
    int64_t img_stride;  // elements between images (Cin_total * H * W)
    int64_t grp_stride;  // elements between groups (Cin_g * H * W)
    int64_t total;       // elements in x (n * Cin_total * H * W); < 2^30 so byte offsets fit 32 bits
};

// Buffer-resource gather: every im2col element is fetched with a raw buffer load whose per-lane byte
// offset is either the real offset or OOB_OFF; the hardware range check returns 0 for the latter, so
// padding taps and tile tails cost no branch and no select.  Offsets are biased by (pt*W + pl)
// elements so they are never negative (the resource base points that far before x).
static constexpr uint32_t OOB_OFF = 0xFFFFFFF0u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(uint32_t)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}

// im2col gather, reduction index r = im2col row, x = output pixel across lanes (coalesced along ow).
// A wave owns NLD consecutive rows of the tile, so its table entries are one contiguous, aligned
// block read with a single wide scalar load -- fetched one tile ahead of use.
template <int BX, int BR>
struct ConvGather {
    using L = LdsKX<BX, BR>;
    static constexpr int NLD = BX * BR / NT;   // rows per wave
    static_assert(BX % 64 == 0, "pixel tile must be a multiple of the wave size");
    static_assert(NLD * (NT / BX) == BR, "rows must split evenly over the wave groups");
    using Params = ConvGeom;
    __amdgpu_buffer_rsrc_t rsrc;
    const int2* tab;     // this wave's first row of the current tile
    int2 ent[NLD];       // entries of the tile about to be loaded (prefetched)
    uint32_t voff;       // biased byte offset of (n, ih0, iw0)
    int ih0, iw0, H, W, row0;
    bool vm;
    __device__ __forceinline__ void init(const Params& P, int x0, int zg) {
        const int m = x0 + threadIdx.x % BX;
        vm = m < P.M;
        const uint32_t mm = vm ? m : 0;
        const uint32_t n = fd_div(mm, P.dOHW);
        const uint32_t p = mm - n * P.OHW;
        const uint32_t oh = fd_div(p, P.dOW);
        const uint32_t ow = p - oh * P.OW;
        ih0 = (int)oh * P.stride - P.pt;
        iw0 = (int)ow * P.stride - P.pl;
        H = P.H;
        W = P.W;
        const int bias = P.pt * P.W + P.pl;
        const int ih0_real = ih0;
        rsrc = make_rsrc(P.x + (int64_t)zg * P.grp_stride - bias, (P.total - (int64_t)zg * P.grp_stride + bias) * 4);
        voff = (uint32_t)((int64_t)n * P.img_stride + (int64_t)ih0_real * P.W + iw0 + bias) * 4u;
        if (!vm) ih0 = 1 << 28;   // a pixel past the end fails every row test below: no separate predicate
        row0 = __builtin_amdgcn_readfirstlane((int)threadIdx.x / BX) * NLD;
        tab = P.ktab + row0;
    }
    __device__ __forceinline__ void prefetch(int rt) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) ent[j] = tab[rt * BR + j];
    }
    __device__ __forceinline__ void load(int rt, float (&v)[NLD]) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int ih = ih0 + (ent[j].y >> 16), iw = iw0 + (ent[j].y & 0xffff);
            const bool ok = ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
            v[j] = buf_load(rsrc, ok ? voff + (uint32_t)ent[j].x * 4u : OOB_OFF);
        }
        prefetch(rt + 1);   // the table is padded by a whole tile, reading one past the end is safe
    }
    __device__ __forceinline__ void store(float* lds, const float (&v)[NLD]) const {
        const int xl = threadIdx.x % BX;
#pragma unroll
        for (int j = 0; j < NLD; ++j) lds[L::idx(xl, row0 + j)] = v[j];
    }
};

// im2col gather for wgrad: x = im2col row (fixed per thread: table entries live in registers),
// reduction index r = output pixel across lanes.
template <int BX, int BR>
struct WgradGather {
    using L = LdsXK<BX, BR>;
    static constexpr int NLD = BX * BR / NT;
    using Params = ConvGeom;
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t off[NLD];   // byte offsets (ci*H + kh)*W + kw
    int khkw[NLD];
    int H, W, M, OHW, OW, stride, pt, pl, bias;
    FastDiv dOHW, dOW;
    int64_t img_stride;
    __device__ __forceinline__ void init(const Params& P, int x0, int zg) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int2 e = P.ktab[x0 + (int)(threadIdx.x + NT * j) / BR];
            off[j] = (uint32_t)e.x * 4u;
            khkw[j] = e.y;
        }
        bias = P.pt * P.W + P.pl;
        rsrc = make_rsrc(P.x + (int64_t)zg * P.grp_stride - bias, (P.total - (int64_t)zg * P.grp_stride + bias) * 4);
        H = P.H; W = P.W; M = P.M; OHW = P.OHW; OW = P.OW;
        stride = P.stride; pt = P.pt; pl = P.pl;
        dOHW = P.dOHW; dOW = P.dOW;
        img_stride = P.img_stride;
    }
    __device__ __forceinline__ void load(int rt, float (&v)[NLD]) const {
        const int m = rt * BR + threadIdx.x % BR;
        const bool vm = m < M;
        const uint32_t mm = vm ? m : 0;
        const uint32_t n = fd_div(mm, dOHW);
        const uint32_t p = mm - n * OHW;
        const uint32_t oh = fd_div(p, dOW);
        const uint32_t ow = p - oh * OW;
        const int ih0r = (int)oh * stride - pt, iw0 = (int)ow * stride - pl;
        const uint32_t voff = (uint32_t)((int64_t)n * img_stride + (int64_t)ih0r * W + iw0 + bias) * 4u;
        const int ih0 = vm ? ih0r : (1 << 28);
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int ih = ih0 + (khkw[j] >> 16), iw = iw0 + (khkw[j] & 0xffff);
            const bool ok = ((unsigned)ih < (unsigned)H) & ((unsigned)iw < (unsigned)W);
            v[j] = buf_load(rsrc, ok ? voff + off[j] : OOB_OFF);
        }
    }
    __device__ __forceinline__ void prefetch(int) {}
    __device__ __forceinline__ void store(float* lds, const float (&v)[NLD]) const {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int e = threadIdx.x + NT * j;
            lds[L::idx(e / BR, e % BR)] = v[j];
        }
    }
};

// dy rows for wgrad: x = output channel (within the group), r = output pixel across lanes.
// dy is NCHW [n][Cout_total][OHW].
template <int BX, int BR>
struct DyRows {
    using L = LdsXK<BX, BR>;
    static constexpr int NLD = BX * BR / NT;
    struct Params {
        const float* dy;
        int M, OHW, Cog, Cout_total;
        FastDiv dOHW;
        int64_t total;   // elements in dy
    };
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t co_off[NLD];   // byte offset of the row's channel plane
    bool co_ok[NLD];        // channel inside the group
    int M, OHW;
    int64_t img_stride;
    FastDiv dOHW;
    __device__ __forceinline__ void init(const Params& P, int x0, int zg) {
        const int64_t goff = (int64_t)zg * P.Cog * P.OHW;
        rsrc = make_rsrc(P.dy + goff, (P.total - goff) * 4);
        M = P.M; OHW = P.OHW;
        img_stride = (int64_t)P.Cout_total * P.OHW;
        dOHW = P.dOHW;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int co = x0 + (int)(threadIdx.x + NT * j) / BR;
            co_ok[j] = co < P.Cog;
            co_off[j] = (uint32_t)co * (uint32_t)P.OHW * 4u;
        }
    }
    __device__ __forceinline__ void load(int rt, float (&v)[NLD]) const {
        const int m = rt * BR + threadIdx.x % BR;
        const bool vm = m < M;
        const uint32_t mm = vm ? m : 0;
        const uint32_t n = fd_div(mm, dOHW);
        const uint32_t p = mm - n * OHW;
        const uint32_t voff = (uint32_t)((int64_t)n * img_stride + p) * 4u;
#pragma unroll
        for (int j = 0; j < NLD; ++j) v[j] = buf_load(rsrc, (vm & co_ok[j]) ? voff + co_off[j] : OOB_OFF);
    }
    __device__ __forceinline__ void prefetch(int) {}
    __device__ __forceinline__ void store(float* lds, const float (&v)[NLD]) const {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int e = threadIdx.x + NT * j;
            lds[L::idx(e / BR, e % BR)] = v[j];
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

// NCHW conv output: i = output channel within the group, j = output pixel.
struct EpiConvNCHW {
    struct Params {
        float* y;
        const float* bias;  // [Cout_total] or null
        const float* mask;  // NCHW like y or null
        int relu;
        int Cog, Cout_total, OHW, M;
        FastDiv dOHW;
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
            const int64_t base = ((int64_t)n * P.Cout_total + (int64_t)zg * P.Cog) * P.OHW + p;
#pragma unroll
            for (int ti = 0; ti < TM; ++ti) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int co = i0 + 32 * ti + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                    if (co < P.Cog) {
                        float v = acc[ti][tj][q];
                        if (P.bias) v += P.bias[zg * P.Cog + co];
                        if (P.relu) v = fmaxf(v, 0.f);
                        const int64_t o = base + (int64_t)co * P.OHW;
                        if (P.mask) v = P.mask[o] > 0.f ? v : 0.f;
                        P.y[o] = v;
                    }
                }
            }
        }
    }
};

// ---- the kernel -------------------------------------------------------------------------------
// grid.x = tiles_i * tiles_j (i fastest), grid.y = groups (zg), grid.z = reduction splits (zs).
template <int BM, int BN, int BR, int WM, int WN, class LA, class LB, class EP>
__global__ __launch_bounds__(NT) void mfma_contract(const typename LA::Params pa, const typename LB::Params pb,
                                                    const typename EP::Params pe, int tiles_i, int rtiles,
                                                    int rt_per_split) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TM * WM * 32 == BM && TN * WN * 32 == BN, "tile shape");
    constexpr int SA = LA::L::SIZE, SB = LB::L::SIZE;
    __shared__ float lds[2 * (SA + SB)];

    const int ti_blk = blockIdx.x % tiles_i, tj_blk = blockIdx.x / tiles_i;
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
    for (int rt = rt0; rt < rt1; ++rt) {
        const int cur = (rt - rt0) & 1;
        const float* A = lds + cur * (SA + SB);
        const float* B = A + SA;
        const bool more = rt + 1 < rt1;
        if (more) {
            la.load(rt + 1, ra);
            lb.load(rt + 1, rb);
        }
#pragma unroll
        for (int kk = 0; kk < BR / 2; ++kk) {
            const int r = 2 * kk + (lane >> 5);
            float af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = A[LA::L::idx(wi0 + 32 * a + (lane & 31), r)];
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = B[LB::L::idx(wj0 + 32 * b + (lane & 31), r)];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (more) {
            float* An = lds + (cur ^ 1) * (SA + SB);
            la.store(An, ra);
            lb.store(An + SA, rb);
        }
        __syncthreads();
    }
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
    int oh, ow, pt, pl;
    int cig, cog;
    int K;           // kh*kw*cig
    int Kd;          // kh*kw*cog (dgrad reduction length)
    int2* ktab_fwd;  // device, padded
    int2* ktab_bwd;  // device, padded (stride 1 only), over dy planes
    int ktab_len_fwd, ktab_len_bwd;
};

static void tf_same_pad(int in, int k, int s, int* out, int* before) {
    *out = (in + s - 1) / s;
    int total = (*out - 1) * s + k - in;
    if (total < 0) total = 0;
    *before = total / 2;
}

static int build_ktab(int2** dev, int* len, int kh, int kw, int cg, int H, int W) {
    const int K = kh * kw * cg;
    const int pad = ((K + 127) / 128) * 128 + 128;
    int2* host = (int2*)malloc(sizeof(int2) * pad);
    if (!host) return 1;
    for (int k = 0; k < pad; ++k) {
        if (k < K) {
            const int c = k % cg, kx = (k / cg) % kw, ky = k / (cg * kw);  // HWIO row order (kh, kw, ci)
            host[k].x = (c * H + ky) * W + kx;
            host[k].y = (ky << 16) | kx;
        } else {
            host[k].x = 0;
            host[k].y = 0x4000 << 16;
        }
    }
    hipError_t e = hipMalloc((void**)dev, sizeof(int2) * pad);
    if (e == hipSuccess) e = hipMemcpy(*dev, host, sizeof(int2) * pad, hipMemcpyHostToDevice);
    free(host);
    *len = pad;
    return e == hipSuccess ? 0 : 2;
}

extern "C" int vl_conv_create(vl_conv_desc** out, int cin, int h, int w, int cout, int kh, int kw, int stride, int groups) {
    VL_CHECK(out, "vl_conv_create: null out");
    VL_CHECK(cin > 0 && cout > 0 && h > 0 && w > 0 && kh > 0 && kw > 0 && stride > 0 && groups > 0, "vl_conv_create: bad geometry");
    VL_CHECK(cin % groups == 0 && cout % groups == 0, "vl_conv_create: channels not divisible by groups");
    VL_CHECK(h < 0x4000 && w < 0x4000 && kh < 256 && kw < 256, "vl_conv_create: plane too large");
    vl_conv_desc* d = (vl_conv_desc*)calloc(1, sizeof(vl_conv_desc));
    VL_CHECK(d, "vl_conv_create: out of host memory");
    d->cin = cin; d->h = h; d->w = w; d->cout = cout; d->kh = kh; d->kw = kw; d->stride = stride; d->groups = groups;
    tf_same_pad(h, kh, stride, &d->oh, &d->pt);
    tf_same_pad(w, kw, stride, &d->ow, &d->pl);
    d->cig = cin / groups;
    d->cog = cout / groups;
    d->K = kh * kw * d->cig;
    d->Kd = kh * kw * d->cog;
    int rc = build_ktab(&d->ktab_fwd, &d->ktab_len_fwd, kh, kw, d->cig, h, w);
    if (rc == 0 && stride == 1) rc = build_ktab(&d->ktab_bwd, &d->ktab_len_bwd, kh, kw, d->cog, d->oh, d->ow);
    if (rc) {
        vl_conv_destroy(d);
        vl_set_error("vl_conv_create: device table allocation failed");
        return 2;
    }
    *out = d;
    return 0;
}

extern "C" void vl_conv_destroy(vl_conv_desc* d) {
    if (!d) return;
    if (d->ktab_fwd) (void)hipFree(d->ktab_fwd);
    if (d->ktab_bwd) (void)hipFree(d->ktab_bwd);
    free(d);
}

extern "C" int vl_conv_out_hw(const vl_conv_desc* d, int* oh, int* ow) {
    VL_CHECK(d, "vl_conv_out_hw: null descriptor");
    if (oh) *oh = d->oh;
    if (ow) *ow = d->ow;
    return 0;
}

// ---- conv forward / dgrad launch ---------------------------------------------------------------
template <int BM, int WM, int WN>
static int launch_conv(const ConvGeom& g, const float* w, int64_t w_ld, int w_grp_stride, int Cog, int Cout_total,
                       const float* bias, const float* mask, int relu, float* y, hipStream_t s) {
    constexpr int BN = 128, BR = 16;
    using LA = DenseKX<BM, BR>;
    using LB = ConvGather<BN, BR>;
    typename LA::Params pa{w, w_ld, Cog, g.K, (int64_t)w_grp_stride};
    EpiConvNCHW::Params pe{y, bias, mask, relu, Cog, Cout_total, g.OHW, g.M, g.dOHW};
    const int tiles_i = ceil_div(Cog, BM), tiles_j = ceil_div(g.M, BN);
    const int rtiles = ceil_div(g.K, BR);
    dim3 grid(tiles_i * tiles_j, (unsigned)(Cout_total / Cog), 1);
    hipLaunchKernelGGL((mfma_contract<BM, BN, BR, WM, WN, LA, LB, EpiConvNCHW>), grid, dim3(NT), 0, s, pa, g, pe, tiles_i,
                       rtiles, rtiles);
    VL_LAUNCH_CHECK();
    return 0;
}

static int dispatch_conv(const ConvGeom& g, const float* w, int64_t w_ld, int w_grp_stride, int Cog, int Cout_total,
                         const float* bias, const float* mask, int relu, float* y, hipStream_t s) {
    // output-channel tile: 128 when it divides well, else 96 (conv1: 96, conv4: 192) or 64 (conv2 dgrad: 48)
    const int w128 = ceil_div(Cog, 128) * 128, w96 = ceil_div(Cog, 96) * 96, w64 = ceil_div(Cog, 64) * 64;
    if (w128 <= w96 && w128 <= w64) return launch_conv<128, 2, 2>(g, w, w_ld, w_grp_stride, Cog, Cout_total, bias, mask, relu, y, s);
    if (w96 <= w64) return launch_conv<96, 1, 4>(g, w, w_ld, w_grp_stride, Cog, Cout_total, bias, mask, relu, y, s);
    return launch_conv<64, 1, 4>(g, w, w_ld, w_grp_stride, Cog, Cout_total, bias, mask, relu, y, s);
}

extern "C" int vl_conv_fwd(const vl_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int n,
                           int relu, vl_stream_t stream) {
    VL_CHECK(d && x && w && y, "vl_conv_fwd: null argument");
    VL_CHECK(n > 0 && (int64_t)n * d->oh * d->ow < (1ll << 31), "vl_conv_fwd: bad batch %d", n);
    ConvGeom g;
    g.x = x; g.ktab = d->ktab_fwd; g.K = d->K; g.M = n * d->oh * d->ow; g.H = d->h; g.W = d->w;
    g.stride = d->stride; g.pt = d->pt; g.pl = d->pl; g.OHW = d->oh * d->ow; g.OW = d->ow;
    g.dOHW = make_fastdiv(g.OHW); g.dOW = make_fastdiv(g.OW);
    g.img_stride = (int64_t)d->cin * d->h * d->w;
    g.grp_stride = (int64_t)d->cig * d->h * d->w;
    g.total = g.img_stride * n;
    VL_CHECK(g.total < (1ll << 30) - (1 << 20), "vl_conv_fwd: input of %lld elements exceeds the 4 GiB buffer-offset range", (long long)g.total);
    // HWIO weights are the [K][Cout_total] GEMM operand as they stand; group g = column block g*cog.
    return dispatch_conv(g, w, d->cout, d->cog, d->cog, d->cout, bias, nullptr, relu, y, (hipStream_t)stream);
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
    g.x = dy; g.ktab = d->ktab_bwd; g.K = d->Kd; g.M = n * d->h * d->w; g.H = d->oh; g.W = d->ow;
    g.stride = 1; g.pt = d->kh - 1 - d->pt; g.pl = d->kw - 1 - d->pl; g.OHW = d->h * d->w; g.OW = d->w;
    g.dOHW = make_fastdiv(g.OHW); g.dOW = make_fastdiv(g.OW);
    g.img_stride = (int64_t)d->cout * d->oh * d->ow;
    g.grp_stride = (int64_t)d->cog * d->oh * d->ow;
    g.total = g.img_stride * n;
    VL_CHECK(g.total < (1ll << 30) - (1 << 20), "vl_conv_dgrad: dy of %lld elements exceeds the 4 GiB buffer-offset range", (long long)g.total);
    return dispatch_conv(g, wt, d->cin, d->cig, d->cig, d->cin, nullptr, relu_mask, 0, dx, (hipStream_t)stream);
}

// ---- conv wgrad -------------------------------------------------------------------------------
static int wgrad_splits(const vl_conv_desc* d, int n) {
    const int64_t M = (int64_t)n * d->oh * d->ow;
    const int rtiles = ceil_div(M, 32);
    const int tiles = ceil_div(d->K, 128) * ceil_div(d->cog, d->cog % 128 == 0 ? 128 : 96) * d->groups;
    int splits = ceil_div(2048, tiles);
    if (splits > rtiles) splits = rtiles;
    if (splits < 1) splits = 1;
    return splits;
}

extern "C" size_t vl_conv_wgrad_ws_bytes(const vl_conv_desc* d, int n) {
    if (!d || n <= 0) return 0;
    return (size_t)wgrad_splits(d, n) * d->K * d->cout * sizeof(float);
}

template <int BN, int WM, int WN>
static int launch_wgrad(const vl_conv_desc* d, const ConvGeom& g, const float* dy, float* dw, float* ws, int splits,
                        hipStream_t s) {
    constexpr int BM = 128, BR = 32;
    using LA = WgradGather<BM, BR>;
    using LB = DyRows<BN, BR>;
    typename LB::Params pb{dy, g.M, g.OHW, d->cog, d->cout, g.dOHW, (int64_t)d->cout * g.M};
    const int64_t slab = (int64_t)d->K * d->cout;
    // slab z: [K][Cout_total], group g = column block
    EpiRowMajor::Params pe{splits > 1 ? ws : dw, d->cout, d->K, d->cog, nullptr, nullptr, 0, d->cog, splits > 1 ? slab : 0};
    const int tiles_i = ceil_div(d->K, BM), tiles_j = ceil_div(d->cog, BN);
    const int rtiles = ceil_div(g.M, BR);
    const int per = ceil_div(rtiles, splits);
    dim3 grid(tiles_i * tiles_j, d->groups, splits);
    hipLaunchKernelGGL((mfma_contract<BM, BN, BR, WM, WN, LA, LB, EpiRowMajor>), grid, dim3(NT), 0, s, g, pb, pe, tiles_i, rtiles,
                       per);
    VL_LAUNCH_CHECK();
    if (splits > 1) {
        const int blocks = (int)((slab + 255) / 256 < 4096 ? (slab + 255) / 256 : 4096);
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, s, ws, dw, slab, splits, slab, nullptr, 0, 0, 0, nullptr);
        VL_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int vl_conv_wgrad(const vl_conv_desc* d, const float* x, const float* dy, float* dw, void* ws, size_t ws_bytes,
                             int n, vl_stream_t stream) {
    VL_CHECK(d && x && dy && dw, "vl_conv_wgrad: null argument");
    VL_CHECK(n > 0 && (int64_t)n * d->oh * d->ow < (1ll << 31), "vl_conv_wgrad: bad batch %d", n);
    const int splits = wgrad_splits(d, n);
    VL_CHECK(splits == 1 || (ws && ws_bytes >= vl_conv_wgrad_ws_bytes(d, n)), "vl_conv_wgrad: workspace too small (%zu < %zu)",
             ws_bytes, vl_conv_wgrad_ws_bytes(d, n));
    ConvGeom g;
    g.x = x; g.ktab = d->ktab_fwd; g.K = d->K; g.M = n * d->oh * d->ow; g.H = d->h; g.W = d->w;
    g.stride = d->stride; g.pt = d->pt; g.pl = d->pl; g.OHW = d->oh * d->ow; g.OW = d->ow;
    g.dOHW = make_fastdiv(g.OHW); g.dOW = make_fastdiv(g.OW);
    g.img_stride = (int64_t)d->cin * d->h * d->w;
    g.grp_stride = (int64_t)d->cig * d->h * d->w;
    g.total = g.img_stride * n;
    VL_CHECK(g.total < (1ll << 30) - (1 << 20) && (int64_t)n * d->cout * d->oh * d->ow < (1ll << 30),
             "vl_conv_wgrad: operand exceeds the 4 GiB buffer-offset range");
    if (d->cog % 128 == 0) return launch_wgrad<128, 2, 2>(d, g, dy, dw, (float*)ws, splits, (hipStream_t)stream);
    return launch_wgrad<96, 4, 1>(d, g, dy, dw, (float*)ws, splits, (hipStream_t)stream);
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

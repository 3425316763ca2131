// LSTM recurrence (tf.contrib.rnn.BasicLSTMCell under tf.nn.dynamic_rnn; reference models/lstm/lstm.py:9-20,34-42,102-143),
// weight-stationary cluster form for gfx950.
//
// The recurrence is T dependent steps of z_t = gx_t + h_{t-1} . Kh (Kh = kernel[D:], [H][4H], 1 MB for H = 256).  The per-clip
// kernel (pointwise.hip: lstm_seq_kernel) streams all of Kh through ONE CU every step: 1 MB at 64 B/clk is 7.8 us per step,
// and only `batch` CUs work (8 of 256 on one rank's shard of the 8-GPU job).  Here the gate COLUMNS are partitioned instead:
//
//   * a group of W = ceil(H/16) workgroups serves up to 8 clips; workgroup w owns hidden units [16 w, 16 w + 16), i.e. the 64
//     gate columns {q H + u} of its units, and keeps that slice of Kh ([H][64] floats = 64 KB for H = 256) in LDS for the
//     whole sequence -- no weight traffic after the prologue;
//   * per step a workgroup computes its 64 columns for its clips (a [clips][H] x [H][64] product on the vector ALU: 131 k MAC
//     for 8 clips -- the step is latency, not throughput), applies the gate math to its 16 units, and PUBLISHES h_t[clip][unit]
//     to the other workgroups of its group; they gather the full h_t before the next step.
//   * backward uses the SAME slice: dz_t of the own columns is local, its contribution to dh_{t-1}[k] for ALL k is
//     sum_j dz[j] Kh[k][j] over the own 64 columns, and the exchange is a reduce-scatter: every workgroup publishes its
//     [clips][H] partial and gathers, for its own 16 units, the W partials (summed in workgroup order: deterministic).
//     No transposed copy of Kh is needed any more.
//
// Exchange = the "data is the flag" granule form of the CDNA guide (Guideline 16, R2): every handed-off float travels as
// one aligned 8-byte {tag = epoch, value} word written by ONE relaxed agent-scope atomic store (write-through, sc1) and read
// by relaxed agent-scope atomic loads (bypass L1) until the tag matches; no flag, no fence, placement independent.
// Epochs count steps within the call (1..T); the TAG of a granule is the launch's base + epoch, bases taken from one process-wide
// counter that advances by T + 1 per launch, so a tag never repeats and no granule of an earlier launch can match (the workspace is
// zeroed once when it is allocated, vltf.h; round 1 - 3 zeroed the granules before every launch: a 5 us memset in front of both
// recurrences of every step).  Buffers alternate by epoch parity (a workgroup can be at most one step ahead of another, see below).
// Spins are bounded: a workgroup that never sees its granules sets the status word and goes on (wrong results, no hang).  The status word is STICKY: no launch clears it,
// vl_lstm_seq_status reads AND resets it, so a forward call's time-out is still there after the backward call and the engine's
// one check per step (engine._finish_step) sees every launch of that step.
//
//   why two buffers suffice: a workgroup publishes epoch e+1 only after it has gathered ALL of epoch e, which includes the
//   slowest workgroup's epoch-e values; the slowest one publishes those only after it finished reading epoch e-1.  So while
//   anyone still reads epoch e-1, nobody can have written epoch e+1 (same parity).
//
// Groups are dealt so that a group's workgroups share blockIdx.x % 8 (= one XCD, one L2) when the group count is a multiple of
// 8: speed only, nothing depends on it.
#include <stdlib.h>

#include "common.h"

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

static constexpr int LU = 16;        // hidden units per workgroup
static constexpr int LC = 4 * LU;    // gate columns per workgroup
static constexpr int CPG = 8;        // clips per group (at most)
static constexpr int LNT = 256;      // threads per workgroup
static constexpr int LH_MAX = 512;   // largest hidden size of this form (LDS: H x 64 floats + 8 x H floats)
static constexpr unsigned SPIN_LIMIT = 1u << 18;   // polls before giving up (~tens of ms; a healthy hop takes microseconds)
// test hooks (vl_lstm_seq_test_hooks): a shorter spin limit, and one workgroup index whose granules are never published, so that
// tests/ can see the time-out path end to end; defaults = production behaviour
static unsigned g_spin_limit = SPIN_LIMIT;
static int g_mute_workgroup = -1;
static unsigned g_epoch_base = 0;                  // tag base of the next launch (see the header comment)
static constexpr size_t STATUS_BYTES = 256;        // status block at the start of the workspace (word 0: timed-out flag)

template <int I> struct IntK { static constexpr int value = I; };

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ void store_granule(u64* g, unsigned epoch, float v) {
    __hip_atomic_store((gu64*)g, ((u64)epoch << 32) | (u64)__float_as_uint(v), RLX_AGENT);
}

// One thread's share of a gather: granules g[idx(m)] for m < n (n <= MAXN) are re-read until every tag equals `epoch`.
// Branch-free inside a pass -- slots past n re-read the thread's last granule -- so that all loads of a pass are in flight together
// (per-slot branches made hipcc wait for every load separately: n dependent L2 round trips per pass).
// Returns false (and raises the status word) when the bounded spin runs out.
template <int MAXN, class IDX>
__device__ __forceinline__ bool gather_granules(const u64* g, int n, IDX idx, unsigned epoch, float (&v)[MAXN], unsigned* status,
                                                unsigned spin_limit) {
    if (n <= 0) return true;
    const gu64* a[MAXN];
#pragma unroll
    for (int m = 0; m < MAXN; ++m) a[m] = (const gu64*)(g + idx(m < n ? m : n - 1));
    for (unsigned spins = 0;; ++spins) {
        u64 x[MAXN];
#pragma unroll
        for (int m = 0; m < MAXN; ++m) x[m] = __hip_atomic_load(a[m], RLX_AGENT);
        bool ok = true;
#pragma unroll
        for (int m = 0; m < MAXN; ++m) {
            v[m] = __uint_as_float((unsigned)x[m]);
            ok &= (unsigned)(x[m] >> 32) == epoch;
        }
        if (ok) return true;
        if (spins >= spin_limit) {
            __hip_atomic_store((gu32*)status, 1u, RLX_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

struct LstmClusterArgs {
    const float* gx;      // fwd: [B T][4H] input projections (+bias)
    const float* kh;      // [H][4H]
    const float* h0;      // nullable [B][H]
    const float* c0;      // nullable [B][H]
    float* act;           // [B T][4H] activated gates (fwd: out, bwd: in)
    float* cseq;          // [B T][H]
    float* hseq;          // fwd out
    float* hprev;         // fwd out
    const float* dout;    // bwd: nullable [B T][H]
    float* dz;            // bwd out [B T][4H]
    float* dh0;           // bwd: nullable out [B][H]
    float* dc0;           // bwd: nullable out [B][H]
    u64* xch;             // exchange granules
    unsigned* status;
    int B, T, H, Hp, G, W, cpg;
    float forget_bias;
    unsigned spin_limit;
    int mute;             // test hook: this workgroup (blockIdx.x) publishes nothing; -1 = none
    unsigned ebase;       // tag of epoch e = ebase + e (unique per launch)
};

// ---- forward ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LNT) void lstm_cluster_fwd_kernel(const LstmClusterArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = p.H, Hp = p.Hp, H4 = 4 * p.H, T = p.T;
    float* Ks = sm;                   // [Hp][LC]  (rows H .. Hp - 1: zeros)
    float* hb = Ks + (size_t)Hp * LC; // [CPG][Hp]   h_{t-1} of the group's clips
    float* zb = hb + CPG * Hp;        // [KP][2 S][LC] recurrent part of the own columns' pre-activations per reduction slice (KP * 2 S = 8 rows)
    const int tid = threadIdx.x;
    const int grp = blockIdx.x % p.G, w = blockIdx.x / p.G;
    const int clip0 = grp * p.cpg;
    const int nclips = min(p.cpg, p.B - clip0);
    // the weight slice, once
    for (int i = tid; i < Hp * LC; i += LNT) {
        const int k = i / LC, j = i - k * LC, q = j / LU, u = w * LU + (j - q * LU);
        Ks[i] = (u < H && k < H) ? p.kh[(int64_t)k * H4 + q * H + u] : 0.f;
    }
    for (int i = tid; i < CPG * Hp; i += LNT) {
        const int c = i / Hp, k = i - c * Hp;
        hb[i] = (p.h0 && c < nclips && k < H) ? p.h0[(int64_t)(clip0 + c) * H + k] : 0.f;
    }
    // gate threads: (clip gc, unit gul) for tid < CPG * LU
    const int gc = tid / LU, gul = tid - gc * LU, gu = w * LU + gul;
    const bool glive = tid < CPG * LU && gc < nclips && gu < H;
    float cst = (glive && p.c0) ? p.c0[(int64_t)(clip0 + gc) * H + gu] : 0.f;
    // matvec threads: wave wv works on column j for the clip pair (slot, slot + S) over the kpart-th slice of the reduction.
    // More than 4 clips: 4 pairs, whole reduction each; 3-4 clips: 2 pairs x 2 slices; 1-2 clips: 1 pair x 4 slices -- so the
    // 8-clip shard of the 8-GPU job (one clip per group) does not leave three waves idle behind one 256-long dot product.
    const int j = tid & (LC - 1), wv = tid >> 6;
    const int S = nclips > 4 ? 4 : (nclips > 2 ? 2 : 1), KP = 4 / S;
    const int slot = wv & (S - 1), kpart = wv / S;
    const int klen = ((H + KP - 1) / KP + 3) / 4 * 4, kbeg = kpart * klen, kend = min(H, kbeg + klen);
    // gather share: granule i = tid + LNT m  ->  (clip i / H, unit i % H) of this group
    const int ngran = nclips * H;
    constexpr int MAXG = CPG * LH_MAX / LNT;
    const int myn = ngran > tid ? (ngran - tid + LNT - 1) / LNT : 0;
    u64* xg = p.xch + (size_t)grp * CPG * Hp;                  // + parity * G * CPG * Hp
    const size_t xpar = (size_t)p.G * CPG * Hp;
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        float zx[4] = {0.f, 0.f, 0.f, 0.f};
        const int64_t r = (int64_t)(clip0 + gc) * T + t;
        if (glive) {
#pragma unroll
            for (int q = 0; q < 4; ++q) zx[q] = p.gx[r * H4 + q * H + gu];
        }
        if (t > 0) {
            const u64* src = xg + ((t - 1) & 1) * xpar;
            auto run = [&](auto tag) {                         // unrolled to the share's size: slots past it would re-read (sc1: L2 traffic)
                constexpr int N = decltype(tag)::value;
                float v[N];
                gather_granules<N>(src, myn, [&](int m) { const int i = tid + LNT * m; const int c = i / H; return c * Hp + (i - c * H); },
                                   p.ebase + (unsigned)t, v, p.status, p.spin_limit);
#pragma unroll
                for (int m = 0; m < N; ++m)
                    if (m < myn) {
                        const int i = tid + LNT * m, c = i / H;
                        hb[c * Hp + (i - c * H)] = v[m];
                    }
            };
            if (myn <= 1) run(IntK<1>{});
            else if (myn <= 2) run(IntK<2>{});
            else if (myn <= 8) run(IntK<8>{});
            else run(IntK<MAXG>{});
        }
        __syncthreads();                                       // h_{t-1} complete in LDS
        const float hp = glive ? hb[gc * Hp + gu] : 0.f;
        {
            float z0 = 0.f, z1 = 0.f;
            const float* h0p = hb + slot * Hp;
            const float* h1p = hb + (slot + S) * Hp;
            const float* kp = Ks + j;
            // Hp is a multiple of 4; hb and Ks hold zeros from H to Hp.  Four reduction quads per pass, their 24 LDS reads issued
            // together: the loop is LDS LATENCY, not bandwidth (one quad per pass: 2.3 us of a 4.6 us step at 64 positions per wave,
            // 4.2 us at 256 -- profiles/r04_lstm_step_parts.txt)
            int k = kbeg;
            for (; k + 16 <= kend; k += 16) {
                float4 a[4], b[4];
                float wq[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a[u] = *reinterpret_cast<const float4*>(h0p + k + 4 * u);
                    b[u] = *reinterpret_cast<const float4*>(h1p + k + 4 * u);
#pragma unroll
                    for (int e = 0; e < 4; ++e) wq[u][e] = kp[(k + 4 * u + e) * LC];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {                  // the order of the sums is the one-quad loop's: same bits
                    z0 += a[u].x * wq[u][0] + a[u].y * wq[u][1] + a[u].z * wq[u][2] + a[u].w * wq[u][3];
                    z1 += b[u].x * wq[u][0] + b[u].y * wq[u][1] + b[u].z * wq[u][2] + b[u].w * wq[u][3];
                }
            }
            for (; k < kend; k += 4) {
                const float4 a = *reinterpret_cast<const float4*>(h0p + k);
                const float4 b = *reinterpret_cast<const float4*>(h1p + k);
                const float w0 = kp[(k + 0) * LC], w1 = kp[(k + 1) * LC], w2 = kp[(k + 2) * LC], w3 = kp[(k + 3) * LC];
                z0 += a.x * w0 + a.y * w1 + a.z * w2 + a.w * w3;
                z1 += b.x * w0 + b.y * w1 + b.z * w2 + b.w * w3;
            }
            zb[(kpart * 2 * S + slot) * LC + j] = z0;
            zb[(kpart * 2 * S + slot + S) * LC + j] = z1;
        }
        __syncthreads();                                       // own columns complete; everyone has consumed h_{t-1}
        if (glive) {
            float zr[4] = {0.f, 0.f, 0.f, 0.f};
            for (int kp_ = 0; kp_ < KP; ++kp_)                   // fixed order: reproducible
#pragma unroll
                for (int q = 0; q < 4; ++q) zr[q] += zb[(kp_ * 2 * S + gc) * LC + q * LU + gul];
            const float zi = zx[0] + zr[0], zj = zx[1] + zr[1], zf = zx[2] + zr[2], zo = zx[3] + zr[3];
            const float gi = sigm(zi), gj = tanhf(zj), gf = sigm(zf + p.forget_bias), go = sigm(zo);
            cst = cst * gf + gi * gj;
            const float h = tanhf(cst) * go;
            if (t + 1 < T && (int)blockIdx.x != p.mute) store_granule(xg + (t & 1) * xpar + gc * Hp + gu, p.ebase + (unsigned)(t + 1), h);
            float* a = p.act + r * H4 + gu;
            a[0] = gi; a[H] = gj; a[2 * H] = gf; a[3 * H] = go;
            p.cseq[r * H + gu] = cst;
            p.hseq[r * H + gu] = h;
            p.hprev[r * H + gu] = hp;
        }
        // no barrier needed here: the next iteration writes hb only after its gather, and every read of hb of this iteration
        // precedes the barrier above; zb is rewritten only after the next iteration's first barrier
    }
}

// ---- backward -----------------------------------------------------------------------------------------------------------
// partial[c][k] = sum over the own 64 columns of dz[c][j] Kh[k][j], for the first NC clips, published as granules
template <int NC>
__device__ __forceinline__ void publish_partials(const float* KsT, const float* zl, u64* dst, int H, int Hp, int nclips, unsigned epoch,
                                                 int tid) {
    for (int k = tid; k < H; k += LNT) {
        float acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = 0.f;
        // eight columns per pass, their LDS reads (8 weights, NC x 2 quads of dz -- broadcast reads) issued together: as in the forward
        // product the loop is LDS latency, not bandwidth; the sums keep the one-column-per-pass order (same bits)
        for (int jj = 0; jj < LC; jj += 8) {
            float wv[8];
            float4 za[NC], zc[NC];
#pragma unroll
            for (int e = 0; e < 8; ++e) wv[e] = KsT[(jj + e) * Hp + k];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                za[c] = *reinterpret_cast<const float4*>(zl + c * LC + jj);
                zc[c] = *reinterpret_cast<const float4*>(zl + c * LC + jj + 4);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                acc[c] += za[c].x * wv[0];
                acc[c] += za[c].y * wv[1];
                acc[c] += za[c].z * wv[2];
                acc[c] += za[c].w * wv[3];
                acc[c] += zc[c].x * wv[4];
                acc[c] += zc[c].y * wv[5];
                acc[c] += zc[c].z * wv[6];
                acc[c] += zc[c].w * wv[7];
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (c < nclips) store_granule(dst + (size_t)c * Hp + k, epoch, acc[c]);
    }
}

__global__ __launch_bounds__(LNT) void lstm_cluster_bwd_kernel(const LstmClusterArgs p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = p.H, Hp = p.Hp, H4 = 4 * p.H, T = p.T, W = p.W;
    float* KsT = sm;                      // [LC][Hp]  KsT[j][k] = kh[k][column j of this workgroup]
    float* zl = KsT + (size_t)LC * Hp;    // [CPG][LC] dz_t of the own columns
    const int tid = threadIdx.x;
    const int grp = blockIdx.x % p.G, w = blockIdx.x / p.G;
    const int clip0 = grp * p.cpg;
    const int nclips = min(p.cpg, p.B - clip0);
    for (int i = tid; i < LC * Hp; i += LNT) {
        const int jj = i / Hp, k = i - jj * Hp, q = jj / LU, u = w * LU + (jj - q * LU);
        KsT[i] = (u < H && k < H) ? p.kh[(int64_t)k * H4 + q * H + u] : 0.f;
    }
    for (int i = tid; i < CPG * LC; i += LNT) zl[i] = 0.f;
    const int gc = tid / LU, gul = tid - gc * LU, gu = w * LU + gul;
    const bool glive = tid < CPG * LU && gc < nclips && gu < H;
    float dc = 0.f, dh = 0.f;
    // exchange: P[parity][group][source workgroup][clip][k]
    const size_t pgrp = (size_t)W * CPG * Hp, ppar = (size_t)p.G * pgrp;
    u64* pg = p.xch + (size_t)grp * pgrp;
    constexpr int MAXW = LH_MAX / LU;
    auto gather_dh = [&](unsigned epoch) {                      // sum over source workgroups of their partial for (gc, gu)
        const u64* src = pg + (epoch & 1) * ppar + (size_t)gc * Hp + gu;
        const size_t wstride = (size_t)CPG * Hp;
        auto run = [&](auto tag) {
            constexpr int N = decltype(tag)::value;
            float v[N];
            gather_granules<N>(src, glive ? W : 0, [&](int m) { return (size_t)m * wstride; }, p.ebase + epoch, v, p.status, p.spin_limit);
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < N; ++m)
                if (m < W) s += v[m];                           // fixed order: reproducible
            return s;
        };
        if (W <= 4) return run(IntK<4>{});
        if (W <= 16) return run(IntK<16>{});
        return run(IntK<MAXW>{});
    };
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        const int64_t r = (int64_t)(clip0 + gc) * T + t;
        const unsigned epoch = (unsigned)(T - t);               // 1 .. T
        float din = 0.f, gi = 0.f, gj = 0.f, gf = 0.f, go = 0.f, cc = 0.f, cp = 0.f;
        if (glive) {
            din = p.dout ? p.dout[r * H + gu] : 0.f;
            const float* a = p.act + r * H4 + gu;
            gi = a[0]; gj = a[H]; gf = a[2 * H]; go = a[3 * H];
            cc = p.cseq[r * H + gu];
            cp = t > 0 ? p.cseq[(r - 1) * H + gu] : (p.c0 ? p.c0[(int64_t)(clip0 + gc) * H + gu] : 0.f);
        }
        if (t < T - 1) dh = gather_dh(epoch - 1);               // published by every workgroup of the group at step t + 1
        if (glive) {
            din += dh;
            const float tc = tanhf(cc);
            const float d_o = din * tc;
            const float dcv = dc + din * go * (1.f - tc * tc);
            const float zi = dcv * gj * gi * (1.f - gi), zj = dcv * gi * (1.f - gj * gj);
            const float zf = dcv * cp * gf * (1.f - gf), zo = d_o * go * (1.f - go);
            dc = dcv * gf;
            float* zp = p.dz + r * H4 + gu;
            zp[0] = zi; zp[H] = zj; zp[2 * H] = zf; zp[3 * H] = zo;
            float* zz = zl + gc * LC + gul;
            zz[0] = zi; zz[LU] = zj; zz[2 * LU] = zf; zz[3 * LU] = zo;
        }
        __syncthreads();                                        // own dz_t in LDS
        if ((t > 0 || p.dh0) && (int)blockIdx.x != p.mute) {
            u64* dst = pg + (epoch & 1) * ppar + (size_t)w * CPG * Hp;
            if (nclips > 4) publish_partials<8>(KsT, zl, dst, H, Hp, nclips, p.ebase + epoch, tid);
            else if (nclips > 2) publish_partials<4>(KsT, zl, dst, H, Hp, nclips, p.ebase + epoch, tid);
            else if (nclips > 1) publish_partials<2>(KsT, zl, dst, H, Hp, nclips, p.ebase + epoch, tid);
            else publish_partials<1>(KsT, zl, dst, H, Hp, nclips, p.ebase + epoch, tid);
        }
        __syncthreads();                                        // zl consumed before the next step rewrites it
    }
    if (p.dh0) {
        dh = gather_dh((unsigned)T);
        if (glive) p.dh0[(int64_t)(clip0 + gc) * H + gu] = dh;
    }
    if (p.dc0 && glive) p.dc0[(int64_t)(clip0 + gc) * H + gu] = dc;
}

// ---- launch ---------------------------------------------------------------------------------------------------------------
static int cluster_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus;
}

struct ClusterPlan {
    int W, Hp, maxG, chunk;      // workgroups per group, padded H, groups per launch (one workgroup per CU), clips per launch
    size_t xch_fwd, xch_bwd;     // granules per launch
};

static ClusterPlan cluster_plan(int H) {
    ClusterPlan c;
    c.W = (H + LU - 1) / LU;
    c.Hp = (H + 3) / 4 * 4;
    c.maxG = cluster_cus() / c.W;                 // every workgroup resident: at most one per CU
    if (c.maxG < 1) c.maxG = 1;
    c.chunk = c.maxG * CPG;
    c.xch_fwd = (size_t)2 * c.maxG * CPG * c.Hp;
    c.xch_bwd = (size_t)2 * c.maxG * c.W * CPG * c.Hp;
    return c;
}

bool vl_lstm_cluster_ok(int H) { return H >= 1 && H <= LH_MAX; }

size_t vl_lstm_cluster_ws_bytes(int H) {
    if (!vl_lstm_cluster_ok(H)) return 0;
    const ClusterPlan c = cluster_plan(H);
    return STATUS_BYTES + sizeof(u64) * (c.xch_fwd > c.xch_bwd ? c.xch_fwd : c.xch_bwd);
}

// Runs one direction over all clips, in chunks of at most maxG * 8 clips (one launch each).
int vl_lstm_cluster_run(bool bwd, LstmClusterArgs a, int batch, void* ws, size_t ws_bytes, hipStream_t s) {
    const ClusterPlan c = cluster_plan(a.H);
    VL_CHECK(ws && ws_bytes >= vl_lstm_cluster_ws_bytes(a.H), "vl_lstm_seq: workspace too small (%zu < %zu bytes)", ws_bytes,
             vl_lstm_cluster_ws_bytes(a.H));
    a.W = c.W;
    a.Hp = c.Hp;
    a.status = (unsigned*)ws;
    a.xch = (u64*)((char*)ws + STATUS_BYTES);
    const size_t lds = bwd ? sizeof(float) * ((size_t)LC * c.Hp + CPG * LC) : sizeof(float) * ((size_t)c.Hp * LC + CPG * c.Hp + CPG * LC);   // zb: KP * clips <= 8 rows
    static bool attr_set[2] = {false, false};
    const void* kern = bwd ? reinterpret_cast<const void*>(lstm_cluster_bwd_kernel) : reinterpret_cast<const void*>(lstm_cluster_fwd_kernel);
    if (!attr_set[bwd]) {
        VL_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_set[bwd] = true;
    }
    a.spin_limit = g_spin_limit;
    a.mute = g_mute_workgroup;
    const float *gx = a.gx, *h0 = a.h0, *c0 = a.c0, *dout = a.dout;
    float *act = a.act, *cseq = a.cseq, *hseq = a.hseq, *hprev = a.hprev, *dz = a.dz, *dh0 = a.dh0, *dc0 = a.dc0;
    for (int b0 = 0; b0 < batch; b0 += c.chunk) {
        const int nb = batch - b0 < c.chunk ? batch - b0 : c.chunk;
        int G = nb < c.maxG ? nb : c.maxG;
        const int cpg = (nb + G - 1) / G;
        G = (nb + cpg - 1) / cpg;
        a.B = nb; a.G = G; a.cpg = cpg;
        const int64_t ro = (int64_t)b0 * a.T, so = (int64_t)b0 * a.H;
        a.gx = gx ? gx + ro * 4 * a.H : nullptr;
        a.h0 = h0 ? h0 + so : nullptr;
        a.c0 = c0 ? c0 + so : nullptr;
        a.act = act + ro * 4 * a.H;
        a.cseq = cseq + ro * a.H;
        a.hseq = hseq ? hseq + ro * a.H : nullptr;
        a.hprev = hprev ? hprev + ro * a.H : nullptr;
        a.dout = dout ? dout + ro * a.H : nullptr;
        a.dz = dz ? dz + ro * 4 * a.H : nullptr;
        a.dh0 = dh0 ? dh0 + so : nullptr;
        a.dc0 = dc0 ? dc0 + so : nullptr;
        a.xch = (u64*)((char*)ws + STATUS_BYTES);                 // the kernels index parity blocks by their own G: [parity][G][...]
        // this launch's tags: base + 1 .. base + T, never used before in this process (any workspace, either direction)
        a.ebase = __atomic_fetch_add(&g_epoch_base, (unsigned)a.T + 1u, __ATOMIC_RELAXED);
        if (a.ebase > 0xfff00000u) {                             // (after ~10^8 launches) start over behind a memset of the whole exchange
            VL_HIP(hipMemsetAsync(a.xch, 0, ws_bytes - STATUS_BYTES, s));
            __atomic_store_n(&g_epoch_base, (unsigned)a.T + 1u, __ATOMIC_RELAXED);
            a.ebase = 0;
        }
        if (bwd) hipLaunchKernelGGL(lstm_cluster_bwd_kernel, dim3(G * c.W), dim3(LNT), lds, s, a);
        else hipLaunchKernelGGL(lstm_cluster_fwd_kernel, dim3(G * c.W), dim3(LNT), lds, s, a);
        VL_LAUNCH_CHECK();
    }
    return 0;
}

static const bool kPerClip = vl_exp_env("VL_LSTM_PERCLIP") != nullptr;
static size_t perclip_ws_bytes(int H) { return STATUS_BYTES + (size_t)4 * H * H * sizeof(float); }   // kh transposed, for the backward

extern "C" size_t vl_lstm_seq_ws_bytes(int batch, int T, int H) {
    (void)batch; (void)T;
    if (H < 1) return 0;
    const size_t a = vl_lstm_cluster_ok(H) ? vl_lstm_cluster_ws_bytes(H) : 0, b = perclip_ws_bytes(H);
    return a > b ? a : b;
}

extern "C" int vl_lstm_seq_fwd(const float* gx, const float* kh, const float* h0, const float* c0, float* act, float* cseq,
                               float* hseq, float* hprev, int batch, int T, int H, float forget_bias, void* ws, size_t ws_bytes,
                               vl_stream_t stream) {
    VL_CHECK(gx && kh && act && cseq && hseq && hprev, "vl_lstm_seq_fwd: null argument");
    VL_CHECK(batch > 0 && T > 0 && H > 0 && H <= 1024, "vl_lstm_seq_fwd: bad shape (hidden size must be <= 1024)");
    VL_CHECK(ws && ws_bytes >= vl_lstm_seq_ws_bytes(batch, T, H), "vl_lstm_seq_fwd: workspace smaller than vl_lstm_seq_ws_bytes");
    if (vl_lstm_cluster_ok(H) && !kPerClip) {
        LstmClusterArgs a = {};
        a.gx = gx; a.kh = kh; a.h0 = h0; a.c0 = c0; a.act = act; a.cseq = cseq; a.hseq = hseq; a.hprev = hprev;
        a.T = T; a.H = H; a.forget_bias = forget_bias;
        return vl_lstm_cluster_run(false, a, batch, ws, ws_bytes, (hipStream_t)stream);
    }
    return vl_lstm_perclip_fwd(gx, kh, h0, c0, act, cseq, hseq, hprev, batch, T, H, forget_bias, (hipStream_t)stream);
}

extern "C" int vl_lstm_seq_bwd(const float* dout, const float* kh, const float* act, const float* cseq, const float* c0, float* dz,
                               float* dh0, float* dc0, int batch, int T, int H, void* ws, size_t ws_bytes, vl_stream_t stream) {
    VL_CHECK(kh && act && cseq && dz, "vl_lstm_seq_bwd: null argument");
    VL_CHECK(batch > 0 && T > 0 && H > 0 && H <= 1024, "vl_lstm_seq_bwd: bad shape (hidden size must be <= 1024)");
    VL_CHECK(ws && ws_bytes >= vl_lstm_seq_ws_bytes(batch, T, H), "vl_lstm_seq_bwd: workspace smaller than vl_lstm_seq_ws_bytes");
    if (vl_lstm_cluster_ok(H) && !kPerClip) {
        LstmClusterArgs a = {};
        a.kh = kh; a.c0 = c0; a.act = const_cast<float*>(act); a.cseq = const_cast<float*>(cseq); a.dout = dout; a.dz = dz;
        a.dh0 = dh0; a.dc0 = dc0; a.T = T; a.H = H;
        return vl_lstm_cluster_run(true, a, batch, ws, ws_bytes, (hipStream_t)stream);
    }
    float* kh_t = (float*)((char*)ws + STATUS_BYTES);
    if (vl_transpose(kh, (int64_t)4 * H, kh_t, H, 4 * H, stream)) return 1;
    return vl_lstm_perclip_bwd(dout, kh_t, act, cseq, c0, dz, dh0, dc0, batch, T, H, (hipStream_t)stream);
}

extern "C" int vl_lstm_seq_status(void* ws, int* timed_out) {
    VL_CHECK(ws && timed_out, "vl_lstm_seq_status: null argument");
    unsigned v = 0;
    VL_HIP(hipMemcpy(&v, ws, sizeof(v), hipMemcpyDeviceToHost));       // synchronises with the launches that may have raised it
    if (v) VL_HIP(hipMemset(ws, 0, sizeof(v)));
    *timed_out = (int)v;
    return 0;
}

extern "C" int vl_lstm_seq_test_hooks(unsigned spin_limit, int mute_workgroup) {
    g_spin_limit = spin_limit ? spin_limit : SPIN_LIMIT;
    g_mute_workgroup = mute_workgroup;
    return 0;
}

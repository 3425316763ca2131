// Round 4, DESIGN 6 (follow-up of pk_modifier_leak.hip, which found nothing between two waves of ONE kernel): the victim
// `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` in one kernel, and on a second stream a PARTNER KERNEL shaped like the split-bf16
// weight gradient's inner loop: LDS reads -> head / tail split (v_cvt_pk_bf16_f32, v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1]) -> bf16 MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o pk_cross_kernel pk_cross_kernel.hip && ./pk_cross_kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// VICTIM 0: v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]   1: plain v_pk_add_f32   2: two v_add_f32   3: op_sel:[0,1] alone (low result takes
// src1's HIGH register)   4: op_sel_hi:[1,0] alone   5: op_sel:[1,0] (src0's high register)   6: v_pk_mul_f32 op_sel:[0,1]
// 7: v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1] (the split kernels' own form)   8: v_pk_fma_f32 op_sel:[0,0,1]
template <int VICTIM>
__global__ __launch_bounds__(256) void victim(unsigned* __restrict__ errs, int iters, float* __restrict__ samples) {
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63;
    unsigned bad[2] = {0, 0}, kind[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        f32x2 a = {1.f + lane + (it & 7), 1000.f + lane}, b = {0.125f * (it & 15), 64.f + (it & 3)};
        asm volatile("" : "+v"(a), "+v"(b));
        f32x2 d, want;
        if constexpr (VICTIM == 0) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); want = f32x2{a.x + b.y, a.y + b.x}; }
        if constexpr (VICTIM == 1) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); want = a + b; }
        if constexpr (VICTIM == 3) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b)); want = f32x2{a.x + b.y, a.y + b.y}; }
        if constexpr (VICTIM == 4) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); want = f32x2{a.x + b.x, a.y + b.x}; }
        if constexpr (VICTIM == 5) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b)); want = f32x2{a.y + b.x, a.y + b.y}; }
        if constexpr (VICTIM == 6) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b)); want = f32x2{a.x * b.y, a.y * b.y}; }
        if constexpr (VICTIM == 7) { asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); want = f32x2{a.x - b.x, a.y - b.y}; }
        if constexpr (VICTIM == 8) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(a)); want = f32x2{__builtin_fmaf(a.x, b.x, a.y), __builtin_fmaf(a.y, b.y, a.y)}; }
        if constexpr (VICTIM == 2) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(d.x) : "v"(a.x), "v"(b.y)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(d.y) : "v"(a.y), "v"(b.x)); want = f32x2{a.x + b.y, a.y + b.x}; }
        const bool w0 = d.x != want.x, w1 = d.y != want.y;
        bad[0] += w0; bad[1] += w1;
        if ((w0 || w1) && samples) {
            const unsigned slot = atomicAdd(errs + 31, 1u);
            if (slot < 6) { float* o = samples + slot * 8; o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; o[4] = d.x; o[5] = d.y; o[6] = want.x; o[7] = want.y; }
        }
        if (w0 || w1) {
            if (d.x == a.x + b.x && d.y == a.y + b.y) kind[0]++;            // plain: operand selection lost
            else if (d.x == a.x - b.y && d.y == a.y - b.x) kind[1]++;       // selection kept, src1 negated
            else if (d.x == a.x - b.x && d.y == a.y - b.y) kind[2]++;       // the partner's modifiers instead of its own
            else kind[3]++;
        }
    }
    const int q = lane >> 4;
    if (bad[0]) atomicAdd(errs + q * 6 + 0, bad[0]);
    if (bad[1]) atomicAdd(errs + q * 6 + 1, bad[1]);
    for (int j = 0; j < 4; ++j) if (kind[j]) atomicAdd(errs + q * 6 + 2 + j, kind[j]);
    if (iters < 0) pad[0] = 1.f;
}

// PARTNER bit 0: LDS reads + head/tail split, bit 1: bf16 MFMAs, bit 2: f32 MFMAs instead
template <int PARTNER>
__global__ __launch_bounds__(256) void partner(float* __restrict__ sink, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.001f * (float)i + 0.37f;
    __syncthreads();
    f32x16 acc = {};
    i32x4 hd = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80}, tl = hd;
    const float* p = lds + (threadIdx.x & 63) * 8;
    for (int it = 0; it < iters; ++it) {
        if constexpr (PARTNER & 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(p + 2 * q + 512 * (it & 3));
                const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
                const f32x2 r = {v[0] - __uint_as_float(h << 16), v[1] - __uint_as_float(h & 0xffff0000u)};
                hd[q] = (int)h;
                tl[q] = (int)__builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
            }
        }
        if constexpr (PARTNER & 4) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(hd[0]), __int_as_float(tl[1]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(hd[2]), __int_as_float(tl[3]), acc, 0, 0, 0);
        } else if constexpr (PARTNER & 2) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hd), __builtin_bit_cast(bf16x8, hd), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hd), __builtin_bit_cast(bf16x8, tl), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, tl), __builtin_bit_cast(bf16x8, hd), acc, 0, 0, 0);
        } else {
            asm volatile("" : "+v"(hd), "+v"(tl));
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s + (float)hd[0] + (float)tl[3] == 123.456f) sink[threadIdx.x] = s;
}

template <int VICTIM, int PARTNER>
static void run(hipStream_t s0, hipStream_t s1, unsigned* errs, float* sink, size_t partner_lds) {
    static const char* vn[] = {"v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_add_f32 (no modifiers)", "2 x v_add_f32", "v_pk_add_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel_hi:[1,0]", "v_pk_add_f32 op_sel:[1,0]", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1]", "v_pk_fma_f32 op_sel:[0,0,1]"};
    static const char* pn[] = {"none", "LDS reads + head/tail split", "bf16 MFMAs", "LDS reads + split + bf16 MFMAs", "f32 MFMAs", "LDS reads + split + f32 MFMAs"};
    CK(hipFuncSetAttribute((const void*)partner<PARTNER>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipMemsetAsync(errs, 0, 32 * sizeof(unsigned), s0));
    CK(hipStreamSynchronize(s0));
    if (PARTNER) hipLaunchKernelGGL((partner<PARTNER>), dim3(256), dim3(256), partner_lds, s1, sink, 60000);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((victim<VICTIM>), dim3(2048), dim3(256), 12544, s0, errs, 20000, sink + 4096);
    CK(hipDeviceSynchronize());
    unsigned h[24];
    CK(hipMemcpy(h, errs, sizeof(h), hipMemcpyDeviceToHost));
    unsigned tot = 0;
    for (int i = 0; i < 24; ++i) tot += h[i];
    printf("victim %-42s | partner kernel (%3zu KB LDS) %-32s | wrong (lo, hi | =plain, =negated, =partner's, other) by lane quarter:", vn[VICTIM], partner_lds >> 10, pn[PARTNER == 4 ? 4 : PARTNER == 5 ? 5 : PARTNER]);
    for (int q = 0; q < 4; ++q) printf("  q%d %u %u | %u %u %u %u", q, h[q * 6], h[q * 6 + 1], h[q * 6 + 2], h[q * 6 + 3], h[q * 6 + 4], h[q * 6 + 5]);
    printf("%s\n", tot ? "   <== WRONG" : "");
    if (tot) {
        float hs[48];
        CK(hipMemcpy(hs, sink + 4096, sizeof(hs), hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; ++i) printf("      a = {%g, %g} b = {%g, %g}: got {%g, %g}, want {%g, %g}\n", hs[8 * i], hs[8 * i + 1], hs[8 * i + 2], hs[8 * i + 3], hs[8 * i + 4], hs[8 * i + 5], hs[8 * i + 6], hs[8 * i + 7]);
    }
    fflush(stdout);
}

int main() {
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    unsigned* errs; float* sink;
    CK(hipMalloc(&errs, 256)); CK(hipMalloc(&sink, 8192 * 4));
    const size_t lds = 16384;
    run<0, 0>(s0, s1, errs, sink, lds); run<0, 1>(s0, s1, errs, sink, lds); run<0, 2>(s0, s1, errs, sink, lds); run<0, 3>(s0, s1, errs, sink, lds);
    run<0, 4>(s0, s1, errs, sink, lds); run<0, 5>(s0, s1, errs, sink, lds);
    run<1, 3>(s0, s1, errs, sink, lds); run<2, 3>(s0, s1, errs, sink, lds); run<3, 3>(s0, s1, errs, sink, lds); run<4, 3>(s0, s1, errs, sink, lds);
    run<5, 3>(s0, s1, errs, sink, lds); run<6, 3>(s0, s1, errs, sink, lds); run<7, 3>(s0, s1, errs, sink, lds); run<8, 3>(s0, s1, errs, sink, lds);
    run<0, 3>(s0, s1, errs, sink, 135168);
    return 0;
}

// Round 4, DESIGN 6: what the pool_lrn_bwd hunt came down to.  Leaving ONE packed instruction of the kernel packed at a time
// (tools/isa_patch_build.py scalarize:keep=i) singled out its only two `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` (low result =
// src0.lo + src1.HI, high result = src0.hi + src1.LO); the kernels that provoke the wrong results all issue
// `v_pk_add_f32 ... neg_lo:[0,1] neg_hi:[0,1]` (the head / tail split of the split-bf16 convolutions); kernels without packed
// instructions (fp32 MFMA wgrad, MFMA-only spinners, copies) never do.  This probe puts the two forms on ONE SIMD: waves 0..3 of a
// 512-thread workgroup (victims) repeat the op_sel form and check every result, waves 4..7 (their SIMD partners) issue another
// VOP3P form back to back.  Wrong results are classified by what they equal.
//   hipcc --offload-arch=gfx950 -O3 -o pk_modifier_leak pk_modifier_leak.hip && ./pk_modifier_leak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// PARTNER: 0 idle (s_sleep loop), 1 v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1], 2 plain v_pk_add_f32, 3 scalar v_add_f32, 4 v_pk_mul_f32,
//          5 v_pk_add_f32 neg on src0, 6 the victim's own form (op_sel)
// VICTIM:  0 v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]; 1 v_pk_add_f32 (no modifiers); 2 v_pk_mul_f32 op_sel_hi:[0,1]; 3 v_pk_fma_f32 op_sel_hi:[1,1,0]
template <int VICTIM, int PARTNER>
__global__ __launch_bounds__(512) void k(unsigned* __restrict__ errs, float* __restrict__ sink, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 4) {                                              // partner waves
        f32x2 a = {1.f + lane, 2.f}, b = {0.5f, 0.25f}, c = {0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            if constexpr (PARTNER == 0) __builtin_amdgcn_s_sleep(8);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if constexpr (PARTNER == 1) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a) : "v"(b));
                if constexpr (PARTNER == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
                if constexpr (PARTNER == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(c.x) : "v"(b.x));
                if constexpr (PARTNER == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
                if constexpr (PARTNER == 5) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[1,0] neg_hi:[1,0]" : "+v"(a) : "v"(b));
                if constexpr (PARTNER == 6) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(a) : "v"(b));
            }
        }
        if (a.x + a.y + c.x == 123.456f) sink[threadIdx.x] = a.x;
        return;
    }
    unsigned bad[2] = {0, 0}, kind[4] = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        f32x2 a = {1.f + lane + (it & 7), 1000.f + lane}, b = {0.125f * (it & 15), 64.f + (it & 3)}, c = {3.f, 5.f};
        asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
        f32x2 d, want;
        if constexpr (VICTIM == 0) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); want = f32x2{a.x + b.y, a.y + b.x}; }
        if constexpr (VICTIM == 1) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); want = a + b; }
        if constexpr (VICTIM == 2) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); want = f32x2{a.x * b.x, a.x * b.y}; }
        if constexpr (VICTIM == 3) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b), "v"(c)); want = f32x2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.x)}; }
        const bool w0 = d.x != want.x, w1 = d.y != want.y;
        bad[0] += w0; bad[1] += w1;
        if (w0 || w1) {
            if (d.x == a.x + b.x && d.y == a.y + b.y) kind[0]++;            // plain (no operand selection)
            else if (d.x == a.x - b.y && d.y == a.y - b.x) kind[1]++;       // selection kept, src1 negated
            else if (d.x == a.x - b.x && d.y == a.y - b.y) kind[2]++;       // the partner's modifiers
            else kind[3]++;
        }
    }
    const int q = lane >> 4;
    if (bad[0]) atomicAdd(errs + q * 6 + 0, bad[0]);
    if (bad[1]) atomicAdd(errs + q * 6 + 1, bad[1]);
    for (int j = 0; j < 4; ++j) if (kind[j]) atomicAdd(errs + q * 6 + 2 + j, kind[j]);
}

template <int VICTIM, int PARTNER>
static void run(unsigned* errs, float* sink) {
    static const char* vn[] = {"v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_add_f32 (no modifiers)", "v_pk_mul_f32 op_sel_hi:[0,1]", "v_pk_fma_f32 op_sel_hi:[1,1,0]"};
    static const char* pn[] = {"idle", "v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1]", "v_pk_add_f32 (no modifiers)", "v_add_f32", "v_pk_mul_f32", "v_pk_add_f32 neg_lo:[1,0] neg_hi:[1,0]",
                               "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]"};
    CK(hipMemset(errs, 0, 24 * sizeof(unsigned)));
    hipLaunchKernelGGL((k<VICTIM, PARTNER>), dim3(1024), dim3(512), 0, 0, errs, sink, 20000);
    CK(hipDeviceSynchronize());
    unsigned h[24];
    CK(hipMemcpy(h, errs, sizeof(h), hipMemcpyDeviceToHost));
    unsigned tot = 0;
    for (int i = 0; i < 24; ++i) tot += h[i];
    printf("victim %-42s | SIMD partner %-40s | wrong (lo, hi | =plain, =negated, =partner's, other) by lane quarter:", vn[VICTIM], pn[PARTNER]);
    for (int q = 0; q < 4; ++q) printf("  q%d %u %u | %u %u %u %u", q, h[q * 6], h[q * 6 + 1], h[q * 6 + 2], h[q * 6 + 3], h[q * 6 + 4], h[q * 6 + 5]);
    printf("%s\n", tot ? "   <== WRONG" : "");
    fflush(stdout);
}

int main() {
    unsigned* errs; float* sink;
    CK(hipMalloc(&errs, 256)); CK(hipMalloc(&sink, 8192));
    run<0, 0>(errs, sink); run<0, 1>(errs, sink); run<0, 2>(errs, sink); run<0, 3>(errs, sink); run<0, 4>(errs, sink); run<0, 5>(errs, sink); run<0, 6>(errs, sink);
    run<1, 1>(errs, sink); run<2, 1>(errs, sink); run<3, 1>(errs, sink);
    run<2, 6>(errs, sink); run<3, 6>(errs, sink); run<1, 6>(errs, sink);
    return 0;
}

// Probe for the mechanism behind round 3's pool_lrn_bwd nondeterminism.  In the failing kernel hipcc's SLP vectoriser had
// paired two fp32 multiplies into `v_pk_mul_f32 v[98:99], v[88:89], v[90:91]`, fed by a transcendental (`v_sqrt_f32 v90`), and the
// register allocator had made the NEXT instruction `ds_read2_b64 v[88:91]`: an LDS read whose destination is the packed
// instruction's sources (write-after-read on the register file).  Lanes 48..63 of the packed result came out computed from the LDS
// data whenever a bf16-MFMA kernel ran on the chip.  This file replays that instruction sequence in isolation (fixed registers
// v100..v103 inside one asm block) and counts wrong results per lane quarter, with a perturber kernel on a second stream.
// Built as a shared library so that tools/pk_f32_war.py can also run the library's real kernels as the perturber:
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libpkwar.so pk_f32_war.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: v_pk_mul_f32 d, v[100:101], v[102:103] ; ds_read2_b64 v[100:103]            (plain WAR, sources set long before)
// MODE 1: v_rsq_f32 v100 ; v_sqrt_f32 v102, v100 ; ... ; v_pk_mul_f32 ; ds_read2_b64   (the failing kernel's sequence)
// MODE 2: as 1 with two scalar v_mul_f32 instead of the packed one (what -fno-slp-vectorize gives)
// MODE 3: as 1 with the LDS read writing OTHER registers (no WAR): control for "the packed op itself is wrong"
// GAP = s_nop wait states between the VALU instruction and the LDS read.
template <int MODE, int GAP>
__global__ __launch_bounds__(256) void victim(unsigned* __restrict__ errs, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    // poison entries: what ds_read2_b64 puts into v100..v103 ({dp-like small float, small integer} pairs, as in the real kernel)
    lds[threadIdx.x * 4 + 0] = 1e-6f * (threadIdx.x + 1);
    lds[threadIdx.x * 4 + 1] = __uint_as_float((unsigned)(threadIdx.x & 7));
    lds[threadIdx.x * 4 + 2] = -2e-6f * (threadIdx.x + 1);
    lds[threadIdx.x * 4 + 3] = __uint_as_float(255u);
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(lds + threadIdx.x * 4);
    unsigned bad_lo = 0, bad_hi = 0, load_bad = 0;
    for (int it = 0; it < iters; ++it) {
        float sc = 1.0f + 0.001f * (float)((it * 7 + lane) & 1023);            // the LRN scale: rq = sc^-1/2, sqrt(rq) = sc^-1/4
        float k = 3e-5f, x = 10.f + (float)(it & 31);
        asm volatile("" : "+v"(sc), "+v"(k), "+v"(x));
        float d0, d1, o0, o1, o2, o3, rq, sq;
        if constexpr (MODE == 0) {
            rq = __builtin_amdgcn_rsqf(sc);
            sq = __builtin_amdgcn_sqrtf(rq);
            asm volatile("v_mov_b32 v100, %6\n\tv_mov_b32 v101, %7\n\tv_mov_b32 v102, %8\n\tv_mov_b32 v103, %9\n\ts_nop 7\n\t"
                         "v_pk_mul_f32 v[104:105], v[100:101], v[102:103]\n\ts_nop %11\n\tds_read2_b64 v[100:103], %10 offset1:1\n\ts_waitcnt lgkmcnt(0)\n\t"
                         "v_mov_b32 %0, v104\n\tv_mov_b32 %1, v105\n\tv_mov_b32 %2, v100\n\tv_mov_b32 %3, v101\n\tv_mov_b32 %4, v102\n\tv_mov_b32 %5, v103"
                         : "=&v"(d0), "=&v"(d1), "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
                         : "v"(rq), "v"(k), "v"(sq), "v"(x), "v"(addr), "n"(GAP)
                         : "memory", "v100", "v101", "v102", "v103", "v104", "v105");
        } else {
            // v100 = rsq(sc); v101 = k; v103 = x; v102 = sqrt(v100); {v104, v105} = {v100 v102, v101 v103}; then the LDS read
            asm volatile("v_mov_b32 v101, %7\n\tv_mov_b32 v103, %9\n\t"
                         "v_rsq_f32 v100, %6\n\t"
                         "v_add_f32 v106, %9, %9\n\tv_add_f32 v107, %7, %7\n\tv_add_f32 v106, v106, v107\n\t"
                         "v_sqrt_f32 v102, v100\n\t"
                         "v_mul_f32 v108, v100, v100\n\tv_add_f32 v107, v106, v107\n\tv_fmac_f32 v106, v107, v107\n\t"
                         ".if %12 == 2\n\tv_mul_f32 v104, v100, v102\n\tv_mul_f32 v105, v101, v103\n\t.else\n\tv_pk_mul_f32 v[104:105], v[100:101], v[102:103]\n\t.endif\n\t"
                         "s_nop %11\n\t"
                         ".if %12 == 3\n\tds_read2_b64 v[110:113], %10 offset1:1\n\t.else\n\tds_read2_b64 v[100:103], %10 offset1:1\n\t.endif\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         ".if %12 == 3\n\tv_mov_b32 v100, v110\n\tv_mov_b32 v101, v111\n\tv_mov_b32 v102, v112\n\tv_mov_b32 v103, v113\n\t.endif\n\t"
                         "v_mov_b32 %0, v104\n\tv_mov_b32 %1, v105\n\tv_mov_b32 %2, v100\n\tv_mov_b32 %3, v101\n\tv_mov_b32 %4, v102\n\tv_mov_b32 %5, v103"
                         : "=&v"(d0), "=&v"(d1), "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
                         : "v"(sc), "v"(k), "v"(sc), "v"(x), "v"(addr), "n"(GAP), "n"(MODE)
                         : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v110", "v111", "v112", "v113");
            rq = __builtin_amdgcn_rsqf(sc);
            sq = __builtin_amdgcn_sqrtf(rq);
        }
        bad_lo += d0 != rq * sq;
        bad_hi += d1 != k * x;
        load_bad += (o0 != lds[threadIdx.x * 4] || __float_as_uint(o1) != (unsigned)(threadIdx.x & 7) || o2 != lds[threadIdx.x * 4 + 2] ||
                     __float_as_uint(o3) != 255u);
    }
    const int q = lane >> 4;                                                   // per lane quarter: [lo half, hi half, load]
    if (bad_lo) atomicAdd(errs + q * 3 + 0, bad_lo);
    if (bad_hi) atomicAdd(errs + q * 3 + 1, bad_hi);
    if (load_bad) atomicAdd(errs + q * 3 + 2, load_bad);
}

template <int BF16>
__global__ __launch_bounds__(256) void mfma_spinner(float* out, int iters) {
    extern __shared__ float pad[];
    f32x16 acc = {};
    const float v = 1.0f + (threadIdx.x & 7) * 0.125f;
    if (BF16) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(v + i); b[i] = (__bf16)(v - i); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v, v + 1.f, acc, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 123.456f) out[threadIdx.x] = s + pad[0];
}

extern "C" int pkwar_victim(int mode, int gap, int grid, int lds_bytes, int iters, unsigned* errs, void* stream) {
#define V(M, G) if (mode == M && gap == G) { hipFuncSetAttribute((const void*)victim<M, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((victim<M, G>), dim3(grid), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, errs, iters); return (int)hipGetLastError(); }
    V(0, 0) V(1, 0) V(2, 0) V(3, 0) V(1, 1) V(1, 3) V(1, 7) V(0, 1)
#undef V
    return -1;
}
extern "C" int pkwar_spin(int bf16, int grid, int lds_bytes, int iters, float* sink, void* stream) {
    hipFuncSetAttribute((const void*)mfma_spinner<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)mfma_spinner<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (bf16) hipLaunchKernelGGL(mfma_spinner<1>, dim3(grid), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, sink, iters);
    else hipLaunchKernelGGL(mfma_spinner<0>, dim3(grid), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, sink, iters);
    return (int)hipGetLastError();
}

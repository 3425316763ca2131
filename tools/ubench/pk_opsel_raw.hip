// Round 4, DESIGN 6: the smallest form of the defect.  `v_pk_add_f32 d, a, b op_sel:[0,1] ...` (low result = a.lo + b.HI) executed
// shortly after a VALU instruction wrote b's high register, in lanes 48..63, while another kernel's waves (LDS reads + v_cvt_pk_bf16 /
// v_pk_add head-tail split + bf16 MFMAs: the split-bf16 weight gradient's loop) share the CU.  Three copies of the instruction in a
// row, GAP wait states after the write of b: which copy is wrong, and how many wait states cure it?
//   hipcc --offload-arch=gfx950 -O3 -o pk_opsel_raw pk_opsel_raw.hip && ./pk_opsel_raw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// WRITER: which instruction produces b.y (the HIGH register of src1) right before: 0 v_cvt_f32_ubyte0 (from an SGPR), 1 v_mov_b32,
// 2 v_add_f32, 3 v_rcp_f32 (quarter rate), 4 written long before (32 wait states).   GAP: s_nop wait states between writer and 1st copy.
template <int WRITER, int GAP>
__global__ __launch_bounds__(256) void victim(unsigned* __restrict__ errs, int iters) {
    extern __shared__ float pad[];
    const int lane = threadIdx.x & 63;
    unsigned bad[3] = {0, 0, 0}, zero_seen = 0;
    for (int it = 0; it < iters; ++it) {
        f32x2 a = {1.f + lane + (it & 7), 1000.f + lane};
        float bx = 0.125f * (float)((it & 15) + 1);
        const int by_i = 64 + (it & 3);
        float seed = (float)by_i;
        if (WRITER == 3) seed = 1.0f / seed;
        asm volatile("" : "+v"(a), "+v"(bx), "+v"(seed));
        f32x2 d1, d2, d3;
        float by_out;
        // b = v[100:101]: v100 = bx (written first), v101 = by written by the instruction under test
        if constexpr (WRITER == 0)
            asm volatile("v_mov_b32 v100, %4\n\ts_nop 7\n\tv_cvt_f32_ubyte0_e32 v101, %5\n\t.if %7 > 0\n\ts_nop %7 - 1\n\t.endif\n\t"
                         "v_pk_add_f32 %0, %6, v[100:101] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_pk_add_f32 %1, %6, v[100:101] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                         "v_pk_add_f32 %2, %6, v[100:101] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32 %3, v101"
                         : "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(by_out) : "v"(bx), "s"(by_i), "v"(a), "n"(GAP) : "v100", "v101");
        else if constexpr (WRITER == 4)
            asm volatile("v_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\ts_nop 15\n\ts_nop 15\n\t"
                         "v_pk_add_f32 %0, %6, v[100:101] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_pk_add_f32 %1, %6, v[100:101] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                         "v_pk_add_f32 %2, %6, v[100:101] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32 %3, v101"
                         : "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(by_out) : "v"(bx), "v"(seed), "v"(a), "n"(GAP) : "v100", "v101");
        else
            asm volatile("v_mov_b32 v100, %4\n\ts_nop 7\n\t"
                         ".if %8 == 1\n\tv_mov_b32 v101, %5\n\t.endif\n\t.if %8 == 2\n\tv_add_f32 v101, %5, %5\n\t.endif\n\t.if %8 == 3\n\tv_rcp_f32 v101, %5\n\t.endif\n\t"
                         ".if %7 > 0\n\ts_nop %7 - 1\n\t.endif\n\t"
                         "v_pk_add_f32 %0, %6, v[100:101] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_pk_add_f32 %1, %6, v[100:101] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                         "v_pk_add_f32 %2, %6, v[100:101] op_sel:[0,1] op_sel_hi:[1,0]\n\tv_mov_b32 %3, v101"
                         : "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(by_out) : "v"(bx), "v"(seed), "v"(a), "n"(GAP), "n"(WRITER) : "v100", "v101");
        float w0, w1;                                                          // the right answer, from scalar instructions on the value b.y ended up with
        asm volatile("s_nop 7\n\tv_add_f32 %0, %2, %3\n\tv_add_f32 %1, %4, %5" : "=&v"(w0), "=&v"(w1) : "v"(a.x), "v"(by_out), "v"(a.y), "v"(bx));
        bad[0] += (d1.x != w0) || (d1.y != w1);
        bad[1] += (d2.x != w0) || (d2.y != w1);
        bad[2] += (d3.x != w0) || (d3.y != w1);
        zero_seen += (d1.x != w0 && d1.x == a.x);                               // the low half came out as a.lo + 0
    }
    const int q = lane >> 4;
    for (int j = 0; j < 3; ++j) if (bad[j]) atomicAdd(errs + q * 4 + j, bad[j]);
    if (zero_seen) atomicAdd(errs + q * 4 + 3, zero_seen);
    if (iters < 0) pad[0] = 1.f;
}

__global__ __launch_bounds__(256) void partner(float* __restrict__ sink, int iters) {   // the split-bf16 weight gradient's loop shape
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.001f * (float)i + 0.37f;
    __syncthreads();
    f32x16 acc = {};
    i32x4 hd = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80}, tl = hd;
    const float* p = lds + (threadIdx.x & 63) * 8;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(p + 2 * q + 512 * (it & 3));
            const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
            const f32x2 r = {v[0] - __uint_as_float(h << 16), v[1] - __uint_as_float(h & 0xffff0000u)};
            hd[q] = (int)h;
            tl[q] = (int)__builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hd), __builtin_bit_cast(bf16x8, hd), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hd), __builtin_bit_cast(bf16x8, tl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, tl), __builtin_bit_cast(bf16x8, hd), acc, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s + (float)hd[0] + (float)tl[3] == 123.456f) sink[threadIdx.x] = s;
}

template <int WRITER, int GAP>
static void run(hipStream_t s0, hipStream_t s1, unsigned* errs, float* sink, bool with_partner) {
    static const char* wn[] = {"v_cvt_f32_ubyte0 (SGPR source)", "v_mov_b32", "v_add_f32", "v_rcp_f32 (quarter rate)", "written 32+ wait states earlier"};
    CK(hipMemsetAsync(errs, 0, 16 * sizeof(unsigned), s0));
    CK(hipStreamSynchronize(s0));
    if (with_partner) hipLaunchKernelGGL(partner, dim3(256), dim3(256), 16384, s1, sink, 60000);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((victim<WRITER, GAP>), dim3(2048), dim3(256), 12544, s0, errs, 20000);
    CK(hipDeviceSynchronize());
    unsigned h[16];
    CK(hipMemcpy(h, errs, sizeof(h), hipMemcpyDeviceToHost));
    unsigned tot = 0;
    for (int i = 0; i < 16; ++i) tot += h[i];
    printf("b.hi written by %-34s then %d wait state(s), partner kernel %-3s | wrong results of copy 1 / 2 / 3 (and: low half = a.lo + 0) by lane quarter:", wn[WRITER], GAP, with_partner ? "yes" : "no");
    for (int q = 0; q < 4; ++q) printf("  q%d %u %u %u (%u)", q, h[q * 4], h[q * 4 + 1], h[q * 4 + 2], h[q * 4 + 3]);
    printf("%s\n", tot ? "   <== WRONG" : "");
    fflush(stdout);
}

int main() {
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    unsigned* errs; float* sink;
    CK(hipMalloc(&errs, 256)); CK(hipMalloc(&sink, 8192));
    run<0, 0>(s0, s1, errs, sink, false); run<1, 0>(s0, s1, errs, sink, false); run<3, 0>(s0, s1, errs, sink, false);
    run<0, 0>(s0, s1, errs, sink, true); run<0, 1>(s0, s1, errs, sink, true); run<0, 2>(s0, s1, errs, sink, true); run<0, 4>(s0, s1, errs, sink, true); run<0, 8>(s0, s1, errs, sink, true);
    run<1, 0>(s0, s1, errs, sink, true); run<1, 1>(s0, s1, errs, sink, true); run<1, 2>(s0, s1, errs, sink, true); run<1, 4>(s0, s1, errs, sink, true);
    run<2, 0>(s0, s1, errs, sink, true); run<2, 2>(s0, s1, errs, sink, true);
    run<3, 0>(s0, s1, errs, sink, true); run<3, 2>(s0, s1, errs, sink, true); run<3, 8>(s0, s1, errs, sink, true);
    run<4, 0>(s0, s1, errs, sink, true);
    return 0;
}

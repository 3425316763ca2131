// Microbenchmark: what limits v_mfma_f32_32x32x2_f32 issue on gfx950?
//   mode 0: pure MFMA, 4 independent accumulators
//   mode 1: + NV independent v_add per MFMA (same wave)
//   mode 2: + 1 ds_read_b32 per MFMA, waited (lgkmcnt(0)) before each group of 4 MFMAs (my GEMM loop)
//   mode 3: ds_reads software-pipelined one group ahead
// Grid = 256 CUs * blocks_per_cu, 256 threads (4 waves = 1 per SIMD per block).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o mfma_issue mfma_issue.hip && ./mfma_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed * i;
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int q = 0; q < 16; ++q) acc[a][q] = 0.f;
    float x = seed + threadIdx.x, y = seed * 2.f;
    float va[8];
    for (int i = 0; i < 8; ++i) va[i] = seed * i;
    int addr = threadIdx.x & 63;
    float r0 = lds[addr], r1 = lds[addr + 64], r2 = lds[addr + 128], r3 = lds[addr + 192];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            float a0 = x, a1 = y, b0 = x, b1 = y;
            if (MODE == 2) {
                const int o = (g * 256 + it * 64) & 4095;
                a0 = lds[addr + o]; a1 = lds[addr + o + 64]; b0 = lds[addr + o + 2048]; b1 = lds[addr + o + 2048 + 64];
            }
            if (MODE == 3) {
                a0 = r0; a1 = r1; b0 = r2; b1 = r3;
                const int o = ((g + 1) * 256 + it * 64) & 4095;
                r0 = lds[addr + o]; r1 = lds[addr + o + 64]; r2 = lds[addr + o + 2048]; r3 = lds[addr + o + 2048 + 64];
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            if (MODE == 1) for (int i = 0; i < NV; ++i) va[i % 8] += y;
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            if (MODE == 1) for (int i = 0; i < NV; ++i) va[(i + 2) % 8] += y;
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            if (MODE == 1) for (int i = 0; i < NV; ++i) va[(i + 4) % 8] += y;
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
            if (MODE == 1) for (int i = 0; i < NV; ++i) va[(i + 6) % 8] += y;
        }
    }
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int q = 0; q < 16; ++q) s += acc[a][q];
    for (int i = 0; i < 8; ++i) s += va[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + r0 + r1 + r2 + r3;
}

template <int MODE, int NV>
void run(const char* name, int bpc, int extra_lds) {
    float* out;
    const int blocks = 256 * bpc, iters = 2000;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, NV>), dim3(blocks), dim3(256), extra_lds, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) {
            double flop = (double)blocks * 4 * iters * 32 * 4096.0;
            printf("%-34s blocks/CU %d : %8.3f ms  %7.1f TFLOP/s\n", name, bpc, ms, flop / ms / 1e9);
        }
    }
    hipFree(out);
}

int main() {
    for (int bpc = 1; bpc <= 3; ++bpc) {
        // extra LDS forces exactly bpc blocks per CU: 160KB / bpc - 32KB static - margin
        int extra = bpc == 1 ? 100000 : (bpc == 2 ? 40000 : 16000);
        run<0, 0>("pure mfma", bpc, extra);
        run<1, 1>("mfma + 1 v_add each", bpc, extra);
        run<1, 2>("mfma + 2 v_add each", bpc, extra);
        run<1, 4>("mfma + 4 v_add each", bpc, extra);
        run<1, 8>("mfma + 8 v_add each", bpc, extra);
        run<2, 0>("mfma + ds_read, wait per group", bpc, extra);
        run<3, 0>("mfma + ds_read pipelined", bpc, extra);
    }
    return 0;
}

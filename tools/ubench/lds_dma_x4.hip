// Probe: semantics of the 16-byte LDS-DMA of gfx950, buffer_load_dwordx4 ... lds (conv_ring_kernel / conv_ring4_kernel rely on
// them): lane l lands at M0 + 16 l; a dword-aligned but 16-byte-misaligned source is fine; the scalar offset IS part of the
// buffer range check (records end -> zeros).  hipcc --offload-arch=gfx950 -O3 -o lds_dma_x4 lds_dma_x4.hip && ./lds_dma_x4
// Output on MI355X: 'shift 0..3: 0 mismatches', 'soffset beyond num_records: values 0 0 0 ... 0', 'straddling: ... [127] 383 [128] 0'.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, float* c, int shift, int nrec = 1 << 20, int soff = 0) {
    extern __shared__ float l[];
    i32x4 rs = {(int)(uintptr_t)a, (int)((uintptr_t)a >> 32) & 0xffff, nrec, 0x00020000};
    uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)l;
    uint32_t voff = threadIdx.x * 16 + shift * 4;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(rs), "s"(soff) : "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = threadIdx.x; j < 256; j += 64) c[j] = l[j];
}
int main() {
    float *a, *c;
    hipMalloc(&a, 4096 * 4); hipMalloc(&c, 256 * 4);
    std::vector<float> h(4096); for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    hipMemcpy(a, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int shift = 0; shift < 4; ++shift) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, a, c, shift);
        std::vector<float> o(256);
        hipMemcpy(o.data(), c, 256 * 4, hipMemcpyDeviceToHost);
        int bad = 0; for (int j = 0; j < 256; ++j) bad += o[j] != (float)(j + shift);
        printf("shift %d: %d mismatches (first values %g %g %g %g %g)\n", shift, bad, o[0], o[1], o[2], o[3], o[4]);
    }
    // is the scalar offset part of the range check?  records end at 1024 B; voffset in range, soffset 2048 B
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, a, c, 0, 1024, 2048);
    std::vector<float> o(256);
    hipMemcpy(o.data(), c, 256 * 4, hipMemcpyDeviceToHost);
    printf("soffset beyond num_records: values %g %g %g ... %g (zeros = range-checked, 512.. = not)\n", o[0], o[1], o[2], o[255]);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, a, c, 0, 1024 + 512, 1024);
    hipMemcpy(o.data(), c, 256 * 4, hipMemcpyDeviceToHost);
    printf("straddling: values %g %g ... [127] %g [128] %g [255] %g\n", o[0], o[1], o[127], o[128], o[255]);
    return 0;
}

// Latency of one hand-off between two workgroups on MI355X, by placement and by cache scope of the accesses -- the number that bounds a
// step of the LSTM cluster kernels (lstm_cluster.hip: publish h_t, gather h_t; one hop per time step).
//   placement: workgroups 0 and 8 of the grid (same XCD: ids are dealt round-robin over the 8 XCDs) or 0 and 1 (neighbouring XCDs)
//   scope:     agent-scope atomics (sc1: what lstm_cluster.hip uses -- placement independent)  |  sc0 loads + sc0 stores (inline asm:
//              coherent in the XCD's L2 only)  |  plain loads (may hit in the CU's L1: expected to time out)
// A round trip = A stores i, B sees i and stores i, A sees i.  Spins are bounded; a time-out is reported, never a hang.
//   hipcc --offload-arch=gfx950 -O3 -o xcd_pingpong xcd_pingpong.hip && ./xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

template <int SCOPE> __device__ __forceinline__ u64 ld(const u64* p) {
    if constexpr (SCOPE == 0) return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if constexpr (SCOPE == 1) {
        u64 v;
        asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    } else {
        u64 v;
        asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
}
template <int SCOPE> __device__ __forceinline__ void st(u64* p, u64 v) {
    if constexpr (SCOPE == 0) __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}

// flags[0]: A -> B, flags[32]: B -> A (different cache lines); out[0] = cycles (wall clock, 100 MHz), out[1] = timed-out flag, out[2..3] = XCC ids
template <int SCOPE>
__global__ void pingpong(u64* flags, u64* out, int partner, int iters, unsigned spin_limit) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0 || (b != 0 && b != partner)) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[2 + (b != 0)] = xcc & 0xf;
    u64* mine = flags + (b == 0 ? 0 : 32);
    const u64* theirs = flags + (b == 0 ? 32 : 0);
    const u64 t0 = wall_clock64();
    for (int i = 1; i <= iters; ++i) {
        if (b == 0) st<SCOPE>(mine, (u64)i);
        unsigned spins = 0;
        while (ld<SCOPE>(theirs) != (u64)i) {
            if (++spins > spin_limit) { out[1] = 1; return; }
        }
        if (b != 0) st<SCOPE>(mine, (u64)i);
    }
    if (b == 0) out[0] = wall_clock64() - t0;
}

template <int SCOPE>
static void run(const char* name, u64* flags, u64* out, int partner) {
    CK(hipMemset(flags, 0, 64 * sizeof(u64)));
    CK(hipMemset(out, 0, 8 * sizeof(u64)));
    const int iters = 2000;
    hipLaunchKernelGGL(pingpong<SCOPE>, dim3(16), dim3(64), 0, 0, flags, out, partner, iters, 1u << 16);
    CK(hipDeviceSynchronize());
    u64 h[8];
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    if (h[1]) printf("%-44s workgroups 0 / %d (XCC %llu / %llu): TIMED OUT (the partner's stores were never seen)\n", name, partner, h[2], h[3]);
    else printf("%-44s workgroups 0 / %d (XCC %llu / %llu): %.0f ns per round trip = %.0f ns per hop\n", name, partner, h[2], h[3],
                h[0] * 10.0 / iters, h[0] * 5.0 / iters);
    fflush(stdout);
}

int main() {
    u64 *flags, *out;
    CK(hipMalloc(&flags, 64 * sizeof(u64)));
    CK(hipMalloc(&out, 8 * sizeof(u64)));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("agent-scope atomics (sc1)", flags, out, 8);
        run<0>("agent-scope atomics (sc1)", flags, out, 1);
        run<1>("sc0 loads, sc0 stores", flags, out, 8);
        run<1>("sc0 loads, sc0 stores", flags, out, 1);
        run<2>("plain loads, sc0 stores", flags, out, 8);
    }
    return 0;
}

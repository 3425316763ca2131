// Shared helpers for libvltf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/vltf.h"

// ---- error convention: int return + thread-local message (include/vltf.h) ----------------------
void vl_set_error(const char* fmt, ...);

#define VL_CHECK(cond, ...)                \
    do {                                   \
        if (!(cond)) {                     \
            vl_set_error(__VA_ARGS__);     \
            return 1;                      \
        }                                  \
    } while (0)

#define VL_HIP(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            vl_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                                            \
        }                                                                                        \
    } while (0)

#define VL_LAUNCH_CHECK() VL_HIP(hipGetLastError())

// ---- experiment / A-B switches -----------------------------------------------------------------
// Every such switch is an environment variable read through this helper into a `static const` (once per process).  The product build
// answers "unset" without looking at the environment; `VL_EXPERIMENTS=1 bash build.sh` builds libvltf_hip_exp.so with them compiled in
// (tools/: VLTF_HIP_LIB=.../libvltf_hip_exp.so).  Nothing in tests/, bench.py or the package depends on one.
#include <stdlib.h>
static inline const char* vl_exp_env(const char* name) {
#ifdef VL_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// ---- division by a runtime constant: q = (mulhi(n, m) + n) >> s, valid for n < 2^31 -------------
struct FastDiv {
    uint32_t d, m, s;
};

static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    f.s = s;
    f.m = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
    return f;
}

__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) { return (__umulhi(n, f.m) + n) >> f.s; }

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// compute units of the current device (256 on MI355X), cached per translation unit: launch plans that size a grid to "one round of
// workgroups" take it from here, not from a literal
static inline int vl_device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    return cus;
}

// one 64-lane wavefront reductions (gfx950: wave = 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- LSTM recurrence: per-clip form (pointwise.hip), the fallback / A-B reference of the cluster form (lstm_cluster.hip) ------
int vl_lstm_perclip_fwd(const float* gx, const float* kh, const float* h0, const float* c0, float* act, float* cseq, float* hseq,
                        float* hprev, int batch, int T, int H, float forget_bias, hipStream_t stream);
int vl_lstm_perclip_bwd(const float* dout, const float* kh_t, const float* act, const float* cseq, const float* c0, float* dz,
                        float* dh0, float* dc0, int batch, int T, int H, hipStream_t stream);

// The convolution descriptor behind include/vltf.h's opaque vl_conv_desc (created by vl_conv_create in mfma_gemm.hip; read by the
// fp32 / split-product kernels there and by the packed-bf16 kernels of conv_c8.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct vl_conv_desc {
    int cin, h, w, cout, kh, kw, stride, groups;
    int oh, ow, pt, pl, pb, pr;   // SAME padding before / after
    int cig, cog;
    int K;    // kh*kw*cig
    int Kd;   // kh*kw*cog (dgrad reduction length)
    int x_halo, y_halo, dy_halo, dx_halo;
    int x_phase;      // 1, or = stride: x is stored column-phase-split (vl_conv_set_x_phase_split)
    int2* ktab2_fwd;  // natural order, checked mode: {byte offset, kh << 16 | kw} (device), halo-aware
    int* ktab_fwd;    // natural order, padded mode: byte offset
    // forward / dgrad run their reduction in the permuted order (build_ktabs): gather tables in that order + first weight
    // row of every 16-row reduction tile.  wgrad keeps the natural-order tables above (its rows ARE the output rows).
    int2* ptab2_fwd;
    int2* ptab2_bwd;
    int* ptab_fwd;
    int* ptab_bwd;
    int* rowtab_fwd;
    int* rowtab_bwd;
    int fwd_padded, bwd_padded;
    uint32_t* wsplit_fwd;   // conv_wsplit_kernel's image of the forward / dgrad weights (split-product arithmetic only)
    uint32_t* wsplit_bwd;
    int* c8_toff_fwd;       // conv_c8.hip: byte offset of every reduction tap (channel block, ky, kx) in the c8 layout, forward / dgrad
    int* c8_toff_bwd;
};

void conv_c8_free_tables(vl_conv_desc* d);   // conv_c8.hip; at destroy
int conv_c8_build_tables(vl_conv_desc* d);    // conv_c8.hip; from rebuild_tables (create, set_halo, phase split): 0, or 2 on allocation failure

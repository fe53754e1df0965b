// Launch arguments and vector types shared by the bf16 convolution translation units (nd_conv_bf16.hip,
// nd_gemm_bf16_quad.hip).
#pragma once
#include "nd_conv_common.h"

namespace nd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

struct ConvArgsH {
    const __bf16* x0;
    const __bf16* x1;
    const __bf16* w;      // packed fragments
    const float* bias;
    const float* rowbias;
    const __bf16* res;
    void* out;            // bf16, or fp32 when out_f32
    int C0, C1, ldx0, ldx1;
    int NI, H, W;         // output (= virtual input) size
    int Hs, Ws;           // stored input size (H >> up)
    int up, res_up;
    int N, ldo, ldr, ld_rowbias;
    int NT32, NC64;
    int thl, twl, nibl;
    int tiles_x, tiles_y, mt, nt, ngroup;
    int silu_out, out_f32;
    // GroupNorm apply fused into the loader: in' = act(in * gnA[img][c] + gnB[img][c]) for real (non-padding) pixels;
    // one image per block (gn_hw > 0: flat pixel list, image = pixel / gn_hw)
    const float* gnA;
    const float* gnB;
    int ld_gn, gn_silu, gn_hw;
    // partial GroupNorm statistics of the OUTPUT (STATS instantiations): rows [img][cs_rows][sum | sum of squares][N],
    // one row per (pixel tile of the image, wave row), every row written by every launch
    float* chstats;
    int cs_rows;
    // split over K (gridDim.y = ksplit > 1): block row s computes channels [s * kchunks * 64, ...) only and writes its raw
    // fp32 accumulators to out + s * ws_stride floats (out is then the workspace; bias / residual / activation are applied
    // by splitk_reduce_kernel, which adds the ksplit partials in order)
    int ksplit, kchunks;
    long ws_stride;
    // conv_bf16_kernel with TN == 2: the wave's output tile goes through a wave-private LDS region (the halo buffers are
    // dead by then) and leaves as 16-byte stores, 8 lanes = one pixel's 128 bytes = one cache line (host-checked: bf16
    // output, N a multiple of the block's channels, 16-byte aligned rows, LDS sized for waves x TM x 4 KiB)
    int coal;
};

__device__ __forceinline__ bf16x8 as_bf16x8(const f32x4& v) {
    union { f32x4 f; bf16x8 h; } u;
    u.f = v;
    return u.h;
}

struct VariantH;
// gemm_bf16q_kernel (nd_gemm_bf16_quad.hip), launched by nd_conv_bf16_nhwc's variant 21
int launch_gemm_bf16q(const ConvArgsH& a, int grid, hipStream_t s);

}  // namespace nd

// Shared by the convolution translation units: launch arguments, block -> tile order, tile planning helpers.
#pragma once
#include "nd_common.h"
#include "nd_weight_stream.h"
#include <stdlib.h>

#ifndef ND_SETPRIO
#define ND_SETPRIO 1      // +0.3..1.8 % on the 1x1 form, measured warm
#endif
#if ND_SETPRIO
#define ND_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define ND_PRIO(x)
#endif

namespace nd {

struct ConvArgs {
    const float* x0;
    const float* x1;
    const float* w;      // packed fragments
    const float* bias;
    const float* rowbias;
    const float* res;
    float* out;
    int C0, C1, ldx0, ldx1;
    int NI, H, W;      // output (= virtual input) size
    int Hs, Ws;        // stored input size (H >> up)
    int up;            // input read through nearest-2x upsampling
    int res_up;        // residual read through nearest-2x upsampling
    int N, ldo, ldr, ld_rowbias;
    int NT32;          // ceil(N / 32)
    int NC32;          // ceil(Cin / 32)
    int thl, twl, nibl;   // log2 of tile height / width / images per block
    int tiles_x, tiles_y, mt, nt;
    int ngroup;        // n tiles per group of the block -> tile order (see tile_of)
    int nhi;           // Winograd: halo items per thread actually needed for this tiling
    const float* zero; // 16 bytes of zeros in device memory (LDS-DMA source for padding)
    float* chstats;    // optional partial output statistics [NI][mbi][4][2][N] (see conv_wino16_kernel's epilogue)
    int mbi;           // m blocks per image (1 when a block holds whole images)
    int vec_ok;        // Winograd epilogue: 16-byte stores / loads are legal (strides and pointers aligned)
    int silu_out;
    // GroupNorm apply fused into the loader: in' = act(in * gnA[img][c] + gnB[img][c]) for real (non-padding) pixels
    const float* gnA;
    const float* gnB;
    int ld_gn, gn_silu, gn_hw;    // gn_hw > 0: flat pixel list, image = pixel / gn_hw
    // split over K (conv_mfma_kernel only; nd_conv_splitk_nhwc): block row s = blockIdx.y runs the channel-chunk range of
    // split s and leaves raw accumulators in out + s * ws_stride; 0 / 1 = one pass
    int ksplit, kchunks;
    long ws_stride;
};

// The arguments of split s, derived from the whole problem's: a convolution over the 32-channel chunks
// [s * kchunks, (s + 1) * kchunks) of the (concatenated) input.  The packed weights are chunk-major
// ([c32][n tile][tap][k-step][lane][4]), so a split's weights are a contiguous range too.
__device__ __forceinline__ void split_k_args_f32(ConvArgs& p, int s, int taps) {
    const int span = p.kchunks * 32;
    const int c_begin = s * span;
    int nc = p.NC32 - s * p.kchunks;
    nc = nc < p.kchunks ? nc : p.kchunks;
    p.w += (size_t)s * p.kchunks * ((size_t)p.NT32 * taps * 4 * 256);
    if (c_begin < p.C0) {
        p.x0 += c_begin;
        const int left0 = p.C0 - c_begin;
        if (left0 >= span) {
            p.C0 = span;
            p.C1 = 0;
        } else {                                  // the concatenation seam lies inside this split's range
            p.C0 = left0;
            p.C1 = (p.C1 < span - left0) ? p.C1 : span - left0;
        }
    } else {
        const int off1 = c_begin - p.C0;
        p.x0 = p.x1 + off1;
        p.ldx0 = p.ldx1;
        p.C0 = (p.C1 - off1 < span) ? p.C1 - off1 : span;
        p.C1 = 0;
    }
    p.NC32 = nc;
    p.out += (size_t)s * p.ws_stride;
}

// Block -> tile order.  The n tiles are taken in groups of `ngroup`; inside a group the walk is m-major with n fastest,
// so the blocks resident at one time on an XCD (consecutive ids) cover a few m tiles x ngroup n tiles: each input tile is
// pulled from HBM once per GROUP and shared through L2 by the ngroup blocks that use it, each weight slab once per m
// row.  ngroup = 1 is the n-major order (weights stay put, inputs re-read nt times), ngroup = nt the m-major one.
__device__ inline void tile_of(int idp, int mt, int nt, int ngroup, int& mblk, int& nblk) {
    const int per_group = ngroup * mt;
    const int g = idp / per_group;
    const int rem = idp - g * per_group;
    const int left = nt - g * ngroup;
    const int gn = left < ngroup ? left : ngroup;
    mblk = rem / gn;
    nblk = g * ngroup + (rem - mblk * gn);
}

static inline int env_ngroup() {
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("ND_NGROUP");
        v = e ? atoi(e) : -1;
    }
    return v;
}

// n tiles per group.  Measured with FETCH_SIZE on B=64 layers (tools/ngroup_fetch.sh): all n tiles when the whole
// weight tensor sits comfortably in an XCD's 4 MiB L2 (input then crosses HBM once: 333 MB instead of 1245 MB per
// launch for 64x64x192->192), otherwise 4 (786 vs 1327 MB for 32x32x384->384; m-major thrashes the weights: 1883 MB).
// Layers whose WEIGHTS outweigh their input (8x8 / 16x16 maps with 1-2 thousand input channels: 38 MB of bf16 weights
// against 8 MB of pixels) take the n-major order instead: the m tiles of one n block run side by side on an XCD and
// share its weight slab through L2 -- m-major made every XCD stream the whole tensor (profiles/r02_pmc_shapes.json:
// 6.97x the algorithmic bytes on those launches), n-major re-reads only the small input once per n block.
// Run time is within 1 % across orders where the layer is MFMA-bound -- the point there is not to burn HBM bandwidth
// and power on re-reads; the weight-dominated layers are not MFMA-bound.
static inline int pick_ngroup(int nt, size_t bytes_per_ntile, size_t input_bytes = 0) {
    int g = env_ngroup();
    if (g <= 0) {
        const size_t wbytes = (size_t)nt * bytes_per_ntile;
        if (wbytes <= ((size_t)3 << 20)) g = nt;
        else if (input_bytes > 0 && wbytes > 2 * input_bytes) g = 1;
        else {
            // as many weight slabs as an XCD's 4 MiB L2 holds next to the streaming input, between 2 and 4 (round 3,
            // tools/ngroup_fetch.sh on conv_wino4_kernel's 64-channel slabs: FETCH_SIZE 32x32x384->384 277 / 307 / 322 /
            // 485 MB x2 for groups of 2 / 3 / 4 / all; 16x16x576 175 / 181 / 214 / 428; 64x64x384 1190 / 1325 / 1393 / 2030)
            g = (int)(((size_t)4 << 20) / (bytes_per_ntile ? bytes_per_ntile : 1));
            g = g < 2 ? 2 : (g > 4 ? 4 : g);
        }
    }
    return g > nt ? nt : g;
}

__host__ __device__ inline int nc32_padded(int C) {
    const int c = (C + 31) / 32;
    return (c + 1) & ~1;      // even number of 32-channel chunks (the 1x1 kernel walks two per barrier)
}

// the split plan shared by nd_conv_splitk_nhwc, nd_conv3x3_winograd_splitk_nhwc and nd_conv_splitk_workspace_floats: whole
// LDS chunks per split (32 channels for 3x3, pairs of them for 1x1); returns the number of splits actually used (< 2:
// cannot split) and the chunks per split
static inline int splitk_plan_f32(int C, int ksize, int splits, int* kchunks) {
    const int nc32 = (C + 31) / 32, unit = ksize == 3 ? 1 : 2;
    int kc = (nc32 + splits - 1) / splits;
    kc = (kc + unit - 1) / unit * unit;
    *kchunks = kc;
    return (nc32 + kc - 1) / kc;
}
// second pass of a split launch (nd_conv_mfma.hip): out = sum_s ws[s] + bias + rowbias[img] + residual, SiLU last
int launch_splitk_reduce_f32(const float* ws, int S, long ws_stride, long M, int N, const float* bias, const float* rowbias,
                             int ld_rowbias, int hw, const float* res, int ldr, float* out, int ldo, int silu, hipStream_t s);
// the same + per-channel partial statistics of the output, one row per run of kSplitkStatsPixels pixels of an image (hw must be
// a multiple of it, N of 4): chstats [NI][hw / kSplitkStatsPixels][sum | sum of squares][N]
constexpr int kSplitkStatsPixels = 16;
int launch_splitk_reduce_stats_f32(const float* ws, int S, long ws_stride, int NI, int hw, int N, const float* bias, const float* rowbias,
                                   int ld_rowbias, const float* res, int ldr, float* out, int ldo, int silu, float* chstats, hipStream_t s);

struct TilePlan {
    int thl, twl, nibl, tiles_x, tiles_y, groups, hp;
    long padded;   // padded pixel count
};

// Eight independent sums over the 32 lanes that share lane >> 5, in a fixed order, valid in lanes 16..31 (48..63) of the
// group: xor 1, xor 2 (quad permutes), mirror within 8, mirror within 16, then lane 15 of the even row broadcast into
// the odd row (row_bcast:15).  DPP operands on the adds themselves; the eight chains are interleaved so that a register
// is read by a DPP operand eight instructions after it was written (the hardware wants two wait states; hipcc does not
// fold v_mov_dpp into a float add and pads every move with s_nop).
#define ND_DPP8(ctrl)                                        \
    "v_add_f32_dpp %0, %0, %0 " ctrl "\n"                    \
    "v_add_f32_dpp %1, %1, %1 " ctrl "\n"                    \
    "v_add_f32_dpp %2, %2, %2 " ctrl "\n"                    \
    "v_add_f32_dpp %3, %3, %3 " ctrl "\n"                    \
    "v_add_f32_dpp %4, %4, %4 " ctrl "\n"                    \
    "v_add_f32_dpp %5, %5, %5 " ctrl "\n"                    \
    "v_add_f32_dpp %6, %6, %6 " ctrl "\n"                    \
    "v_add_f32_dpp %7, %7, %7 " ctrl "\n"
// sum8_over_16_lanes: the first four steps only -- every lane of a 16-lane row ends up with its row's sum.
__device__ __forceinline__ void sum8_over_16_lanes(f32x4& a, f32x4& b) {
    float v0 = a[0], v1 = a[1], v2 = a[2], v3 = a[3], v4 = b[0], v5 = b[1], v6 = b[2], v7 = b[3];
    asm volatile("s_nop 1\n"
                 ND_DPP8("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 ND_DPP8("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 ND_DPP8("row_half_mirror row_mask:0xf bank_mask:0xf")
                 ND_DPP8("row_mirror row_mask:0xf bank_mask:0xf")
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
    a[0] = v0; a[1] = v1; a[2] = v2; a[3] = v3;
    b[0] = v4; b[1] = v5; b[2] = v6; b[3] = v7;
}
__device__ __forceinline__ void sum8_over_32_lanes(f32x4& a, f32x4& b) {
    sum8_over_16_lanes(a, b);
    float v0 = a[0], v1 = a[1], v2 = a[2], v3 = a[3], v4 = b[0], v5 = b[1], v6 = b[2], v7 = b[3];
    asm volatile("s_nop 1\n"
                 ND_DPP8("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
    a[0] = v0; a[1] = v1; a[2] = v2; a[3] = v3;
    b[0] = v4; b[1] = v5; b[2] = v6; b[3] = v7;
}
#undef ND_DPP8

struct ConvArgs;
// conv_wino4_kernel (nd_conv_winograd_quad.hip), launched by nd_conv3x3_winograd_nhwc's variant 12: 4 waves, 128 px x 64 ch,
// two blocks per CU; its halo buffers are filled by kWino4HaloRounds LDS-DMA rounds of 4 waves x 64 lanes x 16 bytes
constexpr int kWino4HaloRounds = 7;
constexpr int kWino4HaloPixels = kWino4HaloRounds * 4 * 64 / 8;      // 224 >= the 208 halo pixels the host admits
int launch_wino4(const ConvArgs& a, int grid, size_t lds, hipStream_t s);
// gemm_f32_kernel (nd_gemm_f32.hip), launched by nd_conv_nhwc's variant 13
int launch_gemm_f32(const ConvArgs& a, int grid, hipStream_t s);
// gemm4_kernel (nd_gemm_f32_quad.hip), launched by nd_conv_nhwc's variant 14
int launch_gemm4(const ConvArgs& a, int grid, hipStream_t s, int tm = 4);
// 256- or 128-pixel blocks for gemm4_kernel (variants 14 / 15 of nd_conv_nhwc) where no caller measured: the smaller tile where
// it fills the chip's block slots (512 at two 256-pixel blocks per CU, 768 at three 128-pixel ones) better -- rounds of blocks
// are what a short GEMM pays for
static inline int gemm4_pick_tm(long M, int nt) {
    if (M % 256) return 2;
    const long b4 = M / 256 * nt, b2 = M / 128 * nt;
    const double e4 = (double)b4 / (double)(((b4 + 511) / 512) * 512), e2 = (double)b2 / (double)(((b2 + 767) / 768) * 768);
    return e2 > e4 + 0.04 ? 2 : 4;
}

static inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

}  // namespace nd

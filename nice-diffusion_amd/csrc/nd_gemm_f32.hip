// K6, GEMM-shaped fp32 1x1 convolution (variant 13 of nd_conv_nhwc; flat pixel lists only):
//   out[M][N] = x[M][K] . w[N][K]^T  with K = Cin of a few hundred (qkv / proj / skip convolutions, model.py:247-253,182).
// The conv-shaped 1x1 forms (nd_conv_mfma.hip) stage one 32-channel chunk per barrier through registers or stream both
// operands straight from global memory; their MFMA + epilogue skeleton alone reaches 125 TFLOP/s and they run at 100-117
// (the vendor GEMM sustains 150 on these shapes: tools/gemm_ceiling.py).  Here BOTH operands go through a ring of THREE
// LDS stages filled by LDS-DMA -- per 32-channel chunk 256 pixel rows x 128 bytes (XOR swizzle applied to the per-lane
// SOURCE address; the LDS side of a DMA is lane-linear) + the chunk's 4 k-steps x 4 n-tile weight fragments of 1 KiB --
// so a chunk has two whole chunks of compute (~16 000 matrix-pipe cycles) to arrive, no operand passes through a VGPR on
// its way in, and per stage every wave issues exactly 4 + 2 DMAs, which makes the one counted wait per chunk
// (vmcnt(6): everything but the youngest stage has landed) uniform.  8 waves = 4 (pixels) x 2 (channels), wave tile
// 64 px x 64 ch, block 256 px x 128 ch, 144 KiB of LDS.  The same layout and DMA schedule as gemm_bf16_kernel
// (nd_conv_bf16.hip), on v_mfma_f32_32x32x2_f32.
// GN = GroupNorm(+AdaGN)(+SiLU) of the input applied to the pixel fragments as they leave LDS (one image per block).
#include "nd_conv_common.h"

namespace nd {

#define ND_GLDS16F(gptr, lptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

template <bool GN>
__global__ void __launch_bounds__(512, 2)
    gemm_f32_kernel(const ConvArgs p) {
    constexpr int BM = 256, BN = 128, TM = 2, TN = 2;
    constexpr int A_W = BM * 32;                      // words of the A part of a stage (256 rows x 128 bytes)
    constexpr int STAGE_W = A_W + 16 * 256;           // + 16 weight fragments of 1 KiB
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 3 stages

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    const int total = gridDim.x;
    const int q = total >> 3, r = total & 7, xcd = blockIdx.x & 7;
    const int idp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    int mblk, nblk;
    tile_of(idp, p.mt, p.nt, p.ngroup, mblk, nblk);
    const int M = p.W;                                 // flat pixel list
    const int m0 = mblk * BM, n0 = nblk * BN;
    const int Ctot = p.C0 + p.C1;
    const int nchunks = p.NC32;

    auto swz = [](int row) -> int { return (row >> 1) & 7; };

    // ---- A DMA descriptors: piece u = j * 8 + wave (j = 0..3) = rows 8u .. 8u+7; lane -> row 8u + lane/8, physical slot lane%8
    int arow[4], asl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (j * 8 + wave) * 8 + (lane >> 3);
        arow[j] = (m0 + row < M) ? (m0 + row) : -1;
        asl[j] = ((lane & 7) ^ swz(row)) << 2;       // first channel (within the chunk) of the logical slot this lane fills
    }
    // ---- weight DMA descriptors: fragment f = j * 8 + wave (j = 0, 1) = (k-step f / 4, n tile f % 4);
    //      packed weights [c32][n tile][kc][lane][4]
    const float* wsrc[2];
    int wdst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int f = j * 8 + wave;
        const int kc = f >> 2, nl = f & 3;
        int ntile = nblk * 4 + nl;
        if (ntile > p.NT32 - 1) ntile = p.NT32 - 1;
        wsrc[j] = p.w + ((size_t)ntile * 4 + kc) * 256 + lane * 4;
        wdst[j] = A_W + f * 256;
    }
    const size_t c32_stride = (size_t)p.NT32 * 4 * 256;
    auto issue_stage = [&](int ch, int buf) {
        float* st = smem + buf * STAGE_W;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = ch * 32 + asl[j];
            const int g = arow[j];
            const float* src = (c < p.C0) ? (p.x0 + (size_t)(g < 0 ? 0 : g) * p.ldx0 + c)
                                          : (p.x1 + (size_t)(g < 0 ? 0 : g) * p.ldx1 + (c - p.C0));
            src = (g >= 0 && c < Ctot) ? src : p.zero;
            ND_GLDS16F(src, st + (j * 8 + wave) * 256);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) ND_GLDS16F(wsrc[j] + (size_t)ch * c32_stride, st + wdst[j]);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    // this lane's two A rows (word offsets inside a stage; the slot is added per k-step) and its two weight fragments
    int aoff[TM], asw[TM];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int row = (wm * TM + mi) * 32 + l31;
        aoff[mi] = row * 32;
        asw[mi] = swz(row);
    }
    const int boff = A_W + ((wn * TN) * 64 + lane) * 4;

    // fused GroupNorm: coefficients of this block's image, channels 8 kc + 4 lh .. + 3 of the current chunk
    const float* ga = nullptr;
    const float* gb = nullptr;
    if constexpr (GN) {
        const int gimg = m0 / p.gn_hw;
        ga = p.gnA + (size_t)gimg * p.ld_gn;
        gb = p.gnB + (size_t)gimg * p.ld_gn;
    }

    issue_stage(0, 0);
    if (nchunks > 1) {
        issue_stage(1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    int buf = 0;                                       // stage of chunk ch = ch % 3
    for (int ch = 0; ch < nchunks; ++ch) {
        int nb2 = buf + 2;
        if (nb2 >= 3) nb2 -= 3;
        if (ch + 2 < nchunks) issue_stage(ch + 2, nb2);            // into the stage chunk ch-1 was read from (barrier passed)
        const float* st = smem + buf * STAGE_W;
        f32x4 a_fr[2][TM], b_fr[2][TN];
        auto read_frags = [&](int slot, int kc) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) b_fr[slot][ni] = *reinterpret_cast<const f32x4*>(st + boff + (kc * 4 + ni) * 256);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                a_fr[slot][mi] = *reinterpret_cast<const f32x4*>(st + aoff[mi] + ((((kc << 1) | lh) ^ asw[mi]) << 2));
        };
        read_frags(0, 0);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int cur = kc & 1, nxt = cur ^ 1;
            if (kc < 3) read_frags(nxt, kc + 1);
            if constexpr (GN) {
                int c = ch * 32 + kc * 8 + 4 * lh;
                c = c < Ctot - 4 ? c : Ctot - 4;           // past the last channel the weights are zero: any finite coefficient will do
                const f32x4 A4 = *reinterpret_cast<const f32x4*>(ga + c);
                const f32x4 B4 = *reinterpret_cast<const f32x4*>(gb + c);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = a_fr[cur][mi][e] * A4[e] + B4[e];
                        if (p.gn_silu) v = fast_silu(v);
                        a_fr[cur][mi][e] = v;
                    }
                }
            }
            ND_PRIO(1);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_fr[cur][ni][j], a_fr[cur][mi][j], acc[mi][ni], 0, 0, 0);
            ND_PRIO(0);
        }
        // publish the NEXT chunk: all but this wave's youngest stage (6 DMAs, issued above) has landed; the last two
        // iterations issue nothing, so everything outstanding is awaited
        if (ch + 2 < nchunks) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        buf = (buf == 2) ? 0 : buf + 1;
    }

    // ---- epilogue: lane = one pixel, register group g4 = 4 consecutive output channels 8*g4 + 4*lh .. +3 of the n tile
    const bool vec_ok = ((p.ldo & 3) == 0) && (!p.res || (p.ldr & 3) == 0);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + (wm * TM + mi) * 32 + l31;
        if (m < M) {
            const size_t opix = (size_t)m;
            const float* rr = p.res ? p.res + opix * p.ldr : nullptr;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int n = n0 + (wn * TN + ni) * 32 + 8 * g4 + 4 * lh;
                    if (n + 3 < p.N && vec_ok) {
                        f32x4 v = {acc[mi][ni][4 * g4 + 0], acc[mi][ni][4 * g4 + 1], acc[mi][ni][4 * g4 + 2],
                                   acc[mi][ni][4 * g4 + 3]};
                        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                        if (rr) v += *reinterpret_cast<const f32x4*>(rr + n);
                        if (p.silu_out) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
                        }
                        *reinterpret_cast<f32x4*>(p.out + opix * p.ldo + n) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (n + e < p.N) {
                                float v = acc[mi][ni][4 * g4 + e];
                                if (p.bias) v += p.bias[n + e];
                                if (rr) v += rr[n + e];
                                if (p.silu_out) v = fast_silu(v);
                                p.out[opix * p.ldo + n + e] = v;
                            }
                        }
                    }
                }
            }
        }
    }
}

int launch_gemm_f32(const ConvArgs& a, int grid, hipStream_t s) {
    const size_t lds = (size_t)3 * 48 * 1024;
    if (a.gnA) {
        auto kern = gemm_f32_kernel<true>;
        static bool attr_set[kMaxDevices] = {};
        if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), attr_set, "nd_conv_nhwc")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
    } else {
        auto kern = gemm_f32_kernel<false>;
        static bool attr_set[kMaxDevices] = {};
        if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), attr_set, "nd_conv_nhwc")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
    }
    return check_launch("nd_conv_nhwc");
}

}  // namespace nd

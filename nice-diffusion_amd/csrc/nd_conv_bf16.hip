// bf16 form of K5/K6 (BASELINE configs[3], [4]): 3x3 (stride 1, pad 1) and 1x1 convolution / linear as an implicit
// GEMM on v_mfma_f32_32x32x16_bf16 -- bf16 activations and weights in HBM, fp32 accumulation, bf16 (or fp32) output.
//
// Same fused surroundings as the fp32 kernel (nd_conv_mfma.hip): two-source input = torch.cat of the skip connection
// (model.py:474), bias, per-image timestep-embedding add (model.py:205), residual add (model.py:211,291), nearest-2x
// upsampling of the input or of the residual (model.py:77-79), SiLU on the output.
//
// The bf16 matrix instruction does 16x the flops per cycle of the fp32 one, so the kernel is shaped by operand delivery,
// not by the matrix pipe:
//   wave tile   TM x TN tiles of 32x32 with TM*TN = 8 for the large layers (128 px x 64 ch: 8 MFMAs = 256 cycles per
//               k-step against 4 ds_read_b128 + 2 global_load_dwordx4), 8 waves = 2 per SIMD, 256 px x 256 ch per block:
//               a block re-reads its weight slab once per 256 pixels (the dominant L2 -> CU stream at this rate).
//   A operand   activations: the (TH+2)x(TW+2) halo tile of a 64-channel chunk (128-byte rows = full cache lines per
//               pixel) is staged in LDS once and read at 9 shifted positions; lane (pixel i, half h) reads ONE 16-byte
//               slot = 8 channels = its whole B-side fragment of a k-step (k = 16 channels).  XOR swizzle as in the
//               fp32 kernel (rows are 128 bytes in both).
//   B operand   weights, pre-packed into fragment order [c64][n tile][tap][k-step][lane][8 bf16]: one coalesced 1 KiB
//               global_load_dwordx4 per fragment, straight to VGPRs, three k-steps ahead (ring of 4).
//   D^T = W.X^T a lane ends up with ONE pixel and 4 consecutive output channels per register group: 8-byte bf16 stores.
#include "nd_conv_bf16_args.h"
#include <type_traits>

#ifndef ND_BF16_SCHED
#define ND_BF16_SCHED 1
#endif
// Prologue and epilogue of a block are vector / memory instruction streams without MFMAs, while the other block of the CU
// is in its MFMA loop at priority 1: left at priority 0 they only issue in the gaps the older block leaves (per-wave
// timeline, tools/bf16_timeline.py: epilogue 15-19 us of a 68 us block life on 128x128 256->256).
#ifndef ND_BF16_EDGEPRIO
#define ND_BF16_EDGEPRIO 3
#endif
#if ND_BF16_EDGEPRIO
#define ND_EDGE_PRIO() __builtin_amdgcn_s_setprio(ND_BF16_EDGEPRIO)
#else
#define ND_EDGE_PRIO()
#endif

namespace nd {

// The arguments of split s, derived from the whole problem's: a convolution over the channel sub-range of that split.
__device__ __forceinline__ void split_k_args(ConvArgsH& p, int s, int taps) {
    const int span = p.kchunks * 64;
    const int c_begin = s * span;
    int nc = p.NC64 - s * p.kchunks;
    nc = nc < p.kchunks ? nc : p.kchunks;
    p.w += (size_t)s * p.kchunks * ((size_t)p.NT32 * taps * 4 * 512);      // both fragment layouts: NT32 * taps * 4 KiB-fragments per chunk
    if (c_begin < p.C0) {
        p.x0 += c_begin;
        const int left0 = p.C0 - c_begin;
        if (left0 >= span) {
            p.C0 = span;
            p.C1 = 0;
        } else {
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
    p.NC64 = nc;
    p.out = static_cast<float*>(p.out) + (size_t)s * p.ws_stride;
}


// K16: the input has <= 16 channels (the UNet's first convolution: 3 image channels padded to 8), so only the first
// k-step of every tap holds anything -- the other three (zero activations times zero-padded weights) are skipped:
// 9 instead of 36 k-steps, the same bits.
template <int WM, int WN, int TM, int TN, int TAPS, bool STATS = false, bool K16 = false>
__global__ void __launch_bounds__(WM* WN * 64, 2)
    conv_bf16_kernel(const ConvArgsH pin) {
    ConvArgsH p = pin;
    if (pin.ksplit > 1) split_k_args(p, blockIdx.y, TAPS);
#if defined(ND_BF_DIAG)
    // diagnostic build only (tools/bf16_timeline.py): stamps go to the buffer passed as `rowbias`, which is then ignored
    unsigned* const dg_buf = reinterpret_cast<unsigned*>(const_cast<float*>(p.rowbias));
    p.rowbias = nullptr;
    const unsigned long long dg_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long dg_t1 = 0, dg_t2 = 0, dg_c1 = 0, dg_c2 = 0;
    unsigned dg_ch[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    ND_EDGE_PRIO();
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int PAD = (TAPS == 9) ? 1 : 0;
    constexpr int NSUB = (TAPS == 9) ? 1 : 2;          // 64-channel sub-chunks per LDS chunk
    constexpr int SPR = 8 * NSUB;                      // 16-byte slots per halo pixel row
    constexpr int ROWF = 32 * NSUB;                    // 4-byte words per halo pixel row
    constexpr int KSTEPS = 4 * NSUB;                   // k-steps (16 channels each) per chunk and tap
    constexpr int STEPS = TAPS * 4;                    // fragments per 64-channel chunk and n tile
    // halo 16-byte items per thread per chunk: 3x3 tiles carry up to 1.5625x their pixels as halo (8x8 maps; fetched in 3 batches,
    // one per tap row); the 1x1 form has exactly BM rows (fetched one group per k-step)
    constexpr int NBI = (TAPS == 9) ? (((BM * 25 + 2 * NT - 1) / (2 * NT) + 2) / 3) : ((BM * SPR / NT + KSTEPS - 1) / KSTEPS);
    constexpr int NBATCH = (TAPS == 9) ? 3 : KSTEPS;
    constexpr int MAXHI = NBI * NBATCH;
    static_assert(NT % SPR == 0, "");

    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 x [HP][ROWF]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave - wm * WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    // ---- XCD-aware block -> tile map (see nd_conv_mfma.hip)
    const int total = gridDim.x;
    const int q = total >> 3, r = total & 7, xcd = blockIdx.x & 7;
    const int idp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    int mblk, nblk;
    tile_of(idp, p.mt, p.nt, p.ngroup, mblk, nblk);
    const int tx = mblk % p.tiles_x;
    const int tmp = mblk / p.tiles_x;
    const int ty = tmp % p.tiles_y;
    const int ig = tmp / p.tiles_y;

    const int TH = 1 << p.thl, TW = 1 << p.twl;
    const int HH = TH + 2 * PAD, HW = TW + 2 * PAD;
    const int HPI = HH * HW;
    const int HP = HPI << p.nibl;
    const int img0 = ig << p.nibl, oy0 = ty << p.thl, ox0 = tx << p.twl;
    const int n0 = nblk * BN;

    // ---- halo descriptors
    const int hslot = tid % SPR;
    const int hrow0 = tid / SPR;
    const float inv_hw = 1.0f / (float)HW, inv_hpi = 1.0f / (float)HPI;
    int gpix[MAXHI];
#pragma unroll
    for (int k = 0; k < MAXHI; ++k) {
        const int hp = hrow0 + k * (NT / SPR);
        int g = -1;
        if (hp < HP) {
            // hp < 4096, HW <= 66: (x + 0.5) * (1 / d) truncates to x / d exactly (the quotient's distance from an integer is
            // >= 0.5 / d, the float error < 1e-3 of that) -- 18 integer divisions less in every block's prologue
            const int li = (p.nibl == 0) ? 0 : (int)(((float)hp + 0.5f) * inv_hpi);
            const int rem = hp - li * HPI;
            const int hy = (int)(((float)rem + 0.5f) * inv_hw);
            const int hx = rem - hy * HW;
            const int img = img0 + li;
            const int iy = oy0 - PAD + hy, ix = ox0 - PAD + hx;
            if (img < p.NI && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                g = (img * p.Hs + (iy >> p.up)) * p.Ws + (ix >> p.up);
        }
        gpix[k] = g;
    }

    const int Ctot = p.C0 + p.C1;
    const int nchunks = (p.NC64 + NSUB - 1) / NSUB;

    auto swz = [](int hp) -> int { return (SPR == 8) ? ((hp >> 1) & 7) : (hp & 15); };
    auto load_halo_pixel = [&](int g, int ch) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int c = ch * (64 * NSUB) + (hslot << 3);
#if defined(ND_HABL_NOHALO)      // timing-only ablation: only chunk 0 is ever fetched
        if (ch > 0) { asm volatile("" :: "v"(g)); return v; }
#endif
        if (g >= 0 && c < Ctot) {
            const __bf16* src = (c < p.C0) ? (p.x0 + (size_t)g * p.ldx0 + c) : (p.x1 + (size_t)g * p.ldx1 + (c - p.C0));
            v = *reinterpret_cast<const f32x4*>(src);
        }
        return v;
    };
    // ---- fused GroupNorm (+SiLU) of the input: per-channel coefficients of this block's image in LDS behind the halo
    //      buffers (A[c] | B[c], c over the padded concatenated channels), applied to a fetched 16-byte item right before
    //      it is parked in LDS -- padding pixels and channels stay exactly zero, as the reference pads the NORMALISED
    //      tensor (model.py:190-194).  vmask bit k = item k holds a real pixel.
    const bool gn = p.gnA != nullptr;
    float* cfA = smem + 2 * (HP * ROWF);
    const int CPAD = nchunks * (64 * NSUB);
    float* cfB = cfA + CPAD;
    unsigned vmask = 0;
#pragma unroll
    for (int k = 0; k < MAXHI; ++k) vmask |= (gpix[k] >= 0 ? 1u : 0u) << k;
    auto stage_gn_coeffs = [&]() {
        if (gn) {
            const int gimg = (p.gn_hw > 0) ? (ox0 / p.gn_hw) : img0;
            for (int c = tid; c < CPAD; c += NT) {
                cfA[c] = (c < Ctot) ? p.gnA[(size_t)gimg * p.ld_gn + c] : 0.f;
                cfB[c] = (c < Ctot) ? p.gnB[(size_t)gimg * p.ld_gn + c] : 0.f;
            }
            __syncthreads();
        }
    };
    auto gn_xform = [&](f32x4 raw, int ch, bool valid) -> f32x4 {
        if (!gn || !valid) return raw;
        const int c = ch * (64 * NSUB) + (hslot << 3);
        if (c >= Ctot) return raw;
        const bf16x8 xv = as_bf16x8(raw);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(cfA + c), a1 = *reinterpret_cast<const f32x4*>(cfA + c + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(cfB + c), b1 = *reinterpret_cast<const f32x4*>(cfB + c + 4);
        union { f32x4 f; bf16x8 h; } o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = (float)xv[e] * (e < 4 ? a0[e & 3] : a1[e & 3]) + (e < 4 ? b0[e & 3] : b1[e & 3]);
            if (p.gn_silu) v = fast_silu(v);
            o.h[e] = (__bf16)v;
        }
        return o.f;
    };
    auto store_halo_item = [&](int k, int buf, f32x4 v, int ch) {
        const int hp = hrow0 + k * (NT / SPR);
        if (hp < HP) {
            float* dst = smem + buf * (HP * ROWF) + hp * ROWF + ((hslot ^ swz(hp)) << 2);
            *reinterpret_cast<f32x4*>(dst) = gn_xform(v, ch, (vmask >> k) & 1u);
        }
    };

    // ---- per-lane operand rows
    int a_hp[TM];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = (wm * TM + mi) * 32 + l31;
        const int li = m >> (p.thl + p.twl);
        const int py = (m >> p.twl) & (TH - 1);
        const int px = m & (TW - 1);
        a_hp[mi] = li * HPI + py * HW + px;
    }
    // weight fragment stream of n tile ni: [c64][n tile][step][lane][8 bf16]; one fragment = 512 bf16 = 1 KiB
    const __bf16* bp[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        int ntile = nblk * (BN / 32) + wn * TN + ni;
        if (ntile > p.NT32 - 1) ntile = p.NT32 - 1;      // N tail: results are discarded in the epilogue
        bp[ni] = p.w + (size_t)ntile * (STEPS * 512) + lane * 8;
    }
    const size_t c64_jump = (size_t)(p.NT32 - 1) * (STEPS * 512);
    int ld_in_c64 = 0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    // operand registers.  Weights: ring of RING fragments per n tile, fetched RING-1 k-steps ahead.  Activations: ONE
    // register set -- the slot of tile mi for the NEXT k-step is re-read right behind the MFMAs that consumed it and has
    // the other TM-1 tiles' MFMAs (>= 190 cycles) to land; 128 accumulator + 16 + 24 operand registers leave room for
    // two waves per SIMD.
    constexpr int RING = wstream::bf16_ring(TAPS);      // measured: deeper rings (6 / 8 on the small register tiles) are slower
    constexpr int BDIST = RING - 1;
    static_assert(STEPS == wstream::bf16_steps(TAPS) &&
                  wstream::pad_chunks(BDIST, STEPS) <= wstream::kBf16PadChunks, "weight read-ahead exceeds the packer's zero padding");
    f32x4 a_fr[TM], b_fr[RING][TN];
    // the packed buffer ends in wstream::kBf16PadChunks zero chunks; the static_assert above is what lets the stream run
    // BDIST fragments past the last real one (split-K shifts the pointer by whole chunks and ends on a chunk boundary)
    static_assert(!K16 || (TAPS == 9 && wstream::pad_chunks(4 * BDIST, STEPS) <= wstream::kBf16PadChunks), "K16: 3x3 only; read-ahead of BDIST taps");
    auto advance_b = [&](f32x4 (&dst)[TN]) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
#if defined(ND_HABL_BL1)          // timing-only ablation: every wave of the chip loads the SAME 2 KiB every k-step (hits in its CU's L1)
            dst[ni] = *reinterpret_cast<const f32x4*>(p.w + (size_t)ni * 512 + lane * 8);
#elif !defined(ND_HABL_NOB)      // timing-only ablation: weight fragments stay whatever the registers hold
            dst[ni] = *reinterpret_cast<const f32x4*>(bp[ni]);
#endif
#if !defined(ND_HABL_BHIT)       // timing-only ablation (BHIT): a wave re-reads its first fragments every k-step (its own L1 / L2 lines)
            bp[ni] += K16 ? 4 * 512 : 512;          // K16: the tap's k-steps 1..3 are skipped
#endif
        }
#if !defined(ND_HABL_BHIT)
        if (++ld_in_c64 == (K16 ? TAPS : STEPS)) {
            ld_in_c64 = 0;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) bp[ni] += c64_jump;
        }
#endif
    };
    // one k-step: TM x TN MFMAs; behind tile mi's MFMAs its fragment for the next step is read from LDS word offset
    // noff[mi] (already swizzled)
    auto mfma_step = [&](const f32x4 (&bw)[TN], const float* hbuf, const int (&noff)[TM]) {
        ND_PRIO(1);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(bw[ni]), as_bf16x8(a_fr[mi]), acc[mi][ni], 0, 0, 0);
#if !defined(ND_HABL_NOA)        // timing-only ablation: no LDS fragment reads
            a_fr[mi] = *reinterpret_cast<const f32x4*>(hbuf + noff[mi]);
#else
            asm volatile("" :: "v"(noff[mi]));
#endif
        }
#if ND_BF16_SCHED && !defined(ND_HABL_NOA)
        // pin the interleave: TN MFMAs, then the one LDS read that refills the fragment they consumed (left alone the
        // scheduler sinks all TM reads behind the last MFMA, so the next step opens waiting for LDS)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            __builtin_amdgcn_sched_group_barrier(0x008, TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#endif
        ND_PRIO(0);
    };

    // ---- prologue: chunk 0 halo, first weight fragments.  All loads are issued before the first of them is used (one
    //      memory latency instead of MAXHI in a row: load -> transform -> LDS store per item was 6 us of a 66 us block)
    {
        f32x4 ph0[MAXHI];
#pragma unroll
        for (int k = 0; k < MAXHI; ++k) ph0[k] = load_halo_pixel(gpix[k], 0);
#pragma unroll
        for (int d = 0; d < BDIST; ++d) advance_b(b_fr[d]);
        __builtin_amdgcn_sched_barrier(0);
        stage_gn_coeffs();      // behind the loads: the coefficient table's own latency hides under theirs
#pragma unroll
        for (int k = 0; k < MAXHI; ++k) store_halo_item(k, 0, ph0[k], 0);
    }
    __syncthreads();
#if defined(ND_BF_DIAG)
    dg_t1 = __builtin_amdgcn_s_memrealtime();
    dg_c1 = __builtin_amdgcn_s_memtime();
#endif

    for (int ch = 0; ch < nchunks; ++ch) {
        const float* hbuf = smem + (ch & 1) * (HP * ROWF);
        const bool halo_next = (ch + 1) < nchunks;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int hp = a_hp[mi];
            a_fr[mi] = *reinterpret_cast<const f32x4*>(hbuf + hp * ROWF + ((lh ^ swz(hp)) << 2));
        }
        if constexpr (TAPS == 9) {
#pragma unroll 1
            for (int dy = 0; dy < 3; ++dy) {
                // next chunk's halo arrives in 3 batches of NBI items (one batch per tap row), fetched at the top of the
                // row and parked in the other LDS buffer after its 12 k-steps (~3000 cycles later)
                int gs[NBI];
#pragma unroll
                for (int i = 0; i < NBI; ++i) {
                    int g = gpix[i];
                    g = (dy == 1) ? gpix[NBI + i] : g;
                    g = (dy == 2) ? gpix[2 * NBI + i] : g;
                    gs[i] = halo_next ? g : -1;
                }
                f32x4 phb[NBI];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int tapoff = dy * HW + dx;
                    const int tapoff_n = (dx < 2) ? tapoff + 1 : ((dy < 2) ? (dy + 1) * HW : 0);
#pragma unroll
                    for (int kc = 0; kc < (K16 ? 1 : 4); ++kc) {
                        const int st = K16 ? dx : dx * 4 + kc;      // 0 .. 11 (K16: 0 .. 2), compile-time: ring slots are static
                        advance_b(b_fr[(st + BDIST) % RING]);
                        if (dx == 0 && kc == 0) {
#pragma unroll
                            for (int i = 0; i < NBI; ++i) phb[i] = load_halo_pixel(gs[i], ch + 1);
                        }
                        int noff[TM];
                        {
                            const int nslot = K16 ? lh : ((((kc + 1) & 3) << 1) | lh);
                            const int toff = (K16 || kc == 3) ? tapoff_n : tapoff;
#pragma unroll
                            for (int mi = 0; mi < TM; ++mi) {
                                const int hp = a_hp[mi] + toff;     // (after the last step of a chunk this reads stale but
                                noff[mi] = hp * ROWF + ((nslot ^ swz(hp)) << 2);   //  in-bounds data that is discarded)
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_step(b_fr[st % RING], hbuf, noff);
                    }
                }
                if (halo_next) {
#pragma unroll
                    for (int i = 0; i < NBI; ++i) {
                        const int k = dy * NBI + i;      // dy is a run-time value: the item index is plain arithmetic
                        const int hp = hrow0 + k * (NT / SPR);
                        if (hp < HP) {
                            float* dst = smem + ((ch + 1) & 1) * (HP * ROWF) + hp * ROWF + ((hslot ^ swz(hp)) << 2);
                            *reinterpret_cast<f32x4*>(dst) = gn_xform(phb[i], ch + 1, (vmask >> k) & 1u);
                        }
                    }
                }
            }
        } else {
            // 1x1: KSTEPS k-steps per chunk; the next chunk's rows are fetched one group of NBI per step and parked two
            // steps later (the last groups behind the loop)
            f32x4 ph[KSTEPS][NBI];
            constexpr int SD = 2;      // k-steps between a group's fetch and its LDS store
            const bool second = (ch * NSUB + 1) < p.NC64;          // the chunk's second 64-channel half holds real channels
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                if (ks < 4 || second) {
                    advance_b(b_fr[(ks + BDIST) % RING]);
#pragma unroll
                    for (int i = 0; i < NBI; ++i) ph[ks][i] = load_halo_pixel(halo_next ? gpix[ks * NBI + i] : -1, ch + 1);
                    int noff[TM];
                    {
                        const int nstep = (ks + 1) & (KSTEPS - 1);      // wraps to 0 after the last step: discarded
                        const int nslot = (nstep << 1) | lh;
#pragma unroll
                        for (int mi = 0; mi < TM; ++mi) noff[mi] = a_hp[mi] * ROWF + ((nslot ^ swz(a_hp[mi])) << 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_step(b_fr[ks % RING], hbuf, noff);
                } else {
#pragma unroll
                    for (int i = 0; i < NBI; ++i) ph[ks][i] = load_halo_pixel(halo_next ? gpix[ks * NBI + i] : -1, ch + 1);
                }
                if (ks >= SD && halo_next) {
#pragma unroll
                    for (int i = 0; i < NBI; ++i) store_halo_item((ks - SD) * NBI + i, (ch + 1) & 1, ph[ks - SD][i], ch + 1);
                }
            }
            if (halo_next) {
#pragma unroll
                for (int ks = KSTEPS - SD; ks < KSTEPS; ++ks)
#pragma unroll
                    for (int i = 0; i < NBI; ++i) store_halo_item(ks * NBI + i, (ch + 1) & 1, ph[ks][i], ch + 1);
            }
        }
        // halo hand-over: only LDS traffic has to be complete; the weight prefetch stays in flight across the barrier
#if !defined(ND_HABL_NOBAR)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#endif
#if defined(ND_BF_DIAG)
        if (ch < 8) dg_ch[ch] = (unsigned)(__builtin_amdgcn_s_memrealtime() - dg_t0);
#endif
    }
#if defined(ND_BF_DIAG)
    dg_t2 = __builtin_amdgcn_s_memrealtime();
    dg_c2 = __builtin_amdgcn_s_memtime();
#endif
#if defined(ND_HABL_NOEPI)
    if (p.N > 0) { if (acc[0][0][0] == 123.456f) static_cast<float*>(p.out)[0] = acc[0][0][1]; return; }
#endif
#if defined(ND_HABL_EPICOAL)
    // timing-only: the wave's output region written ONCE with ideally coalesced 16-byte stores (8 lanes = one pixel's 128
    // bytes), wrong values -- what an LDS-staged epilogue could reach at most
    if (p.N > 0 && !p.out_f32) {
        constexpr int CPR = TN * 4;                      // 16-byte pieces per pixel row of the wave tile
#pragma unroll
        for (int it = 0; it < TM * TN * 2; ++it) {
            const int idx = it * 64 + lane;
            const int pl = idx / CPR, piece = idx - pl * CPR;
            const int m = wm * TM * 32 + pl;
            const int oy = oy0 + ((m >> p.twl) & (TH - 1)), ox = ox0 + (m & (TW - 1));
            const int n = n0 + wn * TN * 32 + piece * 8;
            const int r = it * 8;
            f32x4 lo = {acc[0][0][r & 15], acc[TM - 1][0][(r + 1) & 15], acc[0][TN - 1][(r + 2) & 15], acc[TM - 1][TN - 1][(r + 3) & 15]};
            lo += *reinterpret_cast<const f32x4*>(p.bias + (n & ~3));
            union { bf16x8 h; f32x4 f; } o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.h[e] = (__bf16)(lo[e & 3] + (float)e);
            if (oy < p.H && ox < p.W && n + 7 < p.N)
                *reinterpret_cast<f32x4*>(static_cast<__bf16*>(p.out) + (size_t)((img0 * p.H + oy) * p.W + ox) * p.ldo + n) = o.f;
        }
#if defined(ND_BF_DIAG)
        if (dg_buf && (tid & 63) == 0) {
            unsigned* dg = dg_buf + ((size_t)blockIdx.x * (NT / 64) + wave) * 16;
            const unsigned long long dg_t3 = __builtin_amdgcn_s_memrealtime();
            dg[0] = (unsigned)dg_t0; dg[1] = (unsigned)(dg_t1 - dg_t0); dg[2] = (unsigned)(dg_t2 - dg_t0); dg[3] = (unsigned)(dg_t3 - dg_t0);
            dg[4] = (unsigned)(dg_c2 - dg_c1);
            dg[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            dg[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            dg[7] = (unsigned)(dg_t0 >> 32);
#pragma unroll
            for (int i = 0; i < 8; ++i) dg[8 + i] = dg_ch[i];
        }
#endif
        return;
    }
#endif

    // ---- epilogue: lane = one pixel, register group g4 = 4 consecutive output channels 8*g4 + 4*lh .. +3 of the n tile
    ND_EDGE_PRIO();
    const bool vec_ok = ((p.ldo & 3) == 0) && (!p.res || (p.ldr & 3) == 0) && (!p.rowbias || (p.ld_rowbias & 3) == 0);
    // Two forms.  The COMMON one (whole n block inside N, vector-friendly strides, one image per block, bf16 output, no SiLU;
    // TN == 2: LDS sized for the staging regions, p.coal) is kept small on purpose: the general form below unrolls to
    // ~80 KB of branchy code, and a block that runs it once spends 15-19 us fetching instructions
    // (tools/bf16_timeline.py; the I-cache is 64 KB for two CUs) -- three times what its stores cost.  Everything that
    // can vary per launch is a compile-time copy (residual, timestep row); tiles that reach over the image edge take the
    // general form.
    constexpr bool STAGED = (TN == 2);
    // staged stores: region of this wave = [TM*32 pixels][128 bytes], 16-byte slot s of pixel row r at slot s ^ (r & 7)
    // (several images per block -- the 8x8 maps -- take it too where no statistics are written: the image is per lane then)
    const bool common = vec_ok && (n0 + BN <= p.N) && (p.nibl == 0 || !STATS) && !p.silu_out && !p.out_f32 &&
                        img0 + (1 << p.nibl) <= p.NI && oy0 + TH <= p.H && ox0 + TW <= p.W && (!STAGED || p.coal != 0);
    char* const stg = reinterpret_cast<char*>(smem) + wave * (TM * 32 * 128);
    if (common) {
        // per pixel row of the wave tile: output row, residual row (the whole tile lies inside the image: no masks)
        const __bf16* rrow[TM];
        __bf16* orow[TM];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int m = (wm * TM + mi) * 32 + l31;
            const int oy = oy0 + ((m >> p.twl) & (TH - 1));
            const int ox = ox0 + (m & (TW - 1));
            const int img = img0 + (m >> (p.thl + p.twl));
            const size_t opix = (size_t)(img * p.H + oy) * p.W + ox;
            orow[mi] = static_cast<__bf16*>(p.out) + opix * p.ldo + n0 + wn * (TN * 32) + 4 * lh;
            rrow[mi] = nullptr;
            if (p.res) {
                const size_t rp = p.res_up ? ((size_t)(img * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) : opix;
                rrow[mi] = p.res + rp * p.ldr + n0 + wn * (TN * 32) + 4 * lh;
            }
        }
        const float* const bptr = p.bias + n0 + wn * (TN * 32) + 4 * lh;
        auto rows = [&](auto has_res, auto has_rb) {
            // the additions keep the general form's order (bias, timestep row, residual): results are bit-identical to it
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                f32x4 bv[4], rbv[4], s1[4], s2[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    s1[g4] = s2[g4] = f32x4{0.f, 0.f, 0.f, 0.f};
                    bv[g4] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (p.bias) bv[g4] = *reinterpret_cast<const f32x4*>(bptr + ni * 32 + 8 * g4);
                    if constexpr (decltype(has_rb)::value)
                        rbv[g4] = *reinterpret_cast<const f32x4*>(p.rowbias + (size_t)img0 * p.ld_rowbias + n0 + wn * (TN * 32) + 4 * lh +
                                                                  ni * 32 + 8 * g4);
                }
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const int pxl = mi * 32 + l31;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4 v = {acc[mi][ni][4 * g4 + 0], acc[mi][ni][4 * g4 + 1], acc[mi][ni][4 * g4 + 2], acc[mi][ni][4 * g4 + 3]};
                        v += bv[g4];                          // +0 when there is no bias
                        if constexpr (decltype(has_rb)::value) {
                            if (p.nibl == 0) {
                                v += rbv[g4];
                            } else {          // several images per block: the timestep row is this lane's image's
                                const int img = img0 + (((wm * TM + mi) * 32 + l31) >> (p.thl + p.twl));
                                v += *reinterpret_cast<const f32x4*>(p.rowbias + (size_t)img * p.ld_rowbias + n0 + wn * (TN * 32) + 4 * lh +
                                                                     ni * 32 + 8 * g4);
                            }
                        }
                        if constexpr (decltype(has_res)::value) {
                            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(rrow[mi] + ni * 32 + 8 * g4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                        }
                        const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        if constexpr (STAGED) {
                            *reinterpret_cast<bf16x4*>(stg + pxl * 128 + (((ni * 4 + g4) ^ (pxl & 7)) << 4) + lh * 8) = o;
                        } else {
                            *reinterpret_cast<bf16x4*>(orow[mi] + ni * 32 + 8 * g4) = o;
                        }
                        if constexpr (STATS && !STAGED) {      // sums of what was stored: the statistics are those of the bf16 tensor
                            const f32x4 of = {(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                s1[g4][e] += of[e];
                                s2[g4][e] += of[e] * of[e];
                            }
                        }
                    }
                }
                if constexpr (STATS && !STAGED) {
                    // per-channel sums over this wave's TM x 32 pixels -> row (pixel tile, wave row) of p.chstats (same order of
                    // additions as the general form: mi ascending, then the 32-lane reduction)
                    float* cs = p.chstats + ((size_t)img0 * p.cs_rows + (size_t)(ty * p.tiles_x + tx) * WM + wm) * 2 * p.N +
                                n0 + (wn * TN + ni) * 32 + 4 * lh;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        sum8_over_32_lanes(s1[g4], s2[g4]);
                        if (l31 == 31) {
                            *reinterpret_cast<f32x4*>(cs + 8 * g4) = s1[g4];
                            *reinterpret_cast<f32x4*>(cs + p.N + 8 * g4) = s2[g4];
                        }
                    }
                }
#if defined(ND_BF_DIAG)
                if (ni == 0) dg_ch[5] = (unsigned)(__builtin_amdgcn_s_memrealtime() - dg_t0);
#endif
            }
        };
        if (p.res) {
            if (p.rowbias) rows(std::true_type{}, std::true_type{});
            else rows(std::true_type{}, std::false_type{});
        } else {
            if (p.rowbias) rows(std::false_type{}, std::true_type{});
            else rows(std::false_type{}, std::false_type{});
        }
        if constexpr (STAGED) {
            // the wave reads its region back row-major: 8 lanes = the 8 slots of one pixel row = 128 contiguous bytes of the output
            __bf16* const obase = static_cast<__bf16*>(p.out) + n0 + wn * 64 + (lane & 7) * 8;
            // STATS: the same read feeds the statistics -- a lane sees 8 channels of TM*4 pixels (8 values per 16-byte item
            // instead of 4 per 8-byte store in the register layout, no conversion back from the stored bf16: the bits are the
            // high half of the float), then the 8 lane groups that hold the same channels are added by three exchanges.
            // Fixed order: pixels ascending per lane, then lane ^ 8, ^ 16, ^ 32.
            float sv[8], sq[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) sv[c] = sq[c] = 0.f;
#pragma unroll
            for (int it = 0; it < TM * 4; ++it) {
                const int pxl = it * 8 + (lane >> 3);
                union { f32x4 f; unsigned u[4]; } v;
                v.f = *reinterpret_cast<const f32x4*>(stg + pxl * 128 + (((lane & 7) ^ (pxl & 7)) << 4));
                const int m = wm * TM * 32 + pxl;
                const int oy = oy0 + ((m >> p.twl) & (TH - 1));
                const int ox = ox0 + (m & (TW - 1));
                const int img = img0 + (m >> (p.thl + p.twl));
                *reinterpret_cast<f32x4*>(obase + ((size_t)(img * p.H + oy) * p.W + ox) * p.ldo) = v.f;
                if constexpr (STATS) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {          // dword j = channels 2j (low half), 2j + 1 (high half)
                        const float x0 = __uint_as_float(v.u[j] << 16), x1 = __uint_as_float(v.u[j] & 0xffff0000u);
                        sv[2 * j] += x0;
                        sv[2 * j + 1] += x1;
                        sq[2 * j] += x0 * x0;
                        sq[2 * j + 1] += x1 * x1;
                    }
                }
            }
            if constexpr (STATS) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
#pragma unroll
                    for (int d = 8; d <= 32; d <<= 1) {
                        sv[c] += __shfl_xor(sv[c], d, 64);
                        sq[c] += __shfl_xor(sq[c], d, 64);
                    }
                }
                if (lane < 8) {
                    float* cs = p.chstats + ((size_t)img0 * p.cs_rows + (size_t)(ty * p.tiles_x + tx) * WM + wm) * 2 * p.N +
                                n0 + wn * 64 + lane * 8;
                    *reinterpret_cast<f32x4*>(cs) = f32x4{sv[0], sv[1], sv[2], sv[3]};
                    *reinterpret_cast<f32x4*>(cs + 4) = f32x4{sv[4], sv[5], sv[6], sv[7]};
                    *reinterpret_cast<f32x4*>(cs + p.N) = f32x4{sq[0], sq[1], sq[2], sq[3]};
                    *reinterpret_cast<f32x4*>(cs + p.N + 4) = f32x4{sq[4], sq[5], sq[6], sq[7]};
                }
            }
        }
    } else {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = (wm * TM + mi) * 32 + l31;
        const int li = m >> (p.thl + p.twl);
        const int oy = oy0 + ((m >> p.twl) & (TH - 1));
        const int ox = ox0 + (m & (TW - 1));
        const int img = img0 + li;
        if (img < p.NI && oy < p.H && ox < p.W) {
            const size_t opix = (size_t)(img * p.H + oy) * p.W + ox;
            const float* rb = p.rowbias ? p.rowbias + (size_t)img * p.ld_rowbias : nullptr;
            const __bf16* rr = nullptr;
            if (p.res) {
                const size_t rp = p.res_up ? ((size_t)(img * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) : opix;
                rr = p.res + rp * p.ldr;
            }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int n = n0 + (wn * TN + ni) * 32 + 8 * g4 + 4 * lh;
                    if (n + 3 < p.N && vec_ok) {
                        f32x4 v = {acc[mi][ni][4 * g4 + 0], acc[mi][ni][4 * g4 + 1], acc[mi][ni][4 * g4 + 2],
                                   acc[mi][ni][4 * g4 + 3]};
                        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                        if (rb) v += *reinterpret_cast<const f32x4*>(rb + n);
                        if (rr) {
                            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(rr + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                        }
                        if (p.silu_out) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
                        }
                        if (p.out_f32) {
                            *reinterpret_cast<f32x4*>(static_cast<float*>(p.out) + opix * p.ldo + n) = v;
                        } else {
                            const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                            *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(p.out) + opix * p.ldo + n) = o;
                            if constexpr (STATS) {      // keep what was stored: the statistics below are those of the bf16 tensor
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[mi][ni][4 * g4 + e] = (float)o[e];
                            }
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (n + e < p.N) {
                                float v = acc[mi][ni][4 * g4 + e];
                                if (p.bias) v += p.bias[n + e];
                                if (rb) v += rb[n + e];
                                if (rr) v += (float)rr[n + e];
                                if (p.silu_out) v = fast_silu(v);
                                if (p.out_f32) static_cast<float*>(p.out)[opix * p.ldo + n + e] = v;
                                else static_cast<__bf16*>(p.out)[opix * p.ldo + n + e] = (__bf16)v;
                            }
                        }
                    }
                }
            }
        } else if constexpr (STATS) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        }
#if defined(ND_BF_DIAG)
        if (mi == 0) dg_ch[5] = (unsigned)(__builtin_amdgcn_s_memrealtime() - dg_t0);
#endif
    }
    }
#if defined(ND_BF_DIAG)
    dg_ch[6] = (unsigned)(__builtin_amdgcn_s_memrealtime() - dg_t0);
#if defined(ND_BF_DIAG_DRAIN)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dg_ch[7] = (unsigned)(__builtin_amdgcn_s_memrealtime() - dg_t0);
#endif
#endif
    if (STATS && !common) {
        // per-channel sums / sums of squares of the stored values over this wave's TM x 32 pixels -> row (pixel tile, wave
        // row) of p.chstats.  Host-checked: one image per block, N % 4 == 0, vector-friendly strides, bf16 output.
        float* cs = p.chstats + ((size_t)img0 * p.cs_rows + (size_t)(ty * p.tiles_x + tx) * WM + wm) * 2 * p.N;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = n0 + (wn * TN + ni) * 32 + 8 * g4 + 4 * lh;
                f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float r = acc[mi][ni][4 * g4 + e];
                        s1[e] += r;
                        s2[e] += r * r;
                    }
                }
                sum8_over_32_lanes(s1, s2);
                if (n + 3 < p.N && l31 == 31) {
                    *reinterpret_cast<f32x4*>(cs + n) = s1;
                    *reinterpret_cast<f32x4*>(cs + p.N + n) = s2;
                }
            }
        }
    }
#if defined(ND_BF_DIAG)
    if (dg_buf && (tid & 63) == 0) {
        unsigned* dg = dg_buf + ((size_t)blockIdx.x * (NT / 64) + wave) * 16;
        const unsigned long long dg_t3 = __builtin_amdgcn_s_memrealtime();
        dg[0] = (unsigned)dg_t0; dg[1] = (unsigned)(dg_t1 - dg_t0); dg[2] = (unsigned)(dg_t2 - dg_t0); dg[3] = (unsigned)(dg_t3 - dg_t0);
        dg[4] = (unsigned)(dg_c2 - dg_c1);
        dg[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);          // HW_REG_HW_ID
        dg[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20);         // HW_REG_XCC_ID
        dg[7] = (unsigned)(dg_t0 >> 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) dg[8 + i] = dg_ch[i];
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------
// The same kernel on v_mfma_f32_16x16x32_bf16 (TM, TN count 16-wide tiles; a k-step is 32 channels; weights packed by
// nd_repack_conv_weight_bf16 with layout 1).  Per flop it moves the same operand bytes as the 32x32x16 form, but the
// chip holds a higher clock under it (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.15x the FLOP/s at equal cycles),
// and the bf16 convolutions are limited by the clock the chip holds, not by stalls.
//   A operand (weights)     lane (n = lane & 15, kq = lane >> 4) holds w[n][32*ks + 8*kq + j]
//   B operand (activations) lane (pixel = lane & 15, kq)        holds x[pixel][32*ks + 8*kq + j] = LDS slot 4*ks + kq
//   C/D                     lane (pixel = lane & 15): registers 0..3 = output channels 4*kq .. 4*kq + 3
template <int WM, int WN, int TM, int TN, int TAPS>
__global__ void __launch_bounds__(WM* WN * 64, 2)
    conv_bf16s_kernel(const ConvArgsH pin) {
    ConvArgsH p = pin;
    if (pin.ksplit > 1) split_k_args(p, blockIdx.y, TAPS);
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 16;      // TM, TN count 16-wide tiles of v_mfma_f32_16x16x32_bf16
    constexpr int BN = WN * TN * 16;
    constexpr int PAD = (TAPS == 9) ? 1 : 0;
    constexpr int NSUB = (TAPS == 9) ? 1 : 2;          // 64-channel sub-chunks per LDS chunk
    constexpr int SPR = 8 * NSUB;                      // 16-byte slots per halo pixel row
    constexpr int ROWF = 32 * NSUB;                    // 4-byte words per halo pixel row
    constexpr int KSTEPS = 2 * NSUB;                   // k-steps (32 channels each) per chunk and tap
    constexpr int STEPS = TAPS * 2;                    // fragments per 64-channel chunk and 16-channel n tile
    // halo 16-byte items per thread per chunk: 3x3 tiles carry up to 1.5625x their pixels as halo (8x8 maps; fetched in 3 batches,
    // one per tap row); the 1x1 form has exactly BM rows (fetched one group per k-step)
    constexpr int NBI = (TAPS == 9) ? (((BM * 25 + 2 * NT - 1) / (2 * NT) + 2) / 3) : ((BM * SPR / NT + KSTEPS - 1) / KSTEPS);
    constexpr int NBATCH = (TAPS == 9) ? 3 : KSTEPS;
    constexpr int MAXHI = NBI * NBATCH;
    static_assert(NT % SPR == 0, "");

    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 x [HP][ROWF]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave - wm * WN;
    const int l15 = lane & 15;      // pixel (B operand column) / output channel (A operand row) inside a 16-wide tile
    const int kq = lane >> 4;       // which 8 of the k-step's 32 channels this lane holds

    // ---- XCD-aware block -> tile map (see nd_conv_mfma.hip)
    const int total = gridDim.x;
    const int q = total >> 3, r = total & 7, xcd = blockIdx.x & 7;
    const int idp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    int mblk, nblk;
    tile_of(idp, p.mt, p.nt, p.ngroup, mblk, nblk);
    const int tx = mblk % p.tiles_x;
    const int tmp = mblk / p.tiles_x;
    const int ty = tmp % p.tiles_y;
    const int ig = tmp / p.tiles_y;

    const int TH = 1 << p.thl, TW = 1 << p.twl;
    const int HH = TH + 2 * PAD, HW = TW + 2 * PAD;
    const int HPI = HH * HW;
    const int HP = HPI << p.nibl;
    const int img0 = ig << p.nibl, oy0 = ty << p.thl, ox0 = tx << p.twl;
    const int n0 = nblk * BN;

    // ---- halo descriptors
    const int hslot = tid % SPR;
    const int hrow0 = tid / SPR;
    int gpix[MAXHI];
#pragma unroll
    for (int k = 0; k < MAXHI; ++k) {
        const int hp = hrow0 + k * (NT / SPR);
        int g = -1;
        if (hp < HP) {
            const int li = hp / HPI;
            const int rem = hp - li * HPI;
            const int hy = rem / HW;
            const int hx = rem - hy * HW;
            const int img = img0 + li;
            const int iy = oy0 - PAD + hy, ix = ox0 - PAD + hx;
            if (img < p.NI && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                g = (img * p.Hs + (iy >> p.up)) * p.Ws + (ix >> p.up);
        }
        gpix[k] = g;
    }

    const int Ctot = p.C0 + p.C1;
    const int nchunks = (p.NC64 + NSUB - 1) / NSUB;

    auto swz = [](int hp) -> int { return (SPR == 8) ? ((hp >> 1) & 7) : (hp & 15); };
    auto load_halo_pixel = [&](int g, int ch) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int c = ch * (64 * NSUB) + (hslot << 3);
#if defined(ND_HABL_NOHALO)      // timing-only ablation: only chunk 0 is ever fetched
        if (ch > 0) { asm volatile("" :: "v"(g)); return v; }
#endif
        if (g >= 0 && c < Ctot) {
            const __bf16* src = (c < p.C0) ? (p.x0 + (size_t)g * p.ldx0 + c) : (p.x1 + (size_t)g * p.ldx1 + (c - p.C0));
            v = *reinterpret_cast<const f32x4*>(src);
        }
        return v;
    };
    // ---- fused GroupNorm (+SiLU) of the input: per-channel coefficients of this block's image in LDS behind the halo
    //      buffers (A[c] | B[c], c over the padded concatenated channels), applied to a fetched 16-byte item right before
    //      it is parked in LDS -- padding pixels and channels stay exactly zero, as the reference pads the NORMALISED
    //      tensor (model.py:190-194).  vmask bit k = item k holds a real pixel.
    const bool gn = p.gnA != nullptr;
    float* cfA = smem + 2 * (HP * ROWF);
    const int CPAD = nchunks * (64 * NSUB);
    float* cfB = cfA + CPAD;
    unsigned vmask = 0;
#pragma unroll
    for (int k = 0; k < MAXHI; ++k) vmask |= (gpix[k] >= 0 ? 1u : 0u) << k;
    if (gn) {
        const int gimg = (p.gn_hw > 0) ? (ox0 / p.gn_hw) : img0;
        for (int c = tid; c < CPAD; c += NT) {
            cfA[c] = (c < Ctot) ? p.gnA[(size_t)gimg * p.ld_gn + c] : 0.f;
            cfB[c] = (c < Ctot) ? p.gnB[(size_t)gimg * p.ld_gn + c] : 0.f;
        }
        __syncthreads();
    }
    auto gn_xform = [&](f32x4 raw, int ch, bool valid) -> f32x4 {
        if (!gn || !valid) return raw;
        const int c = ch * (64 * NSUB) + (hslot << 3);
        if (c >= Ctot) return raw;
        const bf16x8 xv = as_bf16x8(raw);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(cfA + c), a1 = *reinterpret_cast<const f32x4*>(cfA + c + 4);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(cfB + c), b1 = *reinterpret_cast<const f32x4*>(cfB + c + 4);
        union { f32x4 f; bf16x8 h; } o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = (float)xv[e] * (e < 4 ? a0[e & 3] : a1[e & 3]) + (e < 4 ? b0[e & 3] : b1[e & 3]);
            if (p.gn_silu) v = fast_silu(v);
            o.h[e] = (__bf16)v;
        }
        return o.f;
    };
    auto store_halo_item = [&](int k, int buf, f32x4 v, int ch) {
        const int hp = hrow0 + k * (NT / SPR);
        if (hp < HP) {
            float* dst = smem + buf * (HP * ROWF) + hp * ROWF + ((hslot ^ swz(hp)) << 2);
            *reinterpret_cast<f32x4*>(dst) = gn_xform(v, ch, (vmask >> k) & 1u);
        }
    };

    // ---- per-lane operand rows
    int a_hp[TM];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = (wm * TM + mi) * 16 + l15;
        const int li = m >> (p.thl + p.twl);
        const int py = (m >> p.twl) & (TH - 1);
        const int px = m & (TW - 1);
        a_hp[mi] = li * HPI + py * HW + px;
    }
    // weight fragment stream of 16-channel n tile ni: [c64][n16 tile][step][lane][8 bf16]; one fragment = 16 ch x 32 k = 1 KiB
    const __bf16* bp[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        int ntile = nblk * (BN / 16) + wn * TN + ni;
        if (ntile > 2 * p.NT32 - 1) ntile = 2 * p.NT32 - 1;      // N tail: results are discarded in the epilogue
        bp[ni] = p.w + (size_t)ntile * (STEPS * 512) + lane * 8;
    }
    const size_t c64_jump = (size_t)(2 * p.NT32 - 1) * (STEPS * 512);
    int ld_in_c64 = 0;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operand registers.  Weights: ring of RING fragments per n tile, fetched RING-1 k-steps ahead.  Activations: ONE
    // register set -- the slot of tile mi for the NEXT k-step is re-read right behind the MFMAs that consumed it and has
    // the other TM-1 tiles' MFMAs (>= 190 cycles) to land; 128 accumulator + 16 + 24 operand registers leave room for
    // two waves per SIMD.
    constexpr int RING = wstream::bf16s_ring(TN, TAPS);      // divides the 6 / 4 unrolled k-steps; k-steps are 32 channels here
    constexpr int BDIST = RING - 1;
    static_assert(STEPS == wstream::bf16s_steps(TAPS) &&
                  wstream::pad_chunks(BDIST, STEPS) <= wstream::kBf16PadChunks, "weight read-ahead exceeds the packer's zero padding");
    f32x4 a_fr[TM], b_fr[RING][TN];
    // the packed buffer ends in wstream::kBf16PadChunks zero chunks (a 1x1 stream of this layout has only 2 fragments per
    // chunk and n tile, so its ring of 4 reaches into the SECOND padding chunk: the static_assert above covers it)
    auto advance_b = [&](f32x4 (&dst)[TN]) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
#if !defined(ND_HABL_NOB)        // timing-only ablation: weight fragments stay whatever the registers hold
            dst[ni] = *reinterpret_cast<const f32x4*>(bp[ni]);
#endif
            bp[ni] += 512;
        }
        if (++ld_in_c64 == STEPS) {
            ld_in_c64 = 0;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) bp[ni] += c64_jump;
        }
    };
    // one k-step: TM x TN MFMAs; behind tile mi's MFMAs its fragment for the next step is read from LDS word offset
    // noff[mi] (already swizzled)
    auto mfma_step = [&](const f32x4 (&bw)[TN], const float* hbuf, const int (&noff)[TM]) {
        ND_PRIO(1);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(bw[ni]), as_bf16x8(a_fr[mi]), acc[mi][ni], 0, 0, 0);
#if !defined(ND_HABL_NOA)        // timing-only ablation: no LDS fragment reads
            a_fr[mi] = *reinterpret_cast<const f32x4*>(hbuf + noff[mi]);
#else
            asm volatile("" :: "v"(noff[mi]));
#endif
        }
#if ND_BF16_SCHED && !defined(ND_HABL_NOA)
        // pin the interleave: TN MFMAs, then the one LDS read that refills the fragment they consumed (left alone the
        // scheduler sinks all TM reads behind the last MFMA, so the next step opens waiting for LDS)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            __builtin_amdgcn_sched_group_barrier(0x008, TN, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#endif
        ND_PRIO(0);
    };

    // ---- prologue: chunk 0 halo, first weight fragments
#pragma unroll
    for (int k = 0; k < MAXHI; ++k) store_halo_item(k, 0, load_halo_pixel(gpix[k], 0), 0);
#pragma unroll
    for (int d = 0; d < BDIST; ++d) advance_b(b_fr[d]);
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
        const float* hbuf = smem + (ch & 1) * (HP * ROWF);
        const bool halo_next = (ch + 1) < nchunks;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int hp = a_hp[mi];
            a_fr[mi] = *reinterpret_cast<const f32x4*>(hbuf + hp * ROWF + ((kq ^ swz(hp)) << 2));
        }
        if constexpr (TAPS == 9) {
#pragma unroll 1
            for (int dy = 0; dy < 3; ++dy) {
                // next chunk's halo arrives in 3 batches of NBI items (one batch per tap row), fetched at the top of the
                // row and parked in the other LDS buffer after its 6 k-steps (~3000 cycles later)
                int gs[NBI];
#pragma unroll
                for (int i = 0; i < NBI; ++i) {
                    int g = gpix[i];
                    g = (dy == 1) ? gpix[NBI + i] : g;
                    g = (dy == 2) ? gpix[2 * NBI + i] : g;
                    gs[i] = halo_next ? g : -1;
                }
                f32x4 phb[NBI];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int tapoff = dy * HW + dx;
                    const int tapoff_n = (dx < 2) ? tapoff + 1 : ((dy < 2) ? (dy + 1) * HW : 0);
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) {
                        const int st = dx * 2 + kc;                 // 0 .. 5, compile-time: ring slots are static
                        advance_b(b_fr[(st + BDIST) % RING]);
                        if (dx == 0 && kc == 0) {
#pragma unroll
                            for (int i = 0; i < NBI; ++i) phb[i] = load_halo_pixel(gs[i], ch + 1);
                        }
                        int noff[TM];
                        {
                            const int nslot = (((kc + 1) & 1) << 2) | kq;
                            const int toff = (kc == 1) ? tapoff_n : tapoff;
#pragma unroll
                            for (int mi = 0; mi < TM; ++mi) {
                                const int hp = a_hp[mi] + toff;     // (after the last step of a chunk this reads stale but
                                noff[mi] = hp * ROWF + ((nslot ^ swz(hp)) << 2);   //  in-bounds data that is discarded)
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_step(b_fr[st % RING], hbuf, noff);
                    }
                }
                if (halo_next) {
#pragma unroll
                    for (int i = 0; i < NBI; ++i) {
                        const int k = dy * NBI + i;      // dy is a run-time value: the item index is plain arithmetic
                        const int hp = hrow0 + k * (NT / SPR);
                        if (hp < HP) {
                            float* dst = smem + ((ch + 1) & 1) * (HP * ROWF) + hp * ROWF + ((hslot ^ swz(hp)) << 2);
                            *reinterpret_cast<f32x4*>(dst) = gn_xform(phb[i], ch + 1, (vmask >> k) & 1u);
                        }
                    }
                }
            }
        } else {
            // 1x1: KSTEPS k-steps per chunk; the next chunk's rows are fetched one group of NBI per step and parked two
            // steps later (the last groups behind the loop)
            f32x4 ph[KSTEPS][NBI];
            constexpr int SD = 2;      // k-steps between a group's fetch and its LDS store
            const bool second = (ch * NSUB + 1) < p.NC64;          // the chunk's second 64-channel half holds real channels
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                if (ks < 2 || second) {
                    advance_b(b_fr[(ks + BDIST) % RING]);
#pragma unroll
                    for (int i = 0; i < NBI; ++i) ph[ks][i] = load_halo_pixel(halo_next ? gpix[ks * NBI + i] : -1, ch + 1);
                    int noff[TM];
                    {
                        const int nstep = (ks + 1) & (KSTEPS - 1);      // wraps to 0 after the last step: discarded
                        const int nslot = (nstep << 2) | kq;
#pragma unroll
                        for (int mi = 0; mi < TM; ++mi) noff[mi] = a_hp[mi] * ROWF + ((nslot ^ swz(a_hp[mi])) << 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_step(b_fr[ks % RING], hbuf, noff);
                } else {
#pragma unroll
                    for (int i = 0; i < NBI; ++i) ph[ks][i] = load_halo_pixel(halo_next ? gpix[ks * NBI + i] : -1, ch + 1);
                }
                if (ks >= SD && halo_next) {
#pragma unroll
                    for (int i = 0; i < NBI; ++i) store_halo_item((ks - SD) * NBI + i, (ch + 1) & 1, ph[ks - SD][i], ch + 1);
                }
            }
            if (halo_next) {
#pragma unroll
                for (int ks = KSTEPS - SD; ks < KSTEPS; ++ks)
#pragma unroll
                    for (int i = 0; i < NBI; ++i) store_halo_item(ks * NBI + i, (ch + 1) & 1, ph[ks][i], ch + 1);
            }
        }
        // halo hand-over: only LDS traffic has to be complete; the weight prefetch stays in flight across the barrier
#if !defined(ND_HABL_NOBAR)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#endif
    }
#if defined(ND_HABL_NOEPI)
    if (p.N > 0) { if (acc[0][0][0] == 123.456f) static_cast<float*>(p.out)[0] = acc[0][0][1]; return; }
#endif

    // ---- epilogue: C/D of the 16x16 tile: lane = pixel l15, registers = the 4 consecutive output channels 4*kq .. +3
    const bool vec_ok = ((p.ldo & 3) == 0) && (!p.res || (p.ldr & 3) == 0) && (!p.rowbias || (p.ld_rowbias & 3) == 0);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = (wm * TM + mi) * 16 + l15;
        const int li = m >> (p.thl + p.twl);
        const int oy = oy0 + ((m >> p.twl) & (TH - 1));
        const int ox = ox0 + (m & (TW - 1));
        const int img = img0 + li;
        if (img < p.NI && oy < p.H && ox < p.W) {
            const size_t opix = (size_t)(img * p.H + oy) * p.W + ox;
            const float* rb = p.rowbias ? p.rowbias + (size_t)img * p.ld_rowbias : nullptr;
            const __bf16* rr = nullptr;
            if (p.res) {
                const size_t rp = p.res_up ? ((size_t)(img * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) : opix;
                rr = p.res + rp * p.ldr;
            }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int n = n0 + (wn * TN + ni) * 16 + 4 * kq;
                if (n + 3 < p.N && vec_ok) {
                    f32x4 v = acc[mi][ni];
                    if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                    if (rb) v += *reinterpret_cast<const f32x4*>(rb + n);
                    if (rr) {
                        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(rr + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                    }
                    if (p.silu_out) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
                    }
                    if (p.out_f32) {
                        *reinterpret_cast<f32x4*>(static_cast<float*>(p.out) + opix * p.ldo + n) = v;
                    } else {
                        const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                        *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(p.out) + opix * p.ldo + n) = o;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e < p.N) {
                            float v = acc[mi][ni][e];
                            if (p.bias) v += p.bias[n + e];
                            if (rb) v += rb[n + e];
                            if (rr) v += (float)rr[n + e];
                            if (p.silu_out) v = fast_silu(v);
                            if (p.out_f32) static_cast<float*>(p.out)[opix * p.ldo + n + e] = v;
                            else static_cast<__bf16*>(p.out)[opix * p.ldo + n + e] = (__bf16)v;
                        }
                    }
                }
            }
        }
    }
}

// LDS-DMA: 16 bytes per lane from a per-lane global address into LDS at M0 + lane * 16 (gemm_bf16_kernel; conv_bf16w_kernel)
#define ND_GLDS16H(gptr, lptr)                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// (Variants 12 / 13 -- conv_bf16w_kernel, both operands through LDS by LDS-DMA -- gave the same bits and measured EQUAL to
//  the register-streamed form; variant 22 -- gemm_bf16x_kernel, 128 px x 128 ch per wave, one block per CU -- measured
//  0.72-0.93 of gemm_bf16q_kernel.  Their code was removed in round 4 (git history); the numbers stay retired.)

// ------------------------------------------------------------------------------------------------------------
// GEMM-shaped 1x1 form (flat pixel list): out[M][N] = x[M][K] . w[N][K]^T with K of a few hundred.  In the conv-shaped
// 1x1 path above a block lives for 2-8 chunks, each chunk's rows are fetched only one chunk ahead through registers, and
// the waves sit parked 60-70 % of the time (PMC: MFMA pipe busy 0.12-0.18).  Here BOTH operands go through a ring of THREE
// LDS stages filled by LDS-DMA (a 64-channel chunk per stage: 256 pixel rows x 128 bytes, swizzle on the source address,
// + the chunk's 4 k-steps x 4 n-tile weight fragments), so a chunk has two whole chunks of compute (~2000 cycles) to arrive;
// per stage every wave issues exactly 4 + 2 DMAs, which makes the one counted wait per chunk (vmcnt(6): everything but the
// youngest stage has landed) uniform.  8 waves = 4 (pixels) x 2 (channels), wave tile 64 px x 64 ch, block 256 px x 128 ch.
template <int DUMMY>
__global__ void __launch_bounds__(512, 2)
    gemm_bf16_kernel(const ConvArgsH p, const __bf16* zero16) {
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
    const int nchunks = p.NC64;

    auto swz = [](int row) -> int { return (row >> 1) & 7; };

    // ---- A DMA descriptors: piece u = j * 8 + wave (j = 0..3) = rows 8u .. 8u+7; lane -> row 8u + lane/8, physical slot lane%8
    int arow[4], asl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (j * 8 + wave) * 8 + (lane >> 3);
        arow[j] = (m0 + row < M) ? (m0 + row) : -1;
        asl[j] = ((lane & 7) ^ swz(row)) << 3;       // first channel (within the chunk) of the logical slot this lane fills
    }
    // ---- weight DMA descriptors: fragment f = j * 8 + wave (j = 0, 1) = (k-step f / 4, n tile f % 4)
    const __bf16* wsrc[2];
    int wdst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int f = j * 8 + wave;
        const int ks = f >> 2, nl = f & 3;
        int ntile = nblk * 4 + nl;
        if (ntile > p.NT32 - 1) ntile = p.NT32 - 1;
        wsrc[j] = p.w + ((size_t)ntile * 4 + ks) * 512 + lane * 8;
        wdst[j] = A_W + f * 256;
    }
    const size_t c64_stride = (size_t)p.NT32 * 4 * 512;
    auto issue_stage = [&](int ch, int buf) {
        float* st = smem + buf * STAGE_W;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = ch * 64 + asl[j];
            const int g = arow[j];
            const __bf16* src = (c < p.C0) ? (p.x0 + (size_t)(g < 0 ? 0 : g) * p.ldx0 + c)
                                           : (p.x1 + (size_t)(g < 0 ? 0 : g) * p.ldx1 + (c - p.C0));
            src = (g >= 0 && c < Ctot) ? src : zero16;
            ND_GLDS16H(src, st + (j * 8 + wave) * 256);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) ND_GLDS16H(wsrc[j] + (size_t)ch * c64_stride, st + wdst[j]);
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
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) b_fr[0][ni] = *reinterpret_cast<const f32x4*>(st + boff + ni * 256);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) a_fr[0][mi] = *reinterpret_cast<const f32x4*>(st + aoff[mi] + ((lh ^ asw[mi]) << 2));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks < 3) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    b_fr[nxt][ni] = *reinterpret_cast<const f32x4*>(st + boff + ((ks + 1) * 4 + ni) * 256);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
                    a_fr[nxt][mi] = *reinterpret_cast<const f32x4*>(st + aoff[mi] + (((((ks + 1) << 1) | lh) ^ asw[mi]) << 2));
            }
            ND_PRIO(1);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(b_fr[cur][ni]), as_bf16x8(a_fr[cur][mi]),
                                                                          acc[mi][ni], 0, 0, 0);
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
            const __bf16* rr = p.res ? p.res + opix * p.ldr : nullptr;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int n = n0 + (wn * TN + ni) * 32 + 8 * g4 + 4 * lh;
                    if (n + 3 < p.N && vec_ok) {
                        f32x4 v = {acc[mi][ni][4 * g4 + 0], acc[mi][ni][4 * g4 + 1], acc[mi][ni][4 * g4 + 2],
                                   acc[mi][ni][4 * g4 + 3]};
                        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
                        if (rr) {
                            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(rr + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                        }
                        if (p.silu_out) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
                        }
                        if (p.out_f32) {
                            *reinterpret_cast<f32x4*>(static_cast<float*>(p.out) + opix * p.ldo + n) = v;
                        } else {
                            const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                            *reinterpret_cast<bf16x4*>(static_cast<__bf16*>(p.out) + opix * p.ldo + n) = o;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (n + e < p.N) {
                                float v = acc[mi][ni][4 * g4 + e];
                                if (p.bias) v += p.bias[n + e];
                                if (rr) v += (float)rr[n + e];
                                if (p.silu_out) v = fast_silu(v);
                                if (p.out_f32) static_cast<float*>(p.out)[opix * p.ldo + n + e] = v;
                                else static_cast<__bf16*>(p.out)[opix * p.ldo + n + e] = (__bf16)v;
                            }
                        }
                    }
                }
            }
        }
    }
}

// Split-K second pass: out[m][n] = bf16(sum_s ws[s][m][n] + bias[n] + rowbias[img(m)][n] + residual[m][n]) (SiLU last), the
// partials added in split order: deterministic.  One thread per 4 channels.
__global__ void __launch_bounds__(256)
    splitk_reduce_kernel(const float* ws, int S, long ws_stride, long M, int N, const float* bias, const float* rowbias,
                         int ld_rowbias, int hw, const __bf16* res, int ldr, __bf16* out, int ldo, int silu) {
    const int nq = N >> 2;
    const long total = M * nq;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const long m = it / nq;
        const int n = (int)(it - m * nq) << 2;
        f32x4 v = *reinterpret_cast<const f32x4*>(ws + m * N + n);
        for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4*>(ws + (size_t)s * ws_stride + m * N + n);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        if (rowbias) v += *reinterpret_cast<const f32x4*>(rowbias + (m / hw) * ld_rowbias + n);
        if (res) {
            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(res + m * ldr + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
        }
        if (silu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
        }
        const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        *reinterpret_cast<bf16x4*>(out + m * ldo + n) = o;
    }
}

// fp32 OIHW [N][C][k][k] (also Conv1d [N][C][1], Linear [N][C]) -> bf16 fragment order
//   out[((((c64*NT32 + ntile)*taps + tap)*4 + ks)*64 + lane)*8 + j] = bf16(w[n = ntile*32 + (lane&31)][c = c64*64 + ks*16 + (lane>>5)*8 + j][tap])
// zero for n >= N, c >= C and for the padding chunks.
__global__ void pack_conv_weight_bf16_kernel(const float* w, __bf16* out, int N, int C, int taps, int NT32, long total) {
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int j = (int)(it & 7);
        const int lane = (int)((it >> 3) & 63);
        long r = it >> 9;
        const int ks = (int)(r & 3);
        r >>= 2;
        const int tap = (int)(r % taps);
        r /= taps;
        const int ntile = (int)(r % NT32);
        const int c64 = (int)(r / NT32);
        const int n = ntile * 32 + (lane & 31);
        const int c = c64 * 64 + ks * 16 + (lane >> 5) * 8 + j;
        out[it] = (__bf16)((n < N && c < C) ? w[((size_t)n * C + c) * taps + tap] : 0.f);
    }
}

// layout 1 (v_mfma_f32_16x16x32_bf16 fragments):
//   out[((((c64*NT16 + nt16)*taps + tap)*2 + ks)*64 + lane)*8 + j] = bf16(w[n = nt16*16 + (lane&15)][c = c64*64 + ks*32 + (lane>>4)*8 + j][tap])
__global__ void pack_conv_weight_bf16s_kernel(const float* w, __bf16* out, int N, int C, int taps, int NT16, long total) {
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int j = (int)(it & 7);
        const int lane = (int)((it >> 3) & 63);
        long r = it >> 9;
        const int ks = (int)(r & 1);
        r >>= 1;
        const int tap = (int)(r % taps);
        r /= taps;
        const int nt = (int)(r % NT16);
        const int c64 = (int)(r / NT16);
        const int n = nt * 16 + (lane & 15);
        const int c = c64 * 64 + ks * 32 + (lane >> 4) * 8 + j;
        out[it] = (__bf16)((n < N && c < C) ? w[((size_t)n * C + c) * taps + tap] : 0.f);
    }
}

__global__ void f32_to_bf16_rows_kernel(const float* x, int ldx, __bf16* out, int ldo, int C, long rows) {
    // [rows][ldx] fp32 -> [rows][ldo] bf16: channels [0, C) converted, [C, ldo) zero
    const long total = rows * ldo;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const long row = it / ldo;
        const int c = (int)(it - row * ldo);
        out[it] = (__bf16)(c < C ? x[row * ldx + c] : 0.f);
    }
}

// ------------------------------------------------------------------------------------------------------------
struct VariantH {
    int wm, wn, tm, tn;
    int ldsw;       // 1: conv_bf16w_kernel (3x3 only; weights and halo through LDS by LDS-DMA)
    int mf;         // 1: conv_bf16s_kernel (v_mfma_f32_16x16x32_bf16: tm, tn count 16-wide tiles, weights in layout 1)
    int bm() const { return wm * tm * (mf ? 16 : 32); }
    int bn() const { return wn * tn * (mf ? 16 : 32); }
    int nt() const { return wm * wn * 64; }
};

static const VariantH kVariantsH[] = {
    {2, 4, 4, 2, 0, 0},   // 0: 256 x 256, 8 waves
    {2, 2, 4, 2, 0, 0},   // 1: 256 x 128, 4 waves
    {2, 4, 4, 1, 0, 0},   // 2: 256 x 128, 8 waves
    {2, 4, 2, 2, 0, 0},   // 3: 128 x 256, 8 waves
    {2, 2, 2, 2, 0, 0},   // 4: 128 x 128, 4 waves
    {2, 4, 2, 1, 0, 0},   // 5: 128 x 128, 8 waves
    {2, 2, 2, 1, 0, 0},   // 6: 128 x  64, 4 waves
    {2, 2, 1, 1, 0, 0},   // 7:  64 x  64, 4 waves
    // one pixel row of waves: every wave owns its own n tile(s), so no weight fragment is fetched twice by a block
    {1, 8, 8, 1, 0, 0},   // 8: 256 x 256, 8 waves, wave tile 256 px x 32 ch
    {1, 4, 8, 1, 0, 0},   // 9: 256 x 128, 4 waves
    {1, 8, 4, 1, 0, 0},   // 10: 128 x 256, 8 waves
    {1, 4, 4, 2, 0, 0},   // 11: 128 x 256, 4 waves, wave tile 128 px x 64 ch
    // RETIRED (conv_bf16w_kernel: both operands through LDS by LDS-DMA, measured equal; removed in round 4)
    {2, 4, 4, 2, 1, 0},   // 12: 256 x 256, 8 waves
    {2, 4, 2, 2, 1, 0},   // 13: 128 x 256, 8 waves
    // v_mfma_f32_16x16x32_bf16 forms (tm, tn in 16-wide tiles)
    {1, 4, 8, 4, 0, 1},   // 14: 128 x 256, 4 waves, wave tile 128 px x 64 ch
    {1, 4, 8, 2, 0, 1},   // 15: 128 x 128, 4 waves, wave tile 128 px x 32 ch
    {2, 2, 4, 4, 0, 1},   // 16: 128 x 128, 4 waves, wave tile  64 px x 64 ch
    {1, 8, 8, 2, 0, 1},   // 17: 128 x 256, 8 waves
    {2, 4, 8, 2, 0, 1},   // 18: 256 x 128, 8 waves
    {2, 2, 4, 2, 0, 1},   // 19: 128 x  64, 4 waves
    // GEMM-shaped 1x1 (flat pixel lists only): three LDS stages filled by LDS-DMA; coded as ldsw = 2
    {4, 2, 2, 2, 2, 0},   // 20: 256 x 128, 8 waves
    // GEMM-shaped 1x1, two blocks per CU (nd_gemm_bf16_quad.hip): pixel rows through 4 LDS stages by LDS-DMA, weights
    // global -> VGPR; coded as ldsw = 3
    {1, 4, 4, 2, 3, 0},   // 21: 128 x 256, 4 waves
    // RETIRED (gemm_bf16x_kernel: 128 px x 128 ch per wave, one block per CU, measured slower; removed in round 4); ldsw = 4
    {2, 2, 4, 4, 4, 0},   // 22: 256 x 256, 4 waves
    // narrow outputs (the UNet's last convolution, 256 -> 6 channels): one 32-channel n tile per block instead of 64
    {4, 1, 1, 1, 0, 0},   // 23: 128 x  32, 4 waves, wave tile 32 px x 32 ch
};
static constexpr int kNumVariantsH = sizeof(kVariantsH) / sizeof(kVariantsH[0]);

static bool coal_epilogue_enabled() {
    static const int v = [] { const char* e = getenv("ND_BF16_COAL_EPI"); return e ? atoi(e) : 1; }();
    return v != 0;
}

static size_t lds_bytes_h(int taps, int hp) { return (size_t)2 * hp * (taps == 9 ? 128 : 256); }
static size_t lds_bytes_w(const VariantH& V, int hp) {
    return (size_t)2 * ((hp * 8 + 63) / 64) * 1024 + (size_t)2 * 4 * (V.bn() / 32) * 1024 + (size_t)V.nt() / 64 * 1024;
}

static int nbi_of(const VariantH& V, int taps) {
    const int ksteps = V.mf ? 4 : 8;
    return taps == 9 ? (((V.bm() * 25 + 2 * V.nt() - 1) / (2 * V.nt()) + 2) / 3) : ((V.bm() * 16 / V.nt() + ksteps - 1) / ksteps);
}

static bool plan_tiles_h(const VariantH& V, int taps, int NI, int H, int W, TilePlan* best) {
    const int bm = V.bm(), nt = V.nt();
    const int lbm = ilog2(bm);
    bool found = false;
    const int pad = taps == 9 ? 1 : 0;
    const int spr = taps == 9 ? 8 : 16;
    const int maxhi = nbi_of(V, taps) * (taps == 9 ? 3 : (V.mf ? 4 : 8));
    for (int twl = 0; twl <= lbm; ++twl) {
        for (int thl = 0; thl + twl <= lbm; ++thl) {
            const int nibl = lbm - twl - thl;
            const int TW = 1 << twl, TH = 1 << thl, NIB = 1 << nibl;
            const int hp = NIB * (TH + 2 * pad) * (TW + 2 * pad);
            if (V.ldsw >= 2) {
                if (thl != 0 || nibl != 0) continue;                       // a flat run of 256 (128) pixels
            } else if (V.ldsw) {
                if ((hp * 8 + 63) / 64 > 9 * (nt / 64)) continue;        // one 1 KiB halo piece per wave and tap
                if (lds_bytes_w(V, hp) > 160 * 1024) continue;
            } else {
                if ((long)hp * spr > (long)maxhi * nt) continue;
                if (lds_bytes_h(taps, hp) > 160 * 1024) continue;
            }
            TilePlan t;
            t.thl = thl; t.twl = twl; t.nibl = nibl;
            t.tiles_x = (W + TW - 1) / TW;
            t.tiles_y = (H + TH - 1) / TH;
            t.groups = (NI + NIB - 1) / NIB;
            t.hp = hp;
            t.padded = (long)t.tiles_x * t.tiles_y * t.groups * bm;
            if (!found || t.padded < best->padded || (t.padded == best->padded && t.hp < best->hp)) {
                *best = t;
                found = true;
            }
        }
    }
    return found;
}

// cost model used when the caller does not pick a variant: rounds of blocks over the CUs x tile area, derated for
// small register tiles (the Python plan builder overrides it by measuring)
static int select_variant_h(int variant, int taps, int pNI, int pH, int pW, int N, TilePlan* out_tp) {
    int best_v = -1;
    TilePlan best_tp{};
    double best_cost = 0;
    for (int v = 0; v < kNumVariantsH; ++v) {
        if (variant >= 0 && v != variant) continue;
        const VariantH& V = kVariantsH[v];
        if (V.ldsw == 1 || V.ldsw == 4) continue;                 // retired variants (kernels removed)
        if (V.ldsw >= 2 && (taps != 1 || variant < 0 || pNI != 1 || pH != 1)) continue;      // flat 1x1 only, explicit choice only
        if (V.mf && variant < 0) continue;                        // needs the layout-1 weights: explicit choice only
        if (V.mf && V.tm * V.tn >= 32) continue;                  // 128 px x 64 ch wave tile on 16x16 MFMAs: 154 registers spill (10x slower); kept only as an index
        TilePlan tp;
        if (!plan_tiles_h(V, taps, pNI, pH, pW, &tp)) continue;
        const long nblk_n = (N + V.bn() - 1) / V.bn();
        const long nblocks = (long)tp.tiles_x * tp.tiles_y * tp.groups * nblk_n;
        const size_t lds = V.ldsw == 4 ? (size_t)128 * 1024 : V.ldsw == 3 ? (size_t)64 * 1024 : V.ldsw == 2 ? (size_t)3 * 48 * 1024 : (V.ldsw ? lds_bytes_w(V, tp.hp) : lds_bytes_h(taps, tp.hp));
        int per_cu = (int)(160 * 1024 / lds);
        const int by_waves = 8 / (V.nt() / 64) > 0 ? 8 / (V.nt() / 64) : 1;      // two waves per SIMD
        if (per_cu > by_waves) per_cu = by_waves;
        if (per_cu < 1) per_cu = 1;
        const long slots = 256L * per_cu;
        const long rounds = (nblocks + slots - 1) / slots;
        double cost = (double)rounds * per_cu * V.bm() * V.bn();
        const int rt = V.tm * V.tn;
        cost /= (rt >= 8) ? 1.0 : (rt >= 4 ? 0.8 : (rt >= 2 ? 0.55 : 0.35));
        if (best_v < 0 || cost < best_cost * 0.999) {
            best_v = v; best_tp = tp; best_cost = cost;
        }
    }
    if (best_v >= 0) *out_tp = best_tp;
    return best_v;
}

template <int WM, int WN, int TM, int TN, int TAPS, bool STATS = false, bool K16 = false>
static int launch_h(const ConvArgsH& a, int grid, size_t lds, hipStream_t s) {
    auto kern = conv_bf16_kernel<WM, WN, TM, TN, TAPS, STATS, K16>;
    static bool attr_set[kMaxDevices] = {};
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), attr_set, "nd_conv_bf16_nhwc")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid, a.ksplit > 1 ? a.ksplit : 1), dim3(WM * WN * 64), lds, s, a);
    return check_launch("nd_conv_bf16_nhwc");
}


template <int WM, int WN, int TM, int TN, int TAPS>
static int launch_s(const ConvArgsH& a, int grid, size_t lds, hipStream_t s) {
    auto kern = conv_bf16s_kernel<WM, WN, TM, TN, TAPS>;
    static bool attr_set[kMaxDevices] = {};
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), attr_set, "nd_conv_bf16_nhwc")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid, a.ksplit > 1 ? a.ksplit : 1), dim3(WM * WN * 64), lds, s, a);
    return check_launch("nd_conv_bf16_nhwc");
}

template <int TAPS>
static int dispatch_h(int v, const ConvArgsH& a, int grid, size_t lds, hipStream_t s) {
    switch (v) {
        case 0: return launch_h<2, 4, 4, 2, TAPS>(a, grid, lds, s);
        case 1: return launch_h<2, 2, 4, 2, TAPS>(a, grid, lds, s);
        case 2: return launch_h<2, 4, 4, 1, TAPS>(a, grid, lds, s);
        case 3: return launch_h<2, 4, 2, 2, TAPS>(a, grid, lds, s);
        case 4: return launch_h<2, 2, 2, 2, TAPS>(a, grid, lds, s);
        case 5: return launch_h<2, 4, 2, 1, TAPS>(a, grid, lds, s);
        case 6: return launch_h<2, 2, 2, 1, TAPS>(a, grid, lds, s);
        case 7: return launch_h<2, 2, 1, 1, TAPS>(a, grid, lds, s);
        case 8: return launch_h<1, 8, 8, 1, TAPS>(a, grid, lds, s);
        case 9: return launch_h<1, 4, 8, 1, TAPS>(a, grid, lds, s);
        case 10: return launch_h<1, 8, 4, 1, TAPS>(a, grid, lds, s);
        case 11:
            if constexpr (TAPS == 9) {
                if (a.C0 + a.C1 <= 16 && a.ksplit <= 1) return launch_h<1, 4, 4, 2, 9, false, true>(a, grid, lds, s);
            }
            return launch_h<1, 4, 4, 2, TAPS>(a, grid, lds, s);
        case 23: return launch_h<4, 1, 1, 1, TAPS>(a, grid, lds, s);
        case 14: return launch_s<1, 4, 8, 4, TAPS>(a, grid, lds, s);
        case 15: return launch_s<1, 4, 8, 2, TAPS>(a, grid, lds, s);
        case 16: return launch_s<2, 2, 4, 4, TAPS>(a, grid, lds, s);
        case 17: return launch_s<1, 8, 8, 2, TAPS>(a, grid, lds, s);
        case 18: return launch_s<2, 4, 8, 2, TAPS>(a, grid, lds, s);
        case 19: return launch_s<2, 2, 4, 2, TAPS>(a, grid, lds, s);
    }
    set_error("nd_conv_bf16_nhwc: bad variant %d", v);
    return ND_E_ARG;
}

// 3x3 launches that also leave the output's per-channel partial sums behind (ConvArgsH::chstats)
static int dispatch_h_stats(int v, const ConvArgsH& a, int grid, size_t lds, hipStream_t s) {
    switch (v) {
        case 0: return launch_h<2, 4, 4, 2, 9, true>(a, grid, lds, s);
        case 1: return launch_h<2, 2, 4, 2, 9, true>(a, grid, lds, s);
        case 2: return launch_h<2, 4, 4, 1, 9, true>(a, grid, lds, s);
        case 3: return launch_h<2, 4, 2, 2, 9, true>(a, grid, lds, s);
        case 4: return launch_h<2, 2, 2, 2, 9, true>(a, grid, lds, s);
        case 5: return launch_h<2, 4, 2, 1, 9, true>(a, grid, lds, s);
        case 6: return launch_h<2, 2, 2, 1, 9, true>(a, grid, lds, s);
        case 7: return launch_h<2, 2, 1, 1, 9, true>(a, grid, lds, s);
        case 8: return launch_h<1, 8, 8, 1, 9, true>(a, grid, lds, s);
        case 9: return launch_h<1, 4, 8, 1, 9, true>(a, grid, lds, s);
        case 10: return launch_h<1, 8, 4, 1, 9, true>(a, grid, lds, s);
        case 11:
            if (a.C0 + a.C1 <= 16) return launch_h<1, 4, 4, 2, 9, true, true>(a, grid, lds, s);
            return launch_h<1, 4, 4, 2, 9, true>(a, grid, lds, s);
    }
    set_error("nd_conv_bf16_stats_nhwc: variant %d cannot leave statistics behind", v);
    return ND_E_ARG;
}

static inline int nc64_padded(int C) { return ((C + 63) / 64 + 1) & ~1; }   // even: the 1x1 form walks two per chunk

static bool use_flat_h(int taps, int flags, const float* rowbias) {
    return taps == 1 && !(flags & (ND_CONV_IN_UP2X | ND_CONV_RES_UP2X)) && rowbias == nullptr;
}

}  // namespace nd

using namespace nd;

extern "C" int nd_conv_bf16_num_variants(void) { return kNumVariantsH; }

extern "C" int nd_conv_bf16_variant_info(int variant, int* bm, int* bn, int* threads) {
    if (variant < 0 || variant >= kNumVariantsH) return ND_E_ARG;
    if (bm) *bm = kVariantsH[variant].bm();
    if (bn) *bn = kVariantsH[variant].bn();
    if (threads) *threads = kVariantsH[variant].nt();
    return ND_OK;
}

extern "C" int64_t nd_conv_bf16_weight_elems(int N, int C, int ksize) {
    if (N <= 0 || C <= 0 || (ksize != 1 && ksize != 3)) return ND_E_ARG;
    const int64_t nt32 = (N + 31) / 32;
    // + kBf16PadChunks chunks of zeros: the fragment stream runs up to 3 fragments ahead of the last real one, and a
    // 16x16x32 1x1 stream has only 2 fragments per chunk and n tile (nd_weight_stream.h; every kernel static_asserts its ring)
    return (int64_t)(nc64_padded(C) + wstream::kBf16PadChunks) * nt32 * ksize * ksize * 4 * 512;
}

// Upper bound (in elements) of what a launch of `variant` (< 0: any) with `splits` splits over K may read of a packed
// tensor.  Split s shifts the weight pointer by s * kchunks whole chunks and consumes at most kchunks of them (a
// multiple of the LDS chunk), so every split ends on a chunk boundary <= nc64_padded(C); the read-ahead goes on from there.
extern "C" int64_t nd_conv_bf16_max_weight_read(int variant, int N, int C, int ksize, int splits) {
    if (N <= 0 || C <= 0 || (ksize != 1 && ksize != 3) || variant >= kNumVariantsH || splits < 1) return ND_E_ARG;
    const int taps = ksize * ksize;
    const int64_t nt32 = (N + 31) / 32;
    int pad = 0;
    for (int v = 0; v < kNumVariantsH; ++v) {
        if (variant >= 0 && v != variant) continue;
        const VariantH& V = kVariantsH[v];
        int p;
        if (V.ldsw >= 3) p = wstream::pad_chunks(wstream::kBf16GemmQAheadSteps, wstream::bf16_steps(1));      // gemm_bf16q_kernel: one chunk
        else if (V.ldsw == 2) p = 0;                                                              // gemm_bf16_kernel: real chunks only
        else if (V.ldsw == 1) p = wstream::pad_chunks(wstream::kBf16DmaAheadTaps, taps);           // conv_bf16w_kernel
        else if (V.mf) p = wstream::pad_chunks(wstream::bf16s_ring(V.tn, taps) - 1, wstream::bf16s_steps(taps));
        else p = wstream::pad_chunks(wstream::bf16_ring(taps) - 1, wstream::bf16_steps(taps));
        pad = p > pad ? p : pad;
    }
    int consumed = nc64_padded(C);
    if (splits > 1) {                                     // the host's split plan (nd_conv_bf16_splitk_nhwc)
        const int nc64 = (C + 63) / 64, unit = taps == 9 ? 1 : 2;
        int kc = (nc64 + splits - 1) / splits;
        kc = (kc + unit - 1) / unit * unit;
        const int S = (nc64 + kc - 1) / kc;
        const int last = nc64 - (S - 1) * kc;                                   // chunks of the last split ...
        const int end = (S - 1) * kc + (last + unit - 1) / unit * unit;         // ... rounded up to whole LDS chunks by the kernel
        consumed = end > consumed ? end : consumed;
    }
    return (int64_t)(consumed + pad) * nt32 * taps * 4 * 512;
}

extern "C" const char* nd_conv_bf16_variant_name(int variant) {
    if (variant < 0 || variant >= kNumVariantsH) return "";
    const VariantH& v = kVariantsH[variant];
    return v.ldsw == 4 ? "(retired) nd::gemm_bf16x_kernel" : v.ldsw == 3 ? "nd::gemm_bf16q_kernel" : v.ldsw == 2 ? "nd::gemm_bf16_kernel" : v.ldsw == 1 ? "(retired) nd::conv_bf16w_kernel" : v.mf ? "nd::conv_bf16s_kernel"
                                                                                               : "nd::conv_bf16_kernel";
}

extern "C" int nd_conv_bf16_variant_layout(int variant) {
    if (variant < 0) return 0;
    if (variant >= kNumVariantsH) return ND_E_ARG;
    return kVariantsH[variant].mf;
}

extern "C" int nd_repack_conv_weight_bf16(const float* w, void* w_out, int N, int C, int ksize, int layout,
                                          nd_stream_t stream) {
    const char* fn = "nd_repack_conv_weight_bf16";
    ND_REQUIRE(w && w_out && N > 0 && C > 0 && (ksize == 1 || ksize == 3) && (layout == 0 || layout == 1), fn, "bad arguments");
    const long total = (long)nd_conv_bf16_weight_elems(N, C, ksize);
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    if (layout == 1) {
        hipLaunchKernelGGL(pack_conv_weight_bf16s_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
                           static_cast<__bf16*>(w_out), N, C, ksize * ksize, 2 * ((N + 31) / 32), total);
        return check_launch(fn);
    }
    hipLaunchKernelGGL(pack_conv_weight_bf16_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
                       static_cast<__bf16*>(w_out), N, C, ksize * ksize, (N + 31) / 32, total);
    return check_launch(fn);
}

extern "C" int nd_f32_to_bf16_rows(const float* x, int ldx, void* out, int ldo, int C, int64_t rows, nd_stream_t stream) {
    const char* fn = "nd_f32_to_bf16_rows";
    ND_REQUIRE(x && out && C > 0 && ldx >= C && ldo >= C && rows > 0, fn, "bad arguments");
    long g = (rows * ldo + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(f32_to_bf16_rows_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, ldx,
                       static_cast<__bf16*>(out), ldo, C, (long)rows);
    return check_launch(fn);
}

// chstats == nullptr: the plain convolution.  Otherwise (3x3, conv_bf16_kernel variants, one image per block) the launch
// also writes the output's partial per-channel statistics; stats_rows_only: no launch, return the rows per image.
static int conv_bf16_impl(const char* fn, const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                          const void* w, const float* bias, const float* rowbias, int ld_rowbias,
                          const void* residual, int ldr, void* out, int ldo,
                          int NI, int H, int W, int N, int ksize, int flags, int variant,
                          const float* gnA, const float* gnB, int ld_gn, float* chstats, bool stats_rows_only,
                          nd_stream_t stream, int splits = 1, float* workspace = nullptr) {
    ND_REQUIRE(x0 && w && out, fn, "null pointer");
    ND_REQUIRE(ksize == 1 || ksize == 3, fn, "ksize must be 1 or 3");
    ND_REQUIRE(NI > 0 && H > 0 && W > 0 && N > 0 && C0 > 0 && C1 >= 0, fn, "bad shape");
    ND_REQUIRE((C0 & 7) == 0 && (C1 & 7) == 0 && (ldx0 & 7) == 0, fn,
               "bf16 channel counts and strides must be multiples of 8 (16-byte loads)");
    ND_REQUIRE(ldx0 >= C0 && ldo >= N, fn, "stride smaller than channel count");
    ND_REQUIRE(aligned16(x0) && aligned16(w), fn, "x0 / w must be 16-byte aligned");
    if (C1 > 0) {
        ND_REQUIRE(x1 != nullptr && (ldx1 & 7) == 0 && ldx1 >= C1 && aligned16(x1), fn,
                   "two-source input needs an aligned x1 with ldx1 >= C1");
    }
    const int up = (flags & ND_CONV_IN_UP2X) ? 1 : 0;
    const int res_up = (flags & ND_CONV_RES_UP2X) ? 1 : 0;
    if (up || res_up) ND_REQUIRE((H & 1) == 0 && (W & 1) == 0, fn, "2x upsampled read needs even H, W");
    if (flags & ND_CONV_SILU_OUT) ND_REQUIRE(residual == nullptr, fn, "SILU_OUT with a residual is not supported");
    if (residual) ND_REQUIRE(ldr >= N, fn, "ldr < N");
    if (rowbias) ND_REQUIRE(ld_rowbias >= N, fn, "ld_rowbias < N");
    ND_REQUIRE((reinterpret_cast<uintptr_t>(out) & 7u) == 0 && (!residual || (reinterpret_cast<uintptr_t>(residual) & 7u) == 0),
               fn, "out / residual must be 8-byte aligned");
    ND_REQUIRE((long)NI * H * W < (1L << 31) / 2, fn, "too many pixels for 32-bit pixel indices");
    ND_REQUIRE(variant < kNumVariantsH, fn, "bad variant");

    const int taps = ksize * ksize;
    const long M = (long)NI * H * W;
    int pNI = NI, pH = H, pW = W;
    if (use_flat_h(taps, flags, rowbias)) { pNI = 1; pH = 1; pW = (int)M; }

    TilePlan tp{};
    const int v = select_variant_h(variant, taps, pNI, pH, pW, N, &tp);
    if (v < 0) return fail_arg(fn, "no tile variant fits this shape");
    const VariantH& V = kVariantsH[v];
    const bool stats = chstats != nullptr || stats_rows_only;
    if (stats) {
        const bool gemmq = taps == 1 && V.ldsw == 3;
        ND_REQUIRE((taps == 9 && !V.ldsw && !V.mf) || gemmq, fn,
                   "statistics: 3x3 convolutions by the conv_bf16_kernel variants, 1x1 by the two-block GEMM form only");
        if (gemmq) {
            ND_REQUIRE(((long)H * W) % 128 == 0 && N % 256 == 0, fn, "statistics (1x1): H*W % 128 == 0 and N % 256 == 0");
            if (stats_rows_only) return H * W / 128;
        }
        ND_REQUIRE(gemmq || tp.nibl == 0, fn, "statistics need one image per block (H*W >= pixel tile)");
        ND_REQUIRE((N & 3) == 0 && (ldo & 3) == 0 && (!residual || (ldr & 3) == 0) && (!rowbias || (ld_rowbias & 3) == 0) &&
                   !(flags & ND_CONV_OUT_F32), fn, "statistics: N and the strides must be multiples of 4, bf16 output");
        if (stats_rows_only) return tp.tiles_x * tp.tiles_y * V.wm;
    }

    ConvArgsH a;
    a.chstats = chstats;
    a.cs_rows = (taps == 1 && V.ldsw == 3) ? H * W / 128 : tp.tiles_x * tp.tiles_y * V.wm;
    a.ksplit = 1; a.kchunks = 0; a.ws_stride = 0;
    const int nc64 = (C0 + C1 + 63) / 64;
    if (splits > 1) {
        // blocks [.., s] run the channel range of split s and leave raw fp32 partials in the workspace; splitk_reduce_kernel
        // finishes the layer.  Chunk ranges are whole LDS chunks (64 channels for 3x3, 128 for 1x1).
        ND_REQUIRE(workspace != nullptr && !V.ldsw && !chstats && gnA == nullptr, fn,
                   "split-K: needs a workspace; not with the LDS-DMA forms, output statistics or a fused GroupNorm");
        ND_REQUIRE(!(flags & (ND_CONV_IN_UP2X | ND_CONV_RES_UP2X | ND_CONV_OUT_F32)) && (N & 3) == 0 && (ldo & 3) == 0 &&
                   (!residual || (ldr & 3) == 0) && (!rowbias || (ld_rowbias & 3) == 0), fn,
                   "split-K: plain bf16 output, N and strides multiples of 4, no 2x-upsampled reads");
        const int unit = taps == 9 ? 1 : 2;
        int kc = (nc64 + splits - 1) / splits;
        kc = (kc + unit - 1) / unit * unit;
        const int S = (nc64 + kc - 1) / kc;
        ND_REQUIRE(S > 1, fn, "split-K: too few input channels for that many splits");
        a.ksplit = S; a.kchunks = kc; a.ws_stride = M * N;
    }
    a.x0 = static_cast<const __bf16*>(x0);
    a.x1 = (C1 > 0) ? static_cast<const __bf16*>(x1) : a.x0;
    a.w = static_cast<const __bf16*>(w);
    a.bias = bias; a.rowbias = rowbias; a.res = static_cast<const __bf16*>(residual); a.out = out;
    if (a.ksplit > 1) { a.bias = nullptr; a.rowbias = nullptr; a.res = nullptr; a.out = workspace; }
    a.C0 = C0; a.C1 = C1; a.ldx0 = ldx0; a.ldx1 = (C1 > 0) ? ldx1 : ldx0;
    a.NI = pNI; a.H = pH; a.W = pW;
    a.up = up; a.res_up = res_up;
    a.Hs = pH >> up; a.Ws = pW >> up;
    a.N = N; a.ldo = (a.ksplit > 1) ? N : ldo; a.ldr = ldr; a.ld_rowbias = ld_rowbias;
    a.NT32 = (N + 31) / 32;
    a.NC64 = (C0 + C1 + 63) / 64;
    a.thl = tp.thl; a.twl = tp.twl; a.nibl = tp.nibl;
    a.tiles_x = tp.tiles_x; a.tiles_y = tp.tiles_y;
    a.mt = tp.tiles_x * tp.tiles_y * tp.groups;
    a.nt = (N + V.bn() - 1) / V.bn();
    a.ngroup = pick_ngroup(a.nt, (size_t)V.bn() * (C0 + C1) * taps * 2, (size_t)M * (C0 + C1) * 2);
    a.silu_out = (flags & ND_CONV_SILU_OUT) ? 1 : 0;
    a.out_f32 = (flags & ND_CONV_OUT_F32) ? 1 : 0;
    if (a.ksplit > 1) { a.silu_out = 0; a.out_f32 = 1; }
    a.gnA = gnA; a.gnB = gnB; a.ld_gn = ld_gn; a.gn_silu = (flags & ND_CONV_GN_SILU) ? 1 : 0; a.gn_hw = 0;
    if (chstats && taps == 1 && V.ldsw == 3) a.gn_hw = H * W;          // statistics rows of the flat GEMM: image = pixel / (H * W)
    size_t lds_gn = 0;
    if (gnA) {
        // one image per block, so that the block's coefficient table in LDS is that image's
        ND_REQUIRE(gnB != nullptr && ld_gn >= C0 + C1, fn, "fused GroupNorm: bad coefficient arrays");
        ND_REQUIRE(!V.ldsw, fn, "fused GroupNorm: not available in the LDS-DMA variants");
        if (pNI == 1 && pH == 1 && taps == 1 && NI * H * W == pW) {
            ND_REQUIRE(((long)H * W) % V.bm() == 0, fn, "fused GroupNorm (1x1): H*W must be a multiple of the pixel tile");
            a.gn_hw = H * W;
        } else {
            ND_REQUIRE(tp.nibl == 0, fn, "fused GroupNorm needs one image per block (H*W >= pixel tile)");
        }
        const int nsub = taps == 9 ? 1 : 2;
        lds_gn = (size_t)2 * ((a.NC64 + nsub - 1) / nsub) * (64 * nsub) * sizeof(float);
        ND_REQUIRE(lds_bytes_h(taps, tp.hp) + lds_gn <= 160 * 1024, fn, "fused GroupNorm: LDS budget exceeded");
    }
    const int grid = a.mt * a.nt;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (V.ldsw == 4 || V.ldsw == 1) return fail_arg(fn, "retired variant (its kernel was removed; see the git history)");
    if (V.ldsw >= 3) {
        ND_REQUIRE(gnA == nullptr && rowbias == nullptr && !a.out_f32 && !a.silu_out && a.ksplit <= 1 && !a.up && !a.res_up, fn,
                   "the two-block GEMM form takes plain 1x1 convolutions with bf16 output only");
        ND_REQUIRE(M % V.bm() == 0 && N % 256 == 0 && C0 % 64 == 0 && C1 % 64 == 0, fn,
                   "the two-block GEMM form needs M % 128 == 0 (256 for the wide form), N % 256 == 0 and whole 64-channel chunks");
        ND_REQUIRE((ldo & 7) == 0 && (!residual || (ldr & 3) == 0) && (reinterpret_cast<uintptr_t>(out) & 15) == 0, fn,
                   "the two-block GEMM form needs 16-byte aligned output rows");
        ND_REQUIRE((double)M * ldx0 * 2 < 2147483648.0 && (double)M * (C1 ? ldx1 : 0) * 2 < 2147483648.0, fn,
                   "the two-block GEMM form addresses its inputs with 32-bit buffer offsets (< 2 GiB)");
        return launch_gemm_bf16q(a, grid, s);
    }
    if (V.ldsw == 2) {
        ND_REQUIRE(gnA == nullptr && rowbias == nullptr, fn, "the GEMM form takes plain 1x1 convolutions only");
        const __bf16* zero16 = a.w + nd_conv_bf16_weight_elems(N, C0 + C1, ksize) - 8;
        auto kern = gemm_bf16_kernel<0>;
        static bool attr_set[kMaxDevices] = {};
        if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), attr_set, fn)) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), (size_t)3 * 48 * 1024, s, a, zero16);
        return check_launch(fn);
    }
    size_t lds = lds_bytes_h(taps, tp.hp) + lds_gn;
    a.coal = 0;
    if (coal_epilogue_enabled() && !V.mf && V.tn == 2 && !a.out_f32 && a.ksplit <= 1 && N % V.bn() == 0 && (ldo & 7) == 0 &&
        (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        // the staging region must not cost a resident block: two 4-wave blocks per CU stay two
        const size_t stage = (size_t)(V.nt() / 64) * V.tm * 32 * 128;
        const size_t lds2 = lds > stage ? lds : stage;
        const size_t cap = 160 * 1024;
        const size_t want = (V.nt() == 256 && cap / lds >= 2) ? 2 : 1;
        if (lds2 <= cap && cap / lds2 >= want) {
            a.coal = 1;
            lds = lds2;
        }
    }
    if (chstats) return dispatch_h_stats(v, a, grid, lds, s);
    const int rc = (taps == 9) ? dispatch_h<9>(v, a, grid, lds, s) : dispatch_h<1>(v, a, grid, lds, s);
    if (rc != ND_OK || a.ksplit <= 1) return rc;
    const long quads = M * (N >> 2);
    long rg = (quads + 255) / 256;
    if (rg > 4096) rg = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)rg), dim3(256), 0, s, workspace, a.ksplit, a.ws_stride, M, N, bias, rowbias,
                       ld_rowbias, H * W, static_cast<const __bf16*>(residual), ldr, static_cast<__bf16*>(out), ldo,
                       (flags & ND_CONV_SILU_OUT) ? 1 : 0);
    return check_launch(fn);
}

extern "C" int nd_conv_bf16_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                                 const void* w, const float* bias, const float* rowbias, int ld_rowbias,
                                 const void* residual, int ldr, void* out, int ldo,
                                 int NI, int H, int W, int N, int ksize, int flags, int variant,
                                 const float* gnA, const float* gnB, int ld_gn, nd_stream_t stream) {
    return conv_bf16_impl("nd_conv_bf16_nhwc", x0, C0, ldx0, x1, C1, ldx1, w, bias, rowbias, ld_rowbias, residual, ldr, out, ldo,
                          NI, H, W, N, ksize, flags, variant, gnA, gnB, ld_gn, nullptr, false, stream);
}

extern "C" int nd_conv_bf16_splitk_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                                        const void* w, const float* bias, const float* rowbias, int ld_rowbias,
                                        const void* residual, int ldr, void* out, int ldo,
                                        int NI, int H, int W, int N, int ksize, int flags, int variant, int splits,
                                        float* workspace, nd_stream_t stream) {
    const char* fn = "nd_conv_bf16_splitk_nhwc";
    ND_REQUIRE(splits >= 2 && splits <= 16 && workspace != nullptr && variant >= 0, fn,
               "2..16 splits, a workspace of nd_conv_bf16_splitk_workspace_floats() floats, and a named tile variant");
    return conv_bf16_impl(fn, x0, C0, ldx0, x1, C1, ldx1, w, bias, rowbias, ld_rowbias, residual, ldr, out, ldo,
                          NI, H, W, N, ksize, flags, variant, nullptr, nullptr, 0, nullptr, false, stream, splits, workspace);
}

// fp32 words nd_conv_bf16_splitk_nhwc writes to (and its reduce kernel reads from) `workspace`: one [NI*H*W][N] slab
// of raw accumulators per split actually used -- fewer than asked for when the layer has fewer LDS chunks
extern "C" int64_t nd_conv_bf16_splitk_workspace_floats(int NI, int H, int W, int N, int C, int ksize, int splits) {
    if (NI <= 0 || H <= 0 || W <= 0 || N <= 0 || C <= 0 || (ksize != 1 && ksize != 3) || splits < 2 || splits > 16) return ND_E_ARG;
    const int nc64 = (C + 63) / 64, unit = ksize == 3 ? 1 : 2;
    int kc = (nc64 + splits - 1) / splits;
    kc = (kc + unit - 1) / unit * unit;
    const int S = (nc64 + kc - 1) / kc;          // the split plan of conv_bf16_impl
    if (S < 2) return ND_E_ARG;
    return (int64_t)S * NI * H * W * N;
}

extern "C" int nd_conv_bf16_stats_rows(int NI, int H, int W, int N, int variant) {
    const char* fn = "nd_conv_bf16_stats_rows";
    ND_REQUIRE(NI > 0 && H > 0 && W > 0 && N > 0 && variant >= 0 && variant < kNumVariantsH, fn, "bad arguments");
    const VariantH& V = kVariantsH[variant];
    if (V.ldsw == 3) return (((long)H * W) % 128 == 0 && N % 256 == 0) ? H * W / 128 : 0;      // gemm_bf16q_kernel (1x1): one row per block
    if (V.ldsw || V.mf || (N & 3)) return 0;
    if (V.wn == 1 && V.tn == 1) return 0;          // the narrow-output form has no statistics instantiation
    TilePlan tp{};
    if (select_variant_h(variant, 9, NI, H, W, N, &tp) < 0 || tp.nibl != 0) return 0;
    return tp.tiles_x * tp.tiles_y * V.wm;
}

extern "C" int nd_conv3x3_bf16_stats_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                                          const void* w, const float* bias, const float* rowbias, int ld_rowbias,
                                          const void* residual, int ldr, void* out, int ldo,
                                          int NI, int H, int W, int N, int flags, int variant,
                                          const float* gnA, const float* gnB, int ld_gn, float* chstats, nd_stream_t stream) {
    const char* fn = "nd_conv3x3_bf16_stats_nhwc";
    ND_REQUIRE(chstats != nullptr && variant >= 0, fn, "chstats is null / the tile variant must be named");
    return conv_bf16_impl(fn, x0, C0, ldx0, x1, C1, ldx1, w, bias, rowbias, ld_rowbias, residual, ldr, out, ldo,
                          NI, H, W, N, 3, flags, variant, gnA, gnB, ld_gn, chstats, false, stream);
}

extern "C" int nd_conv1x1_bf16_stats_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                                          const void* w, const float* bias, const float* rowbias, int ld_rowbias,
                                          const void* residual, int ldr, void* out, int ldo,
                                          int NI, int H, int W, int N, int flags, int variant,
                                          const float* gnA, const float* gnB, int ld_gn, float* chstats, nd_stream_t stream) {
    const char* fn = "nd_conv1x1_bf16_stats_nhwc";
    ND_REQUIRE(chstats != nullptr && variant >= 0, fn, "chstats is null / the tile variant must be named");
    return conv_bf16_impl(fn, x0, C0, ldx0, x1, C1, ldx1, w, bias, rowbias, ld_rowbias, residual, ldr, out, ldo,
                          NI, H, W, N, 1, flags, variant, gnA, gnB, ld_gn, chstats, false, stream);
}


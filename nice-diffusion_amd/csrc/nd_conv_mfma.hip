// K5/K6: 3x3 (stride 1, pad 1) and 1x1 convolution / linear as an implicit GEMM on the fp32 matrix cores.
//
// Replaces nn.Conv2d / nn.Conv1d(k=1) / nn.Linear calls of the reference (model.py:72,169,173,177,180,247,253,
// 349-351,367,448) plus the ops fused around them: torch.cat of the skip connection (model.py:474, two-source
// input), bias, the per-image timestep-embedding add (model.py:205), the residual add (model.py:211,291) and
// the nearest-2x upsampling in front of a conv (model.py:77-79, ND_CONV_IN_UP2X).
//
// Design for gfx950 (wave64, v_mfma_f32_32x32x2_f32 = exact fp32 FMA chains at the fp32 vector rate):
//   GEMM view   M = output pixels, N = output channels, K = taps x input channels.
//   block       WM x WN waves; wave tile = TM x TN MFMA tiles of 32x32; BM = 32*TM*WM pixels, BN = 32*TN*WN channels.
//               The BM pixels are NIB images x (TH x TW) spatial tile.
//   A operand   activations.  K is walked in chunks of 32*NSUB input channels.  Per chunk the input halo tile
//               ((TH+2)x(TW+2) pixels x chunk channels) is staged in LDS ONCE and reused by all 9 taps: the A operand
//               of tap (dy,dx) is the same LDS image read at a shifted pixel, so activations cross L2->LDS ~1.3x
//               instead of 9x.  Lane (i = lane&31, h = lane>>5) reads 16 bytes = k in {8kc+4h..+3} of pixel row i with
//               one ds_read_b128 and feeds them to 4 consecutive MFMAs.  16-byte slots are XOR-swizzled so a 16-lane
//               read group covers 16 distinct slots of the 256-byte bank row.
//   B operand   weights, pre-packed ONCE into MFMA-fragment order [c32][n tile][tap][kc][lane][4]: every B fragment is
//               one fully coalesced 1 KiB global_load_dwordx4 straight into VGPRs (served by L1/L2: co-resident
//               waves stream the same tiles), prefetched one k-step ahead.  Weights never touch LDS, so the only
//               workgroup barrier is the halo hand-over once per chunk (= every 9 taps x 4 k-steps), not per tap.
//   pipeline    next chunk's halo is loaded global->registers one item per tap and parked in the other LDS buffer
//               behind the MFMAs of that tap.
//   grid        one block per (m tile, n tile), remapped so that each XCD gets a contiguous, n-major range of
//               tiles (blocks resident on one XCD stream the same weight fragments through that XCD's L2).
#include "nd_conv_common.h"

namespace nd {

template <int WM, int WN, int TM, int TN, int TAPS, int OCC>
__global__ void __launch_bounds__(WM* WN * 64, OCC)
    conv_mfma_kernel(const ConvArgs pin) {
    ConvArgs p = pin;
    if (pin.ksplit > 1) split_k_args_f32(p, blockIdx.y, TAPS);
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int PAD = (TAPS == 9) ? 1 : 0;
    constexpr int NSUB = (TAPS == 9) ? 1 : 2;          // 32-channel sub-chunks per LDS chunk
    constexpr int SPR = 8 * NSUB;                      // 16-byte slots per halo pixel row
    constexpr int ROWF = 32 * NSUB;                    // floats per halo pixel row
    constexpr int STEPS = TAPS * 4;                    // k-steps (8 channels each) per 32-channel chunk
    constexpr int MAXHI = (TAPS == 9) ? 9 : (BM * SPR / NT);   // halo float4 items per thread per chunk
    static_assert(TAPS == 9 || MAXHI <= NSUB * 4, "one halo item per k-step");
    static_assert(NT % SPR == 0, "");

    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 x [HP][ROWF]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave - wm * WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    // ---- XCD-aware block -> tile map: XCD x (= blockIdx % 8 under round-robin dispatch) gets a contiguous range.
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
    const int HPI = HH * HW;            // halo pixels per image
    const int HP = HPI << p.nibl;       // halo pixels per block
    const int img0 = ig << p.nibl, oy0 = ty << p.thl, ox0 = tx << p.twl;
    const int n0 = nblk * BN;

    // ---- halo descriptors: source pixel index of each halo item (or -1 = zero fill)
    const int hslot = tid % SPR;          // logical 16-byte slot inside the chunk row
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
    const int nchunks = (p.NC32 + NSUB - 1) / NSUB;

    auto swz = [](int hp) -> int { return (SPR == 8) ? ((hp >> 1) & 7) : (hp & 15); };
    // g = source pixel (or -1); one predicated 16-byte load.  Callers with a run-time item index pick g with a
    // select chain over the register array: a load under a per-item branch would make hipcc wait vmcnt(0) per item.
    // fused GroupNorm: this thread always loads the same 4 channels of a chunk, so one coefficient pair per chunk
    const int gimg = (p.gn_hw > 0) ? (ox0 / p.gn_hw) : img0;
    auto load_gn = [&](int ch, f32x4& cA, f32x4& cB) {
        cA = f32x4{1.f, 1.f, 1.f, 1.f};
        cB = f32x4{0.f, 0.f, 0.f, 0.f};
        const int c = ch * ROWF + (hslot << 2);
        if (p.gnA && c < Ctot) {
            cA = *reinterpret_cast<const f32x4*>(p.gnA + (size_t)gimg * p.ld_gn + c);
            cB = *reinterpret_cast<const f32x4*>(p.gnB + (size_t)gimg * p.ld_gn + c);
        }
    };
    f32x4 gA, gB;            // coefficients of the chunk whose halo is being fetched
    auto load_halo_pixel = [&](int g, int ch) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int c = ch * ROWF + (hslot << 2);
        if (g >= 0 && c < Ctot) {
            const float* src = (c < p.C0) ? (p.x0 + (size_t)g * p.ldx0 + c) : (p.x1 + (size_t)g * p.ldx1 + (c - p.C0));
            v = *reinterpret_cast<const f32x4*>(src);
            if (p.gnA) {
                v = v * gA + gB;
                if (p.gn_silu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
                }
            }
        }
        return v;
    };
    auto load_halo_item = [&](int k, int ch) -> f32x4 { return load_halo_pixel(gpix[k], ch); };
    auto store_halo_item = [&](int k, int buf, f32x4 v) {
        const int hp = hrow0 + k * (NT / SPR);
        if (hp < HP) {
            float* dst = smem + buf * (HP * ROWF) + hp * ROWF + ((hslot ^ swz(hp)) << 2);
            *reinterpret_cast<f32x4*>(dst) = v;
        }
    };

    // ---- per-lane operand rows
    int a_hp[TM];     // halo pixel of tap (0,0) for this lane's A row of MFMA tile mi
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = (wm * TM + mi) * 32 + l31;
        const int li = m >> (p.thl + p.twl);
        const int py = (m >> p.twl) & (TH - 1);
        const int px = m & (TW - 1);
        a_hp[mi] = li * HPI + py * HW + px;
    }
    // B fragment stream of n tile ni: [c32][n tile][step][lane][4]; one fragment = 256 floats
    const float* bp[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        int ntile = nblk * (BN / 32) + wn * TN + ni;
        if (ntile > p.NT32 - 1) ntile = p.NT32 - 1;      // N tail: results are discarded in the epilogue
        bp[ni] = p.w + (size_t)ntile * (STEPS * 256) + lane * 4;
    }
    const size_t c32_jump = (size_t)(p.NT32 - 1) * (STEPS * 256);   // from the end of one c32 stream to the next
    int ld_in_c32 = 0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    // operand registers: A ping-pong by k-step parity; B ring of 4 (one per k-step of a tap), fetched BDIST steps ahead
    constexpr int BDIST = wstream::f32_conv_ahead(TAPS);     // measured: deeper prefetch only pays for the short-K 1x1 form
    static_assert(wstream::pad_chunks(BDIST, TAPS * 4) <= wstream::kF32PadChunks, "weight read-ahead exceeds the packer's zero padding");
    f32x4 a_pp[2][TM], b_pp[4][TN];
    // the packed buffer ends in wstream::kF32PadChunks zero chunks: the static_assert above is what lets the stream run ahead
    auto advance_b = [&](f32x4 (&dst)[TN]) {
#if defined(ND_ABL_NOB)
        (void)dst;     // timing-only ablation: B operands stay whatever the registers hold
#else
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            dst[ni] = *reinterpret_cast<const f32x4*>(bp[ni]);
            bp[ni] += 256;
        }
#endif
        if (++ld_in_c32 == STEPS) {
            ld_in_c32 = 0;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) bp[ni] += c32_jump;
        }
    };

    // ---- prologue: chunk 0 halo, first B fragments
    load_gn(0, gA, gB);
#pragma unroll
    for (int k = 0; k < MAXHI; ++k) store_halo_item(k, 0, load_halo_item(k, 0));
#pragma unroll
    for (int d = 0; d < BDIST; ++d) advance_b(b_pp[d]);
    __syncthreads();

    for (int ch = 0; ch < nchunks; ++ch) {
        const float* hbuf = smem + (ch & 1) * (HP * ROWF);
        const bool halo_next = (ch + 1) < nchunks;
        load_gn(ch + 1, gA, gB);                 // every halo fetch of this iteration is for chunk ch + 1
        if constexpr (TAPS == 9) {
            // A and B fragments are fetched one k-step ahead into the other half of a ping-pong register pair
            // (4 k-steps per tap = even, so the roles are compile-time constants and no copies are needed).
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int hp = a_hp[mi];
                a_pp[0][mi] = *reinterpret_cast<const f32x4*>(hbuf + hp * ROWF + ((lh ^ swz(hp)) << 2));
            }
#pragma unroll 1
            for (int dy = 0; dy < 3; ++dy) {
                // Next chunk's halo arrives in 3 batches of 3 items (one batch per tap row).  vmcnt retires in order,
                // so every wait for a B fragment also waits for any older halo load (often an HBM miss): batching
                // exposes that latency 3x per chunk instead of 9x, and the batch is issued behind the B loads of
                // its step so that it has two k-steps of flight before anything younger is waited for.
                int gs[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    int g = gpix[i];
                    g = (dy == 1) ? gpix[3 + i] : g;
                    g = (dy == 2) ? gpix[6 + i] : g;
                    gs[i] = halo_next ? g : -1;
                }
                f32x4 phb[3];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int tapoff = dy * HW + dx;
                    const int tapoff_n = (dx < 2) ? tapoff + 1 : ((dy < 2) ? (dy + 1) * HW : 0);
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc) {
                        const int cur = kc & 1, nxt = cur ^ 1;
                        advance_b(b_pp[(kc + BDIST) & 3]);          // fragments BDIST k-steps ahead
                        if (dx == 0 && kc == 0) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
#if defined(ND_ABL_NOHALO)
                                phb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                                asm volatile("" :: "v"(gs[i]));
#else
                                phb[i] = load_halo_pixel(gs[i], ch + 1);
#endif
                            }
                        }
                        {
                            const int nslot = (((kc + 1) & 3) << 1) | lh;
                            const int noff = (kc == 3) ? tapoff_n : tapoff;
#pragma unroll
                            for (int mi = 0; mi < TM; ++mi) {
                                const int hp = a_hp[mi] + noff;     // (after the last step of a chunk this reads stale but
#if !defined(ND_ABL_NOA)
                                a_pp[nxt][mi] = *reinterpret_cast<const f32x4*>(hbuf + hp * ROWF + ((nslot ^ swz(hp)) << 2));
#else
                                asm volatile("" :: "v"(hp));
#endif
                            }                                       //  in-bounds data that is discarded)
                        }
                        __builtin_amdgcn_sched_barrier(0);         // keep the loads in front of this step's MFMAs
                        ND_PRIO(1);
#if defined(ND_MFMA_INTERLEAVED)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                                for (int ni = 0; ni < TN; ++ni)
                                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_pp[kc][ni][j], a_pp[cur][mi][j], acc[mi][ni], 0, 0, 0);
#else
#pragma unroll
                        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_pp[kc][ni][j], a_pp[cur][mi][j], acc[mi][ni], 0, 0, 0);
#endif
                        ND_PRIO(0);
                    }
                }
#if !defined(ND_ABL_NOHALO)
                if (halo_next) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) store_halo_item(dy * 3 + i, (ch + 1) & 1, phb[i]);
                }
#endif
            }
        } else {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int hp = a_hp[mi];
                a_pp[0][mi] = *reinterpret_cast<const f32x4*>(hbuf + hp * ROWF + ((lh ^ swz(hp)) << 2));
            }
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                if (ch * NSUB + sub < p.NC32) {
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc) {
                        const int item = sub * 4 + kc;
                        const int cur = kc & 1, nxt = cur ^ 1;
                        f32x4 ph = {0.f, 0.f, 0.f, 0.f};
#if !defined(ND_ABL_NOHALO)
                        if (item < MAXHI && halo_next) ph = load_halo_item(item < MAXHI ? item : 0, ch + 1);
#endif
                        advance_b(b_pp[(kc + BDIST) & 3]);
                        {
                            const int nstep = (item + 1) & (NSUB * 4 - 1);      // wraps to 0 after the last step: discarded
                            const int nslot = (nstep << 1) | lh;
#pragma unroll
                            for (int mi = 0; mi < TM; ++mi) {
                                const int hp = a_hp[mi];
#if !defined(ND_ABL_NOA)
                                a_pp[nxt][mi] = *reinterpret_cast<const f32x4*>(hbuf + hp * ROWF + ((nslot ^ swz(hp)) << 2));
#else
                                asm volatile("" :: "v"(hp));
#endif
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        ND_PRIO(1);
#if defined(ND_MFMA_INTERLEAVED)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                                for (int ni = 0; ni < TN; ++ni)
                                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_pp[kc][ni][j], a_pp[cur][mi][j], acc[mi][ni], 0, 0, 0);
#else
#pragma unroll
                        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_pp[kc][ni][j], a_pp[cur][mi][j], acc[mi][ni], 0, 0, 0);
#endif
                        ND_PRIO(0);
#if !defined(ND_ABL_NOHALO)
                        if (item < MAXHI && halo_next) store_halo_item(item < MAXHI ? item : 0, (ch + 1) & 1, ph);
#endif
                    }
                }
            }
        }
        // halo hand-over: only LDS traffic has to be complete; the B prefetch stays in flight across the barrier
#if !defined(ND_ABL_NOBARRIER)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#endif
    }
#if defined(ND_ABL_NOEPI)
    if (p.N > 0) { if (acc[0][0][0] == 123.456f) p.out[0] = acc[0][0][1]; return; }
#endif

    // ---- epilogue.  The MFMAs were issued as D^T = W . X^T (weights as the A operand), so in the C/D layout
    //      (col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) a lane owns ONE pixel and 4 consecutive output
    //      channels per register group: 16-byte stores / residual loads instead of 4-byte ones (the epilogue of a
    //      short-K 1x1 GEMM is store-issue bound otherwise).
    const bool vec_ok = ((p.ldo & 3) == 0) && (!p.res || (p.ldr & 3) == 0) && (!p.rowbias || (p.ld_rowbias & 3) == 0);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = (wm * TM + mi) * 32 + l31;
        const int li = m >> (p.thl + p.twl);
        const int oy = oy0 + ((m >> p.twl) & (TH - 1));
        const int ox = ox0 + (m & (TW - 1));
        const int img = img0 + li;
        if (img < p.NI && oy < p.H && ox < p.W) {
            float* orow = p.out + ((size_t)(img * p.H + oy) * p.W + ox) * p.ldo;
            const float* rb = p.rowbias ? p.rowbias + (size_t)img * p.ld_rowbias : nullptr;
            const float* rr = nullptr;
            if (p.res) {
                const size_t rp = p.res_up ? ((size_t)(img * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1))
                                           : ((size_t)(img * p.H + oy) * p.W + ox);
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
                        if (rr) v += *reinterpret_cast<const f32x4*>(rr + n);
                        if (p.silu_out) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
                        }
                        *reinterpret_cast<f32x4*>(orow + n) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (n + e < p.N) {
                                float v = acc[mi][ni][4 * g4 + e];
                                if (p.bias) v += p.bias[n + e];
                                if (rb) v += rb[n + e];
                                if (rr) v += rr[n + e];
                                if (p.silu_out) v = fast_silu(v);
                                orow[n + e] = v;
                            }
                        }
                    }
                }
            }
        }
    }
}

// Split-K second pass (nd_conv_splitk_nhwc): out[m][n] = sum_s ws[s][m][n] + bias[n] + rowbias[img(m)][n] + residual[m][n]
// (SiLU last), the partials added in split order and then in the one-pass kernel's association: deterministic.  One
// thread per 4 channels.
__global__ void __launch_bounds__(256)
    splitk_reduce_f32_kernel(const float* ws, int S, long ws_stride, long M, int N, const float* bias, const float* rowbias,
                             int ld_rowbias, int hw, const float* res, int ldr, float* out, int ldo, int silu) {
    const int nq = N >> 2;
    const long total = M * nq;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const long m = it / nq;
        const int n = (int)(it - m * nq) << 2;
        f32x4 v = *reinterpret_cast<const f32x4*>(ws + m * N + n);
        for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4*>(ws + (size_t)s * ws_stride + m * N + n);
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        if (rowbias) v += *reinterpret_cast<const f32x4*>(rowbias + (m / hw) * ld_rowbias + n);
        if (res) v += *reinterpret_cast<const f32x4*>(res + m * ldr + n);
        if (silu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
        }
        *reinterpret_cast<f32x4*>(out + m * ldo + n) = v;
    }
}

// The same reduction that also leaves the per-channel partial statistics of what it stores behind (the layout the
// convolutions' epilogues write: chstats [NI][hw / kSplitkStatsPixels][sum | sum of squares][N], fp32, every entry written):
// one block per run of kSplitkStatsPixels pixels of an image, one thread per 4 channels walking the run, so a thread's sums ARE
// the row -- no cross-thread step, a fixed order.  A layer run split over K keeps its output statistics this way instead of
// costing the norms that read it a pass over the tensor (round 5: the 16x16 layers of configs[1] on conv_wf4_kernel).
__global__ void __launch_bounds__(256)
    splitk_reduce_stats_f32_kernel(const float* ws, int S, long ws_stride, int hw, int N, const float* bias, const float* rowbias,
                                   int ld_rowbias, const float* res, int ldr, float* out, int ldo, int silu, float* chstats) {
    const int rpi = hw / kSplitkStatsPixels;          // rows (= blocks) per image
    const int img = blockIdx.x / rpi, rb = blockIdx.x - img * rpi;
    const long m0 = (long)img * hw + (long)rb * kSplitkStatsPixels;
    for (int n = threadIdx.x << 2; n < N; n += blockDim.x << 2) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n);
        if (rowbias) {
            const f32x4 rbv = *reinterpret_cast<const f32x4*>(rowbias + (size_t)img * ld_rowbias + n);
            // (the association of splitk_reduce_f32_kernel: ((sum + bias) + rowbias) + residual)
            f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int i = 0; i < kSplitkStatsPixels; ++i) {
                const long m = m0 + i;
                f32x4 v = *reinterpret_cast<const f32x4*>(ws + m * N + n);
                for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4*>(ws + (size_t)s * ws_stride + m * N + n);
                if (bias) v += bv;
                v += rbv;
                if (res) v += *reinterpret_cast<const f32x4*>(res + m * ldr + n);
                if (silu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
                }
                *reinterpret_cast<f32x4*>(out + m * ldo + n) = v;
                ssum += v;
                ssq += v * v;
            }
            float* ps = chstats + (((size_t)img * rpi + rb) * 2) * N + n;
            *reinterpret_cast<f32x4*>(ps) = ssum;
            *reinterpret_cast<f32x4*>(ps + N) = ssq;
        } else {
            f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int i = 0; i < kSplitkStatsPixels; ++i) {
                const long m = m0 + i;
                f32x4 v = *reinterpret_cast<const f32x4*>(ws + m * N + n);
                for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4*>(ws + (size_t)s * ws_stride + m * N + n);
                if (bias) v += bv;
                if (res) v += *reinterpret_cast<const f32x4*>(res + m * ldr + n);
                if (silu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
                }
                *reinterpret_cast<f32x4*>(out + m * ldo + n) = v;
                ssum += v;
                ssq += v * v;
            }
            float* ps = chstats + (((size_t)img * rpi + rb) * 2) * N + n;
            *reinterpret_cast<f32x4*>(ps) = ssum;
            *reinterpret_cast<f32x4*>(ps + N) = ssq;
        }
    }
}

int launch_splitk_reduce_stats_f32(const float* ws, int S, long ws_stride, int NI, int hw, int N, const float* bias, const float* rowbias,
                                   int ld_rowbias, const float* res, int ldr, float* out, int ldo, int silu, float* chstats, hipStream_t s) {
    const int threads = (N >> 2) >= 256 ? 256 : (((N >> 2) + 63) / 64) * 64;
    hipLaunchKernelGGL(splitk_reduce_stats_f32_kernel, dim3(NI * (hw / kSplitkStatsPixels)), dim3(threads), 0, s, ws, S, ws_stride, hw, N,
                       bias, rowbias, ld_rowbias, res, ldr, out, ldo, silu, chstats);
    return check_launch("split-K reduce + statistics");
}

int launch_splitk_reduce_f32(const float* ws, int S, long ws_stride, long M, int N, const float* bias, const float* rowbias,
                             int ld_rowbias, int hw, const float* res, int ldr, float* out, int ldo, int silu, hipStream_t s) {
    const long quads = M * (N >> 2);
    long rg = (quads + 255) / 256;
    if (rg > 4096) rg = 4096;
    hipLaunchKernelGGL(splitk_reduce_f32_kernel, dim3((int)rg), dim3(256), 0, s, ws, S, ws_stride, M, N, bias, rowbias, ld_rowbias, hw,
                       res, ldr, out, ldo, silu);
    return check_launch("split-K reduce");
}

// ------------------------------------------------------------------------------------------------------------
// 1x1 "stream" form (flat pixel list only): a 1x1 convolution is a plain GEMM, and with K = Cin of a few hundred a
// block lives for ~12 chunks -- too short to amortise the LDS staging + barrier per chunk of the kernel above.  Here
// BOTH operands go straight from global memory into MFMA fragments: lane (i = lane&31, h = lane>>5) loads the 16 bytes
// k in {8s+4h..+3} of ITS pixel row (the fragment order the packed weights already use), so there is no LDS, no
// barrier, and waves run fully decoupled with a 3-step-deep register ring hiding L2 latency.  The 4 k-steps of a
// 32-channel chunk touch the same 128-byte line of a pixel row back to back (served by L1 after the first).
template <int WM, int WN, int TM, int TN, int OCC, bool GN>
__global__ void __launch_bounds__(WM* WN * 64, GN ? 2 : OCC)
    gemm_stream_kernel(const ConvArgs p) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int D = wstream::kF32StreamAhead;        // prefetch distance in k-steps (ring of 4)
    static_assert(wstream::pad_chunks(D, 4) <= wstream::kF32PadChunks, "weight read-ahead exceeds the packer's zero padding");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN;
    const int wn = wave - wm * WN;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    const int total = gridDim.x;
    const int q = total >> 3, r = total & 7, xcd = blockIdx.x & 7;
    const int idp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    int mblk, nblk;
    tile_of(idp, p.mt, p.nt, p.ngroup, mblk, nblk);
    const int M = p.W;                                  // flat pixel list
    const int m0 = mblk * BM;
    const int n0 = nblk * BN;
    const int Ctot = p.C0 + p.C1;

    // per-lane pixel rows (clamped: rows past M compute garbage that the epilogue drops)
    const float* r0[TM];
    const float* r1[TM];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        int m = m0 + (wm * TM + mi) * 32 + l31;
        m = m < M ? m : M - 1;
        r0[mi] = p.x0 + (size_t)m * p.ldx0 + 4 * lh;
        r1[mi] = p.x1 + (size_t)m * p.ldx1 + 4 * lh - p.C0;
    }
    const float* bp[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        int ntile = nblk * (BN / 32) + wn * TN + ni;
        if (ntile > p.NT32 - 1) ntile = p.NT32 - 1;
        bp[ni] = p.w + (size_t)ntile * (4 * 256) + lane * 4;
    }
    const size_t c32_jump = (size_t)(p.NT32 - 1) * (4 * 256);
    const float* ga = nullptr;
    const float* gb = nullptr;
    if constexpr (GN) {
        const int gimg = m0 / p.gn_hw;
        ga = p.gnA + (size_t)gimg * p.ld_gn + 4 * lh;
        gb = p.gnB + (size_t)gimg * p.ld_gn + 4 * lh;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    f32x4 xa[4][TM], wb[4][TN], ca[4], cb[4];
    int cnext = 0;          // first channel of the next k-step to fetch
    int ld_in_c32 = 0;
    auto fetch = [&](int slot) {
        const int c = cnext + 4 * lh;
        const bool ok = c < Ctot;          // also keeps the run-ahead past the last chunk inside the row
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const float* src = (c < p.C0) ? (r0[mi] + cnext) : (r1[mi] + cnext);
            src = ok ? src : r0[mi];
            xa[slot][mi] = *reinterpret_cast<const f32x4*>(src);
        }
        if constexpr (GN) {
            const int cc = ok ? cnext : 0;
            ca[slot] = *reinterpret_cast<const f32x4*>(ga + cc);
            cb[slot] = *reinterpret_cast<const f32x4*>(gb + cc);
        }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            wb[slot][ni] = *reinterpret_cast<const f32x4*>(bp[ni]);
            bp[ni] += 256;
        }
        if (++ld_in_c32 == 4) {
            ld_in_c32 = 0;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) bp[ni] += c32_jump;
        }
        cnext += 8;
    };

#pragma unroll
    for (int d = 0; d < D; ++d) fetch(d);
    const bool ragged = (Ctot & 31) != 0;   // last chunk holds channels >= Ctot: the weights there are zero, but
                                            // whatever was read in their place may be Inf / NaN
    for (int c32 = 0; c32 < p.NC32; ++c32) {
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            fetch((kc + D) & 3);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (GN) {
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    f32x4 v = xa[kc][mi] * ca[kc] + cb[kc];
                    if (p.gn_silu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
                    }
                    xa[kc][mi] = v;
                }
            }
            if (ragged && c32 == p.NC32 - 1) {
                const bool keep = (c32 * 32 + kc * 8 + 4 * lh) < Ctot;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xa[kc][mi][e] = keep ? xa[kc][mi][e] : 0.f;
            }
#if defined(ND_MFMA_INTERLEAVED)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[kc][ni][j], xa[kc][mi][j], acc[mi][ni], 0, 0, 0);
#else
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[kc][ni][j], xa[kc][mi][j], acc[mi][ni], 0, 0, 0);
#endif
        }
    }

    // ---- epilogue (D^T layout, see conv_mfma_kernel): one pixel and 4 consecutive channels per register group
    const bool vec_ok = ((p.ldo & 3) == 0) && (!p.res || (p.ldr & 3) == 0);
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + (wm * TM + mi) * 32 + l31;
        if (m < M) {
            float* orow = p.out + (size_t)m * p.ldo;
            const float* rr = p.res ? p.res + (size_t)m * p.ldr : nullptr;
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
                        *reinterpret_cast<f32x4*>(orow + n) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (n + e < p.N) {
                                float v = acc[mi][ni][4 * g4 + e];
                                if (p.bias) v += p.bias[n + e];
                                if (rr) v += rr[n + e];
                                if (p.silu_out) v = fast_silu(v);
                                orow[n + e] = v;
                            }
                        }
                    }
                }
            }
        }
    }
}

// Weights [N][C][k][k] (OIHW; also Conv1d [N][C][1] and Linear [N][C]) -> fragment order
//   out[((((c32*NT32 + ntile)*taps + tap)*4 + kc)*64 + lane)*4 + j] = w[n = ntile*32 + (lane&31)][c = c32*32 + kc*8 + (lane>>5)*4 + j][tap]
// zero for n >= N, c >= C, and for the padding chunks (c32 in [ceil(C/32), nc32_padded(C)]).
__global__ void pack_conv_weight_kernel(const float* w, float* out, int N, int C, int taps, int NT32, long total) {
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int j = (int)(it & 3);
        const int lane = (int)((it >> 2) & 63);
        long r = it >> 8;
        const int kc = (int)(r & 3);
        r >>= 2;
        const int tap = (int)(r % taps);
        r /= taps;
        const int ntile = (int)(r % NT32);
        const int c32 = (int)(r / NT32);
        const int n = ntile * 32 + (lane & 31);
        const int c = c32 * 32 + kc * 8 + (lane >> 5) * 4 + j;
        out[it] = (n < N && c < C) ? w[((size_t)n * C + c) * taps + tap] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side: tile-shape variants and launch
// ------------------------------------------------------------------------------------------------------------
struct Variant {
    int wm, wn, tm, tn, occ;     // occ = blocks per CU the kernel is compiled for (waves/SIMD = occ * waves / 4)
    int bm() const { return wm * tm * 32; }
    int bn() const { return wn * tn * 32; }
    int nt() const { return wm * wn * 64; }
};

static const Variant kVariants[] = {
    {4, 2, 2, 3, 1},   // 0: 256 x 192, 8 waves, 1 block / CU
    {2, 2, 2, 3, 2},   // 1: 128 x 192, 4 waves, 2 blocks / CU
    {4, 2, 2, 2, 1},   // 2: 256 x 128, 8 waves
    {2, 2, 2, 2, 2},   // 3: 128 x 128, 4 waves, 2 blocks / CU
    {4, 1, 1, 3, 3},   // 4: 128 x  96, 4 waves, 3 blocks / CU
    {2, 2, 1, 3, 3},   // 5:  64 x 192, 4 waves, 3 blocks / CU
    {2, 2, 2, 1, 2},   // 6: 128 x  64, 4 waves
    {2, 2, 1, 1, 4},   // 7:  64 x  64, 4 waves
    {2, 1, 1, 1, 4},   // 8:  64 x  32, 2 waves
    // 1x1 stream form (gemm_stream_kernel; flat pixel lists only, never chosen by the cost model)
    {2, 2, 2, 3, 2},   // 9:  128 x 192
    {2, 2, 2, 2, 3},   // 10: 128 x 128
    {2, 2, 1, 3, 3},   // 11:  64 x 192
    {4, 1, 2, 3, 2},   // 12: 256 x  96
    // GEMM form (gemm_f32_kernel, nd_gemm_f32.hip): both operands through three LDS stages filled by LDS-DMA
    {4, 2, 2, 2, 1},   // 13: 256 x 128, 8 waves
    // GEMM form, two blocks per CU (gemm4_kernel, nd_gemm_f32_quad.hip): pixel rows by buffer_load lds, no vector
    // instruction per DMA, hand-counted waits
    {2, 2, 4, 2, 2},   // 14: 256 x 128, 4 waves
    {2, 2, 2, 2, 3},   // 15: 128 x 128, 4 waves, three blocks per CU: the same kernel where 256-pixel tiles fill the chip badly
};
static constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
static constexpr int kFirstStream = 9;

template <int WM, int WN, int TM, int TN, int OCC>
static int launch_stream(const ConvArgs& a, int grid, hipStream_t s) {
    if (a.gnA)
        hipLaunchKernelGGL((gemm_stream_kernel<WM, WN, TM, TN, OCC, true>), dim3(grid), dim3(WM * WN * 64), 0, s, a);
    else
        hipLaunchKernelGGL((gemm_stream_kernel<WM, WN, TM, TN, OCC, false>), dim3(grid), dim3(WM * WN * 64), 0, s, a);
    return check_launch("nd_conv_nhwc");
}

template <int WM, int WN, int TM, int TN, int TAPS, int OCC>
static int launch_variant(const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
    if constexpr (TAPS == 1 && (WM * TM * 32 * 16) / (WM * WN * 64) > 8) {
        set_error("nd_conv_nhwc: this variant has no 1x1 form");
        return ND_E_ARG;
    } else {
    auto kern = conv_mfma_kernel<WM, WN, TM, TN, TAPS, (OCC * WM * WN + 3) / 4>;
    static bool attr_set[kMaxDevices] = {};
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), attr_set, "nd_conv_nhwc")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid, a.ksplit > 1 ? a.ksplit : 1), dim3(WM * WN * 64), lds, s, a);
    return check_launch("nd_conv_nhwc");
    }
}

template <int TAPS>
static int dispatch(int v, const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
    switch (v) {
        case 0: return launch_variant<4, 2, 2, 3, TAPS, 1>(a, grid, lds, s);
        case 1: return launch_variant<2, 2, 2, 3, TAPS, 2>(a, grid, lds, s);
        case 2: return launch_variant<4, 2, 2, 2, TAPS, 1>(a, grid, lds, s);
        case 3: return launch_variant<2, 2, 2, 2, TAPS, 2>(a, grid, lds, s);
        case 4: return launch_variant<4, 1, 1, 3, TAPS, 3>(a, grid, lds, s);
        case 5: return launch_variant<2, 2, 1, 3, TAPS, 3>(a, grid, lds, s);
        case 6: return launch_variant<2, 2, 2, 1, TAPS, 2>(a, grid, lds, s);
        case 7: return launch_variant<2, 2, 1, 1, TAPS, 4>(a, grid, lds, s);
        case 8: return launch_variant<2, 1, 1, 1, TAPS, 4>(a, grid, lds, s);
    }
    set_error("nd_conv_nhwc: bad variant %d", v);
    return ND_E_ARG;
}


static size_t lds_bytes(int taps, int hp) { return (size_t)2 * hp * (taps == 9 ? 32 : 64) * sizeof(float); }

// Choose TH x TW x NIB = BM minimising padded pixels, then halo size.
static bool plan_tiles(int bm, int nt, int taps, int NI, int H, int W, TilePlan* best) {
    const int lbm = ilog2(bm);
    bool found = false;
    const int pad = taps == 9 ? 1 : 0;
    for (int twl = 0; twl <= lbm; ++twl) {
        for (int thl = 0; thl + twl <= lbm; ++thl) {
            const int nibl = lbm - twl - thl;
            const int TW = 1 << twl, TH = 1 << thl, NIB = 1 << nibl;
            const int hp = NIB * (TH + 2 * pad) * (TW + 2 * pad);
            const int spr = taps == 9 ? 8 : 16;
            const int maxhi = taps == 9 ? 9 : bm * spr / nt;
            if ((long)hp * spr > (long)maxhi * nt) continue;
            if (lds_bytes(taps, hp) > 160 * 1024) continue;
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

// Choose the tile-shape variant (and its spatial tiling) for a problem from a cost model:
//   rounds of blocks over the CU slots x tile cost.  (The Python plan builder overrides this by measuring.)
static int select_variant(int variant, int taps, int pNI, int pH, int pW, int N, TilePlan* out_tp) {
    int best_v = -1;
    TilePlan best_tp{};
    double best_cost = 0;
    for (int v = 0; v < kFirstStream; ++v) {
        if (variant >= 0 && v != variant) continue;
        const Variant& V = kVariants[v];
        if (taps != 9 && V.bm() * 16 / V.nt() > 8) continue;      // 1x1: one halo item per k-step
        TilePlan tp;
        if (!plan_tiles(V.bm(), V.nt(), taps, pNI, pH, pW, &tp)) continue;
        const long nblk_n = (N + V.bn() - 1) / V.bn();
        const long nblocks = (long)tp.tiles_x * tp.tiles_y * tp.groups * nblk_n;
        const size_t lds = lds_bytes(taps, tp.hp);
        int per_cu = (int)(160 * 1024 / lds);
        if (per_cu > V.occ) per_cu = V.occ;
        if (per_cu < 1) per_cu = 1;
        const long slots = 256L * per_cu;
        const long rounds = (nblocks + slots - 1) / slots;
        double cost = (double)rounds * per_cu * V.bm() * V.bn();
        const double eff = (V.tm * V.tn >= 4) ? 1.0 : (V.tm * V.tn >= 2 ? 0.93 : 0.85);
        cost /= eff;
        if (best_v < 0 || cost < best_cost * 0.999) {
            best_v = v; best_tp = tp; best_cost = cost;
        }
    }
    if (best_v >= 0) *out_tp = best_tp;
    return best_v;
}

// flat pixel list for 1x1 (keeps tiles dense for odd image sizes)
static bool use_flat(int taps, int flags, const float* rowbias) {
    return taps == 1 && !(flags & (ND_CONV_IN_UP2X | ND_CONV_RES_UP2X)) && rowbias == nullptr;
}

}  // namespace nd

using namespace nd;

extern "C" int nd_conv_num_variants(void) { return kNumVariants; }

extern "C" int64_t nd_conv_weight_floats(int N, int C, int ksize) {
    if (N <= 0 || C <= 0 || (ksize != 1 && ksize != 3)) return ND_E_ARG;
    const int64_t nt32 = (N + 31) / 32;
    return (int64_t)(nc32_padded(C) + wstream::kF32PadChunks) * nt32 * ksize * ksize * 4 * 256;
}

// Upper bound (in floats) of what a launch of `variant` (< 0: any) may read of a packed tensor: the chunks it consumes plus
// its read-ahead, by the expressions the kernels static_assert on (nd_weight_stream.h)
extern "C" int64_t nd_conv_max_weight_read(int variant, int N, int C, int ksize) {
    if (N <= 0 || C <= 0 || (ksize != 1 && ksize != 3) || variant >= kNumVariants) return ND_E_ARG;
    const int taps = ksize * ksize;
    const int64_t nt32 = (N + 31) / 32;
    int ahead = 0;
    for (int v = 0; v < kNumVariants; ++v) {
        if (variant >= 0 && v != variant) continue;
        const int a = v < kFirstStream ? wstream::f32_conv_ahead(taps)
                                       : (v < 13 ? wstream::kF32StreamAhead : (v == 13 ? wstream::kF32GemmAhead : wstream::kF32Gemm4Ahead));          // 14, 15: gemm4_kernel
        ahead = a > ahead ? a : ahead;
    }
    return (int64_t)(nc32_padded(C) + wstream::pad_chunks(ahead, taps * 4)) * nt32 * taps * 4 * 256;
}

extern "C" int nd_repack_conv_weight(const float* w, float* w_out, int N, int C, int ksize, nd_stream_t stream) {
    const char* fn = "nd_repack_conv_weight";
    ND_REQUIRE(w && w_out && N > 0 && C > 0 && (ksize == 1 || ksize == 3), fn, "bad arguments");
    const long total = (long)nd_conv_weight_floats(N, C, ksize);
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w,
                       w_out, N, C, ksize * ksize, (N + 31) / 32, total);
    return check_launch(fn);
}

static int conv_f32_impl(const char* fn, const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1,
                         const float* w, const float* bias, const float* rowbias, int ld_rowbias,
                         const float* residual, int ldr, float* out, int ldo,
                         int NI, int H, int W, int N, int ksize, int flags, int variant,
                         const float* gnA, const float* gnB, int ld_gn, nd_stream_t stream, int splits = 1,
                         float* workspace = nullptr, float* chstats = nullptr) {
    ND_REQUIRE(x0 && w && out, fn, "null pointer");
    ND_REQUIRE(ksize == 1 || ksize == 3, fn, "ksize must be 1 or 3");
    ND_REQUIRE(NI > 0 && H > 0 && W > 0 && N > 0 && C0 > 0 && C1 >= 0, fn, "bad shape");
    ND_REQUIRE((C0 & 3) == 0 && (C1 & 3) == 0 && (ldx0 & 3) == 0, fn,
               "channel counts and strides must be multiples of 4");
    ND_REQUIRE(ldx0 >= C0 && ldo >= N, fn, "stride smaller than channel count");
    ND_REQUIRE(aligned16(x0) && aligned16(w), fn, "x0 / w must be 16-byte aligned");
    if (C1 > 0) {
        ND_REQUIRE(x1 != nullptr && (ldx1 & 3) == 0 && ldx1 >= C1 && aligned16(x1), fn,
                   "two-source input needs an aligned x1 with ldx1 >= C1");
    }
    const int up = (flags & ND_CONV_IN_UP2X) ? 1 : 0;
    const int res_up = (flags & ND_CONV_RES_UP2X) ? 1 : 0;
    if (up || res_up) ND_REQUIRE((H & 1) == 0 && (W & 1) == 0, fn, "2x upsampled read needs even H, W");
    if (flags & ND_CONV_SILU_OUT) ND_REQUIRE(residual == nullptr, fn, "SILU_OUT with a residual is not supported");
    if (residual) ND_REQUIRE(ldr >= N, fn, "ldr < N");
    if (rowbias) ND_REQUIRE(ld_rowbias >= N, fn, "ld_rowbias < N");
    ND_REQUIRE((long)NI * H * W < (1L << 31) / 2, fn, "too many pixels for 32-bit pixel indices");

    const int taps = ksize * ksize;
    const long M = (long)NI * H * W;
    int pNI = NI, pH = H, pW = W;
    // (a split launch leaves the per-image bias to its reduce pass, so it may use the dense flat pixel list)
    const bool flat = use_flat(taps, flags, splits > 1 ? nullptr : rowbias);
    if (flat) { pNI = 1; pH = 1; pW = (int)M; }

    ND_REQUIRE(variant < kNumVariants, fn, "bad variant");
    const bool stream_form = variant >= kFirstStream;
    if (stream_form) ND_REQUIRE(flat, fn, "stream variants take plain 1x1 convolutions (flat pixel list) only");
    TilePlan best_tp{};
    int best_v = variant;
    if (stream_form) {
        const int bm = kVariants[variant].bm();
        best_tp.thl = 0; best_tp.twl = ilog2(bm); best_tp.nibl = 0;
        best_tp.tiles_x = (int)((M + bm - 1) / bm); best_tp.tiles_y = 1; best_tp.groups = 1; best_tp.hp = bm;
    } else {
        best_v = select_variant(variant, taps, pNI, pH, pW, N, &best_tp);
        if (best_v < 0) return fail_arg(fn, "no tile variant fits this shape");
    }

    const Variant& V = kVariants[best_v];
    ConvArgs a;
    a.ksplit = 1; a.kchunks = 0; a.ws_stride = 0;
    if (splits > 1) {
        // block row s runs the channel range of split s and leaves raw accumulators in the workspace;
        // splitk_reduce_f32_kernel finishes the layer (bias, per-image bias, residual, SiLU)
        ND_REQUIRE(workspace != nullptr && aligned16(workspace) && !stream_form && gnA == nullptr, fn,
                   "split-K: needs a 16-byte aligned workspace; conv_mfma_kernel variants only; no fused GroupNorm");
        ND_REQUIRE(!res_up && (N & 3) == 0 && (ldo & 3) == 0 && aligned16(out) && (!bias || aligned16(bias)) &&
                   (!residual || ((ldr & 3) == 0 && aligned16(residual))) &&
                   (!rowbias || ((ld_rowbias & 3) == 0 && aligned16(rowbias))), fn,
                   "split-K: N and the strides must be multiples of 4 with 16-byte aligned pointers; no 2x-upsampled residual");
        ND_REQUIRE(((C0 + C1) & 31) == 0 && (C1 == 0 || (C0 & 3) == 0), fn, "split-K: whole 32-channel chunks");
        int kc = 0;
        const int S = splitk_plan_f32(C0 + C1, ksize, splits, &kc);
        ND_REQUIRE(S > 1, fn, "split-K: too few input channels for that many splits");
        a.ksplit = S; a.kchunks = kc; a.ws_stride = M * N;
    }
    a.x0 = x0; a.x1 = (C1 > 0) ? x1 : x0; a.w = w; a.bias = bias; a.rowbias = rowbias; a.res = residual; a.out = out;
    if (a.ksplit > 1) { a.bias = nullptr; a.rowbias = nullptr; a.res = nullptr; a.out = workspace; }
    a.C0 = C0; a.C1 = C1; a.ldx0 = ldx0; a.ldx1 = (C1 > 0) ? ldx1 : ldx0;
    a.NI = pNI; a.H = pH; a.W = pW;
    a.up = up; a.res_up = res_up;
    a.Hs = pH >> up; a.Ws = pW >> up;
    a.N = N; a.ldo = (a.ksplit > 1) ? N : ldo; a.ldr = ldr; a.ld_rowbias = ld_rowbias;
    a.NT32 = (N + 31) / 32;
    a.NC32 = (C0 + C1 + 31) / 32;
    a.thl = best_tp.thl; a.twl = best_tp.twl; a.nibl = best_tp.nibl;
    a.tiles_x = best_tp.tiles_x; a.tiles_y = best_tp.tiles_y;
    a.mt = best_tp.tiles_x * best_tp.tiles_y * best_tp.groups;
    a.nt = (N + V.bn() - 1) / V.bn();
    a.ngroup = pick_ngroup(a.nt, (size_t)V.bn() * (C0 + C1) * taps * sizeof(float), (size_t)M * (C0 + C1) * sizeof(float));
    a.vec_ok = 0; a.nhi = 0; a.zero = nullptr; a.chstats = nullptr; a.mbi = 1;
    if (chstats) {
        // output statistics: gemm4_kernel only (variant 14), one fp32 row per (image, 128-pixel run)
        ND_REQUIRE((best_v == 14 || best_v == 15) && flat && gnA == nullptr && splits <= 1, fn, "output statistics: the two-blocks-per-CU GEMM (variants 14 / 15) only, no fused GroupNorm");
        ND_REQUIRE(((long)H * W) % 128 == 0 && (N & 3) == 0 && ldo == N && aligned16(out) && aligned16(chstats) &&
                   (!bias || aligned16(bias)) && (!residual || ((ldr & 3) == 0 && aligned16(residual))), fn,
                   "output statistics: H*W % 128 == 0, N % 4 == 0, ldo == N, 16-byte aligned rows");
        a.chstats = chstats; a.mbi = (int)(((long)H * W) / 128);
    }
    a.silu_out = ((flags & ND_CONV_SILU_OUT) && a.ksplit <= 1) ? 1 : 0;
    a.gnA = gnA; a.gnB = gnB; a.ld_gn = ld_gn; a.gn_silu = (flags & ND_CONV_GN_SILU) ? 1 : 0; a.gn_hw = 0;
    if (gnA) {
        // one image per block, so that a thread's coefficient pair is fixed per chunk
        ND_REQUIRE(gnB != nullptr && ld_gn >= C0 + C1 && (ld_gn & 3) == 0 && aligned16(gnA) && aligned16(gnB), fn,
                   "fused GroupNorm: bad coefficient arrays");
        if (flat) {
            ND_REQUIRE(((long)H * W) % V.bm() == 0, fn, "fused GroupNorm (1x1): H*W must be a multiple of the pixel tile");
            a.gn_hw = H * W;
        } else {
            ND_REQUIRE(best_tp.nibl == 0, fn, "fused GroupNorm needs one image per block (H*W >= pixel tile)");
        }
    }
    const int grid = a.mt * a.nt;
    const size_t lds = lds_bytes(taps, best_tp.hp);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (stream_form ? best_v : -1) {
        case 9: return launch_stream<2, 2, 2, 3, 2>(a, grid, s);
        case 10: return launch_stream<2, 2, 2, 2, 3>(a, grid, s);
        case 11: return launch_stream<2, 2, 1, 3, 3>(a, grid, s);
        case 12: return launch_stream<4, 1, 2, 3, 2>(a, grid, s);
        case 13:
            a.zero = w + (nd_conv_weight_floats(N, C0 + C1, ksize) - 4);      // 16 bytes of the packed weights' zero padding chunk
            return launch_gemm_f32(a, grid, s);
        case 14:
        case 15:
            // gemm4_kernel addresses its input through buffer descriptors with the row advance in the scalar offset
            ND_REQUIRE(M % (best_v == 14 ? 256 : 128) == 0, fn, "the two-blocks-per-CU GEMM needs a multiple of 256 (variant 15: 128) pixels");
            ND_REQUIRE(((C0 + C1) & 31) == 0 && (C1 == 0 || (C0 & 31) == 0), fn, "the two-blocks-per-CU GEMM needs whole 32-channel chunks");
            ND_REQUIRE(M * ldx0 * 4 < (1L << 31) && (C1 == 0 || M * ldx1 * 4 < (1L << 31)), fn,
                       "the two-blocks-per-CU GEMM needs input tensors of less than 2 GiB");
            return launch_gemm4(a, grid, s, best_v == 14 ? 4 : 2);
    }
    const int rc = (taps == 9) ? dispatch<9>(best_v, a, grid, lds, s) : dispatch<1>(best_v, a, grid, lds, s);
    if (rc != ND_OK || a.ksplit <= 1) return rc;
    return launch_splitk_reduce_f32(workspace, a.ksplit, a.ws_stride, M, N, bias, rowbias, ld_rowbias, H * W, residual, ldr, out, ldo,
                                    (flags & ND_CONV_SILU_OUT) ? 1 : 0, s);
}

extern "C" int nd_conv_nhwc(const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1,
                            const float* w, const float* bias, const float* rowbias, int ld_rowbias,
                            const float* residual, int ldr, float* out, int ldo,
                            int NI, int H, int W, int N, int ksize, int flags, int variant,
                            const float* gnA, const float* gnB, int ld_gn, nd_stream_t stream) {
    return conv_f32_impl("nd_conv_nhwc", x0, C0, ldx0, x1, C1, ldx1, w, bias, rowbias, ld_rowbias, residual, ldr, out, ldo, NI, H, W,
                         N, ksize, flags, variant, gnA, gnB, ld_gn, stream);
}

// 1x1 convolution by gemm4_kernel (variant 14) that also leaves the per-channel partial statistics of its output behind:
// chstats [NI][H*W/128][sum | sum of squares][N] in fp32, one row per 128-pixel run of an image, every row written by every
// launch; nd_groupnorm_stats_from_partials folds them (the attention block's output projection + residual feeds the next
// GroupNorm, model.py:291,190).  nd_conv1x1_stats_rows gives the rows per image (0: this shape cannot).
extern "C" int nd_conv1x1_stats_rows(int NI, int H, int W, int N) {
    if (NI <= 0 || H <= 0 || W <= 0 || N <= 0) return ND_E_ARG;
    const long hw = (long)H * W;
    return (hw % 128 == 0 && ((long)NI * hw) % 256 == 0 && (N & 3) == 0) ? (int)(hw / 128) : 0;
}

extern "C" int nd_conv1x1_stats_nhwc(const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1,
                                     const float* w, const float* bias, const float* residual, int ldr, float* out, int ldo,
                                     int NI, int H, int W, int N, int flags, float* chstats, nd_stream_t stream) {
    const char* fn = "nd_conv1x1_stats_nhwc";
    ND_REQUIRE(chstats != nullptr, fn, "chstats is null");
    // 256- or 128-pixel blocks: whichever fills the chip's block slots better (rows of 128 pixels either way)
    const int v = (NI > 0 && H > 0 && W > 0 && N > 0 && gemm4_pick_tm((long)NI * H * W, (N + 127) / 128) == 2) ? 15 : 14;
    return conv_f32_impl(fn, x0, C0, ldx0, x1, C1, ldx1, w, bias, nullptr, 0, residual, ldr, out, ldo, NI, H, W, N, 1, flags, v,
                         nullptr, nullptr, 0, stream, 1, nullptr, chstats);
}

// The same convolution split over K: `splits` (2..16) block rows each run a range of 32-channel chunks of the input and
// leave raw accumulators in `workspace` (nd_conv_splitk_workspace_floats() floats); a second launch adds them in split
// order and applies bias / per-image bias / residual / SiLU.  For layers whose output has too few tiles to fill the chip
// and whose contraction is long (7x7 .. 16x16 maps at small batch); conv_mfma_kernel variants (0..8) only, named explicitly.
extern "C" int nd_conv_splitk_nhwc(const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1,
                                   const float* w, const float* bias, const float* rowbias, int ld_rowbias,
                                   const float* residual, int ldr, float* out, int ldo,
                                   int NI, int H, int W, int N, int ksize, int flags, int variant, int splits,
                                   float* workspace, nd_stream_t stream) {
    const char* fn = "nd_conv_splitk_nhwc";
    ND_REQUIRE(splits >= 2 && splits <= 16 && workspace != nullptr && variant >= 0 && variant < kFirstStream, fn,
               "2..16 splits, a workspace of nd_conv_splitk_workspace_floats() floats, and a named conv_mfma_kernel variant");
    return conv_f32_impl(fn, x0, C0, ldx0, x1, C1, ldx1, w, bias, rowbias, ld_rowbias, residual, ldr, out, ldo, NI, H, W, N, ksize,
                         flags, variant, nullptr, nullptr, 0, stream, splits, workspace);
}

// fp32 words nd_conv_splitk_nhwc writes to (and its reduce kernel reads from) `workspace`: one [NI*H*W][N] slab of raw
// accumulators per split actually used -- fewer than asked for when the layer has fewer chunks
extern "C" int64_t nd_conv_splitk_workspace_floats(int NI, int H, int W, int N, int C, int ksize, int splits) {
    if (NI <= 0 || H <= 0 || W <= 0 || N <= 0 || C <= 0 || (ksize != 1 && ksize != 3) || splits < 2 || splits > 16) return ND_E_ARG;
    int kc = 0;
    const int S = splitk_plan_f32(C, ksize, splits, &kc);
    if (S < 2) return ND_E_ARG;
    return (int64_t)S * NI * H * W * N;
}

extern "C" int nd_conv_select_variant(int NI, int H, int W, int N, int ksize, int flags, int has_rowbias) {
    if (!(ksize == 1 || ksize == 3) || NI <= 0 || H <= 0 || W <= 0 || N <= 0) return ND_E_ARG;
    const int taps = ksize * ksize;
    int pNI = NI, pH = H, pW = W;
    if (use_flat(taps, flags, has_rowbias ? reinterpret_cast<const float*>(1) : nullptr)) {
        pNI = 1; pH = 1; pW = NI * H * W;
    }
    TilePlan tp{};
    const int v = select_variant(-1, taps, pNI, pH, pW, N, &tp);
    return v < 0 ? ND_E_ARG : v;
}

extern "C" int nd_conv_variant_info(int variant, int* bm, int* bn, int* threads) {
    if (variant < 0 || variant >= kNumVariants) return ND_E_ARG;
    if (bm) *bm = kVariants[variant].bm();
    if (bn) *bn = kVariants[variant].bn();
    if (threads) *threads = kVariants[variant].nt();
    return ND_OK;
}


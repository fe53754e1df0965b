// K5/K6: 3x3 (stride 1, pad 1) and 1x1 convolution / linear as an implicit GEMM on the fp32 matrix cores.
//
// Replaces nn.Conv2d / nn.Conv1d(k=1) / nn.Linear calls of the reference (model.py:72,169,173,177,180,247,253,
// 349-351,367,448) plus the ops fused around them: torch.cat of the skip connection (model.py:474, two-source
// input), bias, the per-image timestep-embedding add (model.py:205), the residual add (model.py:211,291) and
// the nearest-2x upsampling in front of a conv (model.py:77-79, ND_CONV_IN_UP2X).
//
// Design for gfx950 (wave64, v_mfma_f32_32x32x2_f32 = exact fp32 FMA chains at the fp32 vector rate):
//   GEMM view   M = output pixels, N = output channels, K = taps x input channels.
//   block       WAVES_M x WAVES_N waves; wave tile = TM x TN MFMA tiles of 32x32; BM = 32*TM*WAVES_M pixels,
//               BN = 32*TN*WAVES_N channels.  The BM pixels are NIB images x (TH x TW) spatial tile.
//   K loop      32 input channels at a time ("chunk").  Per chunk the input halo tile ((TH+2)x(TW+2) pixels x 32
//               channels) is staged in LDS ONCE and reused by all 9 taps (the A operand of tap (dy,dx) is the same
//               LDS image read at a shifted pixel), so activations cross L2->LDS ~1.3x instead of 9x.  Weights
//               stream through a double-buffered [BN][32] LDS tile per (chunk, tap).
//   operands    lane (i = lane&31, h = lane>>5) holds A[i][k] / B[k][i]; one ds_read_b128 fetches the 4 floats
//               k = 8*kc + 4*h + {0..3} of row i, which feed 4 consecutive MFMAs (the k order inside a chunk is a
//               fixed permutation shared by A and B).  16-byte slots are XOR-swizzled by (row>>1)&7 so that a
//               16-lane ds_read_b128 group touches 16 distinct slots of the 256-byte bank row.
//   pipeline    global -> registers for the next weight tile (and 1/9 of the next halo) is issued before the
//               MFMAs of the current tap and written to the other LDS buffer after them; one barrier per tap.
//   grid        one block per (m tile, n tile), remapped so that each XCD gets a contiguous, n-major range of
//               tiles (blocks resident on one XCD stream the same weight tiles through that XCD's L2).
#include "nd_common.h"

namespace nd {

struct ConvArgs {
    const float* x0;
    const float* x1;
    const float* w;
    const float* bias;
    const float* rowbias;
    const float* res;
    float* out;
    int C0, C1, ldx0, ldx1, ldw;
    int NI, H, W;      // output (= virtual input) size
    int Hs, Ws;        // stored input size (H >> up)
    int up;            // input read through nearest-2x upsampling
    int res_up;        // residual read through nearest-2x upsampling
    int N, ldo, ldr, ld_rowbias;
    int thl, twl, nibl;   // log2 of tile height / width / images per block
    int tiles_x, tiles_y, mt, nt;
    int silu_out;
};

template <int TAPS>
struct Halo {
    static constexpr int PAD = (TAPS == 9) ? 1 : 0;
};

// number of float4 halo items a thread may own (item k of a thread = halo pixel (tid>>3) + k*(NT>>3), slot tid&7)
template <int BM, int NT, int TAPS>
struct HaloItems {
    static constexpr int value = (TAPS == 9) ? 9 : (BM * 8 / NT);
};

template <int WAVES_M, int WAVES_N, int TM, int TN, int TAPS>
__global__ void __launch_bounds__(WAVES_M* WAVES_N * 64)
    conv_mfma_kernel(const ConvArgs p) {
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int BM = WAVES_M * TM * 32;
    constexpr int BN = WAVES_N * TN * 32;
    constexpr int PAD = Halo<TAPS>::PAD;
    constexpr int MAXHI = HaloItems<BM, NT, TAPS>::value;
    constexpr int HPF = (TAPS == 9) ? 1 : MAXHI;       // halo float4 prefetched per iteration
    constexpr int WI = BN * 8 / NT;                    // weight float4 per thread per tile
    static_assert(BN * 8 % NT == 0, "weight tile must divide over the block");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N;
    const int wn = wave - wm * WAVES_N;
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    // ---- XCD-aware block -> tile map: XCD x (= blockIdx % 8 under round-robin dispatch) gets a contiguous range.
    const int total = gridDim.x;
    const int q = total >> 3, r = total & 7, xcd = blockIdx.x & 7;
    const int idp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    const int nblk = idp / p.mt;
    const int mblk = idp - nblk * p.mt;
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

    float* const halo_base = smem;                     // 2 x [HP][32]
    float* const w_base = smem + 2 * HP * 32;          // 2 x [BN][32]

    // ---- halo descriptors: source pixel index of each halo item (or -1 = zero fill)
    const int hslot = tid & 7;
    int gpix[MAXHI];
#pragma unroll
    for (int k = 0; k < MAXHI; ++k) {
        const int hp = (tid >> 3) + k * (NT >> 3);
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
    const int nchunks = (Ctot + 31) >> 5;
    const int nit = nchunks * TAPS;

    auto load_halo_item = [&](int k, int ch) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int c = (ch << 5) + (hslot << 2);
        const int g = gpix[k];
        if (g >= 0 && c < Ctot) {
            const float* src = (c < p.C0) ? (p.x0 + (size_t)g * p.ldx0 + c) : (p.x1 + (size_t)g * p.ldx1 + (c - p.C0));
            v = *reinterpret_cast<const f32x4*>(src);
        }
        return v;
    };
    auto store_halo_item = [&](int k, int buf, f32x4 v) {
        const int hp = (tid >> 3) + k * (NT >> 3);
        if (hp < HP) {
            float* dst = halo_base + buf * (HP * 32) + hp * 32 + ((hslot ^ ((hp >> 1) & 7)) << 2);
            *reinterpret_cast<f32x4*>(dst) = v;
        }
    };
    auto load_w_item = [&](int k, int ch, int tap) -> f32x4 {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int n = n0 + (tid >> 3) + k * (NT >> 3);
        const int c = (ch << 5) + (hslot << 2);
        if (n < p.N && c < Ctot) v = *reinterpret_cast<const f32x4*>(p.w + ((size_t)tap * p.N + n) * p.ldw + c);
        return v;
    };
    auto store_w_item = [&](int k, int buf, f32x4 v) {
        const int n = (tid >> 3) + k * (NT >> 3);
        float* dst = w_base + buf * (BN * 32) + n * 32 + ((hslot ^ ((n >> 1) & 7)) << 2);
        *reinterpret_cast<f32x4*>(dst) = v;
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
    int b_row[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) b_row[ni] = ((wn * TN + ni) * 32 + l31) * 32;
    const int b_swz = (l31 >> 1) & 7;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    // ---- prologue: chunk 0 halo + first weight tile
#pragma unroll
    for (int k = 0; k < MAXHI; ++k) store_halo_item(k, 0, load_halo_item(k, 0));
#pragma unroll
    for (int k = 0; k < WI; ++k) store_w_item(k, 0, load_w_item(k, 0, 0));
    __syncthreads();

    int it = 0;
    for (int ch = 0; ch < nchunks; ++ch) {
        const float* hbuf = halo_base + (ch & 1) * (HP * 32);
#pragma unroll 1
        for (int tap = 0; tap < TAPS; ++tap, ++it) {
            // -- issue global loads for the next iteration
            const bool has_next = (it + 1) < nit;
            const int ntap = (tap + 1 == TAPS) ? 0 : tap + 1;
            const int nch = (tap + 1 == TAPS) ? ch + 1 : ch;
            f32x4 pw[WI];
#pragma unroll
            for (int k = 0; k < WI; ++k) {
                pw[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (has_next) pw[k] = load_w_item(k, nch, ntap);
            }
            const bool halo_next = (ch + 1) < nchunks;
            f32x4 ph[HPF];
            if constexpr (TAPS == 9) {
                ph[0] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (halo_next) {
                    // item `tap` of the next chunk's halo (MAXHI == 9: one item per tap)
#pragma unroll
                    for (int k = 0; k < MAXHI; ++k)
                        if (k == tap) ph[0] = load_halo_item(k, ch + 1);
                }
            } else {
#pragma unroll
                for (int k = 0; k < HPF; ++k) {
                    ph[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (halo_next) ph[k] = load_halo_item(k, ch + 1);
                }
            }

            // -- MFMAs of this (chunk, tap)
            const float* wbuf = w_base + (it & 1) * (BN * 32);
            int tapoff = 0;
            if constexpr (TAPS == 9) {
                const int dy = tap / 3;
                tapoff = dy * HW + (tap - dy * 3);
            }
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                const int slot = (kc << 1) | lh;
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const int hp = a_hp[mi] + tapoff;
                    a[mi] = *reinterpret_cast<const f32x4*>(hbuf + hp * 32 + ((slot ^ ((hp >> 1) & 7)) << 2));
                }
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    b[ni] = *reinterpret_cast<const f32x4*>(wbuf + b_row[ni] + ((slot ^ b_swz) << 2));
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < TN; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
            }

            // -- park the prefetched tiles in the other LDS buffers
            if (has_next) {
#pragma unroll
                for (int k = 0; k < WI; ++k) store_w_item(k, (it + 1) & 1, pw[k]);
            }
            if (halo_next) {
                if constexpr (TAPS == 9) {
#pragma unroll
                    for (int k = 0; k < MAXHI; ++k)
                        if (k == tap) store_halo_item(k, (ch + 1) & 1, ph[0]);
                } else {
#pragma unroll
                    for (int k = 0; k < HPF; ++k) store_halo_item(k, (ch + 1) & 1, ph[k]);
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue: C/D layout of v_mfma_f32_32x32x2_f32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int n = n0 + (wn * TN + ni) * 32 + l31;
        const bool nok = n < p.N;
        const float bv = (nok && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = (wm * TM + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int li = m >> (p.thl + p.twl);
                const int oy = oy0 + ((m >> p.twl) & (TH - 1));
                const int ox = ox0 + (m & (TW - 1));
                const int img = img0 + li;
                if (nok && img < p.NI && oy < p.H && ox < p.W) {
                    float v = acc[mi][ni][e] + bv;
                    if (p.rowbias) v += p.rowbias[(size_t)img * p.ld_rowbias + n];
                    if (p.res) {
                        const size_t rp = p.res_up ? ((size_t)(img * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1))
                                                   : ((size_t)(img * p.H + oy) * p.W + ox);
                        v += p.res[rp * p.ldr + n];
                    }
                    if (p.silu_out) v = fast_silu(v);
                    p.out[((size_t)(img * p.H + oy) * p.W + ox) * p.ldo + n] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side: tile-shape variants and launch
// ------------------------------------------------------------------------------------------------------------
struct Variant {
    int wm, wn, tm, tn;
    int bm() const { return wm * tm * 32; }
    int bn() const { return wn * tn * 32; }
    int nt() const { return wm * wn * 64; }
};

static const Variant kVariants[] = {
    {4, 2, 2, 3},   // 0: 256 x 192, 8 waves
    {4, 2, 2, 2},   // 1: 256 x 128, 8 waves
    {2, 2, 2, 3},   // 2: 128 x 192, 4 waves
    {4, 1, 1, 3},   // 3: 128 x  96, 4 waves
    {2, 2, 2, 1},   // 4: 128 x  64, 4 waves
    {2, 2, 1, 1},   // 5:  64 x  64, 4 waves
    {2, 1, 1, 1},   // 6:  64 x  32, 2 waves
};
static constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

template <int WM, int WN, int TM, int TN, int TAPS>
static int launch_variant(const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
    auto kern = conv_mfma_kernel<WM, WN, TM, TN, TAPS>;
    static bool attr_set = false;   // one flag per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            set_error("nd_conv_nhwc: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return ND_E_LAUNCH;
        }
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, s, a);
    return check_launch("nd_conv_nhwc");
}

template <int TAPS>
static int dispatch(int v, const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
    switch (v) {
        case 0: return launch_variant<4, 2, 2, 3, TAPS>(a, grid, lds, s);
        case 1: return launch_variant<4, 2, 2, 2, TAPS>(a, grid, lds, s);
        case 2: return launch_variant<2, 2, 2, 3, TAPS>(a, grid, lds, s);
        case 3: return launch_variant<4, 1, 1, 3, TAPS>(a, grid, lds, s);
        case 4: return launch_variant<2, 2, 2, 1, TAPS>(a, grid, lds, s);
        case 5: return launch_variant<2, 2, 1, 1, TAPS>(a, grid, lds, s);
        case 6: return launch_variant<2, 1, 1, 1, TAPS>(a, grid, lds, s);
    }
    set_error("nd_conv_nhwc: bad variant %d", v);
    return ND_E_ARG;
}

struct TilePlan {
    int thl, twl, nibl, tiles_x, tiles_y, groups, hp;
    long padded;   // padded pixel count
};

static int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

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
            const int maxhi = taps == 9 ? 9 : bm * 8 / nt;
            if (hp * 8 > maxhi * nt) continue;
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

}  // namespace nd


// Choose the tile-shape variant (and its spatial tiling) for a problem: minimise
//   ceil(blocks / slots) * tile cost,   slots = CUs x blocks/CU that fit LDS and the wave budget.
static int select_variant(int variant, int taps, int pNI, int pH, int pW, int N, nd::TilePlan* out_tp) {
    using namespace nd;
    int best_v = -1;
    TilePlan best_tp{};
    double best_cost = 0;
    for (int v = 0; v < kNumVariants; ++v) {
        if (variant >= 0 && v != variant) continue;
        const Variant& V = kVariants[v];
        TilePlan tp;
        if (!plan_tiles(V.bm(), V.nt(), taps, pNI, pH, pW, &tp)) continue;
        const long nblk_n = (N + V.bn() - 1) / V.bn();
        const long nblocks = (long)tp.tiles_x * tp.tiles_y * tp.groups * nblk_n;
        const size_t lds = (size_t)(2 * tp.hp + 2 * V.bn()) * 128;
        if (lds > 160 * 1024) continue;
        int per_cu = (int)(160 * 1024 / lds);
        const int wave_cap = 8 / (V.nt() / 64) > 0 ? 8 / (V.nt() / 64) : 1;   // keep <= 2 waves / SIMD
        if (per_cu > wave_cap) per_cu = wave_cap;
        if (per_cu < 1) per_cu = 1;
        const long slots = 256L * per_cu;
        const long rounds = (nblocks + slots - 1) / slots;
        // cost of one round: a CU runs per_cu blocks concurrently sharing its 4 matrix pipes
        double cost = (double)rounds * per_cu * V.bm() * V.bn();
        // small wave tiles re-read operands from LDS more often and have less MFMA back-to-back: mild penalty
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

using namespace nd;

extern "C" int nd_conv_num_variants(void) { return kNumVariants; }

extern "C" int nd_conv_nhwc(const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1,
                            const float* w, int ldw, const float* bias, const float* rowbias, int ld_rowbias,
                            const float* residual, int ldr, float* out, int ldo,
                            int NI, int H, int W, int N, int ksize, int flags, int variant, nd_stream_t stream) {
    const char* fn = "nd_conv_nhwc";
    ND_REQUIRE(x0 && w && out, fn, "null pointer");
    ND_REQUIRE(ksize == 1 || ksize == 3, fn, "ksize must be 1 or 3");
    ND_REQUIRE(NI > 0 && H > 0 && W > 0 && N > 0 && C0 > 0 && C1 >= 0, fn, "bad shape");
    ND_REQUIRE((C0 & 3) == 0 && (C1 & 3) == 0 && (ldx0 & 3) == 0 && (ldw & 3) == 0, fn,
               "channel counts and strides must be multiples of 4");
    ND_REQUIRE(ldx0 >= C0 && ldw >= C0 + C1 && ldo >= N, fn, "stride smaller than channel count");
    ND_REQUIRE(aligned16(x0) && aligned16(w), fn, "x0 / w must be 16-byte aligned");
    if (C1 > 0) {
        ND_REQUIRE(x1 != nullptr && (C0 & 31) == 0 && (ldx1 & 3) == 0 && ldx1 >= C1 && aligned16(x1), fn,
                   "two-source input needs C0 % 32 == 0 and an aligned x1");
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
    // 1x1: the spatial structure is irrelevant -> flat pixel list (keeps tiles dense for odd image sizes)
    int pNI = NI, pH = H, pW = W;
    const bool flat = use_flat(taps, flags, rowbias);
    if (flat) { pNI = 1; pH = 1; pW = (int)M; }

    TilePlan best_tp{};
    const int best_v = select_variant(variant, taps, pNI, pH, pW, N, &best_tp);
    if (best_v < 0) return fail_arg(fn, "no tile variant fits this shape");

    const Variant& V = kVariants[best_v];
    ConvArgs a;
    a.x0 = x0; a.x1 = (C1 > 0) ? x1 : x0; a.w = w; a.bias = bias; a.rowbias = rowbias; a.res = residual; a.out = out;
    a.C0 = C0; a.C1 = C1; a.ldx0 = ldx0; a.ldx1 = (C1 > 0) ? ldx1 : ldx0; a.ldw = ldw;
    a.NI = pNI; a.H = pH; a.W = pW;
    a.up = up; a.res_up = res_up;
    a.Hs = pH >> up; a.Ws = pW >> up;
    a.N = N; a.ldo = ldo; a.ldr = ldr; a.ld_rowbias = ld_rowbias;
    a.thl = best_tp.thl; a.twl = best_tp.twl; a.nibl = best_tp.nibl;
    a.tiles_x = best_tp.tiles_x; a.tiles_y = best_tp.tiles_y;
    a.mt = best_tp.tiles_x * best_tp.tiles_y * best_tp.groups;
    a.nt = (N + V.bn() - 1) / V.bn();
    a.silu_out = (flags & ND_CONV_SILU_OUT) ? 1 : 0;
    const int grid = a.mt * a.nt;
    const size_t lds = (size_t)(2 * best_tp.hp + 2 * V.bn()) * 128;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return (taps == 9) ? dispatch<9>(best_v, a, grid, lds, s) : dispatch<1>(best_v, a, grid, lds, s);
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

// K5, Winograd F(4x4,3x3): the ResidualBlock 3x3 convolutions (reference: nicediffusion/model.py:173-177,194,209) on a
// quarter of the direct convolution's multiplies -- 36 transform positions per 16 output pixels against 16 per 4 for
// F(2x2,3x3), i.e. 0.5625 x the matrix instructions of conv_wino4_kernel.  Exact-fp32 MFMA throughout; the transforms use
// 0, +-1, +-2, +-4, +-5, +-8 on the data side and 1/4, 1/6, 1/12, 1/24 on the weight side (folded at repack, in float64).
//
// Shape of the kernel (conv_wf4_kernel).  What limits an F(4x4) block on gfx950 is the register file: 36 positions x 256
// output pixels x 48 channels of fp32 accumulators are 108 KiB.  So:
//   * workgroup = 12 waves = TWO half blocks of 6 (a 6-wave workgroup lands on the SIMDs as 2 + 2 + 1 + 1: 0.73 of the MFMA
//     rate, tools/micro/mfma_f32_rate.hip); in a half, wave xi owns ROW xi of the 6x6 transform (6 positions) for 16 tiles (a
//     16x16-pixel region, or four 8x8 images) x 48 output channels on v_mfma_f32_16x16x4_f32: 6 x 3 accumulators of 4
//     registers = 72; <= 168 registers per wave, i.e. THREE waves per SIMD = one workgroup per CU.  The halves work on the
//     same m tile and neighbouring n blocks (256 px x 96 ch per workgroup) and share ONE halo stream;
//   * the input transform B^T d B is computed per wave for its own row only: the row (y) part is 2 (rows 0, 5) or 3 (rows
//     1..4) fused multiply-adds per column with wave-uniform coefficients, the column (x) part 12 operations per 6
//     positions; both work on 8-channel half chunks (ds_read_b64) so that the transform's live values stay at 12 + 12
//     registers;
//   * halo chunks of 16 channels (64 B per pixel) arrive by LDS-DMA (buffer_load ... lds) with one precomputed byte offset
//     per lane and round: zero padding is the buffer's range check.  Pixels are stored in groups of four (256 B) with a
//     4-bit XOR key per group, so that the 16 tiles of a ds_read hit 16 different 16-byte units (tools/wf4_lds_image.py
//     checks the image against the read addresses on the CPU);
//   * weights stream global -> VGPR in fragment order [chunk16][n block][xi][k4][nu][lane][3]: one global_load_dwordx3 per
//     (k4-step, position) feeds the three MFMAs of that position; a fragment is re-loaded in place right behind the MFMAs
//     that consumed it (one k4-step = 18 MFMAs of read-ahead), all as inline ISA with hand-counted vmcnt / lgkmcnt, because
//     hipcc orders every LDS read it can see behind ALL pending LDS-DMA;
//   * ONE barrier per 16-channel chunk: it publishes chunk c + 1 (issued a chunk earlier) and frees chunk c's buffer for
//     chunk c + 2;
//   * epilogue: the row part of A^T M A in registers, the xi part through one LDS round (72 KiB per half), fused bias /
//     per-image bias / residual (+2x) / SiLU and -- STATS -- the per-channel partial sums of what was stored (one row per (m
//     block, output column b), folded by nd_groupnorm_stats_from_partials); split over K as conv_wino4_kernel.  PRE (a launch
//     with a residual): its vectors are requested before the epilogue's barriers.
//   Timing-only ablation builds (-DND_F4ABL_*: NOB NOA NOT NOHALO HALOHIT BHIT L1HIT ACF NOBAR NOEPI NOSTORE COALSTORE NOPROWAIT
//   NOWAITVM NOWAITLGKM) give WRONG results by construction; tools/ab_wf4.py times them (profiles/r05_wf4_ablations.txt).
#include "nd_conv_common.h"
#include <type_traits>

namespace nd {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));

constexpr int kWf4CT = 3;                                  // 16-channel n tiles per block: 48 output channels
constexpr int kWf4BN = 16 * kWf4CT;
constexpr int kWf4Frag = 64 * kWf4CT;                      // floats per weight fragment (one position, one k4-step)
constexpr int kWf4WaveChunk = 4 * 6 * kWf4Frag;            // floats per (chunk, n block, xi): 4 k4-steps x 6 positions
constexpr int kWf4BlockChunk = 6 * kWf4WaveChunk;          // floats per (chunk, n block)
constexpr int kWf4ExchangeBytes = 6 * 4 * kWf4CT * 64 * 16;   // epilogue exchange: [xi][b][ct][lane] float4

template <int GW>
struct Wf4Geo {
    static_assert(GW == 5 || GW == 3, "two block geometries");
    static constexpr int TWL2 = (GW == 5) ? 2 : 1;         // log2 of the tiles of a block along x / y
    static constexpr int THL2 = (GW == 5) ? 2 : 1;
    static constexpr int NIBL = (GW == 5) ? 0 : 2;         // log2 of the images per block
    static constexpr int TW = 4 << TWL2, TH = 4 << THL2;   // output pixels of a block (per image)
    static constexpr int HW = TW + 2, HH = TH + 2;         // halo
    static constexpr int NG = (1 << NIBL) * HH * GW;       // 256-byte groups (4 pixels x 64 B) per halo buffer
    static constexpr int NDMA = (NG + 47) / 48;            // DMA rounds per chunk: 12 waves x 1 KiB (4 groups) each
    static constexpr int BUF = NDMA * 12 * 1024;           // bytes per halo buffer
    static_assert(GW * 4 >= HW, "a halo row fits its groups");
    static_assert(2 * BUF <= 2 * kWf4ExchangeBytes, "the exchange buffers set the LDS size");
    static_assert((5 * GW + 1) * 256 + BUF + 8 < 65536, "ds_read offsets are 16-bit");
};

// Packed fp32 arithmetic for the input transform: every value of the transform is a pair of channels in a register pair,
// so one v_pk_* does what two v_fma_f32 / v_add_f32 would.  Measured beside back-to-back v_mfma_f32_16x16x4_f32 at three
// waves per SIMD (tools/micro/mfma_f32_beside.hip): a v_fma_f32 takes 4.2 cycles from the matrix pipe, a v_pk_fma_f32 5.7.
// ND_F4_PK=0 leaves the arithmetic to hipcc (which emits the scalar forms here).  Same bits either way (IEEE fma / add).
#ifndef ND_F4_PK
#define ND_F4_PK 1
#endif
typedef unsigned long long wf4_u64;
#if ND_F4_PK
#define ND_F4_FMA_C(NAME, CSTR)                                                                              \
    __device__ __forceinline__ f32x2 NAME(f32x2 b, f32x2 a) { /* a + C * b */                                \
        f32x2 r;                                                                                             \
        asm("v_pk_fma_f32 %0, %1, " CSTR ", %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(b), "v"(a));               \
        return r;                                                                                            \
    }
#else
#define ND_F4_FMA_C(NAME, CSTR) \
    __device__ __forceinline__ f32x2 NAME(f32x2 b, f32x2 a) { return (float)(CSTR##f) * b + a; }
#endif
#if ND_F4_PK
ND_F4_FMA_C(wf4_fma_p4, "4.0")
ND_F4_FMA_C(wf4_fma_m4, "-4.0")
ND_F4_FMA_C(wf4_fma_p2, "2.0")
ND_F4_FMA_C(wf4_fma_m2, "-2.0")
// a + s * b with the wave-uniform factor s in both halves of an SGPR pair
__device__ __forceinline__ f32x2 wf4_fma_s(wf4_u64 s, f32x2 b, f32x2 a) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "s"(s), "v"(a));
    return r;
}
__device__ __forceinline__ f32x2 wf4_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 wf4_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
__device__ __forceinline__ f32x2 wf4_fma_p4(f32x2 b, f32x2 a) { return 4.f * b + a; }
__device__ __forceinline__ f32x2 wf4_fma_m4(f32x2 b, f32x2 a) { return -4.f * b + a; }
__device__ __forceinline__ f32x2 wf4_fma_p2(f32x2 b, f32x2 a) { return 2.f * b + a; }
__device__ __forceinline__ f32x2 wf4_fma_m2(f32x2 b, f32x2 a) { return -2.f * b + a; }
__device__ __forceinline__ f32x2 wf4_fma_s(wf4_u64 s, f32x2 b, f32x2 a) { return __uint_as_float((unsigned)s) * b + a; }
__device__ __forceinline__ f32x2 wf4_add(f32x2 a, f32x2 b) { return a + b; }
__device__ __forceinline__ f32x2 wf4_sub(f32x2 a, f32x2 b) { return a - b; }
#endif
__device__ __forceinline__ wf4_u64 wf4_pair(float f) {
    const unsigned u = __float_as_uint(f);
    return ((wf4_u64)u << 32) | u;
}

template <int I, int N, class F>
__device__ __forceinline__ void wf4_sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        wf4_sfor<I + 1, N>(f);
    }
}

// VMEM operations younger than the fragment load of slot i - 6 when slot i of a step starts (5 fragment loads + the halo
// DMAs of step 1, which sit in slots 0 .. NDMA - 1 in front of the slot's load)
#ifndef ND_F4_LDEARLY
#define ND_F4_LDEARLY 1      // a slot's fragment load is issued right behind its MFMAs, in front of the slot's transform work and DMA: +1.3..2.3 % (0: behind them)
#endif
constexpr int wf4_younger(int S, int i, int NDMA) {
    int y = 5;
    if (S == 1)
        for (int k = (i - 5 - ND_F4_LDEARLY > 0 ? i - 5 - ND_F4_LDEARLY : 0); k <= i - 1; ++k)
            if (k < NDMA) ++y;
    return y;
}

// split over K: block row s runs the 16-channel chunks [s * kchunks, (s + 1) * kchunks) of a single-source input and leaves
// raw accumulators in out + s * ws_stride (the packed weights are chunk-major)
__device__ __forceinline__ int split_k_args_wf4(ConvArgs& p, int s) {
    int nc = p.NC32 - s * p.kchunks;
    nc = nc < p.kchunks ? nc : p.kchunks;
    p.w += (size_t)s * p.kchunks * ((size_t)p.nt * kWf4BlockChunk);
    p.NC32 = nc;
    p.out += (size_t)s * p.ws_stride;
    return s * p.kchunks * 16;          // channel shift of the input
}

template <int GW, bool STATS, bool PRE>      // PRE: the launch has a residual -- its rows (and the bias vectors) are requested early
__global__ void __launch_bounds__(768, 3)
    conv_wf4_kernel(const ConvArgs pin) {
    using Geo = Wf4Geo<GW>;
    constexpr int CT = kWf4CT;
    constexpr int NDMA = Geo::NDMA, BUF = Geo::BUF, HH = Geo::HH, HW = Geo::HW;
    constexpr int TWL2 = Geo::TWL2, THL2 = Geo::THL2, NIBL = Geo::NIBL;
    ConvArgs p = pin;
    int cshift = 0;
    if (!STATS && pin.ksplit > 1) cshift = split_k_args_wf4(p, blockIdx.y);
#if defined(ND_F4_DIAG)
    // diagnostic build only (tools/wf4_timeline.py): 10 ns stamps of wave 0 go to the buffer passed as `rowbias`, which is then ignored
    const unsigned long long dg_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long dg_t1 = 0, dg_t2 = 0, dg_t3 = 0, dg_ta = 0, dg_tb = 0;
    unsigned* const dg_buf = reinterpret_cast<unsigned*>(const_cast<float*>(p.rowbias));
    p.rowbias = nullptr;
#endif

    extern __shared__ __attribute__((aligned(16))) float smem_all[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    __builtin_amdgcn_s_setprio(3);                                      // prologue and epilogue: no MFMAs, raised priority
    // Twelve waves = TWO half blocks of six (measured, tools/micro/mfma_f32_rate.hip: a 6-wave workgroup puts its waves on the
    // SIMDs as 2 + 2 + 1 + 1 and so does the next one on the same CU -- 0.73 of the MFMA rate; 12 waves are 3 + 3 + 3 + 3).
    // The halves work on the SAME m tile and neighbouring n blocks (96 output channels per workgroup): ONE halo stream for
    // both (an LDS-DMA instruction costs its wave 60-180 cycles of issue; shared, each wave issues half as many per chunk, and
    // the input crosses L2 half as often); each half has its own weights, accumulators and its own half of the epilogue's LDS.
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int vb = wv >= 6 ? 1 : 0;                                     // half block
    const int xi = wv - 6 * vb;                                         // wave = row of the 6x6 transform
    float* const smem = smem_all;                                       // the two halo buffers (shared by the halves)

    const int total = gridDim.x;
    const int q = total >> 3, r = total & 7, xcd = blockIdx.x & 7;
    const int idp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    int mblk, nblk;
    tile_of(idp, p.mt, (p.nt + 1) >> 1, p.ngroup, mblk, nblk);
    nblk = nblk * 2 + vb;
    const bool active = nblk < p.nt;          // odd number of n blocks: the last half block recomputes the last one and stores nothing
    if (!active) nblk = p.nt - 1;
    const int bx = mblk % p.tiles_x;
    const int btmp = mblk / p.tiles_x;
    const int by = btmp % p.tiles_y;
    const int ig = btmp / p.tiles_y;
    const int img0 = ig << NIBL, oy0 = by * Geo::TH, ox0 = bx * Geo::TW;
    const int n0 = nblk * kWf4BN;

    // swizzle key of a 4-pixel group: a bijection of the block's 16 tiles for every fixed patch element
    auto key_of = [](int li, int hyq, int gxq) -> int {
        if constexpr (GW == 5) return ((hyq & 3) << 2) | (gxq & 3);
        else return ((li & 3) << 2) | ((hyq & 1) << 1) | (gxq & 1);
    };

    // ---- halo DMA descriptors: round k of wave wv fills the 64 16-byte units (k * 12 + wv) * 64 + lane
    constexpr unsigned kOOB = 0x80000000u;          // the host admits tensors of < 2 GiB
    unsigned vo[NDMA];
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
        const int U = (k * 12 + wv) * 64 + lane;
        const int G = U >> 4, u = U & 15;
        const int gx = G % GW;
        const int gt = G / GW;
        const int hy = gt % HH;
        const int li = gt / HH;
        const int s = u ^ key_of(li, hy >> 2, gx);
        const int hx = gx * 4 + (s >> 2);
        const int img = img0 + li;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        const bool ok = G < Geo::NG && hx < HW && img < p.NI && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const unsigned pix = (unsigned)(__mul24(__mul24(img, p.Hs) + (iy >> p.up), p.Ws) + (ix >> p.up));
        vo[k] = ok ? (__umul24(pix, (unsigned)p.ldx0 * 4u) + ((unsigned)(s & 3) << 4)) : kOOB;
    }
    const unsigned npix = (unsigned)(p.NI * p.Hs * p.Ws);
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x0) + cshift, 0,
                                                                        (int)(npix * (unsigned)p.ldx0 * 4u) - cshift * 4, 0x00020000);
    // one DMA round of chunk ch; chunks past the last one are still ISSUED -- the number of VMEM operations per chunk is a constant
    // the hand-counted waits rely on -- but with every lane out of range: zeros into a buffer nobody reads, no memory traffic
    auto halo_issue = [&](int k, int ch, int buf) {
#if defined(ND_F4ABL_NOHALO)
        return;
#endif
#if defined(ND_F4ABL_HALOHIT)
        const int che = 0;                // timing only: every chunk fetches chunk 0 (cache hits)
#else
        const int che = ch < p.NC32 - 1 ? ch : p.NC32 - 1;
#endif
        auto* dst = (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(smem) + buf * BUF + (k * 12 + wv) * 1024);
        const int vof = ch < p.NC32 ? (int)vo[k] : (int)kOOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, vof, che * 64, 0, 0);
    };

    // ---- patch read addresses.  Lane = (tile t = lane & 15, k group kq = lane >> 4): element (r, c) of the tile's 6x6 patch,
    //      channels 4 kq .. 4 kq + 3 of the chunk, lives at A[r >> 2][c >> 2][c & 3] + (r * GW + (c >> 2)) * 256.  Wave 5 reads
    //      rows 1, 3, 5 through the offsets of rows 0, 2, 4 (one row folded into the base).
    const bool typeA = (xi == 0) || (xi == 5);
    int A00[4], A01[2], A10[4], A11[2];
    {
        const int t = lane & 15, kq = lane >> 4;
        const int tli = t >> (THL2 + TWL2);
        const int tty = (t >> TWL2) & ((1 << THL2) - 1);
        const int ttx = t & ((1 << TWL2) - 1);
        // (LDS byte addresses: the dynamic segment starts at 0)
        const int G0 = ((tli * HH + 4 * tty) * GW + ttx + ((xi == 5) ? GW : 0)) * 256;
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            A00[cq] = G0 + ((((cq << 2) | kq) ^ key_of(tli, tty, ttx)) << 4);
            A10[cq] = G0 + ((((cq << 2) | kq) ^ key_of(tli, tty + 1, ttx)) << 4);
            if (cq < 2) {
                A01[cq] = G0 + ((((cq << 2) | kq) ^ key_of(tli, tty, ttx + 1)) << 4);
                A11[cq] = G0 + ((((cq << 2) | kq) ^ key_of(tli, tty + 1, ttx + 1)) << 4);
            }
        }
    }

#if defined(ND_F4ABL_ACF)
    // timing only: linear, conflict-free read addresses (wrong data)
#pragma unroll
    for (int cq = 0; cq < 4; ++cq) {
        A00[cq] = lane * 8 + cq * 512 + vb * kWf4ExchangeBytes;
        A10[cq] = lane * 8 + cq * 512 + 2048 + vb * kWf4ExchangeBytes;
        if (cq < 2) {
            A01[cq] = lane * 8 + cq * 512 + 4096 + vb * kWf4ExchangeBytes;
            A11[cq] = lane * 8 + cq * 512 + 5120 + vb * kWf4ExchangeBytes;
        }
    }
#endif
    // ---- weights: this wave's fragments of a chunk are 4 k4-steps x 6 positions, contiguous
    const size_t cstride = (size_t)p.nt * kWf4BlockChunk;
    const float* wwave = p.w + ((size_t)nblk * 6 + xi) * kWf4WaveChunk;
    const int voff = lane * (CT * 4);                  // byte offset of this lane inside a fragment

    f32x4 acc[6][CT];
#pragma unroll
    for (int nu = 0; nu < 6; ++nu)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[nu][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#if defined(ND_F4_LDX4)
    typedef f32x4 wfrag_t;
#else
    typedef f32x3 wfrag_t;
#endif
    wfrag_t wf[6];        // weight fragments of the current k4-step, re-loaded in place
#if defined(ND_F4ABL_NOB)
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) { wf[nu][0] = 1.f; wf[nu][1] = 0.5f; wf[nu][2] = 0.25f; }
#endif
    f32x2 v[6];           // transformed input of the current step (two k4-steps)

#define ND_SB __builtin_amdgcn_sched_barrier(0)

    auto ldfrag = [&](auto nuc, wfrag_t& d, const float* sbase) {
        constexpr int off = decltype(nuc)::value * kWf4Frag * 4;
        const int vof = voff;             // (named outside the asm statement: clang does not capture a variable a generic lambda only uses as an asm operand)
#if defined(ND_F4_LDX4)
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(vof), "s"(sbase), "i"(off));
#elif !defined(ND_F4ABL_NOB)
        asm volatile("global_load_dwordx3 %0, %1, %2 offset:%3" : "=v"(d) : "v"(vof), "s"(sbase), "i"(off));
#else
        asm volatile("" : "+v"(d) : "v"(vof), "s"(sbase), "i"(off));
#endif
    };
    auto wait_vm = [&](auto nc, wfrag_t& d) {
        constexpr int n = decltype(nc)::value;
#if !defined(ND_F4ABL_NOB) && !defined(ND_F4ABL_NOWAITVM)
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(d) : "i"(n));
#else
        asm volatile("" : "+v"(d) : "i"(n));
#endif
    };
    auto rd64 = [&](auto offc, int a) -> f32x2 {
        f32x2 d;
#if !defined(ND_F4ABL_NOA)
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(a), "i"(decltype(offc)::value));
#else
        d = f32x2{(float)a, 1.f};
#endif
        return d;
    };
#define ND_IC(x) std::integral_constant<int, (x)>{}

    // transform coefficients of this wave's row: rows 1..4: t = (d4 + pc d2) + qc (d3 + pc d1); rows 0 / 5: t = 4 dA - 5 dB + dC
    const wf4_u64 pc = wf4_pair((xi <= 2) ? -4.f : -1.f);
    const wf4_u64 qc = wf4_pair((xi == 1) ? 1.f : ((xi == 2) ? -1.f : ((xi == 3) ? 2.f : -2.f)));
    const wf4_u64 m5 = wf4_pair(-5.f);

    // the whole main loop, instantiated for the two row types (a wave-uniform branch selects)
    auto run = [&](auto tac) {
        constexpr bool TA = decltype(tac)::value;
        constexpr int NR = TA ? 3 : 4;                        // patch rows this wave reads
        f32x2 raw[2][4];          // two columns in flight
        f32x2 tt[6];              // row-transformed columns
        // issue the reads of column c of the patch at byte offset `base` (buffer + k8 half)
        auto issue_col = [&](auto cc, auto basec) {
            constexpr int c = decltype(cc)::value, base = decltype(basec)::value;
            constexpr int coff = (c >> 2) * 256 + base;
            const int a0 = (c < 4) ? A00[c & 3] : A01[c & 1];
            const int a1 = (c < 4) ? A10[c & 3] : A11[c & 1];
            if constexpr (TA) {
                raw[c & 1][0] = rd64(ND_IC(0 * GW * 256 + coff), a0);
                raw[c & 1][1] = rd64(ND_IC(2 * GW * 256 + coff), a0);
                raw[c & 1][2] = rd64(ND_IC(4 * GW * 256 + coff), a1);
            } else {
                raw[c & 1][0] = rd64(ND_IC(1 * GW * 256 + coff), a0);
                raw[c & 1][1] = rd64(ND_IC(2 * GW * 256 + coff), a0);
                raw[c & 1][2] = rd64(ND_IC(3 * GW * 256 + coff), a0);
                raw[c & 1][3] = rd64(ND_IC(4 * GW * 256 + coff), a1);
            }
        };
        // wait until at most `n` younger LDS reads are in flight, then the row part of column c
        auto row_col = [&](auto cc, auto nc) {
            constexpr int c = decltype(cc)::value, n = decltype(nc)::value;
            f32x2* d = raw[c & 1];
            if constexpr (TA) {
#if !defined(ND_F4ABL_NOA) && !defined(ND_F4ABL_NOWAITLGKM)
                asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]) : "i"(n));
#endif
#if !defined(ND_F4ABL_NOT)
                tt[c] = wf4_fma_p4(d[0], wf4_fma_s(m5, d[1], d[2]));
#else
                tt[c] = d[0];
                asm volatile("" :: "v"(d[1]), "v"(d[2]));
#endif
            } else {
#if !defined(ND_F4ABL_NOA) && !defined(ND_F4ABL_NOWAITLGKM)
                asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : "i"(n));
#endif
#if !defined(ND_F4ABL_NOT)
                const f32x2 a = wf4_fma_s(pc, d[1], d[3]);
                const f32x2 b = wf4_fma_s(pc, d[0], d[2]);
                tt[c] = wf4_fma_s(qc, b, a);
#else
                tt[c] = d[0];
                asm volatile("" :: "v"(d[1]), "v"(d[2]), "v"(d[3]));
#endif
            }
        };
        f32x2 ca, cb;             // shared terms of the column part
        auto col_part = [&](auto kc) {
            constexpr int k = decltype(kc)::value;
#if defined(ND_F4ABL_NOT)
            v[k] = tt[k];
            return;
#endif
            if constexpr (k == 0) v[0] = wf4_fma_p4(tt[0], wf4_fma_s(m5, tt[2], tt[4]));
            if constexpr (k == 1) { ca = wf4_fma_m4(tt[2], tt[4]); cb = wf4_fma_m4(tt[1], tt[3]); v[1] = wf4_add(ca, cb); }
            if constexpr (k == 2) v[2] = wf4_sub(ca, cb);
            if constexpr (k == 3) { ca = wf4_sub(tt[4], tt[2]); cb = wf4_sub(tt[3], tt[1]); v[3] = wf4_fma_p2(cb, ca); }
            if constexpr (k == 4) v[4] = wf4_fma_m2(cb, ca);
            if constexpr (k == 5) v[5] = wf4_fma_p4(tt[1], wf4_fma_s(m5, tt[3], tt[5]));
        };

        // ---- prologue: chunks 0 and 1, the fragments of k4-step 0, then the transform of chunk 0's first half
#if defined(ND_F4_DIAG)
        dg_ta = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int k = 0; k < NDMA; ++k) halo_issue(k, 0, 0);
#pragma unroll
        for (int k = 0; k < NDMA; ++k) halo_issue(k, 1, 1);
        wf4_sfor<0, 6>([&](auto nuc) { ldfrag(nuc, wf[decltype(nuc)::value], wwave); });
#if defined(ND_F4_DIAG)
        dg_tb = __builtin_amdgcn_s_memrealtime();
#endif
#if defined(ND_F4ABL_NOPROWAIT)
        asm volatile("s_waitcnt vmcnt(63)" ::: "memory");         // timing only: the prologue does not wait for its halo chunks
#else
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // both chunks have landed; the six fragment loads may be in flight
#endif
        __builtin_amdgcn_s_barrier();
#if defined(ND_F4_DIAG)
        dg_t1 = __builtin_amdgcn_s_memrealtime();
#endif
        wf4_sfor<0, 6>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            issue_col(cc, ND_IC(0));
            row_col(cc, ND_IC(0));
            (void)c;
        });
        wf4_sfor<0, 6>([&](auto kc) { col_part(kc); });
#if !defined(ND_F4_NOPRIO_B)
        __builtin_amdgcn_s_setprio(TA ? 0 : 1);          // the rows with the longer transform run ahead of rows 0 / 5: +1.7..2.3 %, interleaved A/B
#elif defined(ND_F4_PRIO_VB)
        if (vb) __builtin_amdgcn_s_setprio(1);           // experiment: one half block ahead of the other
        else __builtin_amdgcn_s_setprio(0);
#else
        __builtin_amdgcn_s_setprio(0);
#endif

        // one step = two k4-steps of chunk ch (S = 0: channels 0..7, S = 1: 8..15), 36 MFMAs in 12 slots of one position each.
        // Under them: the patch reads and the transform of the NEXT step (S = 0: this chunk's second half, S = 1: the next
        // chunk's first half from the other buffer), the fragment loads of the next k4-step, and in S = 1 the DMA of chunk
        // ch + 2 into this chunk's buffer, which the barrier behind S = 0 has released.
        auto step = [&](auto pcst, auto scst, int ch) {
            constexpr int P = decltype(pcst)::value, S = decltype(scst)::value;
            constexpr int RB = (S == 0) ? (P * BUF + 8) : ((1 - P) * BUF);
#if defined(ND_F4ABL_L1HIT)
            const float* wcur = p.w;          // timing only: every wave of the chip reads the same 18 KiB (L1 hits)
#elif defined(ND_F4ABL_BHIT)
            const float* wcur = wwave;        // timing only: the same fragments every chunk (cache hits)
#else
            const float* wcur = wwave + (size_t)ch * cstride;
#endif
            const float* wn0 = wcur + (2 * S + 1) * (6 * kWf4Frag);                                   // after k4-step 2 S
            const float* wn1 = (S == 0) ? (wcur + 2 * (6 * kWf4Frag)) : (wcur + cstride);             // after k4-step 2 S + 1
            wf4_sfor<0, 12>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int jj = i / 6, nu = i % 6;
                wait_vm(ND_IC(wf4_younger(S, i, NDMA)), wf[nu]);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[nu][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nu][ct], v[nu][jj], acc[nu][ct], 0, 0, 0);
                ND_SB;
#if ND_F4_LDEARLY
                ldfrag(ND_IC(nu), wf[nu], jj == 0 ? wn0 : wn1);
                ND_SB;
#endif
                if constexpr (i >= 2 && i <= 7) row_col(ND_IC(i - 2), ND_IC(i == 7 ? 0 : NR));
                if constexpr (i < 6) issue_col(ND_IC(i), ND_IC(RB));
                if constexpr (i >= 7) col_part(ND_IC(i - 7));
                if constexpr (S == 1 && i < NDMA) halo_issue(i, ch + 2, P);
                ND_SB;
#if !ND_F4_LDEARLY
                ldfrag(ND_IC(nu), wf[nu], jj == 0 ? wn0 : wn1);
                ND_SB;
#endif
            });
            col_part(ND_IC(5));
            ND_SB;
            if constexpr (S == 0) {
                // every read of this chunk's buffer has returned; chunk ch + 1 (issued a chunk ago, in front of 25 - NDMA younger
                // operations) has landed.  Behind the barrier the buffer of chunk ch is free for chunk ch + 2.
#if !defined(ND_F4ABL_NOBAR)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"i"(25 - NDMA - ND_F4_LDEARLY) : "memory");
                __builtin_amdgcn_s_barrier();
#endif
            }
        };
        for (int ch = 0; ch < p.NC32; ch += 2) {          // the host admits whole 32-channel chunks only: NC32 (16-channel chunks) is even
            step(ND_IC(0), ND_IC(0), ch);
            step(ND_IC(0), ND_IC(1), ch);
            step(ND_IC(1), ND_IC(0), ch + 1);
            step(ND_IC(1), ND_IC(1), ch + 1);
        }
    };
    if (typeA) run(std::true_type{});
    else run(std::false_type{});
#undef ND_SB
#if defined(ND_F4_DIAG)
    dg_t2 = __builtin_amdgcn_s_memrealtime();
#endif

    __builtin_amdgcn_s_setprio(3);
#if defined(ND_F4ABL_NOEPI)
    {
        float sacc = 0.f;
#pragma unroll
        for (int nu = 0; nu < 6; ++nu)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) sacc += acc[nu][ct][0] + acc[nu][ct][1] + acc[nu][ct][2] + acc[nu][ct][3];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[4]), "+v"(wf[5]) : : "memory");
        if (sacc == 123.456f) p.out[0] = sacc;
        return;
    }
#endif
    // ---- epilogue.  The run-ahead fragment loads of the last k4-step are still in flight and hipcc cannot know it (inline
    //      ISA): tying the fragments to the wait keeps their registers allocated until every load has returned.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[4]), "+v"(wf[5])
                 :
                 : "memory");
    // ---- the two units this wave finishes below (u = xi, xi + 6: output column b, n tile ct): their bias / per-image bias /
    //      residual vectors are requested NOW, so that their latency (the residual is an HBM miss) passes under the two
    //      barriers and the LDS round instead of in front of each unit's stores
    const int t = lane & 15, kq = lane >> 4;
    const int tli = t >> (THL2 + TWL2);
    const int tty = (t >> TWL2) & ((1 << THL2) - 1);
    const int ttx = t & ((1 << TWL2) - 1);
    const int img = img0 + tli;
    // (a separate instantiation: without a residual the unchanged code measured 1-2 % faster than this one with the
    // requests skipped at run time)
    constexpr bool early = PRE;
    f32x4 pre_b[2], pre_rb[2], pre_r[2][4];
    // (inline ISA: as ordinary loads hipcc moved a loaded register right behind one of them, i.e. put an s_waitcnt vmcnt(0) --
    // a full memory latency -- into the middle of the requests; tools/wf4_timeline.py showed it as 3 us between the loop's end
    // and the exchange barrier)
    auto ld128 = [&](f32x4& d, const float* ptr) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(ptr)); };
    auto load_res = [&](int uu, int nb, int ox) {
        if (!p.res) return;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oy = oy0 + 4 * tty + a;
            if (oy < p.H) {
                const size_t rpx = p.res_up ? ((size_t)(img * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1))
                                            : ((size_t)(img * p.H + oy) * p.W + ox);
                ld128(pre_r[uu][a], p.res + rpx * p.ldr + nb);
            }
        }
    };
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
        const int u = xi + 6 * uu;
        const int b = u & 3, ct = u >> 2;
        const int nb = n0 + ct * 16 + 4 * kq;
        const int ox = ox0 + 4 * ttx + b;
        pre_b[uu] = pre_rb[uu] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < 4; ++a) pre_r[uu][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (early && active && p.vec_ok && nb + 3 < p.N && img < p.NI && ox < p.W) {
            if (p.bias) ld128(pre_b[uu], p.bias + nb);
            if (p.rowbias) ld128(pre_rb[uu], p.rowbias + (size_t)img * p.ld_rowbias + nb);
            if (uu == 0) load_res(0, nb, ox);
        }
    }
    __builtin_amdgcn_s_barrier();                                    // every wave is done with the halo buffers
    f32x4* ex = reinterpret_cast<f32x4*>(smem_all + vb * (kWf4ExchangeBytes / 4));          // this half's [xi][b][ct][lane]
    // M[xi][nu] -> r[b] = sum_nu At[b][nu] M[xi][nu], At = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]]
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const f32x4 s12 = acc[1][ct] + acc[2][ct], d12 = acc[1][ct] - acc[2][ct];
        const f32x4 s34 = acc[3][ct] + acc[4][ct], d34 = acc[3][ct] - acc[4][ct];
        ex[((xi * 4 + 0) * CT + ct) * 64 + lane] = (acc[0][ct] + s12) + s34;
        ex[((xi * 4 + 1) * CT + ct) * 64 + lane] = 2.f * d34 + d12;
        ex[((xi * 4 + 2) * CT + ct) * 64 + lane] = 4.f * s34 + s12;
        ex[((xi * 4 + 3) * CT + ct) * 64 + lane] = (8.f * d34 + d12) + acc[5][ct];
    }
    {
        // the accumulators are dead: the second unit's residual rows fit now
        const int u = xi + 6;
        const int b = u & 3, ct = u >> 2;
        const int nb = n0 + ct * 16 + 4 * kq;
        const int ox = ox0 + 4 * ttx + b;
        if (PRE && active && p.vec_ok && nb + 3 < p.N && img < p.NI && ox < p.W) load_res(1, nb, ox);
    }
    if constexpr (PRE) {
        // (not __syncthreads(): its fence would wait for the prefetched vectors as well)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // the requested vectors: in flight since before the column transform / the LDS round; tied to the wait
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(pre_b[0]), "+v"(pre_b[1]), "+v"(pre_rb[0]), "+v"(pre_rb[1]), "+v"(pre_r[0][0]), "+v"(pre_r[0][1]), "+v"(pre_r[0][2]),
                       "+v"(pre_r[0][3]), "+v"(pre_r[1][0]), "+v"(pre_r[1][1]), "+v"(pre_r[1][2]), "+v"(pre_r[1][3]));
    } else {
        __syncthreads();
    }
#if defined(ND_F4_DIAG)
    dg_t3 = __builtin_amdgcn_s_memrealtime();
#endif
    // wave w finishes units u = w and w + 6 of the 12 (output column b, n tile ct): Y[a][b] = sum_xi At[a][xi] r_xi[b]
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
        const int u = xi + 6 * uu;
        const int b = u & 3, ct = u >> 2;
        f32x4 x[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) x[k] = ex[((k * 4 + b) * CT + ct) * 64 + lane];
        const f32x4 s12 = x[1] + x[2], d12 = x[1] - x[2], s34 = x[3] + x[4], d34 = x[3] - x[4];
        f32x4 yv[4];
        yv[0] = (x[0] + s12) + s34;
        yv[1] = 2.f * d34 + d12;
        yv[2] = 4.f * s34 + s12;
        yv[3] = (8.f * d34 + d12) + x[5];
        const int nb = n0 + ct * 16 + 4 * kq;          // first of this lane's 4 output channels
        const int ox = ox0 + 4 * ttx + b;
        f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
        if (active && nb < p.N && img < p.NI && ox < p.W) {
            const bool vec = p.vec_ok && (nb + 3 < p.N);
            f32x4 bv = pre_b[uu], rbv = pre_rb[uu];
            if (vec && !early) {
                if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + nb);
                if (p.rowbias) rbv = *reinterpret_cast<const f32x4*>(p.rowbias + (size_t)img * p.ld_rowbias + nb);
            }
            if (p.bias) {
                if (!vec) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (nb + c < p.N) bv[c] = p.bias[nb + c];
                }
            }
            if (p.rowbias) {
                const float* rbp = p.rowbias + (size_t)img * p.ld_rowbias + nb;
                if (!vec) {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (nb + c < p.N) rbv[c] = rbp[c];
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int oy = oy0 + 4 * tty + a;
                if (oy < p.H) {
                    f32x4 o = yv[a];
                    float* op = p.out + ((size_t)(img * p.H + oy) * p.W + ox) * p.ldo + nb;
                    const float* rp = nullptr;
                    if (p.res) {
                        const size_t rpx = p.res_up ? ((size_t)(img * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1))
                                                    : ((size_t)(img * p.H + oy) * p.W + ox);
                        rp = p.res + rpx * p.ldr + nb;
                    }
                    if (vec) {
                        // the association of the F(2x2) kernels' vector path: ((y + bias) + rowbias) + residual
                        if (p.bias) o += bv;
                        if (p.rowbias) o += rbv;
                        if (rp) o += pre_r[uu][a];
                        if (p.silu_out) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) o[c] = fast_silu(o[c]);
                        }
#if defined(ND_F4ABL_NOSTORE)
                        if (o[0] == 123.456f) *reinterpret_cast<f32x4*>(op) = o;          // timing only
#elif defined(ND_F4ABL_COALSTORE)
                        // timing only (wrong placement): the same bytes per workgroup, every store instruction one contiguous KiB
                        *reinterpret_cast<f32x4*>(p.out + ((size_t)blockIdx.x * 96 + (wv * 8 + uu * 4 + a)) * 256 + lane * 4) = o;
#else
                        *reinterpret_cast<f32x4*>(op) = o;
#endif
                        if constexpr (STATS) {
                            ssum += o;
                            ssq += o * o;
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (nb + c < p.N) {
                                float v2 = o[c];
                                if (p.bias) v2 += bv[c];
                                if (p.rowbias) v2 += rbv[c];
                                if (rp) v2 += rp[c];
                                if (p.silu_out) v2 = fast_silu(v2);
                                op[c] = v2;
                                if constexpr (STATS) {
                                    ssum[c] += v2;
                                    ssq[c] += v2 * v2;
                                }
                            }
                        }
                    }
                }
            }
        }
        if constexpr (STATS) {
            // GroupNorm statistics of the output: per-channel sum / sum of squares over this block's pixels of output column b
            // of each image, one row per (image, m block, b): chstats [NI][mbi * 4][sum | sum of squares][N], plain stores,
            // every entry written by every launch.  The tiles of an image are the 16 lanes of a DPP row (one image per block)
            // or a quad (four images per block).
            bool writer;
            if constexpr (NIBL == 0) {
                sum8_over_16_lanes(ssum, ssq);
                writer = t == 0;
            } else {
#pragma unroll
                for (int m = 1; m < 4; m <<= 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        ssum[c] += __shfl_xor(ssum[c], m);
                        ssq[c] += __shfl_xor(ssq[c], m);
                    }
                }
                writer = (t & 3) == 0;
            }
            const int mb = (NIBL == 0) ? (by * p.tiles_x + bx) : 0;
            if (active && writer && img < p.NI && nb < p.N) {
                float* ps = p.chstats + ((((size_t)img * p.mbi + mb) * 4 + b) * 2) * p.N + nb;
                if (nb + 3 < p.N && (p.N & 3) == 0) {
                    *reinterpret_cast<f32x4*>(ps) = ssum;
                    *reinterpret_cast<f32x4*>(ps + p.N) = ssq;
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (nb + c < p.N) {
                            ps[c] = ssum[c];
                            ps[p.N + c] = ssq[c];
                        }
                    }
                }
            }
        }
    }
#if defined(ND_F4_DIAG)
    if (wv == 0 && lane == 0 && dg_buf) {
        const unsigned long long dg_t4 = __builtin_amdgcn_s_memrealtime();
        unsigned* d = dg_buf + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        d[0] = (unsigned)dg_t0;
        d[1] = (unsigned)(dg_t1 - dg_t0) | ((unsigned)(dg_ta - dg_t0) << 16);          // (spans < 655 us: 16 bits each)
        d[2] = (unsigned)(dg_t2 - dg_t0);
        d[6] = (__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xffffu) | ((unsigned)(dg_tb - dg_t0) << 16);          // XCC_ID | issue done
        d[3] = (unsigned)(dg_t3 - dg_t0);
        d[4] = (unsigned)(dg_t4 - dg_t0);
        d[5] = __builtin_amdgcn_s_getreg(4 | (31 << 11));           // HW_ID
        d[7] = (unsigned)(dg_t0 >> 32);
    }
#endif
}
#undef ND_IC

// OIHW 3x3 weights -> F(4x4,3x3) domain U = G g G^T (float64, rounded once), fragment order
// [chunk16][n block][xi][k4 step j][nu][lane][ct]: lane = (out channel m = lane & 15 of n tile ct, k group kq = lane >> 4),
// input channel chunk * 16 + 4 kq + j -- the k order of the ds_read_b64 pairs the kernel feeds the MFMAs with
__global__ void pack_wf4_weight_kernel(const float* w, float* out, int N, int C, int nt, long total) {
    const double G[6][3] = {{0.25, 0.0, 0.0},
                            {-1.0 / 6, -1.0 / 6, -1.0 / 6},
                            {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6},
                            {1.0 / 24, -1.0 / 12, 1.0 / 6},
                            {0.0, 0.0, 1.0}};
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        long rr = it;
        const int ct = (int)(rr % kWf4CT);
        rr /= kWf4CT;
        const int lane = (int)(rr & 63);
        rr >>= 6;
        const int nu = (int)(rr % 6);
        rr /= 6;
        const int j = (int)(rr & 3);
        rr >>= 2;
        const int xi = (int)(rr % 6);
        rr /= 6;
        const int nblk = (int)(rr % nt);
        const int cq = (int)(rr / nt);
        const int n = nblk * kWf4BN + ct * 16 + (lane & 15);
        const int c = cq * 16 + 4 * (lane >> 4) + j;
        double u = 0.0;
        if (n < N && c < C) {
            const float* g = w + ((size_t)n * C + c) * 9;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) u += G[xi][a] * (double)g[a * 3 + b] * G[nu][b];
        }
        out[it] = (float)u;
    }
}

template <int GW>
static int launch_wf4(const ConvArgs& a, int grid, hipStream_t s) {
    const char* fn = "nd_conv3x3_winograd_f4_nhwc";
    const size_t lds = 2 * kWf4ExchangeBytes;          // two half blocks
    const dim3 g(grid, a.ksplit > 1 ? a.ksplit : 1);
    auto go = [&](auto kern, bool (&attr_set)[kMaxDevices]) -> int {
        if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), attr_set, fn)) return rc;
        hipLaunchKernelGGL(kern, g, dim3(768), lds, s, a);
        return check_launch(fn);
    };
    static bool set_sr[kMaxDevices] = {}, set_s[kMaxDevices] = {}, set_r[kMaxDevices] = {}, set_0[kMaxDevices] = {};
    if (a.chstats) return a.res ? go(conv_wf4_kernel<GW, true, true>, set_sr) : go(conv_wf4_kernel<GW, true, false>, set_s);
    return a.res ? go(conv_wf4_kernel<GW, false, true>, set_r) : go(conv_wf4_kernel<GW, false, false>, set_0);
}

// block geometry for a map: 5 = 16x16-pixel regions of one image, 3 = four 8x8 images, 0 = this kernel does not take it
static int wf4_geometry(int H, int W) {
    if (H <= 0 || W <= 0 || (H & 3) || (W & 3)) return 0;
    if (H == 8 && W == 8) return 3;
    if (H >= 12 && W >= 12) return 5;
    return 0;
}

}  // namespace nd

using namespace nd;

extern "C" int nd_conv_winograd_f4_num_variants(void) { return 1; }

extern "C" const char* nd_conv_winograd_f4_variant_name(int variant) { return variant == 0 ? "nd::conv_wf4_kernel" : ""; }

extern "C" int nd_conv_winograd_f4_variant_info(int variant, int* bm, int* bn, int* threads) {
    if (variant != 0) return ND_E_ARG;
    if (bm) *bm = 256;
    if (bn) *bn = 2 * kWf4BN;
    if (threads) *threads = 768;
    return ND_OK;
}

extern "C" int64_t nd_conv_winograd_f4_weight_floats(int variant, int N, int C) {
    if (variant != 0 || N <= 0 || C <= 0) return ND_E_ARG;
    return (int64_t)((C + 15) / 16 + wstream::kWf4PadChunks) * ((N + kWf4BN - 1) / kWf4BN) * kWf4BlockChunk;
}

extern "C" int64_t nd_conv_winograd_f4_max_weight_read(int variant, int N, int C) {
    if (variant != 0 || N <= 0 || C <= 0) return ND_E_ARG;
    static_assert(wstream::pad_chunks(wstream::kWf4Ahead, wstream::kWf4StepsPerChunk) <= wstream::kWf4PadChunks,
                  "weight read-ahead exceeds the packer's zero padding");
    return (int64_t)((C + 15) / 16 + wstream::pad_chunks(wstream::kWf4Ahead, wstream::kWf4StepsPerChunk)) *
           ((N + kWf4BN - 1) / kWf4BN) * kWf4BlockChunk;
}

extern "C" int nd_repack_conv_weight_winograd_f4(const float* w_oihw, float* w_out, int N, int C, int variant, nd_stream_t stream) {
    const char* fn = "nd_repack_conv_weight_winograd_f4";
    ND_REQUIRE(w_oihw && w_out && N > 0 && C > 0 && variant == 0, fn, "bad arguments");
    const long total = (long)nd_conv_winograd_f4_weight_floats(variant, N, C);
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_wf4_weight_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w_oihw, w_out,
                       N, C, (N + kWf4BN - 1) / kWf4BN, total);
    return check_launch(fn);
}

// rows per image of the partial statistics the epilogue leaves behind ([NI][rows][sum | sum of squares][N], fp32): one per
// (m block of the image, output column b); 0 = the kernel does not take this map
extern "C" int nd_conv_winograd_f4_stats_rows(int variant, int NI, int H, int W) {
    if (variant != 0 || NI <= 0) return ND_E_ARG;
    const int gw = wf4_geometry(H, W);
    if (gw == 0) return 0;
    return gw == 5 ? ((W + 15) / 16) * ((H + 15) / 16) * 4 : 4;
}

// the same for a launch split over K: the rows come from the reduce pass, one per run of 16 pixels of an image (0: the map's
// pixel count is not a multiple of 16 -- such a launch cannot take chstats)
extern "C" int nd_conv_winograd_f4_splitk_stats_rows(int variant, int NI, int H, int W) {
    if (variant != 0 || NI <= 0) return ND_E_ARG;
    if (wf4_geometry(H, W) == 0 || ((long)H * W) % kSplitkStatsPixels) return 0;
    return (int)(((long)H * W) / kSplitkStatsPixels);
}

extern "C" int nd_conv3x3_winograd_f4_nhwc(const float* x0, int C0, int ldx0, const float* w, const float* bias,
                                           const float* rowbias, int ld_rowbias, const float* residual, int ldr, float* out,
                                           int ldo, int NI, int H, int W, int N, int flags, int variant, float* chstats, int splits,
                                           float* workspace, nd_stream_t stream) {
    const char* fn = "nd_conv3x3_winograd_f4_nhwc";
    ND_REQUIRE(x0 && w && out, fn, "null pointer");
    ND_REQUIRE(variant == 0, fn, "bad variant");
    ND_REQUIRE(NI > 0 && H > 0 && W > 0 && N > 0 && C0 > 0, fn, "bad shape");
    const int gw = wf4_geometry(H, W);
    ND_REQUIRE(gw != 0, fn, "Winograd F(4x4,3x3) takes H and W that are multiples of 4, at least 12x12 or exactly 8x8");
    ND_REQUIRE((C0 & 31) == 0 && (ldx0 & 3) == 0 && ldx0 >= C0 && ldo >= N, fn, "whole 32-channel chunks; strides multiples of 4");
    ND_REQUIRE(aligned16(x0) && aligned16(w), fn, "x0 / w must be 16-byte aligned");
    ND_REQUIRE(!(flags & ~(ND_CONV_IN_UP2X | ND_CONV_RES_UP2X | ND_CONV_SILU_OUT)), fn, "unsupported flag");
    if (flags & ND_CONV_SILU_OUT) ND_REQUIRE(residual == nullptr, fn, "SILU_OUT with a residual is not supported");
    if (residual) ND_REQUIRE(ldr >= N, fn, "ldr < N");
    if (rowbias) ND_REQUIRE(ld_rowbias >= N, fn, "ld_rowbias < N");
    const int up = (flags & ND_CONV_IN_UP2X) ? 1 : 0;
    // the input is addressed through a buffer descriptor: 32-bit byte offsets with the range check as zero padding
    ND_REQUIRE((long)NI * H * W < (1L << 31) / 2, fn, "too many pixels");
    ND_REQUIRE((long)NI * (H >> up) * (W >> up) < (1L << 24) && ldx0 < (1 << 22), fn, "24-bit multiplies for pixel indices");
    ND_REQUIRE((long)NI * (H >> up) * (W >> up) * ldx0 * 4 < (1L << 31), fn, "input tensors of less than 2 GiB");
    ConvArgs a{};
    a.ksplit = 1; a.kchunks = 0; a.ws_stride = 0;
    if (splits > 1) {
        ND_REQUIRE(splits <= 16 && workspace != nullptr && aligned16(workspace), fn, "split-K: 2..16 splits, a 16-byte aligned workspace");
        if (chstats) ND_REQUIRE(((long)H * W) % kSplitkStatsPixels == 0 && aligned16(chstats), fn,
                                "split-K with output statistics: H * W must be a multiple of 16");
        ND_REQUIRE(!(flags & ND_CONV_RES_UP2X) && (N & 3) == 0 && (ldo & 3) == 0 && aligned16(out) && (!bias || aligned16(bias)) &&
                   (!residual || ((ldr & 3) == 0 && aligned16(residual))) &&
                   (!rowbias || ((ld_rowbias & 3) == 0 && aligned16(rowbias))), fn,
                   "split-K: N and the strides must be multiples of 4 with 16-byte aligned pointers; no 2x-upsampled residual");
        int kc = 0;
        const int S = splitk_plan_f32(C0, 3, splits, &kc);      // whole 32-channel chunks per split
        ND_REQUIRE(S > 1, fn, "split-K: too few input channels for that many splits");
        a.ksplit = S; a.kchunks = kc * 2; a.ws_stride = (long)NI * H * W * N;
    }
    a.x0 = x0; a.x1 = x0; a.w = w; a.bias = bias; a.rowbias = rowbias; a.res = residual; a.out = out;
    a.C0 = C0; a.C1 = 0; a.ldx0 = ldx0; a.ldx1 = ldx0;
    a.NI = NI; a.H = H; a.W = W; a.up = up; a.res_up = (flags & ND_CONV_RES_UP2X) ? 1 : 0;
    a.Hs = H >> up; a.Ws = W >> up;
    a.N = N; a.ldo = ldo; a.ldr = ldr; a.ld_rowbias = ld_rowbias;
    a.NT32 = (N + 31) / 32;
    a.NC32 = C0 / 16;                     // this kernel counts 16-channel chunks
    const int TW = gw == 5 ? 16 : 8, NIB = gw == 5 ? 1 : 4;
    a.thl = a.twl = gw == 5 ? 4 : 3; a.nibl = gw == 5 ? 0 : 2;
    a.tiles_x = (W + TW - 1) / TW; a.tiles_y = (H + TW - 1) / TW;
    a.mt = a.tiles_x * a.tiles_y * ((NI + NIB - 1) / NIB);
    a.nt = (N + kWf4BN - 1) / kWf4BN;
    // (the block -> tile order walks PAIRS of n blocks: tile_of's n axis is (nt + 1) / 2 long)
    a.ngroup = pick_ngroup((a.nt + 1) / 2, (size_t)2 * kWf4BN * C0 * 36 * sizeof(float), (size_t)NI * (H >> up) * (W >> up) * C0 * sizeof(float));
    a.vec_ok = ((ldo & 3) == 0 && aligned16(out) && (!bias || aligned16(bias)) &&
                (!residual || ((ldr & 3) == 0 && aligned16(residual))) &&
                (!rowbias || ((ld_rowbias & 3) == 0 && aligned16(rowbias)))) ? 1 : 0;
    a.silu_out = (flags & ND_CONV_SILU_OUT) ? 1 : 0;
    if (a.ksplit > 1) {
        a.bias = nullptr; a.rowbias = nullptr; a.res = nullptr; a.out = workspace; a.ldo = N; a.silu_out = 0; a.res_up = 0;
        a.vec_ok = 1;
    }
    a.chstats = (a.ksplit > 1) ? nullptr : chstats;          // split over K: the reduce pass writes them
    a.mbi = (gw == 5) ? a.tiles_x * a.tiles_y : 1;
    if (chstats) ND_REQUIRE(ldo == N, fn, "output statistics need ldo == N");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = a.mt * ((a.nt + 1) / 2);          // a workgroup = one m tile x two neighbouring n blocks
    const int rc = gw == 5 ? launch_wf4<5>(a, grid, s) : launch_wf4<3>(a, grid, s);
    if (rc != ND_OK || a.ksplit <= 1) return rc;
    if (chstats)
        return launch_splitk_reduce_stats_f32(workspace, a.ksplit, a.ws_stride, NI, H * W, N, bias, rowbias, ld_rowbias, residual, ldr, out, ldo,
                                              (flags & ND_CONV_SILU_OUT) ? 1 : 0, chstats, s);
    return launch_splitk_reduce_f32(workspace, a.ksplit, a.ws_stride, (long)NI * H * W, N, bias, rowbias, ld_rowbias, H * W, residual,
                                    ldr, out, ldo, (flags & ND_CONV_SILU_OUT) ? 1 : 0, s);
}

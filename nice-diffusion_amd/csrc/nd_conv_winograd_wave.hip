// K5, Winograd F(2x2,3x3), whole-transform-per-wave form (variant 11 of nd_conv3x3_winograd_nhwc).
//
// conv_wino16_kernel gives every wave ONE of the 16 transform positions, so the output transform A^T M A has to gather
// 16 waves' accumulators through LDS (three rounds, barriers) and the 16-wave block owns its CU: nothing overlaps that
// epilogue or the next block's prologue, and the MFMA + epilogue skeleton alone tops out at 79 % of the matrix peak
// (DESIGN.md section 6).  Here a wave owns ALL 16 positions of its 32 tiles x 32 channels -- 16 x 16 = 256 accumulator
// registers, which is what the AccVGPR half of the unified register file is for -- so
//   * the output transform is register arithmetic in the epilogue: no exchange, no barrier;
//   * a block is 4 waves (one per SIMD): 2 tile halves x 2 n tiles = 256 output pixels x 64 channels, so its weight
//     slab is re-read once per 64 tiles instead of once per 32 (half the L2 traffic per flop of the position-split form);
//   * per half-step (4 input channels) a wave issues 32 MFMAs (2048 matrix-pipe cycles) against 16 8-byte LDS reads of
//     the raw 4x4 patch, 32 adds for B^T d B, and 16 8-byte weight-fragment loads; everything for half-step h + 1 is
//     issued under the MFMAs of half-step h (one wave per SIMD: the overlap has to come from this wave's own
//     instruction stream).
// Weight layout, halo image in LDS, k order and the association of every sum are those of conv_wino16_kernel: the two
// kernels give the same bits.
#include "nd_conv_common.h"

#if defined(ND_EXPERIMENTAL_KERNELS)      // an experiment that measures slower: built with `make EXPERIMENTAL=1` only

#ifndef ND_WINOW_SCHED
#define ND_WINOW_SCHED 1
#endif

namespace nd {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256, 1)
    conv_winow_kernel(const ConvArgs p) {
    constexpr int NT = 256;
    constexpr int BN = 64;
    constexpr int ROWF = 32, SPR = 8;
    constexpr int HPMAX = 416;
    constexpr int MAXHI = (HPMAX * SPR + NT - 1) / NT;      // 13 halo items (16 bytes each) per thread and chunk
    constexpr int HSTEPS = 8;                               // half-steps (4 channels) per 32-channel chunk

    // halo buffer 0 at byte 0, buffer 1 at byte 65536 (HP * 128 <= 53 KiB): the buffer is ONE bit of a patch read's address, XORed
    // in together with the k-step, so a read costs one VALU instruction for its address
    constexpr int BUF1 = 16384;                             // floats
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1;           // which 32 of the block's 64 tiles
    const int wn = wave & 1;            // which of the block's two n tiles
    const int l31 = lane & 31;
    const int lh = lane >> 5;

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
    const int HH = TH + 2, HW = TW + 2;
    const int HPI = HH * HW;
    const int HP = HPI << p.nibl;
    const int img0 = ig << p.nibl, oy0 = ty << p.thl, ox0 = tx << p.twl;
    const int n0 = nblk * BN + wn * 32;

    // halo image in LDS: the one of conv_wino_kernel / conv_wino16_kernel (128-byte pixel rows, 16-byte slots XOR-swizzled
    // by the pixel's 2x2-tile coordinates so that the 32 lanes of a patch read hit different banks)
    auto lds_off = [&](int hp, int hy, int hx, int slot) -> int {
        const int key = (((hy >> 1) & 3) << 2) | ((hx >> 1) & 3);
        return (hp >> 1) * 64 + (((((hp & 1) << 3) | slot) ^ key) << 2);
    };
    const int hslot = tid % SPR;
    const int hrow0 = tid / SPR;
    // halo descriptors.  The main loop is kept free of branches (one basic block per chunk, so that the instruction order
    // below is the order the hardware sees): a padding pixel, a slot past the last channel and an item past the last halo
    // pixel all LOAD 16 bytes of zeros from p.zero, and an item past the last halo pixel STORES into a spare 16 bytes
    // behind the two halo buffers.
    int gpix[MAXHI];
    unsigned hoff2[(MAXHI + 1) / 2];          // LDS float offsets (buffer 0) of items 2i | 2i + 1, 16 bits each
    const int spare = BUF1 + HP * ROWF;       // < 2^15
#pragma unroll
    for (int k = 0; k < MAXHI; ++k) {
        const int hp = hrow0 + k * (NT / SPR);
        int g = -1, ho = spare;
        if (hp < HP) {
            const int li = hp / HPI;
            const int rem = hp - li * HPI;
            const int hy = rem / HW;
            const int hx = rem - hy * HW;
            const int img = img0 + li;
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            if (img < p.NI && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                g = (img * p.Hs + (iy >> p.up)) * p.Ws + (ix >> p.up);
            ho = lds_off(hp, hy, hx, hslot);
        }
        gpix[k] = g;
        if (k & 1) hoff2[k >> 1] |= (unsigned)ho << 16;
        else hoff2[k >> 1] = (unsigned)ho;
    }
    const int Ctot = p.C0 + p.C1;
    const int nchunks = p.NC32;

    auto load_halo_pixel = [&](int g, int ch) -> f32x4 {
        const int c = ch * ROWF + (hslot << 2);
        const float* src = (c < p.C0) ? (p.x0 + (size_t)g * p.ldx0 + c) : (p.x1 + (size_t)g * p.ldx1 + (c - p.C0));
        src = (g >= 0 && c < Ctot) ? src : p.zero;
        return *reinterpret_cast<const f32x4*>(src);
    };
    auto store_halo_item = [&](int k, int buf, f32x4 v) {
        const int ho = (k & 1) ? (int)(hoff2[k >> 1] >> 16) : (int)(hoff2[k >> 1] & 0xffffu);
        *reinterpret_cast<f32x4*>(smem + (ho == spare ? spare : buf * BUF1 + ho)) = v;
    };

    // this lane's tile and the LDS byte offsets of its 4x4 input patch d[a][b] (16-byte slot `lh`, k-step 0, buffer 0)
    const int twl2 = p.twl - 1, thl2 = p.thl - 1;
    const int te = wm * 32 + l31;
    const int t_li = te >> (thl2 + twl2);
    const int t_y = (te >> twl2) & ((1 << thl2) - 1);
    const int t_x = te & ((1 << twl2) - 1);
    int off16[16];
    {
        const int base = t_li * HPI + (2 * t_y) * HW + 2 * t_x;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) off16[a * 4 + b] = 4 * lds_off(base + a * HW + b, 2 * t_y + a, 2 * t_x + b, lh);      // BYTES
    }

    // weight fragments [c32][n tile][kc][position][lane][4] of this wave's n tile (N tail: clamped, dropped in the epilogue)
    int ntile = nblk * 2 + wn;
    if (ntile > p.NT32 - 1) ntile = p.NT32 - 1;
    const float* bp = p.w + (size_t)ntile * (64 * 256) + lane * 4;
    const size_t c32_stride = (size_t)p.NT32 * (64 * 256);

    f32x16 acc[16];
#pragma unroll
    for (int ps = 0; ps < 16; ++ps)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ps][e] = 0.f;

    // half-step h of chunk ch = channels 32 ch + 8 (h >> 1) + 4 lh + 2 (h & 1) + {0, 1}: the two floats at 2 (h & 1) of
    // the 16-byte fragment / patch slot of k-step h >> 1
    // ONE buffer of weight fragments: position ps's pair is refetched for the next half-step right after this half-step's two
    // MFMAs on it have been issued, i.e. 30 MFMAs (~1900 matrix-pipe cycles) before it is needed again
    f32x2 wb[16];
    auto w_ptr = [&](int c32, int h) -> const float* {
        return bp + (size_t)c32 * c32_stride + (h >> 1) * (16 * 256) + 2 * (h & 1);
    };
    f32x2 raw[16];
    auto load_raw = [&](int buf, int h) {          // prologue only; the main loop spreads these reads over its slots
        const int kx = 4 * (((h >> 1) << 3) | (buf * BUF1));
        const char* base = reinterpret_cast<const char*>(smem) + 8 * (h & 1);
#pragma unroll
        for (int i = 0; i < 16; ++i) raw[i] = *reinterpret_cast<const f32x2*>(base + (off16[i] ^ kx));
    };
#if defined(ND_WINOW_PK)
    auto padd = [](f32x2 a, f32x2 b) -> f32x2 {
        f32x2 d;
        asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
        return d;
    };
    auto psub = [](f32x2 a, f32x2 b) -> f32x2 {
        f32x2 d;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        return d;
    };
#else
    // two scalar adds per pair, spelled out: beside MFMAs a v_pk_add_f32 costs the issuing wave ~13 cycles against 4 for a
    // v_add_f32 (MI355X_MICROARCH.md, 'price of one filler beside MFMAs'), and hipcc packs adjacent scalar adds on its own
    auto padd = [](f32x2 a, f32x2 b) -> f32x2 {
        f32x2 d;
        asm("v_add_f32 %0, %1, %2" : "=v"(d[0]) : "v"(a[0]), "v"(b[0]));
        asm("v_add_f32 %0, %1, %2" : "=v"(d[1]) : "v"(a[1]), "v"(b[1]));
        return d;
    };
    auto psub = [](f32x2 a, f32x2 b) -> f32x2 {
        f32x2 d;
        asm("v_sub_f32 %0, %1, %2" : "=v"(d[0]) : "v"(a[0]), "v"(b[0]));
        asm("v_sub_f32 %0, %1, %2" : "=v"(d[1]) : "v"(a[1]), "v"(b[1]));
        return d;
    };
#endif
    // V = B^T d B, one position at a time straight from the raw patch (no table of row transforms to keep alive):
    // V[xi][nu] = (d[ra][ca] +- d[rb][ca]) +- (d[ra][cb] +- d[rb][cb]) with, for index 0..3 of a row or column,
    // (first, second, sign) = (0,2,-) (1,2,+) (2,1,-) (1,3,-) -- conv_wino16_kernel's formula and association.
    // ONE buffer: position ps of the next half-step overwrites V[ps] once this half-step's two MFMAs on it are issued.
    f32x2 V[16];
    auto transform_pos = [&](int ps) {
        const int xi = ps >> 2, nu = ps & 3;
        const int ra = (xi == 0) ? 0 : ((xi == 2) ? 2 : 1);
        const int rb = (xi == 0) ? 2 : ((xi == 1) ? 2 : ((xi == 2) ? 1 : 3));
        const int ca = (nu == 0) ? 0 : ((nu == 2) ? 2 : 1);
        const int cb = (nu == 0) ? 2 : ((nu == 1) ? 2 : ((nu == 2) ? 1 : 3));
        const f32x2 ta = (xi == 1) ? padd(raw[ra * 4 + ca], raw[rb * 4 + ca]) : psub(raw[ra * 4 + ca], raw[rb * 4 + ca]);
        const f32x2 tb = (xi == 1) ? padd(raw[ra * 4 + cb], raw[rb * 4 + cb]) : psub(raw[ra * 4 + cb], raw[rb * 4 + cb]);
        V[ps] = (nu == 1) ? padd(ta, tb) : psub(ta, tb);
    };

    // ---- prologue: chunk 0's halo, first weights, first transformed patch
#pragma unroll
    for (int k = 0; k < MAXHI; ++k) store_halo_item(k, 0, load_halo_pixel(gpix[k], 0));
    {
        const float* qq = w_ptr(0, 0);
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) wb[ps] = *reinterpret_cast<const f32x2*>(qq + ps * 256);
    }
    __syncthreads();
    load_raw(0, 0);
#pragma unroll
    for (int ps = 0; ps < 16; ++ps) transform_pos(ps);

    for (int ch = 0; ch < nchunks; ++ch) {
#pragma unroll
        for (int h = 0; h < HSTEPS; ++h) {
            // Operands of the NEXT half-step are fetched under this one's MFMAs.  The last half-step of a chunk prefetches
            // from the OTHER halo buffer: every thread stored its part of it during half-steps 0..6, so one barrier in front
            // of half-step 7 publishes it (and orders this chunk's last reads of the current buffer before the next chunk's
            // first stores into it).  After the last chunk these prefetches read the weights' zero padding block / a halo
            // buffer nobody needs.
            const float* wnext;
            int rh, rbuf;
            if (h + 1 < HSTEPS) {
                wnext = w_ptr(ch, h + 1);
                rbuf = ch & 1;
                rh = h + 1;
            } else {
#if !defined(ND_WWABL_NOBAR)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#endif
                wnext = w_ptr(ch + 1, 0);
                rbuf = (ch + 1) & 1;
                rh = 0;
            }
            int kx = 4 * (((rh >> 1) << 3) | (rbuf * BUF1));      // bytes: k-step and buffer bits of a patch read's address
            asm volatile("" : "+s"(kx));      // keeps the XORed offsets from being hoisted out of the chunk loop (64+ registers)
            const char* rsrc = reinterpret_cast<const char*>(smem) + 8 * (rh & 1);
            // next chunk's halo: two 16-byte items per thread and half-step
            constexpr int K0 = 0;
            const int k0 = 2 * h, k1 = 2 * h + 1;
            f32x4 ph0 = {0.f, 0.f, 0.f, 0.f}, ph1 = {0.f, 0.f, 0.f, 0.f};
            // One wave per SIMD: the wave stalls AT an MFMA until the matrix pipe takes it and issues nothing meanwhile, so
            // every MFMA is followed by its share of the other work (<= 64 cycles of issue) inside a fence: a half-step is
            // 32 slots = 16 positions x (MFMA on channel 0 | MFMA on channel 1).  Slot pair ps carries the refetch of the
            // position's weight pair (needed again 30 MFMAs later) and one sixteenth of the rest: two patch reads (ps < 8)
            // or the transforms of positions 2 (ps - 8), + 1 (<= ps: their MFMAs of this half-step are behind us); the two
            // halo fetches ride in pairs 0 and 1, their LDS stores in pairs 14 and 15.
#pragma unroll
            for (int ps = 0; ps < 16; ++ps) {
                acc[ps] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[ps][0], V[ps][0], acc[ps], 0, 0, 0);
                if (ps < 8) {
#if !defined(ND_WWABL_NORAW)
                    raw[2 * ps] = *reinterpret_cast<const f32x2*>(rsrc + (off16[2 * ps] ^ kx));
#else
                    asm volatile("" :: "v"(rsrc), "s"(kx));
#endif
                } else {
#if !defined(ND_WWABL_NOXF)
                    transform_pos(2 * (ps - 8));
#else
                    asm volatile("" :: "v"(raw[2 * (ps - 8)]));
#endif
                }
#if !defined(ND_WWABL_NOHALO)
                if (ps == 0 && k0 < MAXHI) ph0 = load_halo_pixel(gpix[k0 < MAXHI ? k0 : K0], ch + 1);
                if (ps == 1 && k1 < MAXHI) ph1 = load_halo_pixel(gpix[k1 < MAXHI ? k1 : K0], ch + 1);
#endif
#if ND_WINOW_SCHED
                __builtin_amdgcn_sched_barrier(0);
#endif
                acc[ps] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[ps][1], V[ps][1], acc[ps], 0, 0, 0);
#if !defined(ND_WWABL_NOW)
                wb[ps] = *reinterpret_cast<const f32x2*>(wnext + ps * 256);
#else
                asm volatile("" :: "v"(wnext));
#endif
                if (ps < 8) {
#if !defined(ND_WWABL_NORAW)
                    raw[2 * ps + 1] = *reinterpret_cast<const f32x2*>(rsrc + (off16[2 * ps + 1] ^ kx));
#endif
                } else {
#if !defined(ND_WWABL_NOXF)
                    transform_pos(2 * (ps - 8) + 1);
#else
                    asm volatile("" :: "v"(raw[2 * (ps - 8) + 1]));
#endif
                }
#if !defined(ND_WWABL_NOHALO)
                if (ps == 14 && k0 < MAXHI) store_halo_item(k0 < MAXHI ? k0 : K0, (ch + 1) & 1, ph0);
                if (ps == 15 && k1 < MAXHI) store_halo_item(k1 < MAXHI ? k1 : K0, (ch + 1) & 1, ph1);
#endif
#if ND_WINOW_SCHED
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
    }
    static_assert(2 * (HSTEPS - 1) >= MAXHI, "the halo items must all be stored before the last half-step");

    // ---- epilogue: Y[a][b] = sum_xi At[a][xi] (sum_nu At[b][nu] M[xi][nu]),  At = [[1,1,1,0],[0,1,-1,-1]], in registers.
    //      Accumulators are M^T (row = channel, col = tile): register group g4 of a lane = 4 consecutive channels of its tile.
    const int img = img0 + t_li;
    const float* rbp = p.rowbias ? p.rowbias + (size_t)img * p.ld_rowbias : nullptr;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const int nb = n0 + 8 * g4 + 4 * lh;
        f32x4 rx[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float sb = b ? -1.f : 1.f;
                f32x4 m[3];
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    m[i] = f32x4{acc[xi * 4 + b + i][4 * g4 + 0], acc[xi * 4 + b + i][4 * g4 + 1], acc[xi * 4 + b + i][4 * g4 + 2],
                                 acc[xi * 4 + b + i][4 * g4 + 3]};
                rx[xi][b] = (m[0] + sb * m[1]) + sb * m[2];
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float sa = a ? -1.f : 1.f;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x4 yv = (rx[a][b] + sa * rx[a + 1][b]) + sa * rx[a + 2][b];
                const int oy = oy0 + 2 * t_y + a, ox = ox0 + 2 * t_x + b;
                if (img < p.NI && oy < p.H && ox < p.W && nb < p.N) {
                    const bool vec = p.vec_ok && (nb + 3 < p.N);
                    float* op = p.out + ((size_t)(img * p.H + oy) * p.W + ox) * p.ldo + nb;
                    const float* rp = nullptr;
                    if (p.res) {
                        const size_t rpx = p.res_up ? ((size_t)(img * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1))
                                                    : ((size_t)(img * p.H + oy) * p.W + ox);
                        rp = p.res + rpx * p.ldr + nb;
                    }
                    if (vec) {
                        if (p.bias) yv += *reinterpret_cast<const f32x4*>(p.bias + nb);
                        if (rbp) yv += *reinterpret_cast<const f32x4*>(rbp + nb);
                        if (rp) yv += *reinterpret_cast<const f32x4*>(rp);
                        if (p.silu_out) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) yv[c] = fast_silu(yv[c]);
                        }
                        *reinterpret_cast<f32x4*>(op) = yv;
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (nb + c < p.N) {
                                float v2 = yv[c];
                                if (p.bias) v2 += p.bias[nb + c];
                                if (rbp) v2 += rbp[nb + c];
                                if (rp) v2 += rp[c];
                                if (p.silu_out) v2 = fast_silu(v2);
                                op[c] = v2;
                            }
                        }
                    }
                }
            }
        }
    }
}

int launch_winow(const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
    static bool attr_set[kMaxDevices] = {};
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(conv_winow_kernel), attr_set, "nd_conv3x3_winograd_nhwc")) return rc;
    hipLaunchKernelGGL(conv_winow_kernel, dim3(grid), dim3(256), lds, s, a);
    return check_launch("nd_conv3x3_winograd_nhwc");
}

}  // namespace nd

#endif  // ND_EXPERIMENTAL_KERNELS

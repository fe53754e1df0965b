// K6 in bf16, GEMM-shaped 1x1 convolution with two blocks per CU (variant 21 of nd_conv_bf16_nhwc; flat pixel lists only):
//   out[M][N] = x[M][K] . w[N][K]^T (+ bias, + residual) -- the skip / qkv / proj convolutions of BASELINE configs[3], [4]
//   (reference: nicediffusion/model.py:169-170,182,247-253,266-287).
//
// Why a second GEMM form.  gemm_bf16_kernel (nd_conv_bf16.hip) stages BOTH operands through 3 x 48 KiB of LDS: one 8-wave
// block per CU, so every block's ring fill and epilogue is dead time, and with K = 512 ... 2048 that is most of the block
// (qkv 32x32x512->1536: 2.2 us of MFMAs in a 17 us block life, PMC: matrix pipe busy 0.13-0.25).  This form keeps only the
// pixel rows in LDS (4 stages x 16 KiB per block) and streams the weight fragments global -> VGPR, as conv_bf16_kernel does:
// 4 waves at <= 256 registers, 64 KiB of LDS = two blocks per CU, one covering the other's prologue / epilogue / waits.
//   block    128 pixels x 256 channels, 1 x 4 waves, wave tile 128 px x 64 ch (4 x 2 tiles of v_mfma_f32_32x32x16_bf16)
//   rows     buffer_load ... lds, 4 rounds of 1 KiB per wave and 64-channel chunk, row advance and chunk in the scalar
//            offset (no vector instruction per DMA); rows at or past M read as zeros (buffer bounds)
//   weights  the wave's 8 fragments of chunk c + 1 are requested during chunk c (2 per k-step) into the other half of a
//            64-register double buffer: one whole chunk of lead
//   waits    inline ISA with hand-counted vmcnt.  vmcnt retires in issue order, so a DMA's deadline is the first weight
//            wait behind it: the order [weights of k-step 3, DMA rounds] at the end of a chunk gives every DMA five
//            k-steps before a weight wait reaches it and seven before its stage is read (the barrier sits behind k-step 2,
//            so that k-step 3 can already read the next stage).  Every k-step's weight wait is the same vmcnt(12).
//   epilogue the compact staged form of conv_bf16_kernel: wave-private [128 px][128 B] LDS region, 16-byte stores, 8 lanes =
//            one cache line.
// Host-checked: M % 128 == 0, N % 256 == 0, C0 and C1 multiples of 64, tensors < 2 GiB, bf16 output, no SiLU / fused GroupNorm;
// with statistics also H * W % 128 == 0 (a block's 128 pixels belong to one image).
#include "nd_conv_bf16_args.h"
#include <type_traits>

namespace nd {

static_assert(wstream::pad_chunks(wstream::kBf16GemmQAheadSteps, wstream::bf16_steps(1)) <= wstream::kBf16PadChunks,
              "weight read-ahead exceeds the packer's zero padding");

template <bool STATS>      // STATS: also leave the per-channel partial statistics of the output behind (p.chstats, one row per block)
__global__ void __launch_bounds__(256, 2)
    gemm_bf16q_kernel(const ConvArgsH p) {
    constexpr int BM = 128, BN = 256, TM = 4, TN = 2;
    constexpr int STAGE_B = 16384;                     // bytes per stage: 128 rows x 128 bytes (64 channels)
    constexpr int NSTAGE = 4;
    constexpr int NDMA = 4;                            // DMA rounds per wave and chunk: 4 x 4 waves x 1 KiB

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    __builtin_amdgcn_s_setprio(3);                     // prologue / epilogue: vector + memory streams, the other block has the MFMAs
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int lh = lane >> 5;

    const int total = gridDim.x;
    const int q = total >> 3, r = total & 7, xcd = blockIdx.x & 7;
    const int idp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    int mblk, nblk;
    tile_of(idp, p.mt, p.nt, p.ngroup, mblk, nblk);
    const int M = p.W;                                 // flat pixel list
    const int m0 = mblk * BM, n0 = nblk * BN;
    const int nchunks = p.NC64;

    // ---- pixel-row DMAs.  Round k of this wave fills rows (k * 4 + wave) * 8 + lane / 8 of the stage; lane % 8 is the
    //      PHYSICAL 16-byte slot, which holds logical channel slot (lane % 8) ^ swz(row), swz(row) = (row >> 1) & 7 -- the
    //      same in every round (a round is 32 rows further)
    const int row8 = wave * 8 + (lane >> 3);
    const int lslot = (lane & 7) ^ ((row8 >> 1) & 7);
    const unsigned vo0 = __umul24((unsigned)row8, (unsigned)p.ldx0 * 2u) + (unsigned)(lslot << 4);
    const unsigned vo1 = __umul24((unsigned)row8, (unsigned)p.ldx1 * 2u) + (unsigned)(lslot << 4);
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x0), 0, (int)((unsigned)M * (unsigned)p.ldx0 * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x1 ? p.x1 : p.x0), 0, (int)((unsigned)M * (unsigned)p.ldx1 * 2u), 0x00020000);
    // chunks past the last one re-fetch the last chunk into a stage nobody reads any more: the number of VMEM operations
    // per chunk is a constant the hand-counted waits rely on
    auto dma = [&](int k, int ch) {
        const int che = ch < nchunks - 1 ? ch : nchunks - 1;
        const int c0 = che * 64;
        auto* dst = (__attribute__((address_space(3))) void*)(smem + (ch & (NSTAGE - 1)) * (STAGE_B / 4) + (k * 4 + wave) * 256);
        if (c0 < p.C0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, (int)vo0, (m0 + k * 32) * p.ldx0 * 2 + c0 * 2, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, (int)vo1, (m0 + k * 32) * p.ldx1 * 2 + (c0 - p.C0) * 2, 0, 0);
    };

    // ---- fragment reads: row = mi * 32 + l31 (same swizzle for the four mi), 16-byte slot (kc << 1 | lh) ^ swz(row)
    const int aoff = l31 * 128 + ((lh ^ ((l31 >> 1) & 7)) << 4);          // bytes, k-step 0, stage 0; mi adds 4096 per tile
    auto rdA = [&](f32x4 (&a)[TM], int sbits) {          // sbits = stage * 16384 | kc << 5: one v_xor, four reads
        const int addr = aoff ^ sbits;
        asm volatile("ds_read_b128 %0, %1" : "=v"(a[0]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a[1]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(a[2]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(a[3]) : "v"(addr));
    };
    // ---- weight fragments [c64][n tile][k-step][lane][8 bf16] (1 KiB each): scalar base + lane * 16; the wave's two n
    //      tiles are 4 KiB apart
    const int nt0 = nblk * 8 + wave * 2;
    const __bf16* bw0 = p.w + (size_t)nt0 * 2048;
    const __bf16* bw1 = bw0 + 2048;
    const size_t c64_stride = (size_t)p.NT32 * 2048;          // elements
    const int voff = lane * 16;
    auto ldB = [&](f32x4 (&b)[TN], size_t eoff, int kc) {          // eoff: element offset of the chunk
        const __bf16* s0 = bw0 + eoff;
        const __bf16* s1 = bw1 + eoff;
        switch (kc) {
            case 0: asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[0]) : "v"(voff), "s"(s0));
                    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[1]) : "v"(voff), "s"(s1)); break;
            case 1: asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(b[0]) : "v"(voff), "s"(s0));
                    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(b[1]) : "v"(voff), "s"(s1)); break;
            case 2: asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(b[0]) : "v"(voff), "s"(s0));
                    asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(b[1]) : "v"(voff), "s"(s1)); break;
            default: asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(b[0]) : "v"(voff), "s"(s0));
                     asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(b[1]) : "v"(voff), "s"(s1)); break;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    f32x4 afr[2][TM], bfr[2][4][TN];          // weights: [chunk parity][k-step][n tile]

    // ---- prologue.  VMEM order (the order the loop leaves behind): rows of chunks 0 and 1, the 8 weight fragments of
    //      chunk 0, rows of chunk 2
#pragma unroll
    for (int k = 0; k < NDMA; ++k) dma(k, 0);
#pragma unroll
    for (int k = 0; k < NDMA; ++k) dma(k, 1);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) ldB(bfr[0][kc], 0, kc);
#pragma unroll
    for (int k = 0; k < NDMA; ++k) dma(k, 2);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");          // chunks 0 and 1 have landed: 8 fragments + 4 rounds are younger
    __builtin_amdgcn_s_barrier();
    rdA(afr[0], 0);
    __builtin_amdgcn_s_setprio(0);

#define ND_SB __builtin_amdgcn_sched_barrier(0)
    auto chunk = [&](int ch, auto parity) {
        constexpr int cb = decltype(parity)::value, nb = cb ^ 1;
        const int sb = (ch & (NSTAGE - 1)) * STAGE_B, sn = ((ch + 1) & (NSTAGE - 1)) * STAGE_B;
        const size_t en = (size_t)(ch + 1) * c64_stride;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int cur = st & 1, nxt = cur ^ 1;
            // pixel fragments of k-step st + 1 (this stage, or -- behind the barrier of k-step 2 -- the next one) and the
            // weight fragments of the NEXT chunk's k-step st
            int sbits = ((st < 3) ? sb : sn) | (((st + 1) & 3) << 5);
            asm volatile("" : "+s"(sbits));
            rdA(afr[nxt], sbits);
            ldB(bfr[nb][st], en, st);
            // this k-step's weights were requested one chunk ago; younger: the rest of that chunk's requests 2 (3 - st),
            // its 4 DMA rounds, this chunk's requests so far 2 (st + 1) = 12 in every k-step
            asm volatile("s_waitcnt vmcnt(12)" : "+v"(bfr[cb][st][0]), "+v"(bfr[cb][st][1]));
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(afr[cur][0]), "+v"(afr[cur][1]), "+v"(afr[cur][2]), "+v"(afr[cur][3]));
            ND_SB;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(bfr[cb][st][ni]), as_bf16x8(afr[cur][mi]), acc[mi][ni], 0, 0, 0);
                    if (st == 3 && ni == 1) dma(mi, ch + 3);          // stage (ch - 1) % 4: free since the barrier behind k-step 2
                    ND_SB;
                }
            }
            if (st == 2) {
                // chunk ch + 1's rows: requested in k-step 3 of chunk ch - 2; younger than them are 8 + 4 operations of
                // chunk ch - 1 and the 6 weight requests of this chunk so far.  All reads of this stage but k-step 3's
                // (issued above, awaited here) have returned.
                asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    };
    for (int ch = 0; ch < nchunks; ch += 2) {
        chunk(ch, std::integral_constant<int, 0>{});
        if (ch + 1 < nchunks) chunk(ch + 1, std::integral_constant<int, 1>{});
    }
#undef ND_SB
    // run-ahead loads are still in flight: keep their registers allocated until they have returned; every DMA of this wave
    // has landed before the barrier that hands the LDS over to the staging regions
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(bfr[0][0][0]), "+v"(bfr[0][0][1]), "+v"(bfr[0][1][0]), "+v"(bfr[0][1][1]), "+v"(bfr[0][2][0]), "+v"(bfr[0][2][1]),
                   "+v"(bfr[0][3][0]), "+v"(bfr[0][3][1]), "+v"(bfr[1][0][0]), "+v"(bfr[1][0][1]), "+v"(bfr[1][1][0]), "+v"(bfr[1][1][1]),
                   "+v"(bfr[1][2][0]), "+v"(bfr[1][2][1]), "+v"(bfr[1][3][0]), "+v"(bfr[1][3][1])
                 :
                 : "memory");
    asm volatile("" : "+v"(afr[0][0]), "+v"(afr[0][1]), "+v"(afr[0][2]), "+v"(afr[0][3]), "+v"(afr[1][0]), "+v"(afr[1][1]),
                      "+v"(afr[1][2]), "+v"(afr[1][3]));
    __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_s_barrier();

    // ---- epilogue (the compact staged form of conv_bf16_kernel): lane = one pixel, register group g4 = channels
    //      8 g4 + 4 lh .. + 3 of the n tile -> wave-private region [128 px][128 B], slot s of row r at s ^ (r & 7)
    char* const stg = reinterpret_cast<char*>(smem) + wave * (TM * 32 * 128);
    const float* const bptr = p.bias + n0 + wave * 64 + 4 * lh;
    auto rows = [&](auto has_res) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            f32x4 bv[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                bv[g4] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.bias) bv[g4] = *reinterpret_cast<const f32x4*>(bptr + ni * 32 + 8 * g4);
            }
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int pxl = mi * 32 + l31;
                const __bf16* rrow = nullptr;
                if constexpr (decltype(has_res)::value) rrow = p.res + (size_t)(m0 + pxl) * p.ldr + n0 + wave * 64 + 4 * lh + ni * 32;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 v = {acc[mi][ni][4 * g4 + 0], acc[mi][ni][4 * g4 + 1], acc[mi][ni][4 * g4 + 2], acc[mi][ni][4 * g4 + 3]};
                    v += bv[g4];
                    if constexpr (decltype(has_res)::value) {
                        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(rrow + 8 * g4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                    }
                    const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *reinterpret_cast<bf16x4*>(stg + pxl * 128 + (((ni * 4 + g4) ^ (pxl & 7)) << 4) + lh * 8) = o;
                }
            }
        }
    };
    if (p.res) rows(std::true_type{});
    else rows(std::false_type{});
    __bf16* const obase = static_cast<__bf16*>(p.out) + n0 + wave * 64 + (lane & 7) * 8;
    // STATS (the attention block's output feeds the next block's GroupNorm, model.py:291,190): the drain's reads also feed
    // the per-channel sums of what was stored, exactly as in conv_bf16_kernel -- pixels ascending per lane, then the lane
    // groups that hold the same 8 channels by three exchanges; row = this block's 128 pixels of its image (gn_hw = H * W)
    float sv[8], sq[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) sv[c] = sq[c] = 0.f;
#pragma unroll
    for (int it = 0; it < TM * 4; ++it) {
        const int pxl = it * 8 + (lane >> 3);
        union { f32x4 f; unsigned u[4]; } v;
        v.f = *reinterpret_cast<const f32x4*>(stg + pxl * 128 + (((lane & 7) ^ (pxl & 7)) << 4));
        *reinterpret_cast<f32x4*>(obase + (size_t)(m0 + pxl) * p.ldo) = v.f;
        if constexpr (STATS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
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
            const int img = m0 / p.gn_hw, row = (m0 - img * p.gn_hw) >> 7;
            float* cs = p.chstats + ((size_t)img * p.cs_rows + row) * 2 * p.N + n0 + wave * 64 + lane * 8;
            *reinterpret_cast<f32x4*>(cs) = f32x4{sv[0], sv[1], sv[2], sv[3]};
            *reinterpret_cast<f32x4*>(cs + 4) = f32x4{sv[4], sv[5], sv[6], sv[7]};
            *reinterpret_cast<f32x4*>(cs + p.N) = f32x4{sq[0], sq[1], sq[2], sq[3]};
            *reinterpret_cast<f32x4*>(cs + p.N + 4) = f32x4{sq[4], sq[5], sq[6], sq[7]};
        }
    }
}

int launch_gemm_bf16q(const ConvArgsH& a, int grid, hipStream_t s) {
    const size_t lds = (size_t)64 * 1024;
    if (a.chstats) {
        auto kern = gemm_bf16q_kernel<true>;
        static bool attr_set[kMaxDevices] = {};
        if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), attr_set, "nd_conv_bf16_nhwc")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    } else {
        auto kern = gemm_bf16q_kernel<false>;
        static bool attr_set[kMaxDevices] = {};
        if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), attr_set, "nd_conv_bf16_nhwc")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    }
    return check_launch("nd_conv_bf16_nhwc");
}

}  // namespace nd

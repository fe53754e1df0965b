// K6 in bf16, GEMM-shaped 1x1 convolution, ONE block of four 512-register waves per CU (variant 22 of nd_conv_bf16_nhwc):
// the experiment behind DESIGN.md 4.5 'what bounds the bf16 kernels'.  The bf16 convolutions and GEMMs of this library are
// limited by the clock the chip holds under them, and what raises that clock is less energy per MFMA: fewer operand
// bytes from LDS and L2.  A wave tile of 128 px x 128 ch (4 x 4 tiles, 256 accumulator registers) needs 0.25 LDS reads
// and 0.25 global loads per MFMA against 0.5 + 0.25 for the 128 x 64 tile of gemm_bf16x_kernel -- at the price of one
// wave per SIMD (512 registers), so nothing covers a wave's waits or the block's prologue / epilogue.
// Same structure as gemm_bf16x_kernel otherwise: block 256 px x 256 ch (2 x 2 waves), pixel rows by LDS-DMA into 4 stages
// of 32 KiB, weights global -> VGPR two k-steps ahead (ring of four k-steps, 64 registers), hand-counted vmcnt (16 / 16 / 8 / 8 per
// k-step), staged 16-byte stores.  Host-checked: M % 256 == 0, N % 256 == 0, C0 and C1 multiples of
// 64, tensors < 2 GiB, bf16 output, no SiLU / fused GroupNorm.
// Result (tools/ab_gemm_bf16.py, one box, interleaved): bit-identical to variants 20 / 21 and 0.72-0.93 of variant 21's rate on
// every 1x1 shape of configs[3] / [4] (32x32x2048->1024: 1 049 vs 1 217 TFLOP/s; 128x128x512->256: 599 vs 732) -- with one
// wave per SIMD nothing covers the waits, and the fewer operand bytes do not buy that back.  Built with EXPERIMENTAL=1 only.
#include "nd_conv_bf16_args.h"
#include <type_traits>

#if defined(ND_EXPERIMENTAL_KERNELS)      // measured slower than gemm_bf16q_kernel on every shape: not in the product build

namespace nd {

static_assert(wstream::pad_chunks(wstream::kBf16GemmQAheadSteps, wstream::bf16_steps(1)) <= wstream::kBf16PadChunks,
              "weight read-ahead exceeds the packer's zero padding");

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
    gemm_bf16x_kernel(const ConvArgsH p) {
    constexpr int BM = 256, BN = 256, TM = 4, TN = 4;
    constexpr int STAGE_B = 32768;                     // bytes per stage: 256 rows x 128 bytes (64 channels)
    constexpr int NSTAGE = 4;
    constexpr int NDMA = 8;                            // DMA rounds per wave and chunk: 8 x 4 waves x 1 KiB

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    __builtin_amdgcn_s_setprio(3);                     // prologue / epilogue: vector + memory streams, the other block has the MFMAs
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const int aoff = (wm * 128 + l31) * 128 + ((lh ^ ((l31 >> 1) & 7)) << 4);   // bytes, k-step 0, stage 0; mi adds 4096 per tile
    auto rdA = [&](f32x4 (&a)[TM], int sbits) {          // sbits = stage * 32768 | kc << 5: one v_xor, four reads
        const int addr = aoff ^ sbits;
        asm volatile("ds_read_b128 %0, %1" : "=v"(a[0]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a[1]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(a[2]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(a[3]) : "v"(addr));
    };
    // ---- weight fragments [c64][n tile][k-step][lane][8 bf16] (1 KiB each): scalar base + lane * 16; the wave's two n
    //      tiles are 4 KiB apart
    const int nt0 = nblk * 8 + wn * 4;
    const __bf16* bw0 = p.w + (size_t)nt0 * 2048;
    const size_t c64_stride = (size_t)p.NT32 * 2048;          // elements
    const int voff = lane * 16;
    auto ldB = [&](f32x4 (&b)[TN], size_t eoff, int kc) {          // eoff: element offset of the chunk
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const __bf16* sb = bw0 + eoff + ni * 2048;
            switch (kc) {
                case 0: asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[ni]) : "v"(voff), "s"(sb)); break;
                case 1: asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(b[ni]) : "v"(voff), "s"(sb)); break;
                case 2: asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(b[ni]) : "v"(voff), "s"(sb)); break;
                default: asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(b[ni]) : "v"(voff), "s"(sb)); break;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    f32x4 afr[2][TM], bfr[4][TN];             // weights: ring of four k-steps, requested two k-steps ahead

    // ---- prologue.  VMEM order (the order the loop leaves behind): rows of chunks 0 and 1, the weights of k-steps 0 and 1,
    //      rows of chunk 2
#pragma unroll
    for (int k = 0; k < NDMA; ++k) dma(k, 0);
#pragma unroll
    for (int k = 0; k < NDMA; ++k) dma(k, 1);
    ldB(bfr[0], 0, 0);
    ldB(bfr[1], 0, 1);
#pragma unroll
    for (int k = 0; k < NDMA; ++k) dma(k, 2);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");          // chunks 0 and 1 have landed: 8 fragments + 8 rounds are younger
    __builtin_amdgcn_s_barrier();
    rdA(afr[0], 0);
    __builtin_amdgcn_s_setprio(0);

#define ND_SB __builtin_amdgcn_sched_barrier(0)
    for (int ch = 0; ch < nchunks; ++ch) {
        const int sb = (ch & (NSTAGE - 1)) * STAGE_B, sn = ((ch + 1) & (NSTAGE - 1)) * STAGE_B;
        const size_t ec = (size_t)ch * c64_stride, en = ec + c64_stride;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int cur = st & 1, nxt = cur ^ 1;
            // pixel fragments of k-step st + 1 (this stage, or -- behind the barrier of k-step 2 -- the next one) and the
            // weight fragments of k-step st + 2 (this chunk's or the next one's)
            int sbits = ((st < 3) ? sb : sn) | (((st + 1) & 3) << 5);
            asm volatile("" : "+s"(sbits));
            rdA(afr[nxt], sbits);
            if (st < 2) ldB(bfr[st + 2], ec, st + 2);
            else ldB(bfr[st - 2], en, st - 2);
            // this k-step's weights were requested two k-steps ago; younger: the 4 + 4 requests since, and in k-steps 0 and 1
            // the 8 DMA rounds of the previous k-step 3 (issued behind its request)
            if (st < 2) asm volatile("s_waitcnt vmcnt(16)" : "+v"(bfr[st][0]), "+v"(bfr[st][1]), "+v"(bfr[st][2]), "+v"(bfr[st][3]));
            else asm volatile("s_waitcnt vmcnt(8)" : "+v"(bfr[st][0]), "+v"(bfr[st][1]), "+v"(bfr[st][2]), "+v"(bfr[st][3]));
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(afr[cur][0]), "+v"(afr[cur][1]), "+v"(afr[cur][2]), "+v"(afr[cur][3]));
            ND_SB;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(bfr[st][ni]), as_bf16x8(afr[cur][mi]), acc[mi][ni], 0, 0, 0);
                    if (st == 3 && (ni & 1)) dma(mi * 2 + (ni >> 1), ch + 3);   // stage (ch - 1) % 4: free since the barrier behind k-step 2
                    ND_SB;
                }
            }
            if (st == 2) {
                // chunk ch + 1's rows were requested in k-step 3 of chunk ch - 2 and are older than the weights k-step 2 of
                // chunk ch - 1 waited for: landed.  All reads of this stage but k-step 3's (issued above) have returned.
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }
#undef ND_SB
    // run-ahead loads are still in flight: keep their registers allocated until they have returned; every DMA of this wave
    // has landed before the barrier that hands the LDS over to the staging regions
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[0][2]), "+v"(bfr[0][3]), "+v"(bfr[1][0]), "+v"(bfr[1][1]), "+v"(bfr[1][2]), "+v"(bfr[1][3]), "+v"(bfr[2][0]), "+v"(bfr[2][1]), "+v"(bfr[2][2]), "+v"(bfr[2][3]), "+v"(bfr[3][0]), "+v"(bfr[3][1]), "+v"(bfr[3][2]), "+v"(bfr[3][3])
                 :
                 : "memory");
    asm volatile("" : "+v"(afr[0][0]), "+v"(afr[0][1]), "+v"(afr[0][2]), "+v"(afr[0][3]), "+v"(afr[1][0]), "+v"(afr[1][1]),
                      "+v"(afr[1][2]), "+v"(afr[1][3]));
    __builtin_amdgcn_s_setprio(3);
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: lane = one pixel, register group g4 = channels 8 g4 + 4 lh .. + 3 of the n tile -> wave-private region
    //      [128 px][256 B], 16-byte slot s of row r at s ^ (r & 7); read back 16 lanes = one pixel's 256 contiguous bytes
    char* const stg = reinterpret_cast<char*>(smem) + wave * (TM * 32 * 256);
    const int nw = n0 + wn * 128;
    const int mw = m0 + wm * 128;
    const float* const bptr = p.bias + nw + 4 * lh;
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
                if constexpr (decltype(has_res)::value) rrow = p.res + (size_t)(mw + pxl) * p.ldr + nw + 4 * lh + ni * 32;
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
                    *reinterpret_cast<bf16x4*>(stg + pxl * 256 + (((ni * 4 + g4) ^ (pxl & 7)) << 4) + lh * 8) = o;
                }
            }
        }
    };
    if (p.res) rows(std::true_type{});
    else rows(std::false_type{});
    __bf16* const obase = static_cast<__bf16*>(p.out) + nw + (lane & 15) * 8;
#pragma unroll
    for (int it = 0; it < TM * 8; ++it) {
        const int pxl = it * 4 + (lane >> 4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(stg + pxl * 256 + (((lane & 15) ^ (pxl & 7)) << 4));
        *reinterpret_cast<f32x4*>(obase + (size_t)(mw + pxl) * p.ldo) = v;
    }
}

int launch_gemm_bf16x(const ConvArgsH& a, int grid, hipStream_t s) {
    const size_t lds = (size_t)128 * 1024;
    auto kern = gemm_bf16x_kernel;
    static bool attr_set[kMaxDevices] = {};
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), attr_set, "nd_conv_bf16_nhwc")) return rc;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
    return check_launch("nd_conv_bf16_nhwc");
}

}  // namespace nd

#endif  // ND_EXPERIMENTAL_KERNELS

// K6, GEMM-shaped fp32 1x1 convolution, two blocks per CU (variant 14 of nd_conv_nhwc; flat pixel lists only):
//   out[M][N] = x[M][K] . w[N][K]^T  (+ bias, + residual, GroupNorm of x folded in) -- the qkv / proj / skip convolutions
//   of the attention and residual blocks (reference: nicediffusion/model.py:247-253,266-287,169-170,182).
//
// Built on what conv_wino4_kernel (nd_conv_winograd_quad.hip) established for fp32 MFMA kernels on gfx950:
//   * every vector instruction issued on a SIMD takes its cycles from the matrix pipe, so the main loop carries ONE
//     v_xor per k-step (32 MFMAs): pixel rows arrive by buffer_load ... lds with the row advance in the scalar offset
//     (8 DMAs per wave and chunk, no address arithmetic), the four pixel fragments of a k-step are one address + immediate
//     offsets, weight fragments come from a scalar base + a constant lane offset;
//   * 4 waves at <= 256 registers, 64 KiB of LDS: two blocks per CU, one covers the other's prologue / epilogue / waits;
//   * operand loads are inline ISA with hand-counted vmcnt / lgkmcnt (hipcc parks every LDS read behind all pending
//     LDS-DMA otherwise), everything for k-step s + 1 is requested at the top of k-step s.
// Block = 256 pixels x 128 channels (2 x 2 waves, wave tile 4 x 2 MFMA tiles = 8 accumulators), K in 32-channel chunks
// through two LDS stages at 0 and 32 KiB (the stage is one address bit).  TM = 2 is the same kernel on 128 pixels x 128
// channels (wave tile 2 x 2): the launcher takes it where the 256-pixel tiles fill the chip's 512 block slots badly and the
// 128-pixel ones do not (32x32 x 384 -> 1152 at batch 64: 2304 blocks = 4.5 rounds against 4608 = 9; -> 384: 1.5 against 3).  The host requires M % 256 == 0, whole
// 32-channel chunks on either side of a concatenation seam and tensors < 2 GiB (32-bit buffer offsets); with a fused
// GroupNorm also H * W % 256 == 0 (one image per block).
#include "nd_conv_common.h"

namespace nd {

template <bool GN, bool STATS = false, int TM = 4>          // STATS: also leave the per-channel partial statistics of the output behind (p.chstats)
__global__ void __launch_bounds__(256, 2)
    gemm4_kernel(const ConvArgs p) {
    static_assert(TM == 4 || TM == 2, "256- or 128-pixel blocks");
    constexpr int BM = 64 * TM, BN = 128, TN = 2;
    constexpr int STAGE_B = 32768;                     // bytes between the two stages (a stage holds up to 256 rows x 128 bytes)
    constexpr int NDMA = 2 * TM;                       // DMA rounds per wave and chunk: 8 x 4 waves x 1 KiB = 32 KiB (TM = 2: 4, 16 KiB)
    constexpr int LPS = GN ? 4 : 2;                    // register loads per k-step: 2 weight fragments (+ 2 coefficient vectors)

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
    const int nchunks = p.NC32;

    // ---- pixel-row DMAs.  Round k of this wave fills rows (k * 4 + wave) * 8 + lane / 8 of the stage; lane % 8 is the
    //      PHYSICAL 16-byte slot, which holds logical channel slot (lane % 8) ^ swz(row), swz(row) = (row >> 1) & 7 -- the
    //      same for every round (a round is 32 rows further).  So ONE byte offset per lane and source, and round / chunk
    //      live in the scalar offset: no vector instruction per DMA.
    const int row8 = wave * 8 + (lane >> 3);
    const int lslot = (lane & 7) ^ ((row8 >> 1) & 7);
    const unsigned vo0 = __umul24((unsigned)row8, (unsigned)p.ldx0 * 4u) + (unsigned)(lslot << 4);
    const unsigned vo1 = __umul24((unsigned)row8, (unsigned)p.ldx1 * 4u) + (unsigned)(lslot << 4);
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x0), 0, (int)((unsigned)M * (unsigned)p.ldx0 * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x1), 0, (int)((unsigned)M * (unsigned)p.ldx1 * 4u), 0x00020000);
    // chunks past the last one re-fetch the last chunk into a stage nobody reads: the number of VMEM operations per chunk
    // is a constant the hand-counted waits rely on
    auto dma = [&](int k, int ch, int stage) {
        const int che = ch < nchunks - 1 ? ch : nchunks - 1;
        const int c0 = che * 32;
        auto* dst = (__attribute__((address_space(3))) void*)(smem + stage * (STAGE_B / 4) + (k * 4 + wave) * 256);
        if (c0 < p.C0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, (int)vo0, (m0 + k * 32) * p.ldx0 * 4 + c0 * 4, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, (int)vo1, (m0 + k * 32) * p.ldx1 * 4 + (c0 - p.C0) * 4, 0, 0);
    };

    // ---- fragment reads: row = (wm * 4 + mi) * 32 + l31 (same swizzle for the four mi), slot (kc << 1 | lh) ^ swz
    const int arow = wm * (TM * 32) + l31;
    const int aoff = arow * 128 + ((lh ^ ((arow >> 1) & 7)) << 4);          // bytes, k-step 0, stage 0; mi adds 4096 per tile
    auto rdA = [&](f32x4 (&a)[TM], int sbits) {          // sbits = stage bit | (kc << 5): one v_xor, four reads
        const int addr = aoff ^ sbits;
        asm volatile("ds_read_b128 %0, %1" : "=v"(a[0]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(a[1]) : "v"(addr));
        if constexpr (TM == 4) {
            asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(a[2]) : "v"(addr));
            asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(a[3]) : "v"(addr));
        }
    };
    // ---- weight fragments [c32][n tile][kc][lane][4]: scalar base + lane * 16; the wave's two n tiles are 4 KiB apart
    int nt0 = nblk * 4 + wn * 2, nt1 = nt0 + 1;
    nt0 = nt0 > p.NT32 - 1 ? p.NT32 - 1 : nt0;          // N tail: clamped, results dropped in the epilogue
    nt1 = nt1 > p.NT32 - 1 ? p.NT32 - 1 : nt1;
    const float* bw0 = p.w + (size_t)nt0 * 1024;
    const float* bw1 = p.w + (size_t)nt1 * 1024;
    const size_t c32_stride = (size_t)p.NT32 * 1024;
    const int voff = lane * 16;
    static_assert(wstream::pad_chunks(wstream::kF32Gemm4Ahead, 4) <= wstream::kF32PadChunks, "weight read-ahead exceeds the packer's zero padding");
    auto ldB = [&](f32x4 (&b)[TN], size_t foff, int kc) {          // foff: float offset of the chunk
        const float* s0 = bw0 + foff;
        const float* s1 = bw1 + foff;
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
    // ---- fused GroupNorm: x' = act(x * A[img][c] + B[img][c]); this block's image, channels 8 kc + 4 lh .. + 3 of the chunk
    const float* ga = nullptr;
    const float* gb = nullptr;
    const int goff = lh * 16;
    if constexpr (GN) {
        const int gimg = m0 / p.gn_hw;
        ga = p.gnA + (size_t)gimg * p.ld_gn;
        gb = p.gnB + (size_t)gimg * p.ld_gn;
    }
    auto ldC = [&](f32x4 (&c)[2], int ch, int kc) {
        if constexpr (GN) {
            const int che = ch < nchunks - 1 ? ch : nchunks - 1;         // (the run-ahead past the last chunk stays inside the row)
            const float* sa = ga + che * 32 + kc * 8;
            const float* sb = gb + che * 32 + kc * 8;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(c[0]) : "v"(goff), "s"(sa));
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(c[1]) : "v"(goff), "s"(sb));
        }
    };
    auto wait_vm = [&](f32x4& r0, f32x4& r1, int n) {
#define ND_G4CASE(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" : "+v"(r0), "+v"(r1)); break;
        switch (n) {
            ND_G4CASE(2) ND_G4CASE(4) ND_G4CASE(6) ND_G4CASE(8) ND_G4CASE(10) ND_G4CASE(12)
            default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1)); break;
        }
#undef ND_G4CASE
    };
    // GroupNorm of one pixel fragment, in place (the SiLU choice is made once per call, outside the element loop)
    auto fold = [&](f32x4& a, const f32x4 (&c)[2]) {
        if constexpr (GN) {
            f32x4 v = a * c[0] + c[1];
            if (p.gn_silu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
            }
            a = v;
        }
    };
    auto wait_lds4 = [&](f32x4 (&a)[TM]) {          // the TM reads of THIS k-step have returned; the TM just issued may be in flight
        if constexpr (TM == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
        else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a[0]), "+v"(a[1]));
    };
    auto wait_lds0 = [&](f32x4 (&a)[TM]) {
        if constexpr (TM == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]));
    };

    // ---- the residual seeds the accumulators.  Added in the epilogue it was 64 (128) exposed 16-byte loads per lane, 32 bytes of
    //      every line per instruction, with every block of the launch in the same phase: 34-43 us on 32x32 384 -> 384
    //      (tools/time_conv1x1.py).  Requested here, OLDEST of the prologue's VMEM operations, its latency passes under the
    //      first two chunks' DMAs and the wait below covers it.  out = ((res + sum) + bias): the association differs from
    //      the other kernels' ((sum + bias) + res) in the last bit.
    const bool rinit = p.res != nullptr && (p.ldr & 3) == 0 && (p.ldo & 3) == 0 && (p.N & 3) == 0;
    f32x4 seed[TM][TN][4];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) seed[mi][ni][g4] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (rinit) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const float* rrow = p.res + (size_t)(m0 + (wm * TM + mi) * 32 + l31) * p.ldr + n0 + wn * TN * 32 + 4 * lh;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float* ptr = rrow + ni * 32 + 8 * g4;
                    if (n0 + (wn * TN + ni) * 32 + 8 * g4 + 4 * lh + 3 < p.N)
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(seed[mi][ni][g4]) : "v"(ptr));
                }
        }
    }

    f32x4 afr[2][TM], bfr[2][TN], cfr[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) cfr[i][0] = cfr[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue.  VMEM order: (the residual,) chunk 0's rows, the operands of k-step 0, chunk 1's rows (the order the loop
    //      leaves behind)
#pragma unroll
    for (int k = 0; k < NDMA; ++k) dma(k, 0, 0);
    ldC(cfr[0], 0, 0);
    ldB(bfr[0], 0, 0);
#pragma unroll
    for (int k = 0; k < NDMA; ++k) dma(k, 1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPS + NDMA) : "memory");          // chunk 0 (and the residual) has landed: LPS + NDMA younger operations
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)          // (ties the seed registers to the wait: not read before it)
            asm volatile("" : "+v"(seed[mi][ni][0]), "+v"(seed[mi][ni][1]), "+v"(seed[mi][ni][2]), "+v"(seed[mi][ni][3]));
    f32x16 acc[TM][TN];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = seed[mi][ni][e >> 2][e & 3];
    __builtin_amdgcn_s_barrier();
    rdA(afr[0], 0);
    if constexpr (GN) {
        asm volatile("s_waitcnt vmcnt(%2)"          // the coefficients of k-step 0: 2 weight loads + NDMA rounds are younger
                     : "+v"(cfr[0][0]), "+v"(cfr[0][1]) : "i"(2 + NDMA));
        wait_lds0(afr[0]);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) fold(afr[0][mi], cfr[0]);
    }
    __builtin_amdgcn_s_setprio(0);

#define ND_SB __builtin_amdgcn_sched_barrier(0)
    for (int ch = 0; ch < nchunks; ++ch) {
        const int sb = (ch & 1) * STAGE_B, sn = ((ch + 1) & 1) * STAGE_B;
        const size_t fq = (size_t)ch * c32_stride;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int cur = st & 1, nxt = cur ^ 1;
            // ---- everything k-step st + 1 needs is requested now (it has this k-step's 32 MFMAs to arrive): pixel fragments
            //      from this chunk's stage or -- behind the barrier of k-step 2 -- the next chunk's, weights one k-step on
            //      (the stream ends in a zero chunk), coefficients
            int sbits = ((st < 3) ? sb : sn) | (((st + 1) & 3) << 5);
            asm volatile("" : "+s"(sbits));
            rdA(afr[nxt], sbits);
            if (st < 3) { ldC(cfr[nxt], ch, st + 1); ldB(bfr[nxt], fq, st + 1); }
            else { ldC(cfr[nxt], ch + 1, 0); ldB(bfr[nxt], fq + c32_stride, 0); }
            // this k-step's operands: issued one k-step ago; younger = the LPS loads above (+ the 8 DMAs of k-step 3)
            wait_vm(bfr[cur][0], bfr[cur][1], LPS + (st == 0 ? NDMA : 0));
            wait_lds4(afr[cur]);
            ND_SB;
            const int dstage = ch & 1;          // behind the barrier of k-step 2 this chunk's stage is free for chunk ch + 2
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                if constexpr (GN) {
                    // the NEXT k-step's fragments are normalised under this k-step's second half of MFMAs: they and their
                    // coefficients were requested at the top; behind the coefficients only the 2 weight loads (and, in
                    // k-step 3, the DMA rounds issued so far: 2 per mi) are younger
                    if (mi == TM / 2) {
                        wait_vm(cfr[nxt][0], cfr[nxt][1], 2 + (st == 3 ? (TM / 2) * TN : 0));
                        wait_lds0(afr[nxt]);
                    }
                }
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bfr[cur][ni][j], afr[cur][mi][j], acc[mi][ni], 0, 0, 0);
                        if (st == 3 && j == 1) dma(mi * 2 + ni, ch + 2, dstage);          // 8 rounds over the k-step's 32 MFMAs
                        if constexpr (GN) {
                            if (mi >= TM / 2 && j == 3) fold(afr[nxt][(mi - TM / 2) * 2 + ni], cfr[nxt]);      // one fragment per 4 MFMAs
                        }
                        ND_SB;
                    }
                }
            }
            if (st == 2) {
                // every read of this chunk's stage has returned (k-step 3's were waited for above), chunk ch + 1 landed long
                // ago (k-step 1's vmcnt(LPS) left nothing older than its own loads in flight)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    }
#undef ND_SB
    // the run-ahead loads of the last k-step are still in flight: keep their registers allocated until they have returned
    if constexpr (TM == 4)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[1][0]), "+v"(bfr[1][1]), "+v"(cfr[0][0]), "+v"(cfr[0][1]),
                       "+v"(cfr[1][0]), "+v"(cfr[1][1]), "+v"(afr[0][0]), "+v"(afr[0][1]), "+v"(afr[0][TM - 2]), "+v"(afr[0][TM - 1]),
                       "+v"(afr[1][0]), "+v"(afr[1][1]), "+v"(afr[1][TM - 2]), "+v"(afr[1][TM - 1])
                     :
                     : "memory");
    else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[1][0]), "+v"(bfr[1][1]), "+v"(cfr[0][0]), "+v"(cfr[0][1]),
                       "+v"(cfr[1][0]), "+v"(cfr[1][1]), "+v"(afr[0][0]), "+v"(afr[0][1]), "+v"(afr[1][0]), "+v"(afr[1][1])
                     :
                     : "memory");
    __builtin_amdgcn_s_setprio(3);

    // ---- epilogue: lane = one pixel, register group g4 = 4 consecutive output channels 8*g4 + 4*lh .. +3 of the n tile
    const bool vec_ok = ((p.ldo & 3) == 0) && (!p.res || (p.ldr & 3) == 0);
    if constexpr (STATS) {
        // GroupNorm statistics of the output for free (the attention block's output projection + residual feeds the next
        // block's in_norm, model.py:291,190): per channel, the sum / sum of squares of what this wave row (128 consecutive
        // pixels of ONE image: the host requires H*W % 128 == 0) stores, one fp32 row per (image, 128-pixel run):
        // chstats [NI][mbi][sum | sum of squares][N], plain stores, every entry written by every launch (no atomics);
        // folded by nd_groupnorm_stats_from_partials.  The host guarantees M % 256 == 0, N % 4 == 0 and aligned rows.
        const int mrow = m0 + wm * (TM * 32);
        const int hw = p.mbi * 128;
        const int img = mrow / hw;
        float* prow = p.chstats + (((size_t)img * p.mbi + (mrow - img * hw) / 128) * 2) * p.N;
        f32x4 keep_s[TN][4], keep_q[TN][4];          // (TM == 2 only)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = n0 + (wn * TN + ni) * 32 + 8 * g4 + 4 * lh;
                f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
                if (n + 3 < p.N) {
                    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias) bv = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) {
                        const size_t opix = (size_t)(mrow + mi * 32 + l31);
                        f32x4 v = {acc[mi][ni][4 * g4 + 0], acc[mi][ni][4 * g4 + 1], acc[mi][ni][4 * g4 + 2], acc[mi][ni][4 * g4 + 3]};
                        if (p.bias) v += bv;
                        if (p.res && !rinit) v += *reinterpret_cast<const f32x4*>(p.res + opix * p.ldr + n);
                        if (p.silu_out) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
                        }
                        *reinterpret_cast<f32x4*>(p.out + opix * p.ldo + n) = v;
                        ssum += v;
                        ssq += v * v;
                    }
                }
                sum8_over_32_lanes(ssum, ssq);            // DPP adds (nd_conv_common.h); the totals sit in lanes 16..31 / 48..63
                if constexpr (TM == 4) {
                    if (l31 == 31 && n + 3 < p.N) {
                        *reinterpret_cast<f32x4*>(prow + n) = ssum;
                        *reinterpret_cast<f32x4*>(prow + p.N + n) = ssq;
                    }
                } else {
                    // 128-pixel blocks: a wave row is 64 pixels, a statistics row 128 -- the lower wave row hands its sums to the
                    // upper one through LDS behind the stages (bytes 49152..; the stages of a 128-pixel block end at 49152)
                    f32x4* xs = reinterpret_cast<f32x4*>(smem + 49152 / 4) + (((wn * TN + ni) * 4 + g4) * 2 + lh) * 2;
                    if (wm == 1 && l31 == 31) { xs[0] = ssum; xs[1] = ssq; }
                    keep_s[ni][g4] = ssum;
                    keep_q[ni][g4] = ssq;
                }
            }
        }
        if constexpr (TM == 2) {
            __syncthreads();
            if (wm == 0 && l31 == 31) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int n = n0 + (wn * TN + ni) * 32 + 8 * g4 + 4 * lh;
                        if (n + 3 < p.N) {
                            const f32x4* xs = reinterpret_cast<const f32x4*>(smem + 49152 / 4) + (((wn * TN + ni) * 4 + g4) * 2 + lh) * 2;
                            *reinterpret_cast<f32x4*>(prow + n) = keep_s[ni][g4] + xs[0];
                            *reinterpret_cast<f32x4*>(prow + p.N + n) = keep_q[ni][g4] + xs[1];
                        }
                    }
            }
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + (wm * TM + mi) * 32 + l31;
        if (m < M) {
            const size_t opix = (size_t)m;
            const float* rr = (p.res && !rinit) ? p.res + opix * p.ldr : nullptr;
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

template <bool GN, bool STATS, int TM>
static int launch_gemm4_as(const ConvArgs& a, int grid, hipStream_t s) {
    auto kern = gemm4_kernel<GN, STATS, TM>;
    static bool attr_set[kMaxDevices] = {};
    if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), attr_set, "nd_conv_nhwc")) return rc;
    // 256-pixel blocks: two 32 KiB stages; 128-pixel blocks: two 16 KiB stages at 0 and 32 KiB + 1 KiB of statistics hand-over
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), TM == 4 ? (size_t)64 * 1024 : (size_t)50176, s, a);
    return check_launch("nd_conv_nhwc");
}

// tm: 4 = 256-pixel blocks, 2 = 128-pixel blocks (the caller sized a.mt and the grid for it)
int launch_gemm4(const ConvArgs& a, int grid, hipStream_t s, int tm) {
    if (a.chstats)          // (the host admits statistics without a fused GroupNorm only)
        return tm == 2 ? launch_gemm4_as<false, true, 2>(a, grid, s) : launch_gemm4_as<false, true, 4>(a, grid, s);
    if (a.gnA) return tm == 2 ? launch_gemm4_as<true, false, 2>(a, grid, s) : launch_gemm4_as<true, false, 4>(a, grid, s);
    return tm == 2 ? launch_gemm4_as<false, false, 2>(a, grid, s) : launch_gemm4_as<false, false, 4>(a, grid, s);
}

}  // namespace nd

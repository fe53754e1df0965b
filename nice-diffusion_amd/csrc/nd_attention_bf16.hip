// bf16 form of K7: attention core  softmax(q k^T * scale) v  over T = H*W tokens, per (image, head)
// (model.py:266-287), for the bf16 path of BASELINE configs[3], [4].  bf16 q/k/v in, bf16 out; both contractions on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the softmax (max, exp2, running sum, rescale) stays in fp32.
//
// Same flash-style structure as the fp32 kernel (nd_attention.hip):
//   block  = 4 waves x 32 queries of one (image, head); K / V tiles of 64 keys staged in LDS, shared by the waves.
//   S^T    = K . Q^T, keys as MFMA rows, queries as columns: a lane holds ONE query's scores, so the row reduction is
//            15 in-lane ops + one exchange with lane^32, and the probabilities -- converted pairwise to bf16 -- are
//            already the B operand of O^T += V^T . P^T (accumulator as operand; k order 16s + 8(j>>2) + 4h + (j&3)).
//   V^T    the second product contracts over keys, so its A operand needs 4 consecutive KEYS of one d per 8-byte read:
//            V is transposed while staging (4 keys x 8 d per thread: 4 16-byte loads, 8 ds_write_b64) into rows of
//            64 + 4 keys (136-byte stride: 32 lanes reading 32 rows hit all 64 banks).
#include "nd_common.h"

namespace nd {

typedef __bf16 at_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 at_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int AH_KT = 64;        // keys per LDS tile
constexpr int AH_VLD = 68;       // keys per V^T row (padded)

struct AttnArgsH {
    const __bf16* qkv;
    __bf16* out;
    int ld_qkv, ld_out;
    int T, heads, hd;
    int q_off, k_off, v_off, head_stride;
    float scale_log2e;
};

__device__ __forceinline__ at_bf16x8 at_as_bf16x8(const u32x4& v) {
    union { u32x4 u; at_bf16x8 h; } c;
    c.u = v;
    return c.h;
}

// HDP: head dim padded to 64 / 128 / 256 (template), hd: actual head dim (multiple of 8, <= HDP)
template <int HDP>
__global__ void __launch_bounds__(256, (HDP <= 128 ? 2 : 1))      // hd <= 128: two blocks per CU keep their registers <= 256
    attention_bf16_kernel(const AttnArgsH p) {
    constexpr int NT = 256;
    constexpr int BQ = 128;                // queries per block
    constexpr int SPR = HDP / 8;           // 16-byte slots per K row
    constexpr int NKS = HDP / 16;          // k-steps of the first product
    constexpr int NDT = HDP / 32;          // 32-wide d tiles of O^T
    extern __shared__ __attribute__((aligned(16))) unsigned int smem[];
    unsigned int* Ks = smem;                               // [AH_KT][HDP] bf16, 16-byte slots XOR-swizzled
    __bf16* Vt = reinterpret_cast<__bf16*>(smem + AH_KT * HDP / 2);   // [HDP][AH_VLD] bf16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    // XCD-aware order: workgroups go to the 8 XCDs round-robin by their linear id, so the q tiles of one (image, head) --
    // which stream the same K / V -- would land on 8 different L2s (PMC: 6x the algorithmic bytes fetched); give every XCD
    // a contiguous run of (bh, q tile) pairs instead (the map of the convolution kernels, nd_conv_mfma.hip)
    const int lin = blockIdx.y * gridDim.x + blockIdx.x, total = gridDim.x * gridDim.y;
    const int xq = total >> 3, xr = total & 7, xcd = lin & 7;
    const int idp = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    const int bh = idp / (int)gridDim.x, qtile = idp - bh * (int)gridDim.x;
    const int b = bh / p.heads, head = bh - b * p.heads;
    const int q0 = qtile * BQ + wave * 32;
    const size_t rowbase = (size_t)b * p.T;
    const int hoff = head * p.head_stride;

    // ---- Q fragments: lane (query i, half h) holds Q[i][16c + 8h .. +7] for k-step c
    u32x4 q[NKS];
    {
        int qi = q0 + l31;
        if (qi >= p.T) qi = p.T - 1;
        const __bf16* qp = p.qkv + (rowbase + qi) * p.ld_qkv + p.q_off + hoff;
#pragma unroll
        for (int c = 0; c < NKS; ++c) {
            const int d = 16 * c + 8 * lh;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (d < p.hd) v = *reinterpret_cast<const u32x4*>(qp + d);
            q[c] = v;
        }
    }

    f32x16 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int ntiles = (p.T + AH_KT - 1) / AH_KT;
    // K / V tiles are fetched one tile ahead into registers (hd <= 128: 32 registers) and parked in LDS behind the barrier
    // that ends the previous tile, so a tile's global latency runs under the previous tile's products instead of in
    // front of its own (round 3: the kernel sat at 0.15 of the matrix pipe with 0.46 of its wave time parked)
    constexpr bool PREFETCH = (HDP <= 128);
    constexpr int KI = (AH_KT * SPR + NT - 1) / NT;            // K items (16 bytes) per thread and tile
    constexpr int VI = ((AH_KT / 4) * SPR + NT - 1) / NT;      // V items (4 keys x 8 d) per thread and tile
    u32x4 kpre[KI], vpre[VI][4];
    auto fetch = [&](int key0) {
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int it = tid + i * NT;
            const int row = it / SPR;
            const int sl = it - row * SPR;
            const int key = key0 + row;
            kpre[i] = u32x4{0u, 0u, 0u, 0u};
            if (it < AH_KT * SPR && key < p.T && sl * 8 < p.hd)
                kpre[i] = *reinterpret_cast<const u32x4*>(p.qkv + (rowbase + key) * p.ld_qkv + hoff + p.k_off + sl * 8);
        }
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            const int it = tid + i * NT;
            const int kg = it % (AH_KT / 4);
            const int sl = it / (AH_KT / 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = key0 + kg * 4 + r;
                vpre[i][r] = u32x4{0u, 0u, 0u, 0u};
                if (it < (AH_KT / 4) * SPR && key < p.T && sl * 8 < p.hd)
                    vpre[i][r] = *reinterpret_cast<const u32x4*>(p.qkv + (rowbase + key) * p.ld_qkv + hoff + p.v_off + sl * 8);
            }
        }
    };
    // K row-major (swizzled); V transposed: item = (4 keys, 8 d), dword pairs (key k, k+1) of one d are packed
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < KI; ++i) {
            const int it = tid + i * NT;
            const int row = it / SPR;
            const int sl = it - row * SPR;
            const int swz = (SPR >= 16) ? (row & 15) : ((row >> 1) & 7);
            if (it < AH_KT * SPR) *reinterpret_cast<u32x4*>(Ks + row * (HDP / 2) + ((sl ^ swz) << 2)) = kpre[i];
        }
#pragma unroll
        for (int i = 0; i < VI; ++i) {
            const int it = tid + i * NT;
            const int kg = it % (AH_KT / 4);
            const int sl = it / (AH_KT / 4);
            if (it < (AH_KT / 4) * SPR) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int j = e >> 1;
                    u32x2 pk;
                    if (e & 1) {
                        pk[0] = (vpre[i][0][j] >> 16) | (vpre[i][1][j] & 0xffff0000u);
                        pk[1] = (vpre[i][2][j] >> 16) | (vpre[i][3][j] & 0xffff0000u);
                    } else {
                        pk[0] = (vpre[i][0][j] & 0xffffu) | (vpre[i][1][j] << 16);
                        pk[1] = (vpre[i][2][j] & 0xffffu) | (vpre[i][3][j] << 16);
                    }
                    *reinterpret_cast<u32x2*>(Vt + (sl * 8 + e) * AH_VLD + kg * 4) = pk;
                }
            }
        }
    };
    if constexpr (PREFETCH) fetch(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * AH_KT;
        __syncthreads();   // previous tile fully consumed
        if constexpr (!PREFETCH) fetch(key0);
        park();
        __syncthreads();
        if constexpr (PREFETCH) {
            if (kt + 1 < ntiles) fetch(key0 + AH_KT);
        }

#pragma unroll 1
        for (int sub = 0; sub < AH_KT / 32; ++sub) {
            const int kbase = key0 + sub * 32;
            if (kbase >= p.T) break;
            // ---- S^T[key][query] = sum_d K[key][d] * Q[query][d]
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
            const int krow = sub * 32 + l31;
            const int kswz = (SPR >= 16) ? (krow & 15) : ((krow >> 1) & 7);
#pragma unroll
            for (int c = 0; c < NKS; ++c) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(Ks + krow * (HDP / 2) + (((2 * c + lh) ^ kswz) << 2));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_as_bf16x8(a), at_as_bf16x8(q[c]), s, 0, 0, 0);
            }
            // ---- online softmax for this lane's query; register e <-> key kbase + (e&3) + 8*(e>>2) + 4*lh
            float mx = -1e30f;
            if (kbase + 32 <= p.T) {          // whole sub-tile inside the sequence (always, when T % 32 == 0): no masks
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    s[e] *= p.scale_log2e;
                    mx = fmaxf(mx, s[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kbase + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    s[e] = (key < p.T) ? s[e] * p.scale_log2e : -1e30f;
                    mx = fmaxf(mx, s[e]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float ps = 0.f;
            at_bf16x8 pf[2];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe = __builtin_amdgcn_exp2f(s[e] - m_new);
                const __bf16 ph = (__bf16)pe;
                pf[e >> 3][e & 7] = ph;
                ps += (float)ph;          // the sum of what the second product actually multiplies
            }
            ps += __shfl_xor(ps, 32);
            l_run = l_run * alpha + ps;
            m_run = m_new;
            // the rescale is skipped while no lane's running maximum moved (alpha == 1 exactly: same bits)
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
            }
            // ---- O^T[d][query] += sum_key V^T[d][key] * P^T[key][query]; k-step s2 covers keys 16*s2 .. +15 of the subtile,
            //      fragment element j <-> key 16*s2 + 8*(j>>2) + 4*lh + (j&3)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const __bf16* vr = Vt + (dt * 32 + l31) * AH_VLD + sub * 32 + 16 * s2 + 4 * lh;
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(vr);
                    const u32x2 hi = *reinterpret_cast<const u32x2*>(vr + 8);
                    const u32x4 a = {lo[0], lo[1], hi[0], hi[1]};
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_as_bf16x8(a), pf[s2], o[dt], 0, 0, 0);
                }
            }
        }
    }

    // ---- write O[query][head*hd + d] = O^T[d][query] / l
    const int qi = q0 + l31;
    if (qi < p.T) {
        const float inv = 1.0f / l_run;
        __bf16* op = p.out + (rowbase + qi) * p.ld_out + head * p.hd;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = dt * 32 + 8 * g4 + 4 * lh;
                if (d < p.hd) {
                    const at_bf16x4 v = {(__bf16)(o[dt][4 * g4 + 0] * inv), (__bf16)(o[dt][4 * g4 + 1] * inv),
                                         (__bf16)(o[dt][4 * g4 + 2] * inv), (__bf16)(o[dt][4 * g4 + 3] * inv)};
                    *reinterpret_cast<at_bf16x4*>(op + d) = v;
                }
            }
        }
    }
}

template <int HDP>
static int launch_attn_h(const AttnArgsH& a, int B, hipStream_t s) {
    auto kern = attention_bf16_kernel<HDP>;
    const size_t lds = (size_t)AH_KT * HDP * 2 + (size_t)HDP * AH_VLD * 2;
    static bool attr_set[kMaxDevices] = {};
    if (lds > 64 * 1024) {
        if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), attr_set, "nd_attention_bf16_nhwc")) return rc;
    }
    dim3 grid((a.T + 127) / 128, B * a.heads);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return check_launch("nd_attention_bf16_nhwc");
}

}  // namespace nd

using namespace nd;

extern "C" int nd_attention_bf16_nhwc(const void* qkv, int ld_qkv, void* out, int ld_out, int B, int T, int heads,
                                      int hd, int q_off, int k_off, int v_off, int head_stride, float scale,
                                      nd_stream_t stream) {
    const char* fn = "nd_attention_bf16_nhwc";
    ND_REQUIRE(qkv && out && B > 0 && T > 0 && heads > 0, fn, "bad arguments");
    ND_REQUIRE(hd > 0 && (hd & 7) == 0 && hd <= 256, fn, "head dim must be a multiple of 8 and <= 256");
    ND_REQUIRE((ld_qkv & 7) == 0 && (ld_out & 3) == 0 && aligned16(qkv) && (reinterpret_cast<uintptr_t>(out) & 7u) == 0, fn,
               "alignment");
    ND_REQUIRE((q_off & 7) == 0 && (k_off & 7) == 0 && (v_off & 7) == 0 && (head_stride & 7) == 0, fn,
               "offsets must be multiples of 8 (16-byte loads)");
    ND_REQUIRE(ld_out >= heads * hd && (hd & 3) == 0, fn, "ld_out < heads*hd");
    ND_REQUIRE((long)B * heads <= 65535, fn, "too many (image, head) pairs for grid.y");
    AttnArgsH a{static_cast<const __bf16*>(qkv), static_cast<__bf16*>(out), ld_qkv, ld_out, T, heads, hd, q_off, k_off,
                v_off, head_stride, scale * 1.4426950408889634f};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hd <= 64) return launch_attn_h<64>(a, B, s);
    if (hd <= 128) return launch_attn_h<128>(a, B, s);
    return launch_attn_h<256>(a, B, s);
}

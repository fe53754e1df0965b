// K7: attention core  softmax(q k^T * scale) v  over T = H*W tokens, per (image, head)  (model.py:266-287).
//
// Flash-style: the T x T score matrix never leaves the CU.  All contractions run on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32), so the numerics are plain fp32 FMA chains.
//
//   block  = 4 waves = 128 queries of one (image, head); each wave owns 32 queries.
//   K / V  tiles of 64 keys staged in LDS (shared by the 4 waves); K is XOR-swizzled for the b128 row reads.
//   S^T    = K . Q^T is computed with keys as MFMA rows and queries as columns, so a lane holds the scores of ONE
//            query (its column) for 16 keys per 32-key sub-tile: the softmax row reduction is 15 in-lane ops plus
//            one exchange with lane^32, and the probabilities are already in the B-operand layout of the second
//            product O^T += V^T . P^T (accumulator-as-operand: no LDS round trip, no transposes).
//   O^T    accumulators have the query on the lane, so the online-softmax rescale is one multiply per register,
//            and a lane owns 4 consecutive output channels per register group -> 16-byte stores.
#include "nd_common.h"
#include <stdlib.h>

namespace nd {

constexpr int AT_KT = 64;              // keys per LDS tile

struct AttnArgs {
    const float* qkv;
    float* out;
    int ld_qkv, ld_out;
    int T, heads, hd;
    int q_off, k_off, v_off, head_stride;
    float scale_log2e;
};

// HDP: head dim padded to a multiple of 32 (template), hd: actual head dim (multiple of 8, <= HDP)
// WAVES = 4 or 8 waves per block (32 queries each): the K / V tile staged in LDS is shared by all of them
template <int HDP, int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
    attention_kernel(const AttnArgs p) {
    // K / V tiles prefetched through registers: measured only as a register count -- 16 more registers put the 8-wave
    // instantiation at 136 (one block per CU instead of two) or, capped at 128, into scratch -- so it is off; the code path
    // stays for a head dim of 32, where it would fit
    constexpr bool PRE = false;
    constexpr int AT_NT = WAVES * 64;
    constexpr int AT_BQ = WAVES * 32;      // queries per block
    constexpr int SPR = HDP / 4;           // 16-byte slots per K row
    constexpr int NC = HDP / 8;            // b128 chunks along d
    constexpr int NDT = HDP / 32;          // 32-wide d tiles of O^T
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                      // [AT_KT][HDP] swizzled
    float* Vs = smem + AT_KT * HDP;        // [AT_KT][HDP] linear

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
    const int q0 = qtile * AT_BQ + wave * 32;
    const size_t rowbase = (size_t)b * p.T;
    const int hoff = head * p.head_stride;

    // ---- Q fragment: lane (i, h) holds Q[i][8c + 4h + j] in q[4c + j]
    float q[NC * 4];
    {
        int qi = q0 + l31;
        if (qi >= p.T) qi = p.T - 1;
        const float* qp = p.qkv + (rowbase + qi) * p.ld_qkv + p.q_off + hoff;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int d = 8 * c + 4 * lh;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (d < p.hd) v = *reinterpret_cast<const f32x4*>(qp + d);
            // softmax scale (and log2 e) folded into Q once: 16 multiplies per 32-key sub-tile fewer beside the MFMAs
#pragma unroll
            for (int j = 0; j < 4; ++j) q[4 * c + j] = v[j] * p.scale_log2e;
        }
    }

    f32x16 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int ntiles = (p.T + AT_KT - 1) / AT_KT;
    // ---- K / V tiles: a thread's pieces of tile kt + 1 are requested right behind the barrier that publishes tile kt and
    //      sit in registers while tile kt is consumed (the global latency passes under the tile's MFMAs), then go to LDS
    constexpr int NPC = PRE ? (AT_KT * SPR + AT_NT - 1) / AT_NT : 1;          // 16-byte pieces of K (and of V) per thread and tile
    f32x4 kreg[NPC], vreg[NPC];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int it = tid + i * AT_NT;
            const int row = it / SPR;
            const int sl = it - row * SPR;
            const int key = kt * AT_KT + row;
            kreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            vreg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (it < AT_KT * SPR && key < p.T && sl * 4 < p.hd) {
                const float* base = p.qkv + (rowbase + key) * p.ld_qkv + hoff + sl * 4;
                kreg[i] = *reinterpret_cast<const f32x4*>(base + p.k_off);
                vreg[i] = *reinterpret_cast<const f32x4*>(base + p.v_off);
            }
        }
    };
    if constexpr (PRE) fetch(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * AT_KT;
        __syncthreads();   // previous tile fully consumed
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const int it = tid + i * AT_NT;
                if (it < AT_KT * SPR) {
                    const int row = it / SPR;
                    const int sl = it - row * SPR;
                    const int swz = (SPR >= 16) ? (row & 15) : ((row >> 1) & 7);
                    *reinterpret_cast<f32x4*>(Ks + row * HDP + ((sl ^ swz) << 2)) = kreg[i];
                    *reinterpret_cast<f32x4*>(Vs + row * HDP + (sl << 2)) = vreg[i];
                }
            }
        } else {
            // ---- stage K and V tiles (zero-filled beyond T / hd)
            for (int it = tid; it < AT_KT * SPR; it += AT_NT) {
                const int row = it / SPR;
                const int sl = it - row * SPR;
                const int key = key0 + row;
                f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
                if (key < p.T && sl * 4 < p.hd) {
                    const float* base = p.qkv + (rowbase + key) * p.ld_qkv + hoff + sl * 4;
                    kv = *reinterpret_cast<const f32x4*>(base + p.k_off);
                    vv = *reinterpret_cast<const f32x4*>(base + p.v_off);
                }
                const int swz = (SPR >= 16) ? (row & 15) : ((row >> 1) & 7);
                *reinterpret_cast<f32x4*>(Ks + row * HDP + ((sl ^ swz) << 2)) = kv;
                *reinterpret_cast<f32x4*>(Vs + row * HDP + (sl << 2)) = vv;
            }
        }
        __syncthreads();
        if constexpr (PRE) {
            if (kt + 1 < ntiles) fetch(kt + 1);
        }

#pragma unroll 1
        for (int sub = 0; sub < AT_KT / 32; ++sub) {
            const int kbase = key0 + sub * 32;
            if (kbase >= p.T) break;
            // ---- S^T[key][query] = sum_d K[key][d] * Q[query][d]
            f32x16 s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
            const int krow = sub * 32 + l31;
            const int kswz = (SPR >= 16) ? (krow & 15) : ((krow >> 1) & 7);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(Ks + krow * HDP + (((2 * c + lh) ^ kswz) << 2));
#pragma unroll
                for (int j = 0; j < 4; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], q[4 * c + j], s, 0, 0, 0);
            }
            // ---- online softmax for this lane's query; register e <-> key kbase + (e&3) + 8*(e>>2) + 4*lh
            float mx = -1e30f;
            if (kbase + 32 <= p.T) {          // whole sub-tile inside the sequence (always, when T % 32 == 0): no masks
#pragma unroll
                for (int e = 0; e < 16; ++e) mx = fmaxf(mx, s[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kbase + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    s[e] = (key < p.T) ? s[e] : -1e30f;
                    mx = fmaxf(mx, s[e]);
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            // (s - m) and the row sum as packed adds (v_pk_add_f32: two elements per instruction)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 mneg = {-m_new, -m_new};
            f32x2 ps2 = {0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                f32x2 d = f32x2{s[e], s[e + 1]} + mneg;
                d[0] = __builtin_amdgcn_exp2f(d[0]);
                d[1] = __builtin_amdgcn_exp2f(d[1]);
                s[e] = d[0];
                s[e + 1] = d[1];
                ps2 += d;
            }
            float ps = ps2[0] + ps2[1];
            ps += __shfl_xor(ps, 32);
            l_run = l_run * alpha + ps;
            m_run = m_new;
            // the rescale is skipped while no lane's running maximum moved (alpha == 1 exactly: same bits); on the fp32 matrix
            // pipe every vector instruction saved is matrix time
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[dt][e] *= alpha;
            }
            // ---- O^T[d][query] += sum_key V[key][d] * P[key][query]; MFMA step e contracts keys {e-row, e-row + 4}
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int vrow = sub * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const float a = Vs[vrow * HDP + dt * 32 + l31];
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[e], o[dt], 0, 0, 0);
                }
            }
        }
    }

    // ---- write O[query][head*hd + d] = O^T[d][query] / l
    const int qi = q0 + l31;
    if (qi < p.T) {
        const float inv = 1.0f / l_run;
        float* op = p.out + (rowbase + qi) * p.ld_out + head * p.hd;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = dt * 32 + 8 * g4 + 4 * lh;
                if (d < p.hd) {
                    f32x4 v = {o[dt][4 * g4 + 0] * inv, o[dt][4 * g4 + 1] * inv, o[dt][4 * g4 + 2] * inv,
                               o[dt][4 * g4 + 3] * inv};
                    *reinterpret_cast<f32x4*>(op + d) = v;
                }
            }
        }
    }
}

template <int HDP, int WAVES>
static int launch_attn_w(const AttnArgs& a, int B, hipStream_t s) {
    constexpr int AT_NT = WAVES * 64, AT_BQ = WAVES * 32;
    auto kern = attention_kernel<HDP, WAVES>;
    const size_t lds = (size_t)2 * AT_KT * HDP * sizeof(float);
    static bool attr_set[kMaxDevices] = {};
    if (lds > 64 * 1024) {
        if (int rc = ensure_max_lds(reinterpret_cast<const void*>(kern), attr_set, "nd_attention_nhwc")) return rc;
    }
    dim3 grid((a.T + AT_BQ - 1) / AT_BQ, B * a.heads);
    hipLaunchKernelGGL(kern, grid, dim3(AT_NT), lds, s, a);
    return check_launch("nd_attention_nhwc");
}

static int env_attn_waves() {
    static int v = -2;
    if (v == -2) {
        const char* e = getenv("ND_ATTN_WAVES");
        v = e ? atoi(e) : 0;
    }
    return v;
}

template <int HDP>
static int launch_attn(const AttnArgs& a, int B, hipStream_t s) {
    int w = env_attn_waves();
    if (w != 4 && w != 8) w = (HDP <= 64 && a.T >= 512) ? 8 : 4;    // measured: +8 % at T = 1024, -7 % at T = 256
    if (HDP > 64) w = 4;
    if constexpr (HDP <= 64) {
        if (w == 8) return launch_attn_w<HDP, 8>(a, B, s);
    }
    return launch_attn_w<HDP, 4>(a, B, s);
}

}  // namespace nd

using namespace nd;

extern "C" int nd_attention_nhwc(const float* qkv, int ld_qkv, float* out, int ld_out, int B, int T, int heads,
                                 int hd, int q_off, int k_off, int v_off, int head_stride, float scale,
                                 nd_stream_t stream) {
    const char* fn = "nd_attention_nhwc";
    ND_REQUIRE(qkv && out && B > 0 && T > 0 && heads > 0, fn, "bad arguments");
    ND_REQUIRE(hd > 0 && (hd & 7) == 0 && hd <= 256, fn, "head dim must be a multiple of 8 and <= 256");
    ND_REQUIRE((ld_qkv & 3) == 0 && (ld_out & 3) == 0 && aligned16(qkv) && aligned16(out), fn, "alignment");
    ND_REQUIRE((q_off & 3) == 0 && (k_off & 3) == 0 && (v_off & 3) == 0 && (head_stride & 3) == 0, fn,
               "offsets must be multiples of 4");
    ND_REQUIRE(ld_out >= heads * hd, fn, "ld_out < heads*hd");
    ND_REQUIRE((long)B * heads <= 65535, fn, "too many (image, head) pairs for grid.y");
    AttnArgs a{qkv, out, ld_qkv, ld_out, T, heads, hd, q_off, k_off, v_off, head_stride,
               scale * 1.4426950408889634f};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hd <= 32) return launch_attn<32>(a, B, s);
    if (hd <= 64) return launch_attn<64>(a, B, s);
    if (hd <= 128) return launch_attn<128>(a, B, s);
    if (hd <= 192) return launch_attn<192>(a, B, s);
    return launch_attn<256>(a, B, s);
}

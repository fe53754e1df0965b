// K3/K4: GroupNorm (32 groups) over NHWC activations, with the ops the reference wraps around it fused in:
//   silu(in_norm(x))                         model.py:190     (stats + apply|SILU)
//   avg_pool2d(silu(in_norm(x)))             model.py:111,192 (apply|SILU|POOL2)
//   silu(out_norm(h) * (1 + scale) + shift)  model.py:201-207 (apply with scale/shift)
//   silu(out_norm(h + emb))                  model.py:205-207 (addvec)
//   norm(x) of the attention block           model.py:264     (apply, no SiLU)
//   out[0], out[1]                           model.py:446-447
// and the channel concatenation of the skip connection (model.py:474) read as two sources.
//
// HBM-bound.  Pass 1 (stats) streams the tensor once with 16-byte loads along the channel axis (fully coalesced in
// NHWC) and reduces per-channel sums in float64 (the fp64 vector rate is far above what the stream needs, and
// E[x^2]-E[x]^2 in float64 has no cancellation problem for fp32 data); per-(image, group) totals are combined
// across blocks with one float64 atomic pair per block and group.  Pass 2 (apply) folds mean/rstd/gamma/beta and
// the AdaGN scale/shift into one FMA per element (coefficients staged in LDS per block), applies SiLU and
// optionally the 2x2 average pool, and writes 16 bytes per lane.
#include "nd_common.h"

namespace nd {

constexpr int GN_NT = 256;
constexpr int GN_MAXQ = 2;   // channel quads per thread (C <= GN_NT * 4 * GN_MAXQ = 2048)

struct GnSrc {
    const float* x0;
    const float* x1;
    int C0, C1, ldx0, ldx1;
};

__device__ __forceinline__ f32x4 gn_load(const GnSrc& s, size_t pix, int c) {
    const float* p = (c < s.C0) ? (s.x0 + pix * s.ldx0 + c) : (s.x1 + pix * s.ldx1 + (c - s.C0));
    return *reinterpret_cast<const f32x4*>(p);
}

// grid: (pixel chunks, NI).  Threads are laid out as PY pixel rows x QX channel quads.
__global__ void __launch_bounds__(GN_NT)
    gn_stats_kernel(GnSrc s, const float* addvec, int ld_add, double* stats, int HW, int G, int QX, int PY,
                    int pix_per_block) {
    extern __shared__ __attribute__((aligned(16))) double sh[];   // [C][2]
    const int C = s.C0 + s.C1;
    const int CQ = C >> 2;
    const int img = blockIdx.y;
    const int tid = threadIdx.x;
    const int tq = tid % QX;
    const int tp = tid / QX;
    for (int i = tid; i < 2 * C; i += GN_NT) sh[i] = 0.0;
    __syncthreads();

    const int p0 = blockIdx.x * pix_per_block;
    int p1 = p0 + pix_per_block;
    if (p1 > HW) p1 = HW;

    if (tp < PY) {
        double sum[GN_MAXQ][4], ssq[GN_MAXQ][4];
        f32x4 add[GN_MAXQ];
#pragma unroll
        for (int j = 0; j < GN_MAXQ; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[j][e] = ssq[j][e] = 0.0;
            add[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int qd = tq + j * QX;
            if (addvec && qd < CQ) add[j] = *reinterpret_cast<const f32x4*>(addvec + (size_t)img * ld_add + qd * 4);
        }
        const size_t base = (size_t)img * HW;
        for (int p = p0 + tp; p < p1; p += PY) {
#pragma unroll
            for (int j = 0; j < GN_MAXQ; ++j) {
                const int qd = tq + j * QX;
                if (qd < CQ) {
                    f32x4 v = gn_load(s, base + p, qd * 4) + add[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double d = (double)v[e];
                        sum[j][e] += d;
                        ssq[j][e] += d * d;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < GN_MAXQ; ++j) {
            const int qd = tq + j * QX;
            if (qd < CQ) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    atomicAdd(&sh[(qd * 4 + e) * 2 + 0], sum[j][e]);
                    atomicAdd(&sh[(qd * 4 + e) * 2 + 1], ssq[j][e]);
                }
            }
        }
    }
    __syncthreads();
    const int cpg = C / G;
    for (int g = tid; g < G; g += GN_NT) {
        double a = 0.0, b = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            a += sh[c * 2];
            b += sh[c * 2 + 1];
        }
        atomicAdd(&stats[((size_t)img * G + g) * 2 + 0], a);
        atomicAdd(&stats[((size_t)img * G + g) * 2 + 1], b);
    }
}

// grid: (pixel chunks, NI); LDS: coefficient pairs A[c], B[c] so that y = x*A + B
template <bool POOL>
__global__ void __launch_bounds__(GN_NT)
    gn_apply_kernel(GnSrc s, const float* addvec, int ld_add, const double* stats, const float* gamma,
                    const float* beta, const float* scale, const float* shift, int ld_ss, float* out, int ldo,
                    int H, int W, int G, float eps, int silu, int pix_per_block) {
    extern __shared__ __attribute__((aligned(16))) float shf[];   // A[C] | B[C]
    const int C = s.C0 + s.C1;
    const int CQ = C >> 2;
    const int img = blockIdx.y;
    const int tid = threadIdx.x;
    const int HW = H * W;
    const int cpg = C / G;
    const double inv_n = 1.0 / ((double)cpg * (double)HW);
    float* cA = shf;
    float* cB = shf + C;
    for (int c = tid; c < C; c += GN_NT) {
        const int g = c / cpg;
        const double su = stats[((size_t)img * G + g) * 2 + 0];
        const double sq = stats[((size_t)img * G + g) * 2 + 1];
        const double mean = su * inv_n;
        double var = sq * inv_n - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        double a = rstd * (double)gamma[c];
        double b = (double)beta[c] - mean * a;
        if (addvec) b += (double)addvec[(size_t)img * ld_add + c] * a;
        if (scale) {
            const double sc = 1.0 + (double)scale[(size_t)img * ld_ss + c];
            a *= sc;
            b = b * sc + (double)shift[(size_t)img * ld_ss + c];
        }
        cA[c] = (float)a;
        cB[c] = (float)b;
    }
    __syncthreads();

    const int Ho = POOL ? (H >> 1) : H, Wo = POOL ? (W >> 1) : W;
    const int HWo = Ho * Wo;
    const int items0 = blockIdx.x * pix_per_block * CQ;
    int items1 = items0 + pix_per_block * CQ;
    const int total = HWo * CQ;
    if (items1 > total) items1 = total;
    const size_t ibase = (size_t)img * HW;
    const size_t obase = (size_t)img * HWo;
    for (int it = items0 + tid; it < items1; it += GN_NT) {
        const int po = it / CQ;
        const int qd = it - po * CQ;
        const int c = qd * 4;
        const f32x4 a = *reinterpret_cast<const f32x4*>(cA + c);
        const f32x4 b = *reinterpret_cast<const f32x4*>(cB + c);
        f32x4 y;
        if (POOL) {
            const int oy = po / Wo, ox = po - oy * Wo;
            const size_t pi = ibase + (size_t)(2 * oy) * W + 2 * ox;
            f32x4 accv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f32x4 v = gn_load(s, pi + (k >> 1) * W + (k & 1), c) * a + b;
                if (silu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fast_silu(v[e]);
                }
                accv += v;
            }
            y = accv * 0.25f;
        } else {
            y = gn_load(s, ibase + po, c) * a + b;
            if (silu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fast_silu(y[e]);
            }
        }
        *reinterpret_cast<f32x4*>(out + (obase + po) * ldo + c) = y;
    }
}

// coefficients of the fused form: y = x * A[img][c] + B[img][c]  (consumed by the conv loaders)
__global__ void gn_coeffs_kernel(const double* stats, const float* gamma, const float* beta, const float* scale,
                                 const float* shift, int ld_ss, float* coefA, float* coefB, int ld_coef, int C, int HW,
                                 int G, float eps) {
    const int img = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int cpg = C / G;
    const int g = c / cpg;
    const double inv_n = 1.0 / ((double)cpg * (double)HW);
    const double mean = stats[((size_t)img * G + g) * 2 + 0] * inv_n;
    double var = stats[((size_t)img * G + g) * 2 + 1] * inv_n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    double a = rstd * (double)gamma[c];
    double b = (double)beta[c] - mean * a;
    if (scale) {
        const double sc = 1.0 + (double)scale[(size_t)img * ld_ss + c];
        a *= sc;
        b = b * sc + (double)shift[(size_t)img * ld_ss + c];
    }
    coefA[(size_t)img * ld_coef + c] = (float)a;
    coefB[(size_t)img * ld_coef + c] = (float)b;
}

static int check_src(const char* fn, const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1, int G) {
    ND_REQUIRE(x0 != nullptr && C0 > 0 && (C0 & 3) == 0 && (ldx0 & 3) == 0 && ldx0 >= C0 && aligned16(x0), fn,
               "x0: channels/stride must be multiples of 4, pointer 16-byte aligned");
    if (C1 > 0)
        ND_REQUIRE(x1 != nullptr && (C1 & 3) == 0 && (ldx1 & 3) == 0 && ldx1 >= C1 && aligned16(x1), fn,
                   "x1: channels/stride must be multiples of 4, pointer 16-byte aligned");
    ND_REQUIRE(C1 >= 0 && G > 0 && (C0 + C1) % G == 0, fn, "channels not divisible by groups");
    ND_REQUIRE(C0 + C1 <= GN_NT * 4 * GN_MAXQ, fn, "too many channels");
    return ND_OK;
}

// partial rows [img][row][0|1][C] (sum | sum of squares per channel) of a (two-source) tensor -> per-(image, group) sums
// in float64, added to the array gn_stats_kernel fills.  grid (NI, 2 sources); one block per image and source.
__global__ void __launch_bounds__(256)
    gn_from_partials_kernel(const float* p0, int C0, int rows0, const float* p1, int C1, int rows1, double* stats, int G) {
    __shared__ double sh[2 * 64];          // [G][2], G <= 64
    const int img = blockIdx.x;
    const int src = blockIdx.y;
    const float* pp = src ? p1 : p0;
    const int Cs = src ? C1 : C0, rows_all = src ? rows1 : rows0, cbase = src ? C0 : 0;
    // blockIdx.z splits the rows
    const int rchunk = (rows_all + gridDim.z - 1) / gridDim.z;
    const int rbeg = blockIdx.z * rchunk;
    const int rows = (rbeg + rchunk < rows_all) ? rbeg + rchunk : rows_all;   // end (exclusive)
    if (rbeg >= rows) return;
    const int cpg = (C0 + C1) / G;
    for (int i = threadIdx.x; i < 2 * G; i += blockDim.x) sh[i] = 0.0;
    __syncthreads();
    // thread t owns channel c = t % Cs (fixed group) and walks the rows t / Cs, t / Cs + blockDim.x / Cs, ...
    if (Cs <= (int)blockDim.x) {
        const int per = blockDim.x / Cs;
        const int c = threadIdx.x % Cs, r0 = threadIdx.x / Cs;
        if (r0 < per) {
            double a = 0.0, b = 0.0;
            for (int r = rbeg + r0; r < rows; r += per) {
                const float* q = pp + (((size_t)img * rows_all + r) * 2) * Cs + c;
                a += (double)q[0];
                b += (double)q[Cs];
            }
            const int g = (cbase + c) / cpg;
            atomicAdd(&sh[2 * g], a);
            atomicAdd(&sh[2 * g + 1], b);
        }
    } else {
        for (int c = threadIdx.x; c < Cs; c += blockDim.x) {
            double a = 0.0, b = 0.0;
            for (int r = rbeg; r < rows; ++r) {
                const float* q = pp + (((size_t)img * rows_all + r) * 2) * Cs + c;
                a += (double)q[0];
                b += (double)q[Cs];
            }
            const int g = (cbase + c) / cpg;
            atomicAdd(&sh[2 * g], a);
            atomicAdd(&sh[2 * g + 1], b);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * G; i += blockDim.x)
        if (sh[i] != 0.0) atomicAdd(&stats[(size_t)img * G * 2 + i], sh[i]);
}

}  // namespace nd

using namespace nd;

extern "C" int nd_groupnorm_stats_nhwc(const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1,
                                       const float* addvec, int ld_add, double* stats, int NI, int HW, int G,
                                       nd_stream_t stream) {
    const char* fn = "nd_groupnorm_stats_nhwc";
    int rc = check_src(fn, x0, C0, ldx0, x1, C1, ldx1, G);
    if (rc) return rc;
    ND_REQUIRE(stats != nullptr && NI > 0 && HW > 0, fn, "bad arguments");
    if (addvec) ND_REQUIRE((ld_add & 3) == 0 && aligned16(addvec), fn, "addvec alignment");
    const int C = C0 + C1, CQ = C >> 2;
    const int QX = CQ < GN_NT ? CQ : (CQ + GN_MAXQ - 1) / GN_MAXQ;
    ND_REQUIRE(QX <= GN_NT, fn, "too many channels");
    int PY = GN_NT / QX;
    if (PY < 1) PY = 1;
    // enough blocks to fill the chip (~2048), at least 4 pixels per pixel-row of threads
    int chunks = (2048 + NI - 1) / NI;
    int ppb = (HW + chunks - 1) / chunks;
    if (ppb < 4 * PY) ppb = 4 * PY;
    chunks = (HW + ppb - 1) / ppb;
    GnSrc s{x0, C1 > 0 ? x1 : x0, C0, C1, ldx0, C1 > 0 ? ldx1 : ldx0};
    hipLaunchKernelGGL(gn_stats_kernel, dim3(chunks, NI), dim3(GN_NT), (size_t)C * 2 * sizeof(double),
                       reinterpret_cast<hipStream_t>(stream), s, addvec, ld_add, stats, HW, G, QX, PY, ppb);
    return check_launch(fn);
}

extern "C" int nd_groupnorm_apply_nhwc(const float* x0, int C0, int ldx0, const float* x1, int C1, int ldx1,
                                       const float* addvec, int ld_add, const double* stats,
                                       const float* gamma, const float* beta,
                                       const float* scale, const float* shift, int ld_ss,
                                       float* out, int ldo, int NI, int H, int W, int G, float eps, int flags,
                                       nd_stream_t stream) {
    const char* fn = "nd_groupnorm_apply_nhwc";
    int rc = check_src(fn, x0, C0, ldx0, x1, C1, ldx1, G);
    if (rc) return rc;
    const int C = C0 + C1, CQ = C >> 2;
    ND_REQUIRE(stats && gamma && beta && out && NI > 0 && H > 0 && W > 0, fn, "bad arguments");
    ND_REQUIRE((ldo & 3) == 0 && ldo >= C && aligned16(out), fn, "out alignment");
    ND_REQUIRE((scale == nullptr) == (shift == nullptr), fn, "scale and shift go together");
    const bool pool = (flags & ND_GN_POOL2) != 0;
    if (pool) ND_REQUIRE((H & 1) == 0 && (W & 1) == 0, fn, "POOL2 needs even H, W");
    const int HWo = pool ? (H >> 1) * (W >> 1) : H * W;
    int chunks = (2048 + NI - 1) / NI;
    int ppb = (HWo + chunks - 1) / chunks;
    const int min_ppb = (GN_NT * 4 + CQ - 1) / CQ;   // >= 4 items per thread
    if (ppb < min_ppb) ppb = min_ppb;
    chunks = (HWo + ppb - 1) / ppb;
    GnSrc s{x0, C1 > 0 ? x1 : x0, C0, C1, ldx0, C1 > 0 ? ldx1 : ldx0};
    const size_t lds = (size_t)C * 2 * sizeof(float);
    const int silu = (flags & ND_GN_SILU) ? 1 : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (pool)
        hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(chunks, NI), dim3(GN_NT), lds, st, s, addvec, ld_add, stats,
                           gamma, beta, scale, shift, ld_ss, out, ldo, H, W, G, eps, silu, ppb);
    else
        hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(chunks, NI), dim3(GN_NT), lds, st, s, addvec, ld_add, stats,
                           gamma, beta, scale, shift, ld_ss, out, ldo, H, W, G, eps, silu, ppb);
    return check_launch(fn);
}

extern "C" int nd_groupnorm_coeffs(const double* stats, const float* gamma, const float* beta, const float* scale,
                                   const float* shift, int ld_ss, float* coefA, float* coefB, int ld_coef,
                                   int NI, int C, int HW, int G, float eps, nd_stream_t stream) {
    const char* fn = "nd_groupnorm_coeffs";
    ND_REQUIRE(stats && gamma && beta && coefA && coefB && NI > 0 && C > 0 && HW > 0 && G > 0 && C % G == 0, fn,
               "bad arguments");
    ND_REQUIRE((scale == nullptr) == (shift == nullptr) && ld_coef >= C, fn, "scale/shift go together; ld_coef >= C");
    hipLaunchKernelGGL(gn_coeffs_kernel, dim3((C + 127) / 128, NI), dim3(128), 0, reinterpret_cast<hipStream_t>(stream),
                       stats, gamma, beta, scale, shift, ld_ss, coefA, coefB, ld_coef, C, HW, G, eps);
    return check_launch(fn);
}

extern "C" int nd_groupnorm_stats_from_partials(const float* p0, int C0, int rows0, const float* p1, int C1, int rows1,
                                               double* stats, int NI, int G, nd_stream_t stream) {
    const char* fn = "nd_groupnorm_stats_from_partials";
    ND_REQUIRE(p0 && stats && NI > 0 && C0 > 0 && rows0 > 0 && C1 >= 0 && G > 0 && G <= 64 && (C0 + C1) % G == 0, fn,
               "bad arguments");
    if (C1 > 0) ND_REQUIRE(p1 != nullptr && rows1 > 0, fn, "second source");
    const int rmax = rows0 > rows1 ? rows0 : rows1;
    int splits = rmax / 8;
    if (splits < 1) splits = 1;
    if (splits > 16) splits = 16;
    hipLaunchKernelGGL(gn_from_partials_kernel, dim3(NI, C1 > 0 ? 2 : 1, splits), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), p0, C0, rows0, C1 > 0 ? p1 : p0, C1, rows1, stats, G);
    return check_launch(fn);
}

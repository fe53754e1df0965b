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
// NHWC; 4 fp32 or 8 bf16 channels per lane) and accumulates per-channel sums in float64 registers (the fp64 vector
// rate is far above what the stream needs, and E[x^2]-E[x]^2 in float64 has no cancellation problem for fp32 data).
// Everything after the per-thread sums runs in a FIXED order, so the statistics are bitwise reproducible from run
// to run: the pixel rows of a block are folded through LDS in row order, the channels of a group in channel order,
// each block stores its per-(image, group) partial [img][block][G][2] with plain stores and is done; the CONSUMER kernel
// (apply / coefficients, a later launch on the same stream) adds the blocks' partials in block order in its prologue.
// No atomics, no tickets, no fences anywhere (a first version with a last-block ticket paid a per-block tail that cost
// more than the stream itself on 8x8 .. 32x32 maps).  Pass 2 (apply) folds mean/rstd/gamma/beta and the AdaGN scale/shift into one FMA
// per element (coefficients staged in LDS per block), applies SiLU and optionally the 2x2 average pool, and writes
// 16 bytes per lane.
#include "nd_common.h"
#include <stdlib.h>

namespace nd {

constexpr int GN_NT = 256;
constexpr int GN_MAXQ = 2;       // channel vectors per thread
constexpr int GN_MAXCH = 2048;   // channels (LDS budget of the block fold: GN_NT * 8 entries of 16 bytes)

typedef __bf16 gn_bf16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct GnVec;
template <> struct GnVec<float> {
    static constexpr int N = 4;
    typedef f32x4 Raw;
    __device__ static __forceinline__ Raw raw(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    __device__ static __forceinline__ void expand(const Raw& t, float (&v)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = t[e];
    }
    __device__ static __forceinline__ void load(const float* p, float (&v)[4]) { expand(raw(p), v); }
    __device__ static __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    }
};
template <> struct GnVec<__bf16> {
    static constexpr int N = 8;
    typedef gn_bf16x8 Raw;
    __device__ static __forceinline__ Raw raw(const __bf16* p) { return *reinterpret_cast<const gn_bf16x8*>(p); }
    __device__ static __forceinline__ void expand(const Raw& t, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)t[e];
    }
    __device__ static __forceinline__ void load(const __bf16* p, float (&v)[8]) { expand(raw(p), v); }
    __device__ static __forceinline__ void store(__bf16* p, const float (&v)[8]) {
        gn_bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (__bf16)v[e];
        *reinterpret_cast<gn_bf16x8*>(p) = t;
    }
};

template <typename T>
struct GnSrc {
    const T* x0;
    const T* x1;
    int C0, C1, ldx0, ldx1;
};

template <typename T>
__device__ __forceinline__ const T* gn_ptr(const GnSrc<T>& s, size_t pix, int c) {
    return (c < s.C0) ? (s.x0 + pix * s.ldx0 + c) : (s.x1 + pix * s.ldx1 + (c - s.C0));
}

// grid: (pixel chunks, NI).  Threads are laid out as PY pixel rows x QX channel vectors.
// partials: double [NI][gridDim.x][G][2], every entry written.  With `chrows` the block's per-CHANNEL sums are written
// instead, as fp32 rows [NI][gridDim.x][0|1][C] -- the same layout the convolutions' epilogues leave behind
// (nd_conv3x3_winograd_vstats_nhwc), folded into groups by nd_groupnorm_stats_from_partials for ANY grouping: the sums
// of a tensor are then computed once, where it is produced, and re-grouped by every norm that reads it (the up path's
// concatenation norms, model.py:474,190, no longer re-read the skip tensors).
template <typename T>
__global__ void __launch_bounds__(GN_NT)
    gn_stats_kernel(GnSrc<T> s, const float* addvec, int ld_add, double* partials, float* chrows, int HW,
                    int G, int QX, int PY, int pix_per_block) {
    constexpr int V = GnVec<T>::N;
    extern __shared__ __attribute__((aligned(16))) double sh[];   // [PY][C][2], reused for [C][2] and the group sums
    const int C = s.C0 + s.C1;
    const int CQ = C / V;
    const int img = blockIdx.y;
    const int tid = threadIdx.x;
    const int tq = tid % QX;
    const int tp = tid / QX;

    const int p0 = blockIdx.x * pix_per_block;
    int p1 = p0 + pix_per_block;
    if (p1 > HW) p1 = HW;

    if (tp < PY) {
        double sum[GN_MAXQ][V], ssq[GN_MAXQ][V];
        float add[GN_MAXQ][V];
#pragma unroll
        for (int j = 0; j < GN_MAXQ; ++j) {
#pragma unroll
            for (int e = 0; e < V; ++e) { sum[j][e] = ssq[j][e] = 0.0; add[j][e] = 0.f; }
            const int qd = tq + j * QX;
            if (addvec && qd < CQ) {
#pragma unroll
                for (int e = 0; e < V; ++e) add[j][e] = addvec[(size_t)img * ld_add + qd * V + e];
            }
        }
        const size_t base = (size_t)img * HW;
        // 4 pixels per trip: all their loads are issued before the first is consumed (the stream is latency-bound at one
        // 16-byte load in flight per thread)
        constexpr int UP = 4;
        int p = p0 + tp;
        for (; p + (UP - 1) * PY < p1; p += UP * PY) {
            typename GnVec<T>::Raw raw[UP][GN_MAXQ];
#pragma unroll
            for (int u = 0; u < UP; ++u)
#pragma unroll
                for (int j = 0; j < GN_MAXQ; ++j) {
                    const int qd = tq + j * QX;
                    if (qd < CQ) raw[u][j] = GnVec<T>::raw(gn_ptr(s, base + p + u * PY, qd * V));
                }
#pragma unroll
            for (int u = 0; u < UP; ++u)
#pragma unroll
                for (int j = 0; j < GN_MAXQ; ++j) {
                    const int qd = tq + j * QX;
                    if (qd < CQ) {
                        float v[V];
                        GnVec<T>::expand(raw[u][j], v);
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            const double d = (double)(v[e] + add[j][e]);
                            sum[j][e] += d;
                            ssq[j][e] += d * d;
                        }
                    }
                }
        }
        for (; p < p1; p += PY) {
#pragma unroll
            for (int j = 0; j < GN_MAXQ; ++j) {
                const int qd = tq + j * QX;
                if (qd < CQ) {
                    float v[V];
                    GnVec<T>::load(gn_ptr(s, base + p, qd * V), v);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const double d = (double)(v[e] + add[j][e]);
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
                for (int e = 0; e < V; ++e) {
                    double* d = sh + ((size_t)tp * C + qd * V + e) * 2;
                    d[0] = sum[j][e];
                    d[1] = ssq[j][e];
                }
            }
        }
    }
    __syncthreads();
    // fold the PY pixel rows of every channel in row order (row 0's slot receives the channel total)
    for (int c = tid; c < C; c += GN_NT) {
        double a = sh[c * 2], b = sh[c * 2 + 1];
        for (int r = 1; r < PY; ++r) {
            a += sh[((size_t)r * C + c) * 2];
            b += sh[((size_t)r * C + c) * 2 + 1];
        }
        sh[c * 2] = a;
        sh[c * 2 + 1] = b;
    }
    __syncthreads();
    const int nchunks = gridDim.x;
    if (chrows) {
        float* row = chrows + ((size_t)img * nchunks + blockIdx.x) * 2 * C;
        for (int c = tid; c < C; c += GN_NT) {
            row[c] = (float)sh[c * 2];
            row[C + c] = (float)sh[c * 2 + 1];
        }
        return;
    }
    // fold the channels of every group in channel order; one partial per (block, group)
    const int cpg = C / G;
    if (tid < G) {
        double a = 0.0, b = 0.0;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
            a += sh[c * 2];
            b += sh[c * 2 + 1];
        }
        double* dst = partials + (((size_t)img * nchunks + blockIdx.x) * G + tid) * 2;
        dst[0] = a;
        dst[1] = b;
    }
}

// sum of the per-block partials of (image, group g): element s (0 = sum, 1 = sum of squares), in block order.
// Called by the first 2G threads of a consumer block; the result goes to LDS.
__device__ __forceinline__ double gn_fold_partials(const double* partials, int img, int nchunks, int G, int idx) {
    const double* src = partials + (size_t)img * nchunks * G * 2 + idx;
    double acc = 0.0;
    int k = 0;
    for (; k + 8 <= nchunks; k += 8) {
        double r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = src[(size_t)(k + u) * G * 2];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += r[u];
    }
    for (; k < nchunks; ++k) acc += src[(size_t)k * G * 2];
    return acc;
}

// grid: (pixel chunks, NI); LDS: coefficient pairs A[c], B[c] so that y = x*A + B
template <typename T, bool POOL>
__global__ void __launch_bounds__(GN_NT)
    gn_apply_kernel(GnSrc<T> s, const float* addvec, int ld_add, const double* partials, int nchunks, const float* gamma,
                    const float* beta, const float* scale, const float* shift, int ld_ss, T* out, int ldo,
                    int H, int W, int G, float eps, int silu, int pix_per_block) {
    constexpr int V = GnVec<T>::N;
    extern __shared__ __attribute__((aligned(16))) float shf[];   // A[C] | B[C] | group sums double [G][2]
    const int C = s.C0 + s.C1;
    const int CQ = C / V;
    const int img = blockIdx.y;
    const int tid = threadIdx.x;
    const int HW = H * W;
    const int cpg = C / G;
    const double inv_n = 1.0 / ((double)cpg * (double)HW);
    float* cA = shf;
    float* cB = shf + C;
    double* gs = reinterpret_cast<double*>(shf + 2 * C);        // C is a multiple of 4: 8-byte aligned
    for (int i = tid; i < 2 * G; i += GN_NT) gs[i] = gn_fold_partials(partials, img, nchunks, G, i);
    __syncthreads();
    for (int c = tid; c < C; c += GN_NT) {
        const int g = c / cpg;
        const double su = gs[g * 2 + 0];
        const double sq = gs[g * 2 + 1];
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
    typedef typename GnVec<T>::Raw Raw;
    auto one_item = [&](int it, const Raw& r0, const Raw& r1, const Raw& r2, const Raw& r3) {
        float v0[V], v1[V], v2[V], v3[V];
        GnVec<T>::expand(r0, v0);
        if (POOL) { GnVec<T>::expand(r1, v1); GnVec<T>::expand(r2, v2); GnVec<T>::expand(r3, v3); }
        const int po = it / CQ;
        const int c = (it - po * CQ) * V;
        float y[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float a = cA[c + e], b = cB[c + e];
            float t = v0[e] * a + b;
            if (silu) t = fast_silu(t);
            if (POOL) {
                float t1 = v1[e] * a + b, t2 = v2[e] * a + b, t3 = v3[e] * a + b;
                if (silu) { t1 = fast_silu(t1); t2 = fast_silu(t2); t3 = fast_silu(t3); }
                t = (t + t1 + t2 + t3) * 0.25f;
            }
            y[e] = t;
        }
        GnVec<T>::store(out + (obase + po) * ldo + c, y);
    };
    auto src_of = [&](int it, int k) -> const T* {
        const int po = it / CQ;
        const int c = (it - po * CQ) * V;
        if (POOL) {
            const int oy = po / Wo, ox = po - oy * Wo;
            return gn_ptr(s, ibase + (size_t)(2 * oy + (k >> 1)) * W + 2 * ox + (k & 1), c);
        }
        return gn_ptr(s, ibase + po, c);
    };
    constexpr int UA = POOL ? 1 : 4;              // items per trip (a pooled item already has 4 loads in flight)
    int it = items0 + tid;
    for (; it + (UA - 1) * GN_NT < items1; it += UA * GN_NT) {
        Raw v[UA][POOL ? 4 : 1];
#pragma unroll
        for (int u = 0; u < UA; ++u)
#pragma unroll
            for (int k = 0; k < (POOL ? 4 : 1); ++k) v[u][k] = GnVec<T>::raw(src_of(it + u * GN_NT, k));
#pragma unroll
        for (int u = 0; u < UA; ++u)
            one_item(it + u * GN_NT, v[u][0], v[u][POOL ? 1 : 0], v[u][POOL ? 2 : 0], v[u][POOL ? 3 : 0]);
    }
    for (; it < items1; it += GN_NT) {
        Raw v[POOL ? 4 : 1];
#pragma unroll
        for (int k = 0; k < (POOL ? 4 : 1); ++k) v[k] = GnVec<T>::raw(src_of(it, k));
        one_item(it, v[0], v[POOL ? 1 : 0], v[POOL ? 2 : 0], v[POOL ? 3 : 0]);
    }
}

// The same pass with one channel vector per thread for the whole block (no pooling): threads = R pixel rows x CQ channel
// vectors, so a thread's coefficients live in registers and its addresses advance by a constant -- no division and no LDS
// read per item (gn_apply_kernel spends both on every 16 bytes it moves: 5.3-5.5 TB/s against the 6.0-6.2 TB/s an
// elementwise pass reaches on these boxes, tools/hbm_yardstick.py).  Same arithmetic per element, same bits.
// grid: (pixel chunks, NI), block: R * CQ threads (CQ = C / V <= 256).
template <typename T>
__global__ void __launch_bounds__(GN_NT)
    gn_apply_rows_kernel(GnSrc<T> s, const float* addvec, int ld_add, const double* partials, int nchunks, const float* gamma,
                         const float* beta, const float* scale, const float* shift, int ld_ss, T* out, int ldo,
                         int HW, int G, float eps, int silu, int pix_per_block, int R) {
    constexpr int V = GnVec<T>::N;
    extern __shared__ __attribute__((aligned(16))) float shf[];   // A[C] | B[C] | group sums double [G][2]
    const int C = s.C0 + s.C1;
    const int CQ = C / V;
    const int img = blockIdx.y;
    const int tid = threadIdx.x;
    const int nthr = R * CQ;
    const int cpg = C / G;
    const double inv_n = 1.0 / ((double)cpg * (double)HW);
    float* cA = shf;
    float* cB = shf + C;
    double* gs = reinterpret_cast<double*>(shf + 2 * C);
    for (int i = tid; i < 2 * G; i += nthr) gs[i] = gn_fold_partials(partials, img, nchunks, G, i);
    __syncthreads();
    for (int c = tid; c < C; c += nthr) {
        const int g = c / cpg;
        const double su = gs[g * 2 + 0];
        const double sq = gs[g * 2 + 1];
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
    const int r = tid / CQ, q = tid - r * CQ;
    const int c = q * V;
    float a[V], b[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        a[e] = cA[c + e];
        b[e] = cB[c + e];
    }
    const int p0 = blockIdx.x * pix_per_block;
    int p1 = p0 + pix_per_block;
    if (p1 > HW) p1 = HW;
    const size_t ibase = (size_t)img * HW;
    const bool first = c < s.C0;
    const T* src = first ? (s.x0 + ibase * s.ldx0 + c) : (s.x1 + ibase * s.ldx1 + (c - s.C0));
    const size_t lds_ = first ? (size_t)s.ldx0 : (size_t)s.ldx1;
    T* dst = out + ibase * ldo + c;
    typedef typename GnVec<T>::Raw Raw;
    auto one = [&](int px, const Raw& raw) {
        float v[V], y[V];
        GnVec<T>::expand(raw, v);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float t = v[e] * a[e] + b[e];
            if (silu) t = fast_silu(t);
            y[e] = t;
        }
        GnVec<T>::store(dst + (size_t)px * ldo, y);
    };
    int px = p0 + r;
    for (; px + 3 * R < p1; px += 4 * R) {
        Raw v0 = GnVec<T>::raw(src + (size_t)px * lds_), v1 = GnVec<T>::raw(src + (size_t)(px + R) * lds_);
        Raw v2 = GnVec<T>::raw(src + (size_t)(px + 2 * R) * lds_), v3 = GnVec<T>::raw(src + (size_t)(px + 3 * R) * lds_);
        one(px, v0);
        one(px + R, v1);
        one(px + 2 * R, v2);
        one(px + 3 * R, v3);
    }
    for (; px < p1; px += R) one(px, GnVec<T>::raw(src + (size_t)px * lds_));
}

// The apply pass on READY coefficients (nd_groupnorm_coeffs_from_partials in front of it: fold + coefficients in one small
// launch): y = act(x * A[img][c] + B[img][c]).  No block prologue at all -- gn_apply_rows_kernel's 2 048 blocks each fold the
// statistics and form their image's coefficients in float64 before the first byte moves.  Same thread layout, same bits.
template <typename T>
__global__ void __launch_bounds__(GN_NT)
    gn_apply_coeffs_kernel(GnSrc<T> s, const float* coefA, const float* coefB, int ld_coef, T* out, int ldo, int HW, int silu,
                           int pix_per_block, int R) {
    constexpr int V = GnVec<T>::N;
    const int C = s.C0 + s.C1;
    const int CQ = C / V;
    const int img = blockIdx.y;
    const int tid = threadIdx.x;
    const int r = tid / CQ, q = tid - r * CQ;
    const int c = q * V;
    float a[V], b[V];
    {
        const float* pa = coefA + (size_t)img * ld_coef + c;
        const float* pb = coefB + (size_t)img * ld_coef + c;
#pragma unroll
        for (int e = 0; e < V; e += 4) {
            const f32x4 va = *reinterpret_cast<const f32x4*>(pa + e), vb = *reinterpret_cast<const f32x4*>(pb + e);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[e + k] = va[k];
                b[e + k] = vb[k];
            }
        }
    }
    const int p0 = blockIdx.x * pix_per_block;
    int p1 = p0 + pix_per_block;
    if (p1 > HW) p1 = HW;
    const size_t ibase = (size_t)img * HW;
    const bool first = c < s.C0;
    const T* src = first ? (s.x0 + ibase * s.ldx0 + c) : (s.x1 + ibase * s.ldx1 + (c - s.C0));
    const size_t lds_ = first ? (size_t)s.ldx0 : (size_t)s.ldx1;
    T* dst = out + ibase * ldo + c;
    typedef typename GnVec<T>::Raw Raw;
    auto one = [&](int px, const Raw& raw) {
        float v[V], y[V];
        GnVec<T>::expand(raw, v);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float t = v[e] * a[e] + b[e];
            if (silu) t = fast_silu(t);
            y[e] = t;
        }
        GnVec<T>::store(dst + (size_t)px * ldo, y);
    };
    int px = p0 + r;
    for (; px + 3 * R < p1; px += 4 * R) {
        Raw v0 = GnVec<T>::raw(src + (size_t)px * lds_), v1 = GnVec<T>::raw(src + (size_t)(px + R) * lds_);
        Raw v2 = GnVec<T>::raw(src + (size_t)(px + 2 * R) * lds_), v3 = GnVec<T>::raw(src + (size_t)(px + 3 * R) * lds_);
        one(px, v0);
        one(px + R, v1);
        one(px + 2 * R, v2);
        one(px + 3 * R, v3);
    }
    for (; px < p1; px += R) one(px, GnVec<T>::raw(src + (size_t)px * lds_));
}

template <typename T>
static int launch_apply_coeffs(const char* fn, const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                               const float* coefA, const float* coefB, int ld_coef, void* out, int ldo, int NI, int HW,
                               int flags, hipStream_t st) {
    constexpr int V = GnVec<T>::N;
    const int C = C0 + C1, CQ = C / V;
    ND_REQUIRE(C % V == 0 && C0 % V == 0 && CQ <= GN_NT, fn, "channels must be whole vectors on both sides of the seam, at most 256 vectors");
    ND_REQUIRE((ldo & (V - 1)) == 0 && ldo >= C && aligned16(out) && (ld_coef & 3) == 0 && aligned16(coefA) && aligned16(coefB), fn, "alignment");
    int chunks = (2048 + NI - 1) / NI;
    int ppb = (HW + chunks - 1) / chunks;
    const int min_ppb = (GN_NT * 4 + CQ - 1) / CQ;
    if (ppb < min_ppb) ppb = min_ppb;
    chunks = (HW + ppb - 1) / ppb;
    const int R = GN_NT / CQ;
    GnSrc<T> s{static_cast<const T*>(x0), static_cast<const T*>(C1 > 0 ? x1 : x0), C0, C1, ldx0, C1 > 0 ? ldx1 : ldx0};
    hipLaunchKernelGGL((gn_apply_coeffs_kernel<T>), dim3(chunks, NI), dim3(R * CQ), 0, st, s, coefA, coefB, ld_coef,
                       static_cast<T*>(out), ldo, HW, (flags & ND_GN_SILU) ? 1 : 0, ppb, R);
    return check_launch(fn);
}

// ------------------------------------------------------------------------------------------------------------
// Small tensors (launch-bound plans: the EMNIST preset at batch 4 has 54 norms on tensors of 50-400 KB): statistics AND
// apply in ONE launch.  One block per (group, image) reads its slab (HW pixels x C/G channels, a few KB, L2-resident)
// twice: float64 sums in a fixed order (thread-strided items, then a fixed LDS tree: bitwise repeatable, no atomics),
// the group's coefficients y = x * A[c] + B[c] in LDS with gn_apply_kernel's arithmetic, then the second pass writes the
// normalised (+AdaGN, +SiLU, +2x2 average pool) tensor.  Replaces channel-partials + fold + apply (3 launches of ~6 us).
template <typename T, bool POOL, bool VEC>
__global__ void __launch_bounds__(256)
    gn_fused_small_kernel(GnSrc<T> s, const float* gamma, const float* beta, const float* scale, const float* shift, int ld_ss,
                          T* out, int ldo, int H, int W, int G, float eps, int silu) {
    // VEC: the group's channels are whole 16-byte vectors in both sources (C / G, C0, the strides multiples of 4 fp32 / 8
    // bf16 channels, pointers aligned): an item is one vector of one pixel; otherwise one channel of one pixel
    constexpr int V = VEC ? GnVec<T>::N : 1;
    __shared__ double red[2][256];
    __shared__ float cA[64], cB[64];                  // C / G <= GN_MAXCH / 32
    const int g = blockIdx.x, img = blockIdx.y, t = threadIdx.x;
    const int C = s.C0 + s.C1;
    const int cpg = C / G;
    const int q = cpg / V;                            // items per pixel
    const int c_lo = g * cpg;
    const int HW = H * W;
    const int n = HW * q;
    const size_t ibase = (size_t)img * HW;
    auto load = [&](size_t pix, int c, float (&v)[V]) {
        if constexpr (VEC) GnVec<T>::load(gn_ptr(s, pix, c), v);
        else v[0] = (float)*gn_ptr(s, pix, c);
    };
    double a = 0.0, b = 0.0;
    {
        // four items in flight per thread (the pass is latency-bound: a block owns a few KB)
        int i = t;
        for (; i + 3 * 256 < n; i += 4 * 256) {
            float v[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ii = i + u * 256;
                const int pix = ii / q;
                load(ibase + pix, c_lo + (ii - pix * q) * V, v[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const double d = (double)v[u][e];
                    a += d;
                    b += d * d;
                }
        }
        for (; i < n; i += 256) {
            float v[V];
            const int pix = i / q;
            load(ibase + pix, c_lo + (i - pix * q) * V, v);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const double d = (double)v[e];
                a += d;
                b += d * d;
            }
        }
    }
    red[0][t] = a;
    red[1][t] = b;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (t < w) {
            red[0][t] += red[0][t + w];
            red[1][t] += red[1][t + w];
        }
        __syncthreads();
    }
    if (t < cpg) {
        const double inv_n = 1.0 / ((double)HW * (double)cpg);
        const double mean = red[0][0] * inv_n;
        double var = red[1][0] * inv_n - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const int c = c_lo + t;
        double ca = rstd * (double)gamma[c];
        double cb = (double)beta[c] - mean * ca;
        if (scale) {
            const double sc = 1.0 + (double)scale[(size_t)img * ld_ss + c];
            ca *= sc;
            cb = cb * sc + (double)shift[(size_t)img * ld_ss + c];
        }
        cA[t] = (float)ca;
        cB[t] = (float)cb;
    }
    __syncthreads();
    const int Wo = POOL ? (W >> 1) : W, HWo = POOL ? (H >> 1) * Wo : HW;
    const size_t obase = (size_t)img * HWo;
    const int no = HWo * q;
    for (int i = t; i < no; i += 256) {
        const int po = i / q, cc = (i - po * q) * V, c = c_lo + cc;
        float y[V];
        if (POOL) {
            const int oy = po / Wo, ox = po - oy * Wo;
            float v[4][V];
#pragma unroll
            for (int k = 0; k < 4; ++k) load(ibase + (size_t)(2 * oy + (k >> 1)) * W + 2 * ox + (k & 1), c, v[k]);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float ka = cA[cc + e], kb = cB[cc + e];
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float z = v[k][e] * ka + kb;
                    if (silu) z = fast_silu(z);
                    acc += z;
                }
                y[e] = acc * 0.25f;
            }
        } else {
            float v[V];
            load(ibase + po, c, v);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float z = v[e] * cA[cc + e] + cB[cc + e];
                if (silu) z = fast_silu(z);
                y[e] = z;
            }
        }
        if constexpr (VEC) GnVec<T>::store(out + (obase + po) * ldo + c, y);
        else out[(obase + po) * ldo + c] = (T)y[0];
    }
}

// coefficients of the fused form: y = x * A[img][c] + B[img][c]  (consumed by the conv loaders)
__global__ void __launch_bounds__(256)
    gn_coeffs_kernel(const double* partials, int nchunks, const float* gamma, const float* beta, const float* scale,
                     const float* shift, int ld_ss, float* coefA, float* coefB, int ld_coef, int C, int HW, int G, float eps) {
    __shared__ double gs[256];                        // [G][2], G <= 128
    const int img = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * G; i += blockDim.x) gs[i] = gn_fold_partials(partials, img, nchunks, G, i);
    __syncthreads();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int cpg = C / G;
    const int g = c / cpg;
    const double inv_n = 1.0 / ((double)cpg * (double)HW);
    const double mean = gs[g * 2 + 0] * inv_n;
    double var = gs[g * 2 + 1] * inv_n - mean * mean;
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

static int check_src(const char* fn, const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1, int G, int V) {
    const int m = V - 1;
    ND_REQUIRE(x0 != nullptr && C0 > 0 && (C0 & m) == 0 && (ldx0 & m) == 0 && ldx0 >= C0 && aligned16(x0), fn,
               "x0: channels/stride must be multiples of 16 bytes, pointer 16-byte aligned");
    if (C1 > 0)
        ND_REQUIRE(x1 != nullptr && (C1 & m) == 0 && (ldx1 & m) == 0 && ldx1 >= C1 && aligned16(x1), fn,
                   "x1: channels/stride must be multiples of 16 bytes, pointer 16-byte aligned");
    ND_REQUIRE(C1 >= 0 && G > 0 && G <= 128 && (C0 + C1) % G == 0, fn, "channels not divisible by groups");
    ND_REQUIRE(C0 + C1 <= GN_MAXCH && (C0 + C1) / V <= GN_NT * GN_MAXQ, fn, "too many channels");
    return ND_OK;
}

// Thread t's share of the partial rows of one group: items (row r, channel c_lo + j), j < n, in the order t, t + 128, ...
// -- four at a time into four accumulators that are added pairwise at the end (four independent loads in flight: the
// loop is latency-bound, 512 rows x 8 channels per block at 256x256; t + k * GN_FOLD_NT in general), a fixed order shared by gn_from_partials_kernel and
// gn_coeffs_from_partials_kernel so that both produce the same float64 sums.
constexpr int GN_FOLD_NT = 256;      // threads of the two fold kernels (one block per (group, image))
__device__ __forceinline__ void gn_fold_rows(const float* p, int C, int rows, int n, int c_lo, int img, int t, double& a, double& b) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    const int total = rows * n;
    const float* base = p + ((size_t)img * rows * 2) * C + c_lo;
    auto item = [&](int i, double& sa, double& sb) {
        if (i < total) {
            const int r = i / n, cc = i - r * n;
            const float* q = base + (size_t)r * 2 * C + cc;
            sa += (double)q[0];
            sb += (double)q[C];
        }
    };
    for (int i = t; i < total; i += 4 * GN_FOLD_NT) {
        item(i, a0, b0);
        item(i + GN_FOLD_NT, a1, b1);
        item(i + 2 * GN_FOLD_NT, a2, b2);
        item(i + 3 * GN_FOLD_NT, a3, b3);
    }
    a += (a0 + a1) + (a2 + a3);
    b += (b0 + b1) + (b2 + b3);
}

// partial rows [img][row][0|1][C] (sum | sum of squares per channel) of a (two-source) tensor -> per-(image, group) sums
// in float64, written (not added) in a fixed order: one block per (group, image); thread t adds the (row, channel) items
// t, t + 128, ... of the group in float64, the 128 thread sums are added by a fixed tree in LDS.
__global__ void __launch_bounds__(GN_FOLD_NT)
    gn_from_partials_kernel(const float* p0, int C0, int rows0, const float* p1, int C1, int rows1, double* stats, int G) {
    __shared__ double red[2][GN_FOLD_NT];
    const int g = blockIdx.x, img = blockIdx.y;
    const int cpg = (C0 + C1) / G;
    const int t = threadIdx.x;
    double a = 0.0, b = 0.0;
    // the group's channels that live in the first / second source
    const int c_lo = g * cpg, c_hi = c_lo + cpg;
    const int n0 = c_lo < C0 ? ((c_hi < C0 ? c_hi : C0) - c_lo) : 0;      // channels c_lo .. c_lo + n0 - 1 of source 0
    const int n1 = cpg - n0;                                              // then n1 channels of source 1
    const int c1_lo = (c_lo > C0 ? c_lo : C0) - C0;
    if (n0 > 0) gn_fold_rows(p0, C0, rows0, n0, c_lo, img, t, a, b);
    if (n1 > 0) gn_fold_rows(p1, C1, rows1, n1, c1_lo, img, t, a, b);
    red[0][t] = a;
    red[1][t] = b;
    __syncthreads();
    for (int w = GN_FOLD_NT / 2; w > 0; w >>= 1) {
        if (t < w) {
            red[0][t] += red[0][t + w];
            red[1][t] += red[1][t + w];
        }
        __syncthreads();
    }
    if (t == 0) {
        stats[((size_t)img * G + g) * 2 + 0] = red[0][0];
        stats[((size_t)img * G + g) * 2 + 1] = red[1][0];
    }
}

// The two kernels above in one launch, for norms that are applied by a convolution's loader (nd_groupnorm_coeffs right
// behind nd_groupnorm_stats_from_partials: 38 pairs of 5-6 us launches per forward of BASELINE configs[3]): the block of
// (group, image) folds the partial rows exactly as gn_from_partials_kernel does -- same items per thread, same tree, so
// the float64 sums are the same bits -- and its first C / G threads write the group's channels' coefficients with
// gn_coeffs_kernel's arithmetic.
__global__ void __launch_bounds__(GN_FOLD_NT)
    gn_coeffs_from_partials_kernel(const float* p0, int C0, int rows0, const float* p1, int C1, int rows1, const float* gamma,
                                   const float* beta, const float* scale, const float* shift, int ld_ss, float* coefA,
                                   float* coefB, int ld_coef, int HW, int G, float eps) {
    __shared__ double red[2][GN_FOLD_NT];
    const int g = blockIdx.x, img = blockIdx.y;
    const int C = C0 + C1;
    const int cpg = C / G;
    const int t = threadIdx.x;
    double a = 0.0, b = 0.0;
    const int c_lo = g * cpg, c_hi = c_lo + cpg;
    const int n0 = c_lo < C0 ? ((c_hi < C0 ? c_hi : C0) - c_lo) : 0;
    const int n1 = cpg - n0;
    const int c1_lo = (c_lo > C0 ? c_lo : C0) - C0;
    if (n0 > 0) gn_fold_rows(p0, C0, rows0, n0, c_lo, img, t, a, b);
    if (n1 > 0) gn_fold_rows(p1, C1, rows1, n1, c1_lo, img, t, a, b);
    red[0][t] = a;
    red[1][t] = b;
    __syncthreads();
    for (int w = GN_FOLD_NT / 2; w > 0; w >>= 1) {
        if (t < w) {
            red[0][t] += red[0][t + w];
            red[1][t] += red[1][t + w];
        }
        __syncthreads();
    }
    const double inv_n = 1.0 / ((double)cpg * (double)HW);
    const double mean = red[0][0] * inv_n;
    double var = red[1][0] * inv_n - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int c = c_lo + t; c < c_hi; c += GN_FOLD_NT) {
        double ca = rstd * (double)gamma[c];
        double cb = (double)beta[c] - mean * ca;
        if (scale) {
            const double sc = 1.0 + (double)scale[(size_t)img * ld_ss + c];
            ca *= sc;
            cb = cb * sc + (double)shift[(size_t)img * ld_ss + c];
        }
        coefA[(size_t)img * ld_coef + c] = (float)ca;
        coefB[(size_t)img * ld_coef + c] = (float)cb;
    }
}

// pixel chunks per image of the statistics pass: enough blocks to fill the chip (~2048), at least 4 pixels per
// pixel-row of threads
static void stats_geometry(int NI, int HW, int CQ, int* QX, int* PY, int* ppb, int* chunks) {
    // threads = PY pixel rows x QX channel vectors, each thread owning up to GN_MAXQ vectors QX apart: of the splits that
    // cover CQ, take the one that keeps the most threads busy (e.g. 96 vectors: 5 rows x 48 x 2 = 240 threads, not 2 x 96)
    int best_q = 0, best_active = -1;
    for (int j = 1; j <= GN_MAXQ; ++j) {
        const int qx = (CQ + j - 1) / j;
        if (qx > GN_NT) continue;
        const int active = (GN_NT / qx) * qx;
        if (active > best_active) { best_active = active; best_q = qx; }
    }
    *QX = best_q > 0 ? best_q : GN_NT + 1;       // > GN_NT: rejected by the caller
    *PY = best_q > 0 ? GN_NT / best_q : 1;
    // blocks per launch: a block's fixed tail (LDS fold, partial store, ticket) costs about as much as streaming 100 KB,
    // so blocks get >= 128 KiB each, between one and four per CU (measured: 64x64x192 fp32 x 64 images 45 us with 2048
    // blocks, 36 us with 1024; 16x16x576 20 us -> 15 us with 512).  ND_GN_BLOCKS overrides (tuning knob).
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("ND_GN_BLOCKS");
        forced = e ? atoi(e) : 0;
        if (forced < 0) forced = 0;
    }
    long target = (long)NI * HW * CQ * 16 / (128 << 10);
    if (target < 256) target = 256;
    if (target > 1024) target = 1024;
    if (forced > 0) target = forced;
    int ch = (int)((target + NI - 1) / NI);
    int pb = (HW + ch - 1) / ch;
    if (pb < 4 * *PY) pb = 4 * *PY;
    *ppb = pb;
    *chunks = (HW + pb - 1) / pb;
}

template <typename T>
static int launch_stats(const char* fn, const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                        const float* addvec, int ld_add, double* partials, float* chrows, int NI, int HW, int G, hipStream_t st) {
    constexpr int V = GnVec<T>::N;
    const int C = C0 + C1, CQ = C / V;
    int QX, PY, ppb, chunks;
    stats_geometry(NI, HW, CQ, &QX, &PY, &ppb, &chunks);
    ND_REQUIRE(QX <= GN_NT, fn, "too many channels");
    GnSrc<T> s{static_cast<const T*>(x0), static_cast<const T*>(C1 > 0 ? x1 : x0), C0, C1, ldx0, C1 > 0 ? ldx1 : ldx0};
    const size_t lds = (size_t)PY * C * 2 * sizeof(double);
    hipLaunchKernelGGL(gn_stats_kernel<T>, dim3(chunks, NI), dim3(GN_NT), lds, st, s, addvec, ld_add, partials, chrows, HW, G, QX,
                       PY, ppb);
    return check_launch(fn);
}

// ND_GN_APPLY_ROWS=0: the item-strided kernel everywhere (A/B switch)
static bool apply_rows_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ND_GN_APPLY_ROWS");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

template <typename T>
static int launch_apply(const char* fn, const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                        const float* addvec, int ld_add, const double* stats, int nchunks, const float* gamma,
                        const float* beta, const float* scale, const float* shift, int ld_ss, void* out, int ldo, int NI,
                        int H, int W, int G, float eps, int flags, hipStream_t st) {
    constexpr int V = GnVec<T>::N;
    const int C = C0 + C1, CQ = C / V;
    ND_REQUIRE((ldo & (V - 1)) == 0 && ldo >= C && aligned16(out), fn, "out alignment");
    const bool pool = (flags & ND_GN_POOL2) != 0;
    const int HWo = pool ? (H >> 1) * (W >> 1) : H * W;
    int chunks = (2048 + NI - 1) / NI;
    int ppb = (HWo + chunks - 1) / chunks;
    const int min_ppb = (GN_NT * 4 + CQ - 1) / CQ;   // >= 4 items per thread
    if (ppb < min_ppb) ppb = min_ppb;
    chunks = (HWo + ppb - 1) / ppb;
    GnSrc<T> s{static_cast<const T*>(x0), static_cast<const T*>(C1 > 0 ? x1 : x0), C0, C1, ldx0, C1 > 0 ? ldx1 : ldx0};
    const size_t lds = (size_t)C * 2 * sizeof(float) + (size_t)G * 2 * sizeof(double);
    const int silu = (flags & ND_GN_SILU) ? 1 : 0;
    if (pool)
        hipLaunchKernelGGL((gn_apply_kernel<T, true>), dim3(chunks, NI), dim3(GN_NT), lds, st, s, addvec, ld_add, stats,
                           nchunks, gamma, beta, scale, shift, ld_ss, static_cast<T*>(out), ldo, H, W, G, eps, silu, ppb);
    else if (CQ <= GN_NT && C0 % V == 0 && apply_rows_enabled()) {
        const int R = GN_NT / CQ;          // pixel rows of threads; R * CQ threads (>= 129)
        hipLaunchKernelGGL((gn_apply_rows_kernel<T>), dim3(chunks, NI), dim3(R * CQ), lds, st, s, addvec, ld_add, stats,
                           nchunks, gamma, beta, scale, shift, ld_ss, static_cast<T*>(out), ldo, H * W, G, eps, silu, ppb, R);
    } else
        hipLaunchKernelGGL((gn_apply_kernel<T, false>), dim3(chunks, NI), dim3(GN_NT), lds, st, s, addvec, ld_add, stats,
                           nchunks, gamma, beta, scale, shift, ld_ss, static_cast<T*>(out), ldo, H, W, G, eps, silu, ppb);
    return check_launch(fn);
}

}  // namespace nd

using namespace nd;

extern "C" int nd_groupnorm_stats_blocks(int NI, int HW, int C, int dtype) {
    if (NI <= 0 || HW <= 0 || C <= 0 || (dtype != ND_DT_F32 && dtype != ND_DT_BF16)) return ND_E_ARG;
    int QX, PY, ppb, chunks;
    stats_geometry(NI, HW, C / (dtype == ND_DT_BF16 ? 8 : 4), &QX, &PY, &ppb, &chunks);
    return chunks;
}

extern "C" int nd_groupnorm_stats_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                                       const float* addvec, int ld_add, double* partials, int NI, int HW,
                                       int G, int dtype, nd_stream_t stream) {
    const char* fn = "nd_groupnorm_stats_nhwc";
    ND_REQUIRE(dtype == ND_DT_F32 || dtype == ND_DT_BF16, fn, "dtype must be ND_DT_F32 or ND_DT_BF16");
    int rc = check_src(fn, x0, C0, ldx0, x1, C1, ldx1, G, dtype == ND_DT_BF16 ? 8 : 4);
    if (rc) return rc;
    ND_REQUIRE(partials != nullptr && NI > 0 && HW > 0, fn, "bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == ND_DT_BF16)
        return launch_stats<__bf16>(fn, x0, C0, ldx0, x1, C1, ldx1, addvec, ld_add, partials, nullptr, NI, HW, G, st);
    return launch_stats<float>(fn, x0, C0, ldx0, x1, C1, ldx1, addvec, ld_add, partials, nullptr, NI, HW, G, st);
}

// Per-channel partial sums of ONE tensor: rows [NI][nd_groupnorm_stats_blocks(NI, HW, C, dtype)][sum | sum of squares][C]
// in fp32 (each a float64 sum over the block's pixels, rounded once), every entry written by every launch.
extern "C" int nd_groupnorm_channel_partials_nhwc(const void* x, int C, int ldx, float* rows, int NI, int HW, int dtype,
                                                  nd_stream_t stream) {
    const char* fn = "nd_groupnorm_channel_partials_nhwc";
    ND_REQUIRE(dtype == ND_DT_F32 || dtype == ND_DT_BF16, fn, "dtype must be ND_DT_F32 or ND_DT_BF16");
    int rc = check_src(fn, x, C, ldx, nullptr, 0, 0, 1, dtype == ND_DT_BF16 ? 8 : 4);
    if (rc) return rc;
    ND_REQUIRE(rows != nullptr && NI > 0 && HW > 0, fn, "bad arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == ND_DT_BF16) return launch_stats<__bf16>(fn, x, C, ldx, nullptr, 0, 0, nullptr, 0, nullptr, rows, NI, HW, 1, st);
    return launch_stats<float>(fn, x, C, ldx, nullptr, 0, 0, nullptr, 0, nullptr, rows, NI, HW, 1, st);
}

extern "C" int nd_groupnorm_apply_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                                       const float* addvec, int ld_add, const double* partials, int nblocks,
                                       const float* gamma, const float* beta,
                                       const float* scale, const float* shift, int ld_ss,
                                       void* out, int ldo, int NI, int H, int W, int G, float eps, int flags, int dtype,
                                       nd_stream_t stream) {
    const char* fn = "nd_groupnorm_apply_nhwc";
    ND_REQUIRE(dtype == ND_DT_F32 || dtype == ND_DT_BF16, fn, "dtype must be ND_DT_F32 or ND_DT_BF16");
    int rc = check_src(fn, x0, C0, ldx0, x1, C1, ldx1, G, dtype == ND_DT_BF16 ? 8 : 4);
    if (rc) return rc;
    ND_REQUIRE(partials && nblocks > 0 && gamma && beta && out && NI > 0 && H > 0 && W > 0, fn, "bad arguments");
    ND_REQUIRE((scale == nullptr) == (shift == nullptr), fn, "scale and shift go together");
    if (flags & ND_GN_POOL2) ND_REQUIRE((H & 1) == 0 && (W & 1) == 0, fn, "POOL2 needs even H, W");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == ND_DT_BF16)
        return launch_apply<__bf16>(fn, x0, C0, ldx0, x1, C1, ldx1, addvec, ld_add, partials, nblocks, gamma, beta, scale,
                                    shift, ld_ss, out, ldo, NI, H, W, G, eps, flags, st);
    return launch_apply<float>(fn, x0, C0, ldx0, x1, C1, ldx1, addvec, ld_add, partials, nblocks, gamma, beta, scale, shift,
                               ld_ss, out, ldo, NI, H, W, G, eps, flags, st);
}

extern "C" int nd_groupnorm_coeffs(const double* stats, int nblocks, const float* gamma, const float* beta,
                                   const float* scale, const float* shift, int ld_ss, float* coefA, float* coefB,
                                   int ld_coef, int NI, int C, int HW, int G, float eps, nd_stream_t stream) {
    const char* fn = "nd_groupnorm_coeffs";
    ND_REQUIRE(stats && nblocks > 0 && gamma && beta && coefA && coefB && NI > 0 && C > 0 && HW > 0 && G > 0 && G <= 128 &&
               C % G == 0, fn, "bad arguments");
    ND_REQUIRE((scale == nullptr) == (shift == nullptr) && ld_coef >= C, fn, "scale/shift go together; ld_coef >= C");
    hipLaunchKernelGGL(gn_coeffs_kernel, dim3((C + 255) / 256, NI), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       stats, nblocks, gamma, beta, scale, shift, ld_ss, coefA, coefB, ld_coef, C, HW, G, eps);
    return check_launch(fn);
}

extern "C" int nd_groupnorm_stats_from_partials(const float* p0, int C0, int rows0, const float* p1, int C1, int rows1,
                                               double* stats, int NI, int G, nd_stream_t stream) {
    const char* fn = "nd_groupnorm_stats_from_partials";
    ND_REQUIRE(p0 && stats && NI > 0 && C0 > 0 && rows0 > 0 && C1 >= 0 && G > 0 && G <= 128 && (C0 + C1) % G == 0, fn,
               "bad arguments");
    if (C1 > 0) ND_REQUIRE(p1 != nullptr && rows1 > 0, fn, "second source");
    hipLaunchKernelGGL(gn_from_partials_kernel, dim3(G, NI), dim3(GN_FOLD_NT), 0, reinterpret_cast<hipStream_t>(stream), p0, C0,
                       rows0, C1 > 0 ? p1 : p0, C1, C1 > 0 ? rows1 : 0, stats, G);
    return check_launch(fn);
}

extern "C" int nd_groupnorm_coeffs_from_partials(const float* p0, int C0, int rows0, const float* p1, int C1, int rows1,
                                                const float* gamma, const float* beta, const float* scale, const float* shift,
                                                int ld_ss, float* coefA, float* coefB, int ld_coef, int NI, int HW, int G,
                                                float eps, nd_stream_t stream) {
    const char* fn = "nd_groupnorm_coeffs_from_partials";
    ND_REQUIRE(p0 && gamma && beta && coefA && coefB && NI > 0 && HW > 0 && C0 > 0 && rows0 > 0 && C1 >= 0 && G > 0 && G <= 128 &&
               (C0 + C1) % G == 0, fn, "bad arguments");
    if (C1 > 0) ND_REQUIRE(p1 != nullptr && rows1 > 0, fn, "second source");
    ND_REQUIRE((scale == nullptr) == (shift == nullptr) && ld_coef >= C0 + C1, fn, "scale/shift go together; ld_coef >= C");
    hipLaunchKernelGGL(gn_coeffs_from_partials_kernel, dim3(G, NI), dim3(GN_FOLD_NT), 0, reinterpret_cast<hipStream_t>(stream), p0, C0,
                       rows0, C1 > 0 ? p1 : p0, C1, C1 > 0 ? rows1 : 0, gamma, beta, scale, shift, ld_ss, coefA, coefB, ld_coef,
                       HW, G, eps);
    return check_launch(fn);
}

extern "C" int nd_groupnorm_fused_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                                       const float* gamma, const float* beta, const float* scale, const float* shift, int ld_ss,
                                       void* out, int ldo, int NI, int H, int W, int G, float eps, int flags, int dtype,
                                       nd_stream_t stream) {
    const char* fn = "nd_groupnorm_fused_nhwc";
    ND_REQUIRE(dtype == ND_DT_F32 || dtype == ND_DT_BF16, fn, "dtype must be ND_DT_F32 or ND_DT_BF16");
    ND_REQUIRE(x0 && gamma && beta && out && NI > 0 && NI <= 65535 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && G > 0 && G <= 128 &&
               (C0 + C1) % G == 0 && (C0 + C1) / G <= 64, fn, "bad arguments (channels per group <= 64)");
    ND_REQUIRE(ldx0 >= C0 && (C1 == 0 || (x1 && ldx1 >= C1)) && ldo >= C0 + C1, fn, "strides");
    ND_REQUIRE((scale == nullptr) == (shift == nullptr), fn, "scale and shift go together");
    ND_REQUIRE(!scale || ld_ss >= C0 + C1, fn, "ld_ss < C0 + C1 (the AdaGN rows would be read past their end)");
    ND_REQUIRE((long)H * W * ((C0 + C1) / G) < (1L << 30), fn, "tensor too large for the one-launch form");
    const bool pool = (flags & ND_GN_POOL2) != 0;
    if (pool) ND_REQUIRE((H & 1) == 0 && (W & 1) == 0, fn, "POOL2 needs even H, W");
    const int silu = (flags & ND_GN_SILU) ? 1 : 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(G, NI), blk(256);
    const int vq = dtype == ND_DT_BF16 ? 8 : 4;
    const int cpg = (C0 + C1) / G;
    const bool vec = (cpg % vq) == 0 && (C0 % vq) == 0 && (ldx0 % vq) == 0 && (C1 == 0 || (ldx1 % vq) == 0) && (ldo % vq) == 0 &&
                     aligned16(x0) && (C1 == 0 || aligned16(x1)) && aligned16(out);
#define ND_GNF_LAUNCH(TY, POOLV, VECV)                                                                                             \
    hipLaunchKernelGGL((gn_fused_small_kernel<TY, POOLV, VECV>), grid, blk, 0, st, s, gamma, beta, scale, shift, ld_ss,             \
                       static_cast<TY*>(out), ldo, H, W, G, eps, silu)
    if (dtype == ND_DT_BF16) {
        GnSrc<__bf16> s{static_cast<const __bf16*>(x0), static_cast<const __bf16*>(C1 > 0 ? x1 : x0), C0, C1, ldx0, C1 > 0 ? ldx1 : ldx0};
        if (pool) { if (vec) ND_GNF_LAUNCH(__bf16, true, true); else ND_GNF_LAUNCH(__bf16, true, false); }
        else { if (vec) ND_GNF_LAUNCH(__bf16, false, true); else ND_GNF_LAUNCH(__bf16, false, false); }
    } else {
        GnSrc<float> s{static_cast<const float*>(x0), static_cast<const float*>(C1 > 0 ? x1 : x0), C0, C1, ldx0, C1 > 0 ? ldx1 : ldx0};
        if (pool) { if (vec) ND_GNF_LAUNCH(float, true, true); else ND_GNF_LAUNCH(float, true, false); }
        else { if (vec) ND_GNF_LAUNCH(float, false, true); else ND_GNF_LAUNCH(float, false, false); }
    }
#undef ND_GNF_LAUNCH
    return check_launch(fn);
}

extern "C" int nd_groupnorm_apply_coeffs_nhwc(const void* x0, int C0, int ldx0, const void* x1, int C1, int ldx1,
                                              const float* coefA, const float* coefB, int ld_coef, void* out, int ldo, int NI,
                                              int HW, int flags, int dtype, nd_stream_t stream) {
    const char* fn = "nd_groupnorm_apply_coeffs_nhwc";
    ND_REQUIRE(dtype == ND_DT_F32 || dtype == ND_DT_BF16, fn, "dtype must be ND_DT_F32 or ND_DT_BF16");
    ND_REQUIRE(x0 && coefA && coefB && out && NI > 0 && NI <= 65535 && HW > 0 && C0 > 0 && C1 >= 0, fn, "bad arguments");
    ND_REQUIRE(ldx0 >= C0 && (C1 == 0 || (x1 && ldx1 >= C1)) && ld_coef >= C0 + C1, fn, "strides");
    ND_REQUIRE((flags & ~ND_GN_SILU) == 0, fn, "only ND_GN_SILU (no pooling in this form)");
    const int V = dtype == ND_DT_BF16 ? 8 : 4;
    ND_REQUIRE((ldx0 & (V - 1)) == 0 && (C1 == 0 || (ldx1 & (V - 1)) == 0) && aligned16(x0) && (C1 == 0 || aligned16(x1)), fn,
               "input rows must be 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == ND_DT_BF16)
        return launch_apply_coeffs<__bf16>(fn, x0, C0, ldx0, x1, C1, ldx1, coefA, coefB, ld_coef, out, ldo, NI, HW, flags, st);
    return launch_apply_coeffs<float>(fn, x0, C0, ldx0, x1, C1, ldx1, coefA, coefB, ld_coef, out, ldo, NI, HW, flags, st);
}

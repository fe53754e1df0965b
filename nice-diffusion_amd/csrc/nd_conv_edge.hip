// K5 at the two ends of the UNet, as what they are (HBM-bound, not GEMM-shaped):
//   first  Conv2d(in_channels <= 4, model_channels, 3, padding=1)      model.py:427-431 (the first downsampling layer)
//   last   GroupNorm -> SiLU -> Conv2d(model_channels, out_channels <= 7, 3, padding=1)      model.py:446-449
//
// conv3x3_first_kernel.  K = 4 channels x 9 taps: one v_mfma_f32_16x16x4_f32 per (tap, 16 output channels, 16 pixels) with
//   the NHWC4 pixel's four channels as the k index -- no im2col, no Winograd tiles (the F(2x2) kernel spends its time on
//   transforms of a 4-channel input and then needs a separate pass over the 200 MB output for the next GroupNorm's
//   statistics).  A = weights [16 ch][4 c] (registers, loaded once per pass), B = pixels [4 c][16 px] (one float per lane from
//   global: the input is 4 MB and stays in L2), D = [16 ch][16 px]: lane = pixel l & 15, channels 4 (l >> 4) + i -> one
//   16-byte store per lane and 16-channel tile.  A block owns RB image rows (RB * W / 16 tiles, round-robin over 4 waves);
//   output channels in passes of 4 tiles (36 weight + 16 accumulator + 32 statistics registers; the next tile's nine
//   pixel loads are issued before this tile's MFMAs).  STATS: per-channel sum | sum of
//   squares of what was stored, per lane over its tiles, 16-lane rows folded by four exchanges, the four waves added in
//   wave order in LDS: row [img][rb][2][N], every row written by every launch (no zeroing, no atomics).
// taps_gather_kernel.  The last convolution has N = 3 or 6 output channels: as a GEMM over K = 9 C it would waste 10 of
//   every 16 MFMA columns.  Instead the taps go into the N dimension: P[px][tap * N + n] = sum_c act(x[px][c]) w[n][c][tap]
//   is one 1x1 GEMM (M = pixels, K = C, 9 N <= 64 columns; nd_conv_nhwc with the GroupNorm + SiLU in its loader: one n
//   block, so the fold is evaluated once and the normalised tensor is never written), and
//   out[y][x][n] = bias[n] + sum_tap P[y + ky - 1][x + kx - 1][tap * N + n] (zero outside the image) is this kernel:
//   one thread per pixel, 9 neighbours x N floats.
#include "nd_common.h"

namespace nd {

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct FirstArgs {
    const float* x;
    const float* w;          // [9 taps][NT][64 lanes]
    const float* bias;
    float* out;
    float* chstats;          // [NI][H / RB][2][N] or null
    int ldo, H, W, N, NT, RB;
};

constexpr int kFirstPass = 4;      // 16-channel tiles per pass: 36 weight registers, 16 accumulators, 32 statistics registers

template <bool STATS>
__global__ void __launch_bounds__(256) conv3x3_first_kernel(const FirstArgs p) {
    extern __shared__ __attribute__((aligned(16))) float red[];          // STATS: [4 waves][2][N]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int img = blockIdx.y, rb = blockIdx.x;
    const int NT = p.NT, H = p.H, W = p.W, N = p.N;
    const int px = lane & 15, kk = lane >> 4;
    const int tpr = W >> 4, ntiles = p.RB * tpr;
    const float* const ximg = p.x + (size_t)img * H * W * 4;

    // the nine B operands of tile t (one float per lane and tap; zero outside the image)
    auto load_b = [&](int t, float (&b)[9]) {
        const int ty = t / tpr;
        const int y = rb * p.RB + ty, x0 = (t - ty * tpr) << 4;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int yy = y + dy - 1, xx = x0 + px + dx - 1;
                const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
                b[dy * 3 + dx] = ok ? ximg[((size_t)yy * W + xx) * 4 + kk] : 0.f;
            }
        }
    };

    for (int p0 = 0; p0 < NT; p0 += kFirstPass) {
        // this pass's weights stay in registers for all of the wave's tiles
        float a[9][kFirstPass];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int j = 0; j < kFirstPass; ++j) a[tap][j] = (p0 + j < NT) ? p.w[(size_t)(tap * NT + p0 + j) * 64 + lane] : 0.f;
        f32x4 bv[kFirstPass];
#pragma unroll
        for (int j = 0; j < kFirstPass; ++j) {
            bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && p0 + j < NT) bv[j] = *reinterpret_cast<const f32x4*>(p.bias + (p0 + j) * 16 + 4 * kk);
        }
        float s[kFirstPass][4], q[kFirstPass][4];
#pragma unroll
        for (int j = 0; j < kFirstPass; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) s[j][i] = q[j][i] = 0.f;
        float b[9], bn[9];
        if (wave < ntiles) load_b(wave, b);
        for (int t = wave; t < ntiles; t += 4) {
            if (t + 4 < ntiles) load_b(t + 4, bn);          // the next tile's pixels are in flight behind this tile's MFMAs
            f32x4 acc[kFirstPass];
#pragma unroll
            for (int j = 0; j < kFirstPass; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int j = 0; j < kFirstPass; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tap][j], b[tap], acc[j], 0, 0, 0);
            const int ty = t / tpr;
            const int y = rb * p.RB + ty, x0 = (t - ty * tpr) << 4;
            float* const orow = p.out + ((size_t)(img * H + y) * W + x0 + px) * p.ldo + 4 * kk;
#pragma unroll
            for (int j = 0; j < kFirstPass; ++j) {
                if (p0 + j < NT) {
                    const f32x4 v = acc[j] + bv[j];
                    *reinterpret_cast<f32x4*>(orow + (p0 + j) * 16) = v;
                    if constexpr (STATS) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            s[j][i] += v[i];
                            q[j][i] += v[i] * v[i];
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) b[k] = bn[k];
        }
        if constexpr (STATS) {
#pragma unroll
            for (int j = 0; j < kFirstPass; ++j) {
                if (p0 + j < NT) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
#pragma unroll
                        for (int d = 1; d <= 8; d <<= 1) {
                            s[j][i] += __shfl_xor(s[j][i], d, 64);
                            q[j][i] += __shfl_xor(q[j][i], d, 64);
                        }
                    }
                    if (px == 0) {
                        float* r = red + (size_t)wave * 2 * N + (p0 + j) * 16 + 4 * kk;
                        *reinterpret_cast<f32x4*>(r) = f32x4{s[j][0], s[j][1], s[j][2], s[j][3]};
                        *reinterpret_cast<f32x4*>(r + N) = f32x4{q[j][0], q[j][1], q[j][2], q[j][3]};
                    }
                }
            }
        }
    }
    if constexpr (STATS) {
        __syncthreads();
        float* const row = p.chstats + ((size_t)img * gridDim.x + rb) * 2 * N;
        for (int c = tid; c < 2 * N; c += 256) row[c] = ((red[c] + red[2 * N + c]) + red[4 * N + c]) + red[6 * N + c];
    }
}

__global__ void pack_first_weight_kernel(const float* w, float* out, int N, int C0, int NT) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)9 * NT * 64) return;
    const int lane = (int)(i & 63);
    const int nt = (int)((i >> 6) % NT), tap = (int)((i >> 6) / NT);
    const int n = nt * 16 + (lane & 15), c = lane >> 4;
    out[i] = (n < N && c < C0) ? w[((size_t)n * C0 + c) * 9 + tap] : 0.f;
}

// P [NI*H*W][ldp] (column tap * N + n) -> out [NI*H*W][ldo] channels 0 .. N-1
template <int N>
__global__ void __launch_bounds__(256) taps_gather_kernel(const float* P, int ldp, const float* bias, float* out, int ldo,
                                                          int H, int W, long npix) {
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    float acc[N];
#pragma unroll
    for (int n = 0; n < N; ++n) acc[n] = bias ? bias[n] : 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
            const float* src = P + (pix + (long)(ky - 1) * W + (kx - 1)) * ldp + (ky * 3 + kx) * N;
            if constexpr ((N & 1) == 0) {
#pragma unroll
                for (int n = 0; n < N; n += 2) {
                    const f32x2 v = *reinterpret_cast<const f32x2*>(src + n);
                    acc[n] += v[0];
                    acc[n + 1] += v[1];
                }
            } else {
#pragma unroll
                for (int n = 0; n < N; ++n) acc[n] += src[n];
            }
        }
    }
    float* o = out + pix * ldo;
#pragma unroll
    for (int n = 0; n < N; ++n) o[n] = acc[n];
}

// rows per block: 8 where the image allows (64x64 at batch 64: 512 blocks = two 168-register waves per SIMD, all resident)
static int first_rows_per_block(int H) { return (H % 8 == 0) ? 8 : ((H % 4 == 0) ? 4 : ((H % 2 == 0) ? 2 : 1)); }

}  // namespace nd

using namespace nd;

extern "C" int64_t nd_conv_first_weight_floats(int N) {
    if (N <= 0) return -1;
    return (int64_t)9 * ((N + 15) / 16) * 64;
}

extern "C" int nd_repack_conv_first_weight(const float* w_oihw, float* out, int N, int C0, nd_stream_t stream) {
    const char* fn = "nd_repack_conv_first_weight";
    ND_REQUIRE(w_oihw && out && N > 0 && C0 > 0 && C0 <= 4, fn, "bad arguments (1..4 input channels)");
    const int NT = (N + 15) / 16;
    const long n = (long)9 * NT * 64;
    hipLaunchKernelGGL(pack_first_weight_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), w_oihw, out, N, C0, NT);
    return check_launch(fn);
}

extern "C" int nd_conv3x3_first_stats_rows(int NI, int H, int W, int N) {
    if (NI <= 0 || H <= 0 || W <= 0 || N <= 0 || (W & 15) || (N & 15) || N > 256) return 0;
    return H / first_rows_per_block(H);
}

extern "C" int nd_conv3x3_first_nhwc(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo, int NI,
                                     int H, int W, int N, float* chstats, nd_stream_t stream) {
    const char* fn = "nd_conv3x3_first_nhwc";
    ND_REQUIRE(x && w && out, fn, "null pointer");
    ND_REQUIRE(ldx == 4 && aligned16(x), fn, "input must be NHWC4 (ldx == 4), 16-byte aligned");
    ND_REQUIRE(NI > 0 && H > 0 && W > 0 && (W & 15) == 0, fn, "W must be a multiple of 16");
    ND_REQUIRE(N > 0 && (N & 15) == 0 && N <= 256, fn, "N must be a multiple of 16, at most 256");
    ND_REQUIRE(ldo >= N && (ldo & 3) == 0 && aligned16(out) && (!bias || aligned16(bias)) && aligned16(w), fn,
               "ldo must be >= N and a multiple of 4; pointers 16-byte aligned");
    ND_REQUIRE(NI <= 65535, fn, "too many images");
    FirstArgs a;
    a.x = x; a.w = w; a.bias = bias; a.out = out; a.chstats = chstats; a.ldo = ldo; a.H = H; a.W = W; a.N = N;
    a.NT = N / 16; a.RB = first_rows_per_block(H);
    const size_t lds = chstats ? (size_t)32 * N : 0;
    const dim3 grid(H / a.RB, NI);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (chstats) hipLaunchKernelGGL(conv3x3_first_kernel<true>, grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL(conv3x3_first_kernel<false>, grid, dim3(256), lds, s, a);
    return check_launch(fn);
}

extern "C" int nd_conv3x3_taps_gather_nhwc(const float* P, int ldp, const float* bias, float* out, int ldo, int NI, int H,
                                           int W, int N, nd_stream_t stream) {
    const char* fn = "nd_conv3x3_taps_gather_nhwc";
    ND_REQUIRE(P && out && NI > 0 && H > 0 && W > 0, fn, "bad arguments");
    ND_REQUIRE(N >= 1 && N <= 7 && ldp >= 9 * N && ldo >= N, fn, "1..7 output channels, ldp >= 9 N, ldo >= N");
    ND_REQUIRE((ldp & 1) == 0 && (reinterpret_cast<uintptr_t>(P) & 7u) == 0, fn, "P rows must be 8-byte aligned");
    const long npix = (long)NI * H * W;
    const dim3 grid((unsigned)((npix + 255) / 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define ND_GATHER_CASE(NN) \
    case NN: hipLaunchKernelGGL(taps_gather_kernel<NN>, grid, dim3(256), 0, s, P, ldp, bias, out, ldo, H, W, npix); break;
    switch (N) {
        ND_GATHER_CASE(1)
        ND_GATHER_CASE(2)
        ND_GATHER_CASE(3)
        ND_GATHER_CASE(4)
        ND_GATHER_CASE(5)
        ND_GATHER_CASE(6)
        ND_GATHER_CASE(7)
    }
#undef ND_GATHER_CASE
    return check_launch(fn);
}

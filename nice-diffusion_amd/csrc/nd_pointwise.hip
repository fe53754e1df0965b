// HBM-bound / tiny kernels around the UNet: timestep embedding (K1), resampling (K8), layout conversion at the
// API edge, weight repack, the direct stride-2 convolution, and uint8 image conversion (N2).
#include "nd_common.h"

namespace nd {

// ---- K1: model.py:514-523 ------------------------------------------------------------------------------------
// freqs[k] = exp(k * -(ln(10000)/half)) is a constant fp32 table built once by the host (model.py:516-517)
__global__ void timestep_embed_kernel(const int64_t* t, const float* freqs, int B, int dim, float* out, int ld) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dim) return;
    const int half = dim >> 1;
    float v = 0.f;   // zero pad for odd dims
    if (i < 2 * half) {
        const int k = (i < half) ? i : i - half;
        // rounded fp32 product (the reference materialises t*f before cos/sin); no FMA contraction into the range reduction
        const float arg = __fmul_rn((float)t[b], freqs[k]);
        v = (i < half) ? cosf(arg) : sinf(arg);
    }
    out[(size_t)b * ld + i] = v;
}

__global__ void embedding_add_silu_kernel(float* emb, const float* table, const int64_t* y, int num_rows, int D,
                                          float* silu_out) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    float v = emb[(size_t)b * D + i];
    if (table) {
        int64_t r = y[b];
        r = r < 0 ? 0 : (r >= num_rows ? num_rows - 1 : r);
        v += table[(size_t)r * D + i];
        emb[(size_t)b * D + i] = v;
    }
    if (silu_out) {
        // accurate SiLU here: 38 linears hang off this vector
        silu_out[(size_t)b * D + i] = v / (1.0f + expf(-v));
    }
}

// ---- K8 ------------------------------------------------------------------------------------------------------
__global__ void upsample2x_kernel(const float* x, int ldx, float* out, int ldo, int H, int W, int CQ, long total) {
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int qd = (int)(it % CQ);
        long pix = it / CQ;                       // output pixel index over [NI, 2H, 2W]
        const int ox = (int)(pix % (2 * W));
        pix /= (2 * W);
        const int oy = (int)(pix % (2 * H));
        const long img = pix / (2 * H);
        const size_t src = ((size_t)(img * H + (oy >> 1)) * W + (ox >> 1)) * ldx + qd * 4;
        const size_t dst = ((size_t)(img * 2 * H + oy) * (2 * W) + ox) * ldo + qd * 4;
        *reinterpret_cast<f32x4*>(out + dst) = *reinterpret_cast<const f32x4*>(x + src);
    }
}

// space-to-depth by 2: out[img][y][x][(p*2+q)*C + c] = x[img][2y+p][2x+q][c].  A stride-2 3x3 convolution (Downsample,
// model.py:103-108) is a stride-1 3x3 convolution of this tensor with rearranged (and partly zero) weights, so it runs on
// the MFMA / Winograd kernels instead of a one-thread-per-output loop.
__global__ void space_to_depth2_kernel(const float* x, int ldx, float* out, int ldo, int H, int W, int CQ, long total) {
    const int Ho = H >> 1, Wo = W >> 1;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int qd = (int)(it % CQ);
        long r = it / CQ;
        const int pq = (int)(r & 3);
        r >>= 2;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const long img = r / Ho;
        const size_t src = ((size_t)(img * H + 2 * oy + (pq >> 1)) * W + 2 * ox + (pq & 1)) * ldx + qd * 4;
        const size_t dst = ((size_t)(img * Ho + oy) * Wo + ox) * ldo + (size_t)pq * CQ * 4 + qd * 4;
        *reinterpret_cast<f32x4*>(out + dst) = *reinterpret_cast<const f32x4*>(x + src);
    }
}

__global__ void avgpool2x_kernel(const float* x, int ldx, float* out, int ldo, int H, int W, int CQ, long total) {
    const int Ho = H >> 1, Wo = W >> 1;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int qd = (int)(it % CQ);
        long pix = it / CQ;
        const int ox = (int)(pix % Wo);
        pix /= Wo;
        const int oy = (int)(pix % Ho);
        const long img = pix / Ho;
        const size_t s0 = ((size_t)(img * H + 2 * oy) * W + 2 * ox) * ldx + qd * 4;
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + s0);
        const f32x4 b = *reinterpret_cast<const f32x4*>(x + s0 + ldx);
        const f32x4 c = *reinterpret_cast<const f32x4*>(x + s0 + (size_t)W * ldx);
        const f32x4 d = *reinterpret_cast<const f32x4*>(x + s0 + (size_t)W * ldx + ldx);
        const size_t dst = ((size_t)(img * Ho + oy) * Wo + ox) * ldo + qd * 4;
        *reinterpret_cast<f32x4*>(out + dst) = (a + b + c + d) * 0.25f;
    }
}

// bf16 form: 8 channels (16 bytes) per thread, averaged in fp32
typedef __bf16 pw_bf16x8 __attribute__((ext_vector_type(8)));
__global__ void avgpool2x_bf16_kernel(const __bf16* x, int ldx, __bf16* out, int ldo, int H, int W, int CQ, long total) {
    const int Ho = H >> 1, Wo = W >> 1;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int qd = (int)(it % CQ);
        long pix = it / CQ;
        const int ox = (int)(pix % Wo);
        pix /= Wo;
        const int oy = (int)(pix % Ho);
        const long img = pix / Ho;
        const size_t s0 = ((size_t)(img * H + 2 * oy) * W + 2 * ox) * ldx + qd * 8;
        const pw_bf16x8 a = *reinterpret_cast<const pw_bf16x8*>(x + s0);
        const pw_bf16x8 b = *reinterpret_cast<const pw_bf16x8*>(x + s0 + ldx);
        const pw_bf16x8 c = *reinterpret_cast<const pw_bf16x8*>(x + s0 + (size_t)W * ldx);
        const pw_bf16x8 d = *reinterpret_cast<const pw_bf16x8*>(x + s0 + (size_t)W * ldx + ldx);
        pw_bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)(((float)a[e] + (float)b[e] + (float)c[e] + (float)d[e]) * 0.25f);
        *reinterpret_cast<pw_bf16x8*>(out + ((size_t)(img * Ho + oy) * Wo + ox) * ldo + qd * 8) = o;
    }
}

// ---- layout --------------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* src, float* dst, int C, int HW, int ld, long total) {
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int c = (int)(it % ld);
        const long pix = it / ld;                 // img*HW + p
        const long img = pix / HW;
        const int p = (int)(pix - img * HW);
        dst[it] = (c < C) ? src[((size_t)img * C + c) * HW + p] : 0.f;
    }
}

__global__ void nhwc_to_nchw_kernel(const float* src, float* dst, int C, int HW, int ld, long total) {
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int p = (int)(it % HW);
        const long ic = it / HW;                  // img*C + c
        const long img = ic / C;
        const int c = (int)(ic - img * C);
        dst[it] = src[((size_t)img * HW + p) * ld + c];
    }
}

// image forms (ld = 4, C <= 4): one thread per pixel, the NHWC side moves as one 16-byte access, the NCHW side as C
// plane accesses that are coalesced across the threads of a wave
__global__ void nchw_to_nhwc4_kernel(const float* src, float* dst, int C, int HW, long npix) {
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
        const long img = pix / HW;
        const int p = (int)(pix - img * HW);
        const float* sp = src + (size_t)img * C * HW + p;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C) v[c] = sp[(size_t)c * HW];
        *reinterpret_cast<f32x4*>(dst + pix * 4) = v;
    }
}

__global__ void nhwc4_to_nchw_kernel(const float* src, float* dst, int C, int HW, long npix) {
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
        const long img = pix / HW;
        const int p = (int)(pix - img * HW);
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + pix * 4);
        float* dp = dst + (size_t)img * C * HW + p;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C) dp[(size_t)c * HW] = v[c];
    }
}

// ---- direct convolution (one thread per output element; only for the non-preset stride-2 Downsample conv) -----
__global__ void conv_direct_kernel(const float* x, int C, int ldx, const float* w, const float* bias,
                                   float* out, int ldo, int H, int W, int Ho, int Wo, int N, int ks, int stride,
                                   int pad, long total) {
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int n = (int)(it % N);
        long pix = it / N;
        const int ox = (int)(pix % Wo);
        pix /= Wo;
        const int oy = (int)(pix % Ho);
        const long img = pix / Ho;
        float acc = bias ? bias[n] : 0.f;
        for (int ky = 0; ky < ks; ++ky) {
            const int iy = oy * stride - pad + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < ks; ++kx) {
                const int ix = ox * stride - pad + kx;
                if (ix < 0 || ix >= W) continue;
                const float* xp = x + ((size_t)(img * H + iy) * W + ix) * ldx;
                const float* wp = w + (size_t)n * C * ks * ks + ky * ks + kx;      // OIHW
                for (int c = 0; c < C; ++c) acc = fmaf(xp[c], wp[(size_t)c * ks * ks], acc);
            }
        }
        out[((size_t)(img * Ho + oy) * Wo + ox) * ldo + n] = acc;
    }
}

// ---- N2: scripts/sample.py:94-100,164-171 --------------------------------------------------------------------
__global__ void to_uint8_kernel(const float* x, int ldx, uint8_t* out, int C, int invert, long total) {
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long)gridDim.x * blockDim.x) {
        const int c = (int)(it % C);
        const long pix = it / C;
        float v = (x[(size_t)pix * ldx + c] + 1.0f) * 127.5f;
        v = fminf(fmaxf(v, 0.0f), 255.0f);
        if (invert) v = 255.0f - v;
        out[it] = (uint8_t)v;   // truncation, as tensor.to(torch.uint8)
    }
}

__device__ __forceinline__ uint32_t to_u8(float x, int invert) {
    float v = (x + 1.0f) * 127.5f;
    v = fminf(fmaxf(v, 0.0f), 255.0f);
    if (invert) v = 255.0f - v;
    return (uint32_t)(uint8_t)v;            // truncation, as tensor.to(torch.uint8)
}

// image form (ldx = 4, C = 3 or 1, pixel count a multiple of 4): one thread per 4 pixels = 64 bytes in, 12 / 4 out
template <int C>
__global__ void to_uint8_pix4_kernel(const float* x, uint32_t* out, int invert, long nquads) {
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nquads; q += (long)gridDim.x * blockDim.x) {
        uint32_t b[4 * C];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (q * 4 + i) * 4);
#pragma unroll
            for (int c = 0; c < C; ++c) b[i * C + c] = to_u8(v[c], invert);
        }
#pragma unroll
        for (int w = 0; w < C; ++w)
            out[q * C + w] = b[4 * w] | (b[4 * w + 1] << 8) | (b[4 * w + 2] << 16) | (b[4 * w + 3] << 24);
    }
}

static inline int grid_for(long total, int block) {
    long g = (total + block - 1) / block;
    if (g > 16384) g = 16384;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace nd

using namespace nd;

#define ND_STREAM(s) reinterpret_cast<hipStream_t>(s)

extern "C" int nd_timestep_embed(const int64_t* t, const float* freqs, int B, int dim, float* out, int ld_out,
                                 nd_stream_t stream) {
    const char* fn = "nd_timestep_embed";
    ND_REQUIRE(t && freqs && out && B > 0 && dim > 1 && ld_out >= dim, fn, "bad arguments");
    hipLaunchKernelGGL(timestep_embed_kernel, dim3((dim + 127) / 128, B), dim3(128), 0, ND_STREAM(stream), t, freqs, B,
                       dim, out, ld_out);
    return check_launch(fn);
}

extern "C" int nd_embedding_add_silu(float* emb, const float* table, const int64_t* y, int num_rows, int B, int D,
                                     float* silu_out, nd_stream_t stream) {
    const char* fn = "nd_embedding_add_silu";
    ND_REQUIRE(emb && B > 0 && D > 0, fn, "bad arguments");
    ND_REQUIRE((table == nullptr) == (y == nullptr), fn, "table and y go together");
    hipLaunchKernelGGL(embedding_add_silu_kernel, dim3((D + 127) / 128, B), dim3(128), 0, ND_STREAM(stream), emb,
                       table, y, num_rows, D, silu_out);
    return check_launch(fn);
}

extern "C" int nd_upsample2x_nhwc(const float* x, int ldx, float* out, int ldo, int NI, int H, int W, int C,
                                  nd_stream_t stream) {
    const char* fn = "nd_upsample2x_nhwc";
    ND_REQUIRE(x && out && NI > 0 && H > 0 && W > 0 && C > 0, fn, "bad arguments");
    ND_REQUIRE((C & 3) == 0 && (ldx & 3) == 0 && (ldo & 3) == 0 && aligned16(x) && aligned16(out), fn, "alignment");
    const long total = (long)NI * 4 * H * W * (C >> 2);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ND_STREAM(stream), x, ldx, out,
                       ldo, H, W, C >> 2, total);
    return check_launch(fn);
}

extern "C" int nd_space_to_depth2_nhwc(const float* x, int ldx, float* out, int ldo, int NI, int H, int W, int C,
                                      nd_stream_t stream) {
    const char* fn = "nd_space_to_depth2_nhwc";
    ND_REQUIRE(x && out && NI > 0 && H > 1 && W > 1 && C > 0, fn, "bad arguments");
    ND_REQUIRE((H & 1) == 0 && (W & 1) == 0, fn, "H and W must be even");
    ND_REQUIRE((C & 3) == 0 && (ldx & 3) == 0 && (ldo & 3) == 0 && ldx >= C && ldo >= 4 * C && aligned16(x) && aligned16(out),
               fn, "alignment / strides");
    const long total = (long)NI * (H >> 1) * (W >> 1) * 4 * (C >> 2);
    hipLaunchKernelGGL(space_to_depth2_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ND_STREAM(stream), x, ldx, out, ldo,
                       H, W, C >> 2, total);
    return check_launch(fn);
}

extern "C" int nd_avgpool2x_nhwc(const void* xv, int ldx, void* outv, int ldo, int NI, int H, int W, int C, int dtype,
                                 nd_stream_t stream) {
    const char* fn = "nd_avgpool2x_nhwc";
    ND_REQUIRE(xv && outv && NI > 0 && H > 1 && W > 1 && C > 0, fn, "bad arguments");
    ND_REQUIRE((H & 1) == 0 && (W & 1) == 0, fn, "H and W must be even");
    ND_REQUIRE(dtype == ND_DT_F32 || dtype == ND_DT_BF16, fn, "dtype must be ND_DT_F32 or ND_DT_BF16");
    if (dtype == ND_DT_BF16) {
        ND_REQUIRE((C & 7) == 0 && (ldx & 7) == 0 && (ldo & 7) == 0 && aligned16(xv) && aligned16(outv), fn, "alignment");
        const long total = (long)NI * (H >> 1) * (W >> 1) * (C >> 3);
        hipLaunchKernelGGL(avgpool2x_bf16_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ND_STREAM(stream),
                           static_cast<const __bf16*>(xv), ldx, static_cast<__bf16*>(outv), ldo, H, W, C >> 3, total);
        return check_launch(fn);
    }
    const float* x = static_cast<const float*>(xv);
    float* out = static_cast<float*>(outv);
    ND_REQUIRE((C & 3) == 0 && (ldx & 3) == 0 && (ldo & 3) == 0 && aligned16(x) && aligned16(out), fn, "alignment");
    const long total = (long)NI * (H >> 1) * (W >> 1) * (C >> 2);
    hipLaunchKernelGGL(avgpool2x_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ND_STREAM(stream), x, ldx, out,
                       ldo, H, W, C >> 2, total);
    return check_launch(fn);
}

extern "C" int nd_nchw_to_nhwc(const float* src, float* dst, int NI, int C, int HW, int ld, nd_stream_t stream) {
    const char* fn = "nd_nchw_to_nhwc";
    ND_REQUIRE(src && dst && NI > 0 && C > 0 && HW > 0 && ld >= C, fn, "bad arguments");
    if (ld == 4 && aligned16(dst)) {
        const long npix = (long)NI * HW;
        hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, ND_STREAM(stream), src, dst, C, HW,
                           npix);
        return check_launch(fn);
    }
    const long total = (long)NI * HW * ld;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ND_STREAM(stream), src, dst, C,
                       HW, ld, total);
    return check_launch(fn);
}

extern "C" int nd_nhwc_to_nchw(const float* src, float* dst, int NI, int C, int HW, int ld, nd_stream_t stream) {
    const char* fn = "nd_nhwc_to_nchw";
    ND_REQUIRE(src && dst && NI > 0 && C > 0 && HW > 0 && ld >= C, fn, "bad arguments");
    if (ld == 4 && aligned16(src)) {
        const long npix = (long)NI * HW;
        hipLaunchKernelGGL(nhwc4_to_nchw_kernel, dim3(grid_for(npix, 256)), dim3(256), 0, ND_STREAM(stream), src, dst, C, HW,
                           npix);
        return check_launch(fn);
    }
    const long total = (long)NI * C * HW;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ND_STREAM(stream), src, dst, C,
                       HW, ld, total);
    return check_launch(fn);
}

extern "C" int nd_conv_direct_nhwc(const float* x, int C, int ldx, const float* w_oihw, const float* bias,
                                   float* out, int ldo, int NI, int H, int W, int N, int ksize, int stride, int pad,
                                   nd_stream_t stream) {
    const char* fn = "nd_conv_direct_nhwc";
    ND_REQUIRE(x && w_oihw && out && NI > 0 && H > 0 && W > 0 && N > 0 && C > 0 && ldx >= C, fn, "bad arguments");
    ND_REQUIRE(ksize >= 1 && stride >= 1 && pad >= 0, fn, "bad conv geometry");
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    ND_REQUIRE(Ho > 0 && Wo > 0 && ldo >= N, fn, "bad output shape");
    const long total = (long)NI * Ho * Wo * N;
    hipLaunchKernelGGL(conv_direct_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ND_STREAM(stream), x, C, ldx,
                       w_oihw, bias, out, ldo, H, W, Ho, Wo, N, ksize, stride, pad, total);
    return check_launch(fn);
}

extern "C" int nd_to_uint8_hwc(const float* x, int ldx, uint8_t* out, int NI, int HW, int C, int invert,
                               nd_stream_t stream) {
    const char* fn = "nd_to_uint8_hwc";
    ND_REQUIRE(x && out && NI > 0 && HW > 0 && C > 0 && ldx >= C, fn, "bad arguments");
    const long npix = (long)NI * HW;
    if (ldx == 4 && (C == 3 || C == 1) && (npix & 3) == 0 && aligned16(x) && (reinterpret_cast<uintptr_t>(out) & 3u) == 0) {
        uint32_t* o32 = reinterpret_cast<uint32_t*>(out);
        if (C == 3)
            hipLaunchKernelGGL(to_uint8_pix4_kernel<3>, dim3(grid_for(npix / 4, 256)), dim3(256), 0, ND_STREAM(stream), x, o32,
                               invert, npix / 4);
        else
            hipLaunchKernelGGL(to_uint8_pix4_kernel<1>, dim3(grid_for(npix / 4, 256)), dim3(256), 0, ND_STREAM(stream), x, o32,
                               invert, npix / 4);
        return check_launch(fn);
    }
    const long total = (long)NI * HW * C;
    hipLaunchKernelGGL(to_uint8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ND_STREAM(stream), x, ldx, out, C,
                       invert, total);
    return check_launch(fn);
}

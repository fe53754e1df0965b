// K10: per-step sampler updates of the reverse loop, fused into one HBM pass each.
//   nd_ddim_step  <- Diffusion.ddim_denoising_step     diffusion.py:324-367 (everything after the model call)
//   nd_ddpm_step  <- Diffusion.denoising_step + get_eps_and_log_var   diffusion.py:248-314
//   nd_qsample / nd_qsample_steps  <- Diffusion.diffusion_step   diffusion.py:232-240 (one step for all / one per image)
//   nd_eps_log_var <- Diffusion.get_eps_and_log_var    diffusion.py:242-264 (after the model call)
//   nd_fill_timestep / nd_step_advance / nd_copy_row_by_step: the device step word and what the captured step body reads
//   through it (the model timestep; this step's row of the chain's precomputed K1/K2 table, model.py:197,346-352)
// Under classifier-free guidance the step kernels also write the unconditional half's copy of x_{t-1} (x_dup).
// The reference gathers 4-6 per-step scalars with extract() (diffusion.py:478-496: a host->device copy of a whole
// float64 table, .float(), gather) per call.  Here the tables live on the device as one fp32 row per step and the
// step index itself is a device word, so the whole step body can be captured in a hipGraph and replayed.
// Arithmetic follows the reference's fp32 operation order so teacher-forced steps agree to rounding.
#include "nd_common.h"

// keep the reference's one-rounding-per-op fp32 arithmetic (no mul+add contraction) in this file
#pragma clang fp contract(off)

namespace nd {

// ---- Philox4x32-10 + Box-Muller: counter-based N(0,1), no state in memory --------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32_10(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr_lo, (uint32_t)(ctr_lo >> 32), (uint32_t)ctr_hi, (uint32_t)(ctr_hi >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

// N(0,1) draws for the two elements 2*pair and 2*pair + 1 of step `t` (Box-Muller on one Philox block)
__device__ __forceinline__ void philox_normal_pair(uint64_t seed, int t, uint64_t pair, float& n_even, float& n_odd) {
    uint32_t r[4];
    philox4x32_10(seed, pair, (uint64_t)(uint32_t)t, r);
    const float u1 = ((float)r[0] + 1.0f) * 2.3283064365386963e-10f;   // (0, 1]
    const float u2 = (float)r[1] * 2.3283064365386963e-10f;            // [0, 1)
    const float rad = sqrtf(-2.0f * __logf(u1));
    const float ang = 6.283185307179586f * u2;
    n_even = rad * __cosf(ang);
    n_odd = rad * __sinf(ang);
}

// one N(0,1) draw for element `idx` of step `t`
__device__ __forceinline__ float philox_normal(uint64_t seed, int t, uint64_t idx) {
    float a, b;
    philox_normal_pair(seed, t, idx >> 1, a, b);
    return (idx & 1) ? b : a;
}

struct StepArgs {
    const float* x;
    float* x_out;
    float* x_dup;      // if set, a second copy of x_out (the unconditional half of a classifier-free batch reads it)
    float* pred_x0;    // if set, the (clipped unless ND_STEP_NO_CLIP) x_0 estimate of the step, laid out like x_out
    int flags;         // ND_STEP_NO_CLIP | ND_STEP_PER_IMAGE
    const float* eps;
    const float* eps_u;
    const float* coef;
    const int32_t* step;
    const float* noise;
    int64_t noise_stride;
    uint64_t seed;
    const uint64_t* seed_dev;   // if set, the Philox key is read from this device word (one captured graph, any seed)
    uint64_t idx0;     // index of this call's first element in the unpadded [B][HW][C] order of the WHOLE (unsharded) batch
    int ldx, ld_eps, HW, C;
    float w, eta;
    int var_kind;
    long total;   // B*HW*C
};

__device__ __forceinline__ float mix_eps(const StepArgs& a, float e, float eu) {
    return a.eps_u ? (1.0f + a.w) * e - a.w * eu : e;                  // diffusion.py:284 / :347
}

// per-step scalars of the DDIM update (diffusion.py:359-360, fp32 like the reference's tensors)
struct DdimScal {
    float c_rec, c_recm1, s_abp, s_dir, sigma;
};
__device__ __forceinline__ DdimScal ddim_scalars(const StepArgs& a, int t) {
    const float* cf = a.coef + (size_t)t * ND_COEF_COLS;
    const float ab = cf[2], abp = cf[3];
    const float var = a.eta * a.eta * (1.0f - abp) * (1.0f - ab / abp) / (1.0f - ab);
    DdimScal k;
    k.c_rec = cf[0]; k.c_recm1 = cf[1];
    k.s_abp = sqrtf(abp);
    k.s_dir = sqrtf(1.0f - abp - var);
    k.sigma = (t != 0) ? sqrtf(var) : 0.0f;                             // mask = (t != 0)
    return k;
}
// one element; px = offset of the element in x / noise, it = its index in the unpadded [B][HW][C] order (Philox counter)
template <bool CLIP>
__device__ __forceinline__ float ddim_elem(const DdimScal& k, float xt, float e, float nz, float& x0) {
    x0 = k.c_rec * xt - k.c_recm1 * e;                                  // :350-351
    if (CLIP) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);                       // :352-353
    float v = x0 * k.s_abp + k.s_dir * e;                               // :360
    if (k.sigma != 0.0f) v += k.sigma * nz;                             // :366
    return v;
}
// log-variance of get_eps_and_log_var (diffusion.py:248-261) from the raw second half of the model output
__device__ __forceinline__ float step_log_var(int var_kind, const float* cf, float lv_raw) {
    const float lv_a = cf[6], lv_b = cf[7];
    if (var_kind == ND_VAR_LEARNED) return lv_raw;                      // :249
    if (var_kind == ND_VAR_LEARNED_INTERP) {
        const float frac = (lv_raw + 1.0f) / 2.0f;                      // :256
        return frac * lv_b + (1.0f - frac) * lv_a;                      // :257 (max_log = lv_b, min_log = lv_a)
    }
    return lv_a;                                                        // :259 / :261
}
template <bool CLIP>
__device__ __forceinline__ float ddpm_elem(const StepArgs& a, const float* cf, int t, float xt, float e, float lv_raw,
                                           float nz, float& x0) {
    const float c_rec = cf[0], c_recm1 = cf[1], c_x0 = cf[4], c_xt = cf[5];
    const float log_var = step_log_var(a.var_kind, cf, lv_raw);
    x0 = c_rec * xt - c_recm1 * e;                                      // :287-288
    if (CLIP) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);                       // :289-290
    float v = c_x0 * x0 + c_xt * xt;                                    // :293-294
    if (t != 0) v += expf(0.5f * log_var) * nz;                         // :313
    return v;
}

// EX = the extended form behind the public per-step methods (Diffusion.ddim_denoising_step / denoising_step,
// diffusion.py:266-369): optional pred_x0 output, clip_x=False, one step index PER IMAGE (the reference's `t` is a [B]
// tensor).  The loop's own launches (one step word, clipped, no pred_x0) take EX = false: same code as before.
template <bool EX>
__device__ __forceinline__ int step_of(const StepArgs& a, long pix) {
    if (EX && (a.flags & ND_STEP_PER_IMAGE)) return a.step[pix / a.HW];
    return *a.step;
}

// generic form: one thread per element, any strides
template <bool DDIM, bool EX>
__global__ void __launch_bounds__(256) step_elem_kernel(const StepArgs a) {
    int t = *a.step;
    const float* cf = a.coef + (size_t)t * ND_COEF_COLS;
    DdimScal k = ddim_scalars(a, t);
    const bool learned = !DDIM && a.var_kind != ND_VAR_FIXED;
    const bool clip = !(EX && (a.flags & ND_STEP_NO_CLIP));
    const uint64_t seed = a.seed_dev ? *a.seed_dev : a.seed;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < a.total; it += (long)gridDim.x * blockDim.x) {
        const int c = (int)(it % a.C);
        const long pix = it / a.C;
        if (EX && (a.flags & ND_STEP_PER_IMAGE)) {
            t = step_of<EX>(a, pix);
            cf = a.coef + (size_t)t * ND_COEF_COLS;
            k = ddim_scalars(a, t);
        }
        const bool need_nz = DDIM ? (k.sigma != 0.0f) : (t != 0);
        const size_t px = (size_t)pix * a.ldx + c;
        const size_t pe = (size_t)pix * a.ld_eps;
        const float xt = a.x[px];
        const float e = mix_eps(a, a.eps[pe + c], a.eps_u ? a.eps_u[pe + c] : 0.f);
        float nz = 0.f;
        if (need_nz) nz = a.noise ? a.noise[(size_t)t * a.noise_stride + px] : philox_normal(seed, t, a.idx0 + (uint64_t)it);
        const float lv = learned ? a.eps[pe + a.C + c] : 0.f;
        float x0;
        float o;
        if (clip) o = DDIM ? ddim_elem<true>(k, xt, e, nz, x0) : ddpm_elem<true>(a, cf, t, xt, e, lv, nz, x0);
        else o = DDIM ? ddim_elem<false>(k, xt, e, nz, x0) : ddpm_elem<false>(a, cf, t, xt, e, lv, nz, x0);
        a.x_out[px] = o;
        if (a.x_dup) a.x_dup[px] = o;
        if (EX && a.pred_x0) a.pred_x0[px] = x0;
    }
}

// image form (ldx = 4, ld_eps = 4 or 8, C <= 4: every shipped configuration): one thread per pixel, 16-byte accesses;
// same per-element arithmetic and Philox counters as the generic form
template <bool DDIM, int C, bool EX>
__global__ void __launch_bounds__(256) step_pixel_kernel(const StepArgs a) {
    int t = *a.step;
    const float* cf = a.coef + (size_t)t * ND_COEF_COLS;
    DdimScal k = ddim_scalars(a, t);
    const bool learned = !DDIM && a.var_kind != ND_VAR_FIXED;
    const bool clip = !(EX && (a.flags & ND_STEP_NO_CLIP));
    const uint64_t seed = a.seed_dev ? *a.seed_dev : a.seed;
    const long npix = a.total / C;
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += (long)gridDim.x * blockDim.x) {
        if (EX && (a.flags & ND_STEP_PER_IMAGE)) {
            t = step_of<EX>(a, pix);
            cf = a.coef + (size_t)t * ND_COEF_COLS;
            k = ddim_scalars(a, t);
        }
        const bool need_nz = DDIM ? (k.sigma != 0.0f) : (t != 0);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(a.x + pix * 4);
        const float* er = a.eps + pix * a.ld_eps;
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(er);
        f32x4 e1 = {0.f, 0.f, 0.f, 0.f}, u0 = {0.f, 0.f, 0.f, 0.f}, nz = {0.f, 0.f, 0.f, 0.f};
        if (a.ld_eps == 8 && learned) e1 = *reinterpret_cast<const f32x4*>(er + 4);
        if (a.eps_u) u0 = *reinterpret_cast<const f32x4*>(a.eps_u + pix * a.ld_eps);
        if (need_nz) {
            if (a.noise) {
                nz = *reinterpret_cast<const f32x4*>(a.noise + (size_t)t * a.noise_stride + pix * 4);
            } else {
                // the C elements of a pixel span at most two Philox pairs
                const uint64_t it0 = a.idx0 + (uint64_t)pix * C, p0 = it0 >> 1;
                float g[4];
                philox_normal_pair(seed, t, p0, g[0], g[1]);
                g[2] = g[3] = 0.f;
                if (((it0 + C - 1) >> 1) != p0) philox_normal_pair(seed, t, p0 + 1, g[2], g[3]);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int j = (int)(it0 + c - 2 * p0);       // 0 .. 3
                    nz[c] = (j == 0) ? g[0] : ((j == 1) ? g[1] : ((j == 2) ? g[2] : g[3]));
                }
            }
        }
        f32x4 o = xv, p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float e = mix_eps(a, e0[c], u0[c]);
            const float lv = (C + c < 4) ? e0[(C + c) & 3] : e1[(C + c) & 3];
            float x0;
            if (clip) o[c] = DDIM ? ddim_elem<true>(k, xv[c], e, nz[c], x0) : ddpm_elem<true>(a, cf, t, xv[c], e, lv, nz[c], x0);
            else o[c] = DDIM ? ddim_elem<false>(k, xv[c], e, nz[c], x0) : ddpm_elem<false>(a, cf, t, xv[c], e, lv, nz[c], x0);
            p[c] = x0;
        }
        *reinterpret_cast<f32x4*>(a.x_out + pix * 4) = o;
        if (a.x_dup) *reinterpret_cast<f32x4*>(a.x_dup + pix * 4) = o;
        if (EX && a.pred_x0) *reinterpret_cast<f32x4*>(a.pred_x0 + pix * 4) = p;
    }
}

// get_eps_and_log_var (diffusion.py:242-264) after the model call: NHWC model output -> eps and log-variance, both NCHW
// [B][C][HW]; one step index per image
__global__ void __launch_bounds__(256) eps_log_var_kernel(const float* out_nhwc, int ld, const float* coef, const int32_t* steps,
                                                          int var_kind, float* eps, float* log_var, int B, int HW, int C) {
    const long total = (long)B * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const int c = (int)((i / HW) % C);
        const int b = (int)(i / ((long)HW * C));
        const float* row = out_nhwc + ((size_t)b * HW + p) * ld;
        const float* cf = coef + (size_t)steps[b] * ND_COEF_COLS;
        eps[i] = row[c];
        log_var[i] = step_log_var(var_kind, cf, var_kind != ND_VAR_FIXED ? row[C + c] : 0.f);
    }
}

// diffusion_step with one step index per image (diffusion.py:232-240): out = sa[t_b] x0 + sb[t_b] noise
__global__ void __launch_bounds__(256) qsample_steps_kernel(const float* x0, const float* noise, float* out, int B, long per_image,
                                                            const float* sa, const float* sb, const int32_t* steps) {
    const long total = (long)B * per_image;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = steps[i / per_image];
        out[i] = sa[t] * x0[i] + sb[t] * noise[i];
    }
}

__global__ void qsample_kernel(const float* x0, const float* noise, float* out, long n, float a, float b) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = a * x0[i] + b * noise[i];
}

__global__ void qsample4_kernel(const f32x4* x0, const f32x4* noise, f32x4* out, long n4, float a, float b) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        out[i] = a * x0[i] + b * noise[i];
}

__global__ void fill_timestep_kernel(const int64_t* tmap, const int32_t* step, int64_t* t_out, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) t_out[i] = tmap[*step];
}

__global__ void step_advance_kernel(int32_t* step, int delta) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *step += delta;
}

// row (*step - lo) of a [rows][n4] table -> out; a step word outside [lo, lo + rows) copies the nearest row (the host
// checks the range of the chain before it starts; the device never reads outside the table)
__global__ void __launch_bounds__(256) copy_row_by_step_kernel(const f32x4* table, const int32_t* step, int lo, int rows,
                                                                long n4, f32x4* out) {
    int r = *step - lo;
    r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
    const f32x4* src = table + (size_t)r * n4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) out[i] = src[i];
}

static int launch_step(bool ddim, const char* fn, const float* x, float* x_out, float* x_dup, float* pred_x0, int flags, int ldx, const float* eps,
                       const float* eps_u, int ld_eps, float w, const float* coef, const int32_t* step, float eta,
                       int var_kind, const float* noise, int64_t noise_stride, uint64_t seed, const uint64_t* seed_dev, uint64_t first_elem, int B,
                       int HW, int C, nd_stream_t stream) {
    ND_REQUIRE(x && x_out && eps && coef && step, fn, "null pointer");
    ND_REQUIRE(B > 0 && HW > 0 && C > 0 && ldx >= C, fn, "bad shape");
    const int need = (!ddim && var_kind != ND_VAR_FIXED) ? 2 * C : C;
    ND_REQUIRE(ld_eps >= need, fn, "model output has too few channels for this variance kind");
    ND_REQUIRE(var_kind >= 0 && var_kind <= 2, fn, "bad var_kind");
    ND_REQUIRE((flags & ~(ND_STEP_NO_CLIP | ND_STEP_PER_IMAGE)) == 0, fn, "unknown flag");
    ND_REQUIRE(!pred_x0 || (pred_x0 != x && pred_x0 != x_out && pred_x0 != x_dup), fn, "pred_x0 must be its own buffer");
    const bool ex = pred_x0 != nullptr || flags != 0;
    StepArgs a;
    a.pred_x0 = pred_x0; a.flags = flags;
    a.x = x; a.x_out = x_out; a.x_dup = x_dup; a.eps = eps; a.eps_u = eps_u; a.coef = coef; a.step = step; a.noise = noise;
    a.noise_stride = noise_stride; a.seed = seed; a.seed_dev = seed_dev; a.idx0 = first_elem; a.ldx = ldx; a.ld_eps = ld_eps; a.HW = HW; a.C = C; a.w = w;
    a.eta = eta; a.var_kind = var_kind; a.total = (long)B * HW * C;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ND_REQUIRE(!x_dup || (x_dup != x && x_dup != x_out), fn, "x_dup must be a third buffer");
    const bool image_form = ldx == 4 && (ld_eps == 4 || ld_eps == 8) && C <= 4 && aligned16(x) && aligned16(x_out) &&
                            (!x_dup || aligned16(x_dup)) && (!pred_x0 || aligned16(pred_x0)) &&
                            aligned16(eps) && (!eps_u || aligned16(eps_u)) &&
                            (!noise || (aligned16(noise) && (noise_stride & 3) == 0));
    if (image_form) {
        long g = ((long)B * HW + 255) / 256;
        if (g > 16384) g = 16384;
#define ND_STEP_CASE(CC)                                                                                        \
    case CC:                                                                                                    \
        if (ddim && ex) hipLaunchKernelGGL((step_pixel_kernel<true, CC, true>), dim3((int)g), dim3(256), 0, s, a);        \
        else if (ddim) hipLaunchKernelGGL((step_pixel_kernel<true, CC, false>), dim3((int)g), dim3(256), 0, s, a);       \
        else if (ex) hipLaunchKernelGGL((step_pixel_kernel<false, CC, true>), dim3((int)g), dim3(256), 0, s, a);         \
        else hipLaunchKernelGGL((step_pixel_kernel<false, CC, false>), dim3((int)g), dim3(256), 0, s, a);                \
        break;
        switch (C) {
            ND_STEP_CASE(1)
            ND_STEP_CASE(2)
            ND_STEP_CASE(3)
            ND_STEP_CASE(4)
        }
#undef ND_STEP_CASE
    } else {
        long g = (a.total + 255) / 256;
        if (g > 8192) g = 8192;
        if (ddim && ex) hipLaunchKernelGGL((step_elem_kernel<true, true>), dim3((int)g), dim3(256), 0, s, a);
        else if (ddim) hipLaunchKernelGGL((step_elem_kernel<true, false>), dim3((int)g), dim3(256), 0, s, a);
        else if (ex) hipLaunchKernelGGL((step_elem_kernel<false, true>), dim3((int)g), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((step_elem_kernel<false, false>), dim3((int)g), dim3(256), 0, s, a);
    }
    return check_launch(fn);
}

}  // namespace nd

using namespace nd;

extern "C" int nd_ddim_step(const float* x, float* x_out, float* x_dup, float* pred_x0, int flags, int ldx, const float* eps,
                            const float* eps_uncond, int ld_eps, float guidance_w, const float* coef, const int32_t* step, float eta,
                            const float* noise, int64_t noise_step_stride, uint64_t seed, const uint64_t* seed_dev,
                            uint64_t first_elem, int B, int HW, int C, nd_stream_t stream) {
    return launch_step(true, "nd_ddim_step", x, x_out, x_dup, pred_x0, flags, ldx, eps, eps_uncond, ld_eps, guidance_w, coef, step, eta,
                       ND_VAR_FIXED, noise, noise_step_stride, seed, seed_dev, first_elem, B, HW, C, stream);
}

extern "C" int nd_ddpm_step(const float* x, float* x_out, float* x_dup, float* pred_x0, int flags, int ldx, const float* eps,
                            const float* eps_uncond, int ld_eps, float guidance_w, const float* coef, const int32_t* step, int var_kind,
                            const float* noise, int64_t noise_step_stride, uint64_t seed, const uint64_t* seed_dev,
                            uint64_t first_elem, int B, int HW, int C, nd_stream_t stream) {
    return launch_step(false, "nd_ddpm_step", x, x_out, x_dup, pred_x0, flags, ldx, eps, eps_uncond, ld_eps, guidance_w, coef, step, 0.f,
                       var_kind, noise, noise_step_stride, seed, seed_dev, first_elem, B, HW, C, stream);
}

extern "C" int nd_qsample(const float* x0, const float* noise, float* out, int64_t n, float sqrt_ab, float sqrt_1mab,
                          nd_stream_t stream) {
    const char* fn = "nd_qsample";
    ND_REQUIRE(x0 && noise && out && n > 0, fn, "bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if ((n & 3) == 0 && aligned16(x0) && aligned16(noise) && aligned16(out)) {
        long g = (n / 4 + 255) / 256;
        if (g > 16384) g = 16384;
        hipLaunchKernelGGL(qsample4_kernel, dim3((int)g), dim3(256), 0, s, reinterpret_cast<const f32x4*>(x0),
                           reinterpret_cast<const f32x4*>(noise), reinterpret_cast<f32x4*>(out), (long)(n / 4), sqrt_ab,
                           sqrt_1mab);
    } else {
        long g = (n + 255) / 256;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(qsample_kernel, dim3((int)g), dim3(256), 0, s, x0, noise, out, (long)n, sqrt_ab, sqrt_1mab);
    }
    return check_launch(fn);
}

extern "C" int nd_qsample_steps(const float* x0, const float* noise, float* out, int B, int64_t per_image,
                                const float* sqrt_ab, const float* sqrt_1mab, const int32_t* steps, nd_stream_t stream) {
    const char* fn = "nd_qsample_steps";
    ND_REQUIRE(x0 && noise && out && sqrt_ab && sqrt_1mab && steps && B > 0 && per_image > 0, fn, "bad arguments");
    long g = ((long)B * per_image + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(qsample_steps_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x0, noise, out, B,
                       (long)per_image, sqrt_ab, sqrt_1mab, steps);
    return check_launch(fn);
}

extern "C" int nd_eps_log_var(const float* model_out, int ld_out, const float* coef, const int32_t* steps, int var_kind,
                              float* eps, float* log_var, int B, int HW, int C, nd_stream_t stream) {
    const char* fn = "nd_eps_log_var";
    ND_REQUIRE(model_out && coef && steps && eps && log_var && B > 0 && HW > 0 && C > 0, fn, "bad arguments");
    ND_REQUIRE(var_kind >= 0 && var_kind <= 2, fn, "bad var_kind");
    ND_REQUIRE(ld_out >= (var_kind != ND_VAR_FIXED ? 2 * C : C), fn, "model output has too few channels for this variance kind");
    long g = ((long)B * HW * C + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(eps_log_var_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), model_out, ld_out, coef,
                       steps, var_kind, eps, log_var, B, HW, C);
    return check_launch(fn);
}

extern "C" int nd_fill_timestep(const int64_t* timestep_map, const int32_t* step, int64_t* t_out, int B,
                                nd_stream_t stream) {
    const char* fn = "nd_fill_timestep";
    ND_REQUIRE(timestep_map && step && t_out && B > 0, fn, "bad arguments");
    hipLaunchKernelGGL(fill_timestep_kernel, dim3((B + 255) / 256), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), timestep_map, step, t_out, B);
    return check_launch(fn);
}

extern "C" int nd_copy_row_by_step(const float* table, const int32_t* step, int lo, int rows, int64_t row_floats,
                                   float* out, nd_stream_t stream) {
    const char* fn = "nd_copy_row_by_step";
    ND_REQUIRE(table && step && out && rows > 0 && row_floats > 0, fn, "bad arguments");
    ND_REQUIRE((row_floats & 3) == 0 && aligned16(table) && aligned16(out), fn, "rows must be multiples of 16 bytes, 16-byte aligned");
    long g = (row_floats / 4 + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(copy_row_by_step_kernel, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const f32x4*>(table), step, lo, rows, (long)(row_floats / 4),
                       reinterpret_cast<f32x4*>(out));
    return check_launch(fn);
}

extern "C" int nd_step_advance(int32_t* step, int delta, nd_stream_t stream) {
    const char* fn = "nd_step_advance";
    ND_REQUIRE(step != nullptr, fn, "null pointer");
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), step, delta);
    return check_launch(fn);
}

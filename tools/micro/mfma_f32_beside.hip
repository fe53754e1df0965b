// Diagnostic (not part of the product): what does one more instruction of each kind cost BESIDE back-to-back fp32 MFMAs
// (v_mfma_f32_16x16x4_f32, 12 waves per CU = 3 per SIMD, the shape of conv_wf4_kernel)?  Cycles per MFMA = 32 when free.
// hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_beside.hip -o gpurun_variants/mfma_f32_beside
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE: 0 none, 1: K v_fma_f32, 2: K v_pk_fma_f32, 3: K ds_read_b64 (waited one group later), 4: K global_load_dwordx3 (L2 hits,
// waited six groups later), 5: K v_pk_add_f32, 6: K v_add_f32, 7: K s_nop 0
template <int MODE, int K>
__global__ void __launch_bounds__(768) kb(float* out, const float* src, int iters) {
    extern __shared__ float lds[];
    f32x4 acc[18];
    for (int i = 0; i < 18; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f32x2{a + i, b - i};
    f32x2 d[4];
    f32x3 w[6];
    for (int i = 0; i < 4; ++i) d[i] = f32x2{0.f, 0.f};
    for (int i = 0; i < 6; ++i) w[i] = f32x3{0.f, 0.f, 0.f};
    const int laddr = (threadIdx.x & 63) * 8;
    const int voff = (threadIdx.x & 63) * 12;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 6; ++g) {
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[g * 3 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[g * 3 + i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if constexpr (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[k & 7][0]) : "v"(a), "v"(b));
                if constexpr (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[k & 7]) : "v"(x[(k + 1) & 7]), "v"(x[(k + 2) & 7]));
                if constexpr (MODE == 5) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(x[k & 7]) : "v"(x[(k + 1) & 7]));
                if constexpr (MODE == 6) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[k & 7][0]) : "v"(a));
                if constexpr (MODE == 7) asm volatile("s_nop 0");
                if constexpr (MODE == 3) {
                    if (k == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d[k & 3]) : "v"(laddr), "i"(512 * (k & 3)));
                }
                if constexpr (MODE == 4) {
                    asm volatile("s_waitcnt vmcnt(5)" : "+v"(w[g]));
                    asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(w[g]) : "v"(voff), "s"(src + g * 192));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
    float s = 0.f;
    for (int i = 0; i < 18; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    for (int i = 0; i < 4; ++i) s += d[i][0];
    for (int i = 0; i < 6; ++i) s += w[i][0];
    if (s == 123.456f) out[0] = s;
}

template <int MODE, int K>
void run(const char* name, float* out, const float* src) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto launch = [&] { hipLaunchKernelGGL((kb<MODE, K>), dim3(256), dim3(768), 16384, 0, out, src, iters); };
    for (int w = 0; w < 3; ++w) launch();
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(e0);
        launch();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    // per SIMD: 3 waves x iters x 18 MFMAs
    const double cyc = best * 1e-3 * 2.4e9 / (3.0 * iters * 18);
    printf("%-46s %7.3f ms  %6.2f cycles per MFMA at 2.4 GHz (%.2f of the rate)  -> %.2f cycles per extra instruction\n", name, best, cyc, 32.0 / cyc,
           K ? (cyc - 32.6) * 3.0 / K : 0.0);
}

int main() {
    float* out; float* src;
    (void)hipMalloc(&out, 1024); (void)hipMalloc(&src, 1 << 20);
    (void)hipMemset(src, 0, 1 << 20);
    run<0, 0>("MFMA only", out, src);
    run<1, 2>("+ 2 v_fma_f32 per 3 MFMAs", out, src);
    run<1, 5>("+ 5 v_fma_f32 per 3 MFMAs", out, src);
    run<6, 5>("+ 5 v_add_f32 per 3 MFMAs", out, src);
    run<2, 2>("+ 2 v_pk_fma_f32 per 3 MFMAs", out, src);
    run<2, 5>("+ 5 v_pk_fma_f32 per 3 MFMAs", out, src);
    run<5, 5>("+ 5 v_pk_add_f32 per 3 MFMAs", out, src);
    run<7, 5>("+ 5 s_nop 0 per 3 MFMAs", out, src);
    run<3, 2>("+ 2 ds_read_b64 per 3 MFMAs", out, src);
    run<3, 4>("+ 4 ds_read_b64 per 3 MFMAs", out, src);
    run<4, 1>("+ 1 global_load_dwordx3 per 3 MFMAs", out, src);
    return 0;
}

// Diagnostic (not part of the product): what rate do back-to-back fp32 MFMAs reach on this box, by instruction shape and
// by waves per SIMD?  hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_f32_rate.hip -o gpurun_variants/mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(384) k16(float* out, int iters, unsigned long long* stamps) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int NACC>
__global__ void __launch_bounds__(256) k32(float* out, int iters, unsigned long long* stamps) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5];
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <class F>
void run(const char* name, F launch, int blocks, int threads, double flop_per_thread_iter, int iters, unsigned long long* st) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) launch();
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> h(blocks * 2);
    hipMemcpy(h.data(), st, blocks * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (int b = 0; b < blocks; ++b) clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);   // GHz (memrealtime: 100 MHz)
    std::sort(clk.begin(), clk.end());
    const double flops = flop_per_thread_iter * iters * (double)blocks * threads / 64.0;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)  in-kernel clock median %.3f GHz\n", name, best, flops / best / 1e9, flops / best / 1e9 / 157.3, clk[clk.size() / 2]);
}

int main() {
    float* out; unsigned long long* st;
    hipMalloc(&out, 1024); hipMalloc(&st, 8192 * 16);
    const int iters = 20000;
    // flops per WAVE per iteration: NACC MFMAs x (16*16*4*2 | 32*32*2*2)
    run("16x16x4, 18 acc, 6-wave blocks x2/CU (3/SIMD)", [&] { hipLaunchKernelGGL(k16<18>, dim3(512), dim3(384), 0, 0, out, iters, st); }, 512, 384, 18 * 2048.0, iters, st);
    run("16x16x4, 18 acc, 6-wave blocks x1/CU", [&] { hipLaunchKernelGGL(k16<18>, dim3(256), dim3(384), 0, 0, out, iters, st); }, 256, 384, 18 * 2048.0, iters, st);
    run("16x16x4, 18 acc, 4-wave blocks x1/CU (1/SIMD)", [&] { hipLaunchKernelGGL(k16<18>, dim3(256), dim3(256), 0, 0, out, iters, st); }, 256, 256, 18 * 2048.0, iters, st);
    run("16x16x4, 18 acc, 4-wave blocks x2/CU (2/SIMD)", [&] { hipLaunchKernelGGL(k16<18>, dim3(512), dim3(256), 0, 0, out, iters, st); }, 512, 256, 18 * 2048.0, iters, st);
    run("16x16x4, 18 acc, 4-wave blocks x3/CU (3/SIMD)", [&] { hipLaunchKernelGGL(k16<18>, dim3(768), dim3(256), 0, 0, out, iters, st); }, 768, 256, 18 * 2048.0, iters, st);
    run("16x16x4, 2 acc, 4-wave blocks x1/CU", [&] { hipLaunchKernelGGL(k16<2>, dim3(256), dim3(256), 0, 0, out, iters * 4, st); }, 256, 256, 2 * 2048.0, iters * 4, st);
    run("32x32x2, 8 acc, 4-wave blocks x1/CU (1/SIMD)", [&] { hipLaunchKernelGGL(k32<8>, dim3(256), dim3(256), 0, 0, out, iters, st); }, 256, 256, 8 * 4096.0, iters, st);
    run("32x32x2, 8 acc, 4-wave blocks x2/CU (2/SIMD)", [&] { hipLaunchKernelGGL(k32<8>, dim3(512), dim3(256), 0, 0, out, iters, st); }, 512, 256, 8 * 4096.0, iters, st);
    return 0;
}

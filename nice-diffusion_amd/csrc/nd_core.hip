// Library bookkeeping: version, last-error string (thread-local), device architecture.
#include "nd_common.h"

#include <string.h>

namespace nd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// flags registered by translation units compiled with a timing-only / diagnostic macro (nd_common.h); function-local
// storage: registration happens during static initialisation, in no particular order
static char* build_flags_buf() {
    static char buf[1024] = "";
    return buf;
}

void register_build_flag(const char* flag) {
    char* b = build_flags_buf();
    if (strstr(b, flag)) return;
    const size_t n = strlen(b), m = strlen(flag);
    if (n + m + 2 >= 1024) return;
    if (n) b[n] = ' ';
    memcpy(b + n + (n ? 1 : 0), flag, m + 1);
}

}  // namespace nd

namespace nd {

// Order-independent 64-bit digest of a list of device buffers: sum over 32-bit words of mix(word, global word index).
// Every block walks all segments and takes every gridDim.x-th 1 KiB slice of each; one 64-bit atomic add per block.
__global__ void __launch_bounds__(256) checksum_kernel(const void* const* ptrs, const int64_t* nbytes, int nseg,
                                                       unsigned long long* out) {
    unsigned long long h = 0, base = 0;
    for (int sgm = 0; sgm < nseg; ++sgm) {
        const uint32_t* w = static_cast<const uint32_t*>(ptrs[sgm]);
        const long nw = nbytes[sgm] >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nw; i += (long)gridDim.x * 256) {
            unsigned long long z = ((unsigned long long)w[i] << 32 | (uint32_t)w[i]) ^ ((base + (unsigned long long)i) * 0x9E3779B97F4A7C15ull);
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;       // splitmix64 finaliser
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            h += z ^ (z >> 31);
        }
        base += (unsigned long long)nw;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace nd

extern "C" int nd_checksum_segments(const void* const* ptrs, const int64_t* nbytes, int nseg, uint64_t* out,
                                    nd_stream_t stream) {
    const char* fn = "nd_checksum_segments";
    ND_REQUIRE(ptrs && nbytes && out && nseg > 0, fn, "bad arguments");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(out, 0, sizeof(uint64_t), s) != hipSuccess) return nd::fail_arg(fn, "hipMemsetAsync failed");
    hipLaunchKernelGGL(nd::checksum_kernel, dim3(1024), dim3(256), 0, s, ptrs, nbytes, nseg,
                       reinterpret_cast<unsigned long long*>(out));
    return nd::check_launch(fn);
}

extern "C" int nd_version(void) { return 130; }

#ifndef ND_SRC_HASH
#define ND_SRC_HASH "unhashed"
#endif
extern "C" const char* nd_build_id(void) { return ND_SRC_HASH; }

extern "C" const char* nd_build_flags(void) { return nd::build_flags_buf(); }

extern "C" const char* nd_last_error(void) { return nd::g_err; }

extern "C" const char* nd_device_arch(void) {
    static thread_local char arch[256];
    arch[0] = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return arch;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return arch;
    strncpy(arch, prop.gcnArchName, sizeof(arch) - 1);
    arch[sizeof(arch) - 1] = 0;
    return arch;
}

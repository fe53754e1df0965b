// Shared helpers for libnd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "nd_hip.h"

namespace nd {

void set_error(const char* fmt, ...);

inline int fail_arg(const char* fn, const char* what) {
    set_error("%s: %s", fn, what);
    return ND_E_ARG;
}

inline int check_launch(const char* fn) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", fn, hipGetErrorString(e));
        return ND_E_LAUNCH;
    }
    return ND_OK;
}

// Raise a kernel's dynamic-LDS limit to the CU's 160 KiB.  The attribute belongs to the (kernel, device) pair, so the
// "already done" flag is kept per device ordinal; `flags` is one static array per kernel instantiation.
constexpr int kMaxDevices = 64;
inline int ensure_max_lds(const void* kern, bool (&flags)[kMaxDevices], const char* fn) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = kMaxDevices - 1;   // no caching
    if (flags[dev] && dev != kMaxDevices - 1) return ND_OK;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
        set_error("%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
        return ND_E_LAUNCH;
    }
    flags[dev] = true;
    return ND_OK;
}

// A library compiled with any timing-only / diagnostic macro defined (nd_variant_flags.inc lists every one the sources
// test) reports it through nd_build_flags(): such builds give wrong results by construction and must not pass for the
// product library.  The registration runs at load time, once per translation unit that saw the macro.
void register_build_flag(const char* flag);
struct BuildFlagReg {
    explicit BuildFlagReg(const char* flag) { register_build_flag(flag); }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define ND_REQUIRE(cond, fn, msg) \
    do {                          \
        if (!(cond)) return nd::fail_arg(fn, msg); \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float fast_silu(float v) {
    // v * sigmoid(v); v_exp_f32 + v_rcp_f32 (about 1 ulp each).  __frcp_rn would expand to the correctly rounded division
    // sequence (v_div_scale / v_div_fmas / v_div_fixup: 10 more instructions per element)
    return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}

}  // namespace nd

#define ND_VF_STR(x) #x
#if !defined(__HIP_DEVICE_COMPILE__)
#define ND_VARIANT_FLAG(n) namespace { const nd::BuildFlagReg nd_vf_##n(#n "=" ND_VF_STR(n)); }
#else
#define ND_VARIANT_FLAG(n)
#endif
#include "nd_variant_flags.inc"

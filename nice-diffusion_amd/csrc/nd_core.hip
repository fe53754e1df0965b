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

}  // namespace nd

extern "C" int nd_version(void) { return 100; }

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

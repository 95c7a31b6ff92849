// Error reporting and version of the C ABI.
#include "common.h"

#include <mutex>
#include <vector>

namespace dvq {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DVQ_ELAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return DVQ_OK;
}

int ensure_dynamic_lds(const void* fn, size_t bytes, const char* what) {
    struct Granted { int device; const void* fn; size_t bytes; };
    static std::mutex mu;
    static std::vector<Granted> table;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return fail(DVQ_ENODEV, "%s: no current HIP device", what);
    std::lock_guard<std::mutex> lock(mu);
    for (Granted& g : table)
        if (g.device == device && g.fn == fn) {
            if (g.bytes >= bytes) return DVQ_OK;
            if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
                return fail(DVQ_ELAUNCH, "%s: cannot raise the dynamic LDS limit to %zu bytes", what, bytes);
            g.bytes = bytes;
            return DVQ_OK;
        }
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return fail(DVQ_ELAUNCH, "%s: cannot raise the dynamic LDS limit to %zu bytes", what, bytes);
    table.push_back({device, fn, bytes});
    return DVQ_OK;
}

}  // namespace dvq

extern "C" int dvq_version(void) { return DVQ_VERSION; }
extern "C" const char* dvq_last_error(void) { return dvq::g_err; }

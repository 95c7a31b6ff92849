// Error reporting and version of the C ABI.
#include "common.h"

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

}  // namespace dvq

extern "C" int dvq_version(void) { return DVQ_VERSION; }
extern "C" const char* dvq_last_error(void) { return dvq::g_err; }

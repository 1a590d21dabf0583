// Library-level entry points: version, error string, device check.
#include <stdarg.h>
#include <string.h>
#include "common.h"

static thread_local char g_err[512] = "";

void ml_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int ml_version(void) { return ML_ABI_VERSION; }

extern "C" const char *ml_last_error(void) { return g_err; }

extern "C" int ml_device_check(void) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        ml_set_error("device_check: hipGetDevice failed: %s", hipGetErrorString(e));
        return ML_E_NOGPU;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) {
        ml_set_error("device_check: hipGetDeviceProperties failed: %s", hipGetErrorString(e));
        return ML_E_NOGPU;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        ml_set_error("device_check: device %d is %s, this library is built for gfx950 only", dev, prop.gcnArchName);
        return ML_E_NOGPU;
    }
    return ML_OK;
}

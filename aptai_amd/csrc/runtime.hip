// Error plumbing, version and device check for libaptai_hip.so.
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void aptai_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* aptai_last_error(void) { return g_err; }
extern "C" int aptai_version(void) { return 100; }

extern "C" int aptai_device_check(char* name, int name_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) APTAI_FAIL(APTAI_ERR_NO_DEVICE, "no HIP device is current");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) APTAI_FAIL(APTAI_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (name && name_len > 0) {
        strncpy(name, prop.gcnArchName, (size_t)name_len - 1);
        name[name_len - 1] = 0;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        APTAI_FAIL(APTAI_ERR_NO_DEVICE, "device is %s; this library is built for gfx950 only", prop.gcnArchName);
    return APTAI_OK;
}

// Optional per-step salt for every dropout mask: a device pointer to two uint32 words that the seeded kernels XOR into
// their (seed0, seed1).  Lets a captured hipGraph draw fresh masks on every replay (the host rewrites the two words
// before the replay) while forward and backward of one step still regenerate identical masks.
static const uint32_t* g_seed_salt = nullptr;
extern "C" int aptai_set_seed_salt(const void* device_ptr_2xu32) {
    g_seed_salt = (const uint32_t*)device_ptr_2xu32;
    return APTAI_OK;
}
const uint32_t* aptai_seed_salt(void) { return g_seed_salt; }

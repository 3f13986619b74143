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

// Optional per-step salt for every dropout mask, bound to a STREAM: a device pointer to two uint32 words that the seeded
// kernels launched on that stream XOR into their (seed0, seed1).  Lets a captured hipGraph draw fresh masks on every replay
// (the host rewrites the two words before the replay) while forward and backward of one step still regenerate identical
// masks.  The binding is looked up at launch (= capture) time and travels as a kernel argument, so two runners that capture
// on two streams, or a runner next to eager launches on another stream, never see each other's salt.
#include <mutex>
namespace {
constexpr int SALT_SLOTS = 64;
struct SaltSlot { const void* stream; const uint32_t* ptr; bool used; };
SaltSlot g_salts[SALT_SLOTS];
std::mutex g_salt_mu;
}  // namespace

extern "C" int aptai_set_seed_salt(void* stream, const void* device_ptr_2xu32) {
    std::lock_guard<std::mutex> lk(g_salt_mu);
    int free_slot = -1;
    for (int i = 0; i < SALT_SLOTS; ++i) {
        if (g_salts[i].used && g_salts[i].stream == stream) {
            if (device_ptr_2xu32) g_salts[i].ptr = (const uint32_t*)device_ptr_2xu32;
            else g_salts[i].used = false;
            return APTAI_OK;
        }
        if (!g_salts[i].used && free_slot < 0) free_slot = i;
    }
    if (!device_ptr_2xu32) return APTAI_OK;                    // clearing a stream that has no binding
    if (free_slot < 0) APTAI_FAIL(APTAI_ERR_INVALID, "aptai_set_seed_salt: more than %d streams carry a salt", SALT_SLOTS);
    g_salts[free_slot] = {stream, (const uint32_t*)device_ptr_2xu32, true};
    return APTAI_OK;
}
const uint32_t* aptai_seed_salt(const void* stream) {
    std::lock_guard<std::mutex> lk(g_salt_mu);
    for (int i = 0; i < SALT_SLOTS; ++i)
        if (g_salts[i].used && g_salts[i].stream == stream) return g_salts[i].ptr;
    return nullptr;
}


// Optional per-step FRAME BOUNDS, bound to a stream like the salt: a device pointer to two int32 words
//   [0] frames of the first conv layer that count for its GroupNorm statistics (HF:317-323 normalises over the frames of the batch AS
//       COLLATED: the reference pads every batch to its own longest utterance, train/train_aptai.py:268-285),
//   [1] frames of the regression head's output that exist for LowPassFilterLayer's 'same' zero padding (models/modules.py:46-61).
// A hipGraph captured for a BUCKET length (longer than the batch) replays with the bounds of the batch it is fed, so its results
// equal the eager run on the batch's own padded length.  Unbound streams use the static sizes they are called with.
namespace {
SaltSlot g_bounds[SALT_SLOTS];
}  // namespace

extern "C" int aptai_set_frame_bounds(void* stream, const void* device_ptr_2xi32) {
    std::lock_guard<std::mutex> lk(g_salt_mu);
    int free_slot = -1;
    for (int i = 0; i < SALT_SLOTS; ++i) {
        if (g_bounds[i].used && g_bounds[i].stream == stream) {
            if (device_ptr_2xi32) g_bounds[i].ptr = (const uint32_t*)device_ptr_2xi32;
            else g_bounds[i].used = false;
            return APTAI_OK;
        }
        if (!g_bounds[i].used && free_slot < 0) free_slot = i;
    }
    if (!device_ptr_2xi32) return APTAI_OK;
    if (free_slot < 0) APTAI_FAIL(APTAI_ERR_INVALID, "aptai_set_frame_bounds: more than %d streams carry bounds", SALT_SLOTS);
    g_bounds[free_slot] = {stream, (const uint32_t*)device_ptr_2xi32, true};
    return APTAI_OK;
}
const int32_t* aptai_frame_bounds(const void* stream) {
    std::lock_guard<std::mutex> lk(g_salt_mu);
    for (int i = 0; i < SALT_SLOTS; ++i)
        if (g_bounds[i].used && g_bounds[i].stream == stream) return (const int32_t*)g_bounds[i].ptr;
    return nullptr;
}

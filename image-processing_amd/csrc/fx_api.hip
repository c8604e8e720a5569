// libfx: library state, error text, per-device scratch, launch timing.
#include "fx_common.h"

#include <mutex>
#include <string.h>

namespace {

constexpr int kMaxDevices = 16;
constexpr int kScratchSlots = 6;

thread_local char g_err[512] = "";

struct DeviceState {
    void *scratch[kScratchSlots] = {};
    size_t scratch_bytes[kScratchSlots] = {};
};

std::mutex g_mu;
DeviceState g_dev[kMaxDevices];

bool g_profile = false;
thread_local hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
thread_local bool g_ev_valid = false;

}  // namespace

void fx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void *fx_scratch(size_t bytes, int slot)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices || slot < 0 ||
        slot >= kScratchSlots) {
        fx_set_error("fx_scratch: no current device");
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceState &st = g_dev[dev];
    if (st.scratch_bytes[slot] >= bytes && st.scratch[slot])
        return st.scratch[slot];
    if (st.scratch[slot]) {
        // The old block may still be in use by work already enqueued.
        (void)hipDeviceSynchronize();
        (void)hipFree(st.scratch[slot]);
        st.scratch[slot] = nullptr;
        st.scratch_bytes[slot] = 0;
    }
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        fx_set_error("fx_scratch: hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return nullptr;
    }
    st.scratch[slot] = p;
    st.scratch_bytes[slot] = want;
    return p;
}

FxProfileScope::FxProfileScope(hipStream_t s) : stream(s), active(g_profile)
{
    if (!active)
        return;
    if (!g_ev0) {
        if (hipEventCreate(&g_ev0) != hipSuccess || hipEventCreate(&g_ev1) != hipSuccess) {
            active = false;
            return;
        }
    }
    g_ev_valid = false;
    if (hipEventRecord(g_ev0, stream) != hipSuccess)
        active = false;
}

FxProfileScope::~FxProfileScope()
{
    if (!active)
        return;
    if (hipEventRecord(g_ev1, stream) == hipSuccess)
        g_ev_valid = true;
}

extern "C" {

int fx_init(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        fx_set_error("fx_init: no HIP device (%s)", hipGetErrorString(e));
        return FX_ENODEV;
    }
    FX_REQUIRE(device >= 0 && device < n && device < kMaxDevices, "fx_init: device %d out of range (%d present)",
               device, n);
    FX_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    FX_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fx_set_error("fx_init: device %d is %s; this library carries gfx950 code only", device,
                     prop.gcnArchName);
        return FX_ENODEV;
    }
    return FX_OK;
}

int fx_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int d = 0; d < kMaxDevices; d++) {
        for (int s = 0; s < kScratchSlots; s++) {
            if (g_dev[d].scratch[s]) {
                (void)hipSetDevice(d);
                (void)hipDeviceSynchronize();
                (void)hipFree(g_dev[d].scratch[s]);
                g_dev[d].scratch[s] = nullptr;
                g_dev[d].scratch_bytes[s] = 0;
            }
        }
    }
    (void)hipSetDevice(cur);
    return FX_OK;
}

const char *fx_last_error(void) { return g_err; }

int fx_abi_version(void) { return 1; }

int fx_profile(int enable)
{
    g_profile = enable != 0;
    return FX_OK;
}

double fx_profile_last_ms(void)
{
    if (!g_ev_valid)
        return -1.0;
    if (hipEventSynchronize(g_ev1) != hipSuccess)
        return -1.0;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_ev0, g_ev1) != hipSuccess)
        return -1.0;
    return (double)ms;
}

}  // extern "C"

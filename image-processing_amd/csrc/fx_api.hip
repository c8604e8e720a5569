// libfx: library state, error text, per-device scratch, launch timing.
#include "fx_common.h"

#include <map>
#include <memory>
#include <mutex>
#include <string.h>

namespace {

constexpr int kMaxDevices = 16;
constexpr int kScratchSlots = 7;

thread_local char g_err[512] = "";

// Scratch is owned per (device, stream): work that two streams enqueue never
// shares a buffer, and work on one stream is ordered by the stream itself.
struct StreamState {
    std::mutex call_mu;  // held by an entry point from its first scratch request until it returns
    void *scratch[kScratchSlots] = {};
    size_t scratch_bytes[kScratchSlots] = {};
    void *pin = nullptr;          // page-locked staging block for host-side arguments
    size_t pin_bytes = 0;
    hipEvent_t pin_ev = nullptr;  // recorded behind the copies out of `pin`
    bool pin_busy = false;
    hipStream_t side = nullptr;   // second stream of calls that overlap two halves of their work
    hipEvent_t fork_ev = nullptr, join_ev = nullptr;
};

void release_side(StreamState &st)
{
    if (st.side) {
        (void)hipStreamSynchronize(st.side);
        (void)hipStreamDestroy(st.side);
    }
    st.side = nullptr;
    if (st.fork_ev)
        (void)hipEventDestroy(st.fork_ev);
    if (st.join_ev)
        (void)hipEventDestroy(st.join_ev);
    st.fork_ev = st.join_ev = nullptr;
}

void release_pin(StreamState &st)
{
    if (st.pin_busy && st.pin_ev)
        (void)hipEventSynchronize(st.pin_ev);
    st.pin_busy = false;
    if (st.pin)
        (void)hipHostFree(st.pin);
    st.pin = nullptr;
    st.pin_bytes = 0;
    if (st.pin_ev)
        (void)hipEventDestroy(st.pin_ev);
    st.pin_ev = nullptr;
}

std::mutex g_mu;
std::map<std::pair<int, hipStream_t>, std::unique_ptr<StreamState>> g_streams;

bool g_profile = false;
// two event pairs: a call that runs two launches of its dominant kernel side by side brackets both
// A ring of timed launches, so that a caller can read a launch's time some calls later (reading it right away
// makes the host wait for the launch before it enqueues anything else).
constexpr int kProfRing = 32;
thread_local hipEvent_t g_ev[kProfRing][2][2] = {};
thread_local bool g_ev_valid[kProfRing][2] = {};
thread_local int64_t g_prof_units[kProfRing] = {};
thread_local int g_prof_at = 0;   // the latest scope's place in the ring

StreamState *stream_state(hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices)
        return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    std::unique_ptr<StreamState> &st = g_streams[std::make_pair(dev, stream)];
    if (!st)
        st.reset(new StreamState());
    return st.get();
}

}  // namespace

void fx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

FxCallGuard::FxCallGuard(hipStream_t s) : stream(s), state(stream_state(s))
{
    if (state)
        static_cast<StreamState *>(state)->call_mu.lock();
}

FxCallGuard::~FxCallGuard()
{
    if (state)
        static_cast<StreamState *>(state)->call_mu.unlock();
}

void *FxCallGuard::scratch(size_t bytes, int slot)
{
    if (!state || slot < 0 || slot >= kScratchSlots) {
        fx_set_error("fx scratch: no current device");
        return nullptr;
    }
    StreamState &st = *static_cast<StreamState *>(state);
    if (st.scratch_bytes[slot] >= bytes && st.scratch[slot])
        return st.scratch[slot];
    // Stream-ordered growth: the new block exists from this point of the stream
    // on, the old one is released once the work already enqueued has passed.
    // Neither call waits for the device.
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    void *p = nullptr;
    hipError_t e = hipMallocAsync(&p, want, stream);
    if (e != hipSuccess) {
        fx_set_error("fx scratch: hipMallocAsync(%zu) failed: %s", want, hipGetErrorString(e));
        return nullptr;
    }
    if (st.scratch[slot])
        (void)hipFreeAsync(st.scratch[slot], stream);
    st.scratch[slot] = p;
    st.scratch_bytes[slot] = want;
    return p;
}

void *FxCallGuard::staging(size_t bytes)
{
    if (!state) {
        fx_set_error("fx staging: no current device");
        return nullptr;
    }
    StreamState &st = *static_cast<StreamState *>(state);
    if (st.pin_busy) {
        const hipError_t e = hipEventSynchronize(st.pin_ev);
        if (e != hipSuccess) {
            fx_set_error("fx staging: hipEventSynchronize failed: %s", hipGetErrorString(e));
            return nullptr;
        }
        st.pin_busy = false;
    }
    if (st.pin_bytes < bytes || !st.pin) {
        if (st.pin)
            (void)hipHostFree(st.pin);
        st.pin = nullptr;
        st.pin_bytes = 0;
        const size_t want = bytes < (64u << 10) ? (64u << 10) : bytes;
        const hipError_t e = hipHostMalloc(&st.pin, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            st.pin = nullptr;
            fx_set_error("fx staging: hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return nullptr;
        }
        st.pin_bytes = want;
    }
    return st.pin;
}

int FxCallGuard::staged()
{
    StreamState &st = *static_cast<StreamState *>(state);
    if (!st.pin_ev)
        FX_HIP_CHECK(hipEventCreateWithFlags(&st.pin_ev, hipEventDisableTiming));
    FX_HIP_CHECK(hipEventRecord(st.pin_ev, stream));
    st.pin_busy = true;
    return FX_OK;
}

hipStream_t FxCallGuard::fork()
{
    if (!state) {
        fx_set_error("fx fork: no current device");
        return nullptr;
    }
    StreamState &st = *static_cast<StreamState *>(state);
    hipError_t e = hipSuccess;
    if (!st.side) {
        hipStream_t side = nullptr;
        // non-blocking: ordered against `stream` by the two events only, also when `stream` is the null stream
        if ((e = hipStreamCreateWithFlags(&side, hipStreamNonBlocking)) == hipSuccess &&
            (e = hipEventCreateWithFlags(&st.fork_ev, hipEventDisableTiming)) == hipSuccess &&
            (e = hipEventCreateWithFlags(&st.join_ev, hipEventDisableTiming)) == hipSuccess) {
            st.side = side;
        } else {
            if (side)
                (void)hipStreamDestroy(side);
            release_side(st);
            fx_set_error("fx fork: creating the second stream failed: %s", hipGetErrorString(e));
            return nullptr;
        }
    }
    if ((e = hipEventRecord(st.fork_ev, stream)) != hipSuccess ||
        (e = hipStreamWaitEvent(st.side, st.fork_ev, 0)) != hipSuccess) {
        fx_set_error("fx fork: %s", hipGetErrorString(e));
        return nullptr;
    }
    return st.side;
}

int FxCallGuard::join()
{
    StreamState &st = *static_cast<StreamState *>(state);
    FX_HIP_CHECK(hipEventRecord(st.join_ev, st.side));
    FX_HIP_CHECK(hipStreamWaitEvent(stream, st.join_ev, 0));
    return FX_OK;
}

FxProfileScope::FxProfileScope(hipStream_t s, int64_t units, int pair) : stream(s), active(g_profile), pair(pair)
{
    if (!active)
        return;
    if (pair == 0)
        g_prof_at = (g_prof_at + 1) % kProfRing;
    const int at = g_prof_at;
    if (!g_ev[at][pair][0]) {
        if (hipEventCreate(&g_ev[at][pair][0]) != hipSuccess || hipEventCreate(&g_ev[at][pair][1]) != hipSuccess) {
            g_ev[at][pair][0] = nullptr;
            active = false;
            return;
        }
    }
    if (pair == 0) {
        g_prof_units[at] = units;
        g_ev_valid[at][1] = false;
    }
    g_ev_valid[at][pair] = false;
    if (hipEventRecord(g_ev[at][pair][0], stream) != hipSuccess)
        active = false;
}

FxProfileScope::~FxProfileScope()
{
    if (!active)
        return;
    if (hipEventRecord(g_ev[g_prof_at][pair][1], stream) == hipSuccess)
        g_ev_valid[g_prof_at][pair] = true;
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
    return fx_hu_noalloc_setup();
}

int fx_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto &kv : g_streams) {
        StreamState &st = *kv.second;
        std::lock_guard<std::mutex> call(st.call_mu);
        bool synced = false;
        release_pin(st);
        if (st.side) {
            (void)hipSetDevice(kv.first.first);
            release_side(st);
        }
        for (int s = 0; s < kScratchSlots; s++) {
            if (!st.scratch[s])
                continue;
            if (!synced) {
                (void)hipSetDevice(kv.first.first);
                (void)hipDeviceSynchronize();
                synced = true;
            }
            (void)hipFree(st.scratch[s]);
            st.scratch[s] = nullptr;
            st.scratch_bytes[s] = 0;
        }
    }
    (void)hipSetDevice(cur);
    return FX_OK;
}

int fx_release_stream(void *stream)
{
    StreamState *st = stream_state((hipStream_t)stream);
    if (!st)
        return FX_OK;
    std::lock_guard<std::mutex> call(st->call_mu);
    release_pin(*st);
    release_side(*st);
    for (int s = 0; s < kScratchSlots; s++) {
        if (st->scratch[s])
            FX_HIP_CHECK(hipFreeAsync(st->scratch[s], (hipStream_t)stream));
        st->scratch[s] = nullptr;
        st->scratch_bytes[s] = 0;
    }
    return FX_OK;
}

const char *fx_last_error(void) { return g_err; }

int fx_abi_version(void) { return 2; }

int fx_profile(int enable)
{
    g_profile = enable != 0;
    return FX_OK;
}

long long fx_profile_last_units(void) { return (long long)g_prof_units[g_prof_at]; }

double fx_profile_ms_back(int back)
{
    if (back < 0 || back >= kProfRing)
        return -1.0;
    const int at = ((g_prof_at - back) % kProfRing + kProfRing) % kProfRing;
    if (!g_ev_valid[at][0])
        return -1.0;
    double sum = 0.0;
    int n = 0;
    for (int pair = 0; pair < 2; pair++) {
        if (!g_ev_valid[at][pair])
            continue;
        float ms = 0.f;
        if (hipEventSynchronize(g_ev[at][pair][1]) != hipSuccess ||
            hipEventElapsedTime(&ms, g_ev[at][pair][0], g_ev[at][pair][1]) != hipSuccess)
            return -1.0;
        sum += (double)ms;
        n++;
    }
    return sum / n;
}

double fx_profile_last_ms(void) { return fx_profile_ms_back(0); }

}  // extern "C"

// ------------------------------------------------------------- collective
// The one exchange step of the sharded path for callers that hold an RCCL
// communicator of their own.  RCCL is looked up at the first call (the copy
// already loaded into the process, e.g. by PyTorch, if there is one), so the
// library itself has no link-time dependency on it.
#include <dlfcn.h>

namespace {

typedef int (*fx_nccl_allgather_t)(const void *, void *, size_t, int, void *, hipStream_t);
typedef const char *(*fx_nccl_errstr_t)(int);
fx_nccl_allgather_t g_nccl_allgather = nullptr;
fx_nccl_errstr_t g_nccl_errstr = nullptr;
constexpr int kNcclFloat32 = 7;  // ncclFloat32 in rccl.h

bool load_rccl()
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_nccl_allgather)
        return true;
    void *h = nullptr;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names)
        if (!h)
            h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char *n : names)
        if (!h)
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h)
        return false;
    g_nccl_allgather = (fx_nccl_allgather_t)dlsym(h, "ncclAllGather");
    g_nccl_errstr = (fx_nccl_errstr_t)dlsym(h, "ncclGetErrorString");
    return g_nccl_allgather != nullptr;
}

}  // namespace

extern "C" int fx_allgather_features(const void *d_local, int64_t local_rows, int row_floats, void *d_global,
                                     void *rccl_comm, void *stream)
{
    FX_REQUIRE(local_rows >= 0 && row_floats >= 1, "fx_allgather_features: %lld rows of %d floats",
               (long long)local_rows, row_floats);
    FX_REQUIRE(rccl_comm != nullptr, "fx_allgather_features: rccl_comm is NULL");
    if (local_rows == 0)
        return FX_OK;
    FX_REQUIRE(d_local && d_global, "fx_allgather_features: null buffer");
    if (!load_rccl()) {
        fx_set_error("fx_allgather_features: librccl.so.1 could not be loaded (%s)", dlerror());
        return FX_ENODEV;
    }
    const int rc = g_nccl_allgather(d_local, d_global, (size_t)local_rows * (size_t)row_floats, kNcclFloat32, rccl_comm,
                                    (hipStream_t)stream);
    if (rc != 0) {
        fx_set_error("fx_allgather_features: ncclAllGather failed: %s", g_nccl_errstr ? g_nccl_errstr(rc) : "?");
        return FX_EHIP;
    }
    return FX_OK;
}

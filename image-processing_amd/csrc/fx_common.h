// Shared host/device helpers of libfx (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/fx.h"

#define FX_WAVE 64
#define FX_BLOCK 256
#define FX_WAVES_PER_BLOCK (FX_BLOCK / FX_WAVE)

// ---------------------------------------------------------------- host side
void fx_set_error(const char *fmt, ...);
// An entry point that needs scratch creates one guard for the stream it
// enqueues on and keeps it until it returns.  Scratch belongs to (device,
// stream, slot): calls on different streams never share a buffer and may
// overlap; calls on one stream reuse the buffer in stream order.  The guard
// also serialises host threads that enqueue on the same stream.  A buffer
// grows with the stream-ordered allocator (no device-wide wait) and is
// released by fx_release_stream / fx_shutdown.
struct FxCallGuard {
    hipStream_t stream;
    void *state;
    explicit FxCallGuard(hipStream_t s);
    ~FxCallGuard();
    FxCallGuard(const FxCallGuard &) = delete;
    FxCallGuard &operator=(const FxCallGuard &) = delete;
    // NULL (and the error text set) on failure
    void *scratch(size_t bytes, int slot);
    // Page-locked host block of the stream for small arguments the caller hands over in host memory
    // (offset tables, work lists): the entry point fills it, enqueues its copies to the device from it and
    // calls staged(); nothing waits for those copies.  The block is reused by the stream's next call, which
    // first waits until the previous call's copies have left it (a wait for copies, not for kernels enqueued
    // after them).  NULL (and the error text set) on failure.
    void *staging(size_t bytes);
    int staged();
    // A second stream of the call's own, for work that may run beside what the call enqueues on `stream`.
    // fork() makes it wait for everything enqueued on `stream` so far and returns it (NULL and the error text
    // set on failure); join() makes `stream` wait for everything enqueued on it since.  Scratch of the call may
    // be used on both between the two.  The stream is created on first use and kept with the stream's scratch.
    hipStream_t fork();
    int join();
};

// Defined beside the block it looks up (fx_hu.hip); fx_init calls it so that
// fx_hu_batch_u8_noalloc finds the address ready.
int fx_hu_noalloc_setup(void);

struct FxProfileScope {
    hipStream_t stream;
    bool active;
    int pair;
    // units: how many of the call's units the launch in the scope processes (0: all of them).  pair 1 brackets a
    // second launch of the same kernel that runs beside the first on another stream (opened after pair 0's
    // scope): fx_profile_last_ms then reports the mean of the two.
    explicit FxProfileScope(hipStream_t s, int64_t units = 0, int pair = 0);
    ~FxProfileScope();
};

#define FX_HIP_CHECK(call)                                                        \
    do {                                                                          \
        hipError_t fx_e_ = (call);                                                \
        if (fx_e_ != hipSuccess) {                                                \
            fx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(fx_e_), \
                         __FILE__, __LINE__);                                     \
            return FX_EHIP;                                                       \
        }                                                                         \
    } while (0)

#define FX_REQUIRE(cond, ...)        \
    do {                             \
        if (!(cond)) {               \
            fx_set_error(__VA_ARGS__); \
            return FX_EINVAL;        \
        }                            \
    } while (0)

// -------------------------------------------------------------- device side
#ifdef __HIPCC__

typedef unsigned long long u64;
typedef long long i64;

__device__ __forceinline__ int fx_lane() { return threadIdx.x & (FX_WAVE - 1); }

// 16 bytes of an input that the launch reads exactly once: a non-temporal load (global_load_dwordx4 ... nt), so
// the stream does not displace what the caches hold for re-use.  p must be 16-byte aligned.
__device__ __forceinline__ uint4 fx_load_once16(const void *p)
{
    typedef uint32_t fx_u32x4 __attribute__((ext_vector_type(4)));
    const fx_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const fx_u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 fx_load_once_f4(const float4 *p)
{
    typedef float fx_f32x4 __attribute__((ext_vector_type(4)));
    const fx_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const fx_f32x4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int fx_wave() { return threadIdx.x >> 6; }

// Butterfly all-reduce over the 64 lanes of a wave: every lane ends with the
// result.
__device__ __forceinline__ u64 fx_wave_sum_u64(u64 v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += (u64)__shfl_xor((unsigned long long)v, o, FX_WAVE);
    return v;
}

__device__ __forceinline__ double fx_wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o, FX_WAVE);
    return v;
}

__device__ __forceinline__ double fx_wave_max_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = fmax(v, __shfl_xor(v, o, FX_WAVE));
    return v;
}

// Wave-wide maximum through DPP row operations (a few cycles each) instead of
// LDS-routed shuffles: pairs, quads, half rows, rows of 16, then the last lane
// of a row broadcast into the next rows; lane 63 ends with the result, which
// every lane receives as a wave-uniform value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float fx_dpp_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double fx_dpp_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = (int)b, hi = (int)(b >> 32);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi2 << 32) | (unsigned int)lo2);
}
__device__ __forceinline__ float fx_wave_max_f32_dpp(float v)
{
    v = fmaxf(v, fx_dpp_f32<0xB1, 0xf>(v));   // quad_perm [1,0,3,2]
    v = fmaxf(v, fx_dpp_f32<0x4E, 0xf>(v));   // quad_perm [2,3,0,1]
    v = fmaxf(v, fx_dpp_f32<0x141, 0xf>(v));  // row_half_mirror
    v = fmaxf(v, fx_dpp_f32<0x140, 0xf>(v));  // row_mirror
    v = fmaxf(v, fx_dpp_f32<0x142, 0xa>(v));  // row_bcast:15 into rows 1 and 3
    v = fmaxf(v, fx_dpp_f32<0x143, 0xc>(v));  // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float fx_wave_min_f32_dpp(float v)
{
    v = fminf(v, fx_dpp_f32<0xB1, 0xf>(v));
    v = fminf(v, fx_dpp_f32<0x4E, 0xf>(v));
    v = fminf(v, fx_dpp_f32<0x141, 0xf>(v));
    v = fminf(v, fx_dpp_f32<0x140, 0xf>(v));
    v = fminf(v, fx_dpp_f32<0x142, 0xa>(v));
    v = fminf(v, fx_dpp_f32<0x143, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ double fx_wave_max_f64_dpp(double v)
{
    v = fmax(v, fx_dpp_f64<0xB1, 0xf>(v));
    v = fmax(v, fx_dpp_f64<0x4E, 0xf>(v));
    v = fmax(v, fx_dpp_f64<0x141, 0xf>(v));
    v = fmax(v, fx_dpp_f64<0x140, 0xf>(v));
    v = fmax(v, fx_dpp_f64<0x142, 0xa>(v));
    v = fmax(v, fx_dpp_f64<0x143, 0xc>(v));
    const long long b = __double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)b, 63);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Wave-wide sum of doubles through DPP row operations (the lanes a step does not write take 0): partners in
// pairs, quads, half rows and rows of 16, then the rows' totals passed on; every lane receives lane 63's total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double fx_dpp_f64_z(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = (int)b, hi = (int)(b >> 32);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi2 << 32) | (unsigned int)lo2);
}
__device__ __forceinline__ double fx_wave_sum_f64_dpp(double v)
{
    v += fx_dpp_f64_z<0xB1, 0xf>(v);    // quad_perm [1,0,3,2]
    v += fx_dpp_f64_z<0x4E, 0xf>(v);    // quad_perm [2,3,0,1]
    v += fx_dpp_f64_z<0x141, 0xf>(v);   // row_half_mirror
    v += fx_dpp_f64_z<0x140, 0xf>(v);   // row_mirror
    v += fx_dpp_f64_z<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v += fx_dpp_f64_z<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    const long long b = __double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)b, 63);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Wave-wide maximum of unsigned integers.  The lanes a DPP step does not write take 0, the identity of the
// maximum, so move and maximum fold into one DPP instruction per step.
__device__ __forceinline__ uint32_t fx_wave_max_u32_dpp(uint32_t v)
{
    const auto step = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// Maximum of floats / doubles that are >= 0 (or NaN, which wins): their bit patterns order like unsigned
// integers.  For doubles the high words first, then the low words of the lanes that hold the maximal high word.
__device__ __forceinline__ float fx_wave_max_nonneg_f32_dpp(float f)
{
    return __int_as_float((int)fx_wave_max_u32_dpp((uint32_t)__float_as_int(f)));
}
__device__ __forceinline__ double fx_wave_max_nonneg_f64_dpp(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t hi = fx_wave_max_u32_dpp((uint32_t)(b >> 32));
    const uint32_t lo = fx_wave_max_u32_dpp((uint32_t)(b >> 32) == hi ? (uint32_t)b : 0u);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Round an unsigned 128-bit integer (hi:lo) to the nearest double, ties to
// even.
__device__ __forceinline__ double fx_u128_to_double(u64 hi, u64 lo)
{
    if (hi == 0 && lo < (1ull << 53))
        return (double)lo;
    int top = hi ? (127 - __clzll((i64)hi)) : (63 - __clzll((i64)lo));
    int shift = top - 52;  // >= 1
    u64 keep, rem_hi, rem_lo;
    if (shift >= 64) {
        int s = shift - 64;
        keep = hi >> s;
        rem_hi = s ? (hi & ((1ull << s) - 1)) : 0ull;
        rem_lo = lo;
    } else {
        keep = (hi << (64 - shift)) | (lo >> shift);
        rem_hi = 0;
        rem_lo = lo & ((1ull << shift) - 1);
    }
    // half = 2^(shift-1)
    u64 half_hi = 0, half_lo = 0;
    if (shift - 1 >= 64)
        half_hi = 1ull << (shift - 1 - 64);
    else
        half_lo = 1ull << (shift - 1);
    bool gt = (rem_hi > half_hi) || (rem_hi == half_hi && rem_lo > half_lo);
    bool eq = (rem_hi == half_hi) && (rem_lo == half_lo);
    if (gt || (eq && (keep & 1)))
        keep += 1;
    return ldexp((double)keep, shift);
}

#endif  // __HIPCC__

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
// Per-device scratch owned by the library; grows, never shrinks until
// fx_shutdown.  Returns NULL (and sets the error text) on failure.
void *fx_scratch(size_t bytes, int slot);

struct FxProfileScope {
    hipStream_t stream;
    bool active;
    explicit FxProfileScope(hipStream_t s);
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

// Per-column binarisation thresholds of a feature table and their application
// (reference src/models/train_bernoulli_nb.py:307-385: fit_thresholds with
// np.median / np.quantile / the histogram Otsu fallback, apply_thresholds).
//
// The table is row-major float64 [rows, cols] in device memory.  An order
// statistic of a column is found exactly by a most-significant-digit radix
// select over order-preserving 64-bit keys: eight passes of 8 bits, each pass a
// 256-bin histogram (LDS per block, then global) of the elements that still
// match the digits chosen so far.  A column needs two ranks (the two middle
// elements, or the two neighbours of a quantile position); up to BIN_GROUP
// columns go through the passes together.  Rows can be restricted to those
// whose label equals a given one (thresholds per class).
#include "fx_common.h"
#include <string.h>

namespace {

constexpr int BIN_GROUP = 16;            // columns per pass group
constexpr int BIN_Q = 2 * BIN_GROUP;     // rank queries per group
constexpr int BIN_DIGITS = 256;
constexpr u64 BIN_NO_ROW = ~0ull;

struct BinCol {         // per column of the group
    u64 count;          // participating rows without NaN
    u64 kmin, kmax;     // smallest / largest key
    u64 first_row;      // first participating row
    u64 nan_count;
    u64 rank[2];        // zero-based ranks wanted
    u64 prefix[2];      // digits chosen so far (high bits)
    u64 left[2];        // rank among the elements that match the prefix
    double t;           // interpolation weight of the quantile
};

struct BinArgs {
    const double *x;
    int64_t rows;
    int cols;
    int64_t stride;        // elements between rows
    const int32_t *labels; // or NULL
    int label;
    int c0, gw;            // first column and width of the group
    BinCol *col;           // [BIN_GROUP]
    uint32_t *hist;        // [BIN_Q][256], or [BIN_GROUP][256] for the Otsu histogram
};

__device__ __forceinline__ u64 key_of(double v)
{
    const u64 b = (u64)__double_as_longlong(v);
    return b ^ ((b >> 63) ? ~0ull : (1ull << 63));
}
__device__ __forceinline__ double value_of(u64 k)
{
    const u64 b = k ^ ((k >> 63) ? (1ull << 63) : ~0ull);
    return __longlong_as_double((long long)b);
}

__global__ void bin_reset_kernel(BinCol *col, uint32_t *hist)
{
    const int t = threadIdx.x;
    if (t < BIN_GROUP) {
        BinCol c;
        memset(&c, 0, sizeof(c));
        c.kmin = ~0ull;
        c.first_row = BIN_NO_ROW;
        col[t] = c;
    }
    for (int i = t; i < BIN_Q * BIN_DIGITS; i += blockDim.x)
        hist[i] = 0;
}

// count, min, max, NaN count and first participating row of every column
__global__ __launch_bounds__(FX_BLOCK) void bin_prepass_kernel(BinArgs A)
{
    __shared__ u64 s_count[BIN_GROUP], s_min[BIN_GROUP], s_max[BIN_GROUP], s_first[BIN_GROUP], s_nan[BIN_GROUP];
    const int t = threadIdx.x;
    if (t < BIN_GROUP) {
        s_count[t] = 0, s_min[t] = ~0ull, s_max[t] = 0, s_first[t] = BIN_NO_ROW, s_nan[t] = 0;
    }
    __syncthreads();
    const int64_t total = A.rows * A.gw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + t; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / A.gw;
        const int c = (int)(i - r * A.gw);
        if (A.labels && A.labels[r] != A.label)
            continue;
        atomicMin(&s_first[c], (u64)r);
        const double v = A.x[r * A.stride + A.c0 + c];
        if (v != v) {
            atomicAdd(&s_nan[c], 1ull);
            continue;
        }
        const u64 k = key_of(v);
        atomicAdd(&s_count[c], 1ull);
        atomicMin(&s_min[c], k);
        atomicMax(&s_max[c], k);
    }
    __syncthreads();
    if (t < A.gw) {
        BinCol *c = A.col + t;
        atomicAdd(&c->count, s_count[t]);
        atomicMin(&c->kmin, s_min[t]);
        atomicMax(&c->kmax, s_max[t]);
        atomicMin(&c->first_row, s_first[t]);
        atomicAdd(&c->nan_count, s_nan[t]);
    }
}

// ranks of the two order statistics a column needs
//   median (np.median): elements (n-1)/2 and n/2
//   quantile (np.quantile, method "linear"): h = (n-1) q, elements floor(h) and floor(h)+1, weight h - floor(h)
__global__ void bin_ranks_kernel(BinCol *col, int gw, int method, double q)
{
    const int t = threadIdx.x;
    if (t >= gw)
        return;
    BinCol *c = col + t;
    const u64 n = c->count;
    u64 lo = 0, hi = 0;
    double w = 0.0;
    if (n > 0) {
        if (method == 0) {
            lo = (n - 1) / 2;
            hi = n / 2;
        } else {
            const double h = (double)(n - 1) * q;
            const double f = floor(h);
            lo = (u64)f;
            if (lo > n - 1)
                lo = n - 1;
            hi = lo + 1 < n ? lo + 1 : n - 1;
            w = h - f;
        }
    }
    c->rank[0] = c->left[0] = lo;
    c->rank[1] = c->left[1] = hi;
    c->prefix[0] = c->prefix[1] = 0;
    c->t = w;
}

// one digit of the radix select: histogram of the digit at `shift` over the
// elements whose higher digits equal the prefix of the query
__global__ __launch_bounds__(FX_BLOCK) void bin_digit_hist_kernel(BinArgs A, int shift)
{
    __shared__ uint32_t s_hist[BIN_Q * BIN_DIGITS];
    __shared__ u64 s_prefix[BIN_Q];
    const int t = threadIdx.x;
    for (int i = t; i < BIN_Q * BIN_DIGITS; i += FX_BLOCK)
        s_hist[i] = 0;
    if (t < BIN_Q)
        s_prefix[t] = A.col[t >> 1].prefix[t & 1];
    __syncthreads();
    const bool first = shift == 56;
    const int64_t total = A.rows * A.gw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + t; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / A.gw;
        const int c = (int)(i - r * A.gw);
        if (A.labels && A.labels[r] != A.label)
            continue;
        const double v = A.x[r * A.stride + A.c0 + c];
        if (v != v)
            continue;
        const u64 k = key_of(v);
        const uint32_t digit = (uint32_t)(k >> shift) & 255u;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int qi = 2 * c + j;
            if (first || ((k ^ s_prefix[qi]) >> (shift + 8)) == 0)
                atomicAdd(&s_hist[qi * BIN_DIGITS + digit], 1u);
        }
    }
    __syncthreads();
    for (int i = t; i < 2 * A.gw * BIN_DIGITS; i += FX_BLOCK)
        if (s_hist[i])
            atomicAdd(&A.hist[i], s_hist[i]);
}

// choose the digit that holds the wanted rank; clears the histogram for the next pass
__global__ void bin_digit_pick_kernel(BinCol *col, uint32_t *hist, int gw, int shift)
{
    const int t = threadIdx.x;
    if (t < 2 * gw) {
        BinCol *c = col + (t >> 1);
        const int j = t & 1;
        uint32_t *h = hist + t * BIN_DIGITS;
        if (c->count > 0) {
            u64 left = c->left[j], cum = 0;
            int d = 0;
            for (; d < BIN_DIGITS - 1; d++) {
                if (cum + h[d] > left)
                    break;
                cum += h[d];
            }
            c->left[j] = left - cum;
            c->prefix[j] |= (u64)d << shift;
        }
    }
    __syncthreads();
    for (int i = t; i < 2 * gw * BIN_DIGITS; i += blockDim.x)
        hist[i] = 0;
}

// thresholds of the group from the two selected elements
__global__ void bin_finish_kernel(BinArgs A, int method, int constant_rule, double *thr, int32_t *flags, int64_t *count)
{
    const int t = threadIdx.x;
    if (t >= A.gw)
        return;
    const BinCol *c = A.col + t;
    const u64 n = c->count;
    const bool has_nan = c->nan_count != 0;
    int f = has_nan ? 2 : 0;
    double r;
    if (n == 0 || has_nan) {
        r = __builtin_nan("");
    } else {
        const double a = value_of(c->prefix[0]), b = value_of(c->prefix[1]);
        if (method == 0) {
            r = (n & 1) ? a : (a + b) / 2.0;
        } else {
            // numpy's _lerp: a + (b - a) t, and b - (b - a)(1 - t) from t = 0.5 on
            const double diff = b - a, w = c->t;
            r = a + diff * w;
            if (w >= 0.5)
                r = b - diff * (1.0 - w);
        }
        // fit_thresholds: a column whose values all equal the first one keeps that value
        if (value_of(c->kmin) == value_of(c->kmax)) {
            f |= 1;
            if (constant_rule)
                r = A.x[(int64_t)c->first_row * A.stride + A.c0 + t];
        }
    }
    thr[A.c0 + t] = r;
    if (flags)
        flags[A.c0 + t] = f;
    if (count && A.c0 + t == 0)
        count[0] = (int64_t)(n + c->nan_count);
}

// ---- Otsu threshold of a column: np.histogram(values, bins=256) and the
// between-class variance of its cumulative sums (train_bernoulli_nb.py:318-328)
struct OtsuRange {
    double first, last, step, denom;
};
__device__ __forceinline__ OtsuRange otsu_range(const BinCol *c)
{
    OtsuRange R;
    R.first = value_of(c->kmin);
    R.last = value_of(c->kmax);
    if (R.first == R.last) {  // np.histogram widens an empty range by half a unit each side
        R.first = R.first - 0.5;
        R.last = R.last + 0.5;
    }
    R.denom = R.last - R.first;
    R.step = R.denom / 256.0;  // np.linspace(first, last, 257): i * step + first, last entry = last
    return R;
}
__device__ __forceinline__ double otsu_edge(const OtsuRange &R, int i)
{
    return i == 256 ? R.last : (double)i * R.step + R.first;
}

__global__ __launch_bounds__(FX_BLOCK) void bin_otsu_hist_kernel(BinArgs A)
{
    __shared__ uint32_t s_hist[BIN_GROUP * BIN_DIGITS];
    __shared__ OtsuRange s_range[BIN_GROUP];
    const int t = threadIdx.x;
    for (int i = t; i < BIN_GROUP * BIN_DIGITS; i += FX_BLOCK)
        s_hist[i] = 0;
    if (t < A.gw)
        s_range[t] = otsu_range(A.col + t);
    __syncthreads();
    const int64_t total = A.rows * A.gw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + t; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / A.gw;
        const int c = (int)(i - r * A.gw);
        if (A.labels && A.labels[r] != A.label)
            continue;
        const double v = A.x[r * A.stride + A.c0 + c];
        if (v != v)
            continue;
        const OtsuRange R = s_range[c];
        // np.histogram's uniform-bin rule: scaled position, truncated, then corrected against the edges
        int idx = (int)(((v - R.first) / R.denom) * 256.0);
        if (idx == 256)
            idx = 255;
        if (v < otsu_edge(R, idx))
            idx -= 1;
        if (idx != 255 && v >= otsu_edge(R, idx + 1))
            idx += 1;
        atomicAdd(&s_hist[c * BIN_DIGITS + idx], 1u);
    }
    __syncthreads();
    for (int i = t; i < A.gw * BIN_DIGITS; i += FX_BLOCK)
        if (s_hist[i])
            atomicAdd(&A.hist[i], s_hist[i]);
}

__global__ void bin_otsu_finish_kernel(BinArgs A, int constant_rule, double *thr, int32_t *flags, int64_t *count)
{
    const int t = threadIdx.x;
    if (t < A.gw) {
        const BinCol *c = A.col + t;
        const u64 n = c->count;
        const bool has_nan = c->nan_count != 0;
        int f = has_nan ? 2 : 0;
        double r;
        if (has_nan) {
            r = __builtin_nan("");
        } else if (n == 0) {
            r = 0.0;  // _otsu_threshold_1d of an empty array
        } else {
            const OtsuRange R = otsu_range(c);
            const uint32_t *h = A.hist + t * BIN_DIGITS;
            const double total = (double)n;
            // first sweep: mu_t = last entry of cumsum(p * mid)
            double mu_t = 0.0;
            for (int i = 0; i < 256; i++) {
                const double p = (double)h[i] / total;
                const double mid = (otsu_edge(R, i) + otsu_edge(R, i + 1)) / 2.0;
                mu_t = i == 0 ? p * mid : mu_t + p * mid;
            }
            double omega = 0.0, mu = 0.0, best = 0.0;
            int arg = 0;
            for (int i = 0; i < 256; i++) {
                const double p = (double)h[i] / total;
                const double mid = (otsu_edge(R, i) + otsu_edge(R, i + 1)) / 2.0;
                omega = i == 0 ? p : omega + p;
                mu = i == 0 ? p * mid : mu + p * mid;
                const double d = mu_t * omega - mu;
                const double s = d * d / (omega * (1.0 - omega) + 1e-12);
                if (i == 0 || s > best) {  // nanargmax: first of the largest (s is never NaN here)
                    best = s;
                    arg = i;
                }
            }
            r = (otsu_edge(R, arg) + otsu_edge(R, arg + 1)) / 2.0;
            if (value_of(c->kmin) == value_of(c->kmax)) {
                f |= 1;
                if (constant_rule)
                    r = A.x[(int64_t)c->first_row * A.stride + A.c0 + t];
            }
        }
        thr[A.c0 + t] = r;
        if (flags)
            flags[A.c0 + t] = f;
        if (count && A.c0 + t == 0)
            count[0] = (int64_t)(n + c->nan_count);
    }
    __syncthreads();
    for (int i = t; i < BIN_GROUP * BIN_DIGITS; i += blockDim.x)
        A.hist[i] = 0;
}

__global__ __launch_bounds__(FX_BLOCK) void bin_apply_kernel(const double *x, int64_t rows, int cols, int64_t stride,
                                                           const double *thr, uint8_t *out)
{
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        out[i] = x[r * stride + c] >= thr[c] ? 1 : 0;
    }
}

unsigned bin_grid(int64_t elements)
{
    const int64_t want = (elements + FX_BLOCK - 1) / FX_BLOCK;
    return (unsigned)(want < 1 ? 1 : want > 2048 ? 2048 : want);
}

}  // namespace

extern "C" int fx_column_thresholds_f64(const double *d_x, int64_t rows, int cols, int64_t row_stride,
                                        const int32_t *d_labels, int label, int method, double q, int constant_rule,
                                        double *d_thr, int32_t *d_flags, int64_t *d_count, void *stream)
{
    FX_REQUIRE(rows >= 0 && cols >= 1, "fx_column_thresholds_f64: table %lld x %d", (long long)rows, cols);
    FX_REQUIRE(d_x != nullptr || rows == 0, "fx_column_thresholds_f64: d_x is NULL");
    FX_REQUIRE(d_thr != nullptr, "fx_column_thresholds_f64: d_thr is NULL");
    FX_REQUIRE(row_stride >= cols, "fx_column_thresholds_f64: row stride %lld < %d columns", (long long)row_stride, cols);
    FX_REQUIRE(method >= 0 && method <= 2, "fx_column_thresholds_f64: method %d (0 median, 1 quantile, 2 otsu)", method);
    FX_REQUIRE(method != 1 || (q >= 0.0 && q <= 1.0), "fx_column_thresholds_f64: quantile %g outside [0, 1]", q);
    hipStream_t s = (hipStream_t)stream;
    FxCallGuard guard(s);
    unsigned char *base = (unsigned char *)guard.scratch(BIN_GROUP * sizeof(BinCol) + BIN_Q * BIN_DIGITS * 4, 6);
    if (!base)
        return FX_ENOMEM;
    BinArgs A;
    A.x = d_x;
    A.rows = rows;
    A.cols = cols;
    A.stride = row_stride;
    A.labels = d_labels;
    A.label = label;
    A.col = (BinCol *)base;
    A.hist = (uint32_t *)(base + BIN_GROUP * sizeof(BinCol));
    for (int c0 = 0; c0 < cols; c0 += BIN_GROUP) {
        A.c0 = c0;
        A.gw = cols - c0 < BIN_GROUP ? cols - c0 : BIN_GROUP;
        const unsigned grid = bin_grid(rows * A.gw);
        hipLaunchKernelGGL(bin_reset_kernel, dim3(1), dim3(FX_BLOCK), 0, s, A.col, A.hist);
        if (rows > 0)
            hipLaunchKernelGGL(bin_prepass_kernel, dim3(grid), dim3(FX_BLOCK), 0, s, A);
        if (method == 2) {
            if (rows > 0)
                hipLaunchKernelGGL(bin_otsu_hist_kernel, dim3(grid), dim3(FX_BLOCK), 0, s, A);
            hipLaunchKernelGGL(bin_otsu_finish_kernel, dim3(1), dim3(FX_BLOCK), 0, s, A, constant_rule, d_thr, d_flags, d_count);
        } else {
            hipLaunchKernelGGL(bin_ranks_kernel, dim3(1), dim3(FX_WAVE), 0, s, A.col, A.gw, method, q);
            for (int shift = 56; shift >= 0 && rows > 0; shift -= 8) {
                hipLaunchKernelGGL(bin_digit_hist_kernel, dim3(grid), dim3(FX_BLOCK), 0, s, A, shift);
                hipLaunchKernelGGL(bin_digit_pick_kernel, dim3(1), dim3(FX_BLOCK), 0, s, A.col, A.hist, A.gw, shift);
            }
            hipLaunchKernelGGL(bin_finish_kernel, dim3(1), dim3(FX_WAVE), 0, s, A, method, constant_rule, d_thr, d_flags, d_count);
        }
        FX_HIP_CHECK(hipGetLastError());
    }
    return FX_OK;
}

extern "C" int fx_binarize_ge_f64(const double *d_x, int64_t rows, int cols, int64_t row_stride, const double *d_thr,
                                  uint8_t *d_out, void *stream)
{
    FX_REQUIRE(rows >= 0 && cols >= 1, "fx_binarize_ge_f64: table %lld x %d", (long long)rows, cols);
    if (rows == 0)
        return FX_OK;
    FX_REQUIRE(d_x && d_thr && d_out, "fx_binarize_ge_f64: NULL argument");
    FX_REQUIRE(row_stride >= cols, "fx_binarize_ge_f64: row stride %lld < %d columns", (long long)row_stride, cols);
    hipLaunchKernelGGL(bin_apply_kernel, dim3(bin_grid(rows * cols)), dim3(FX_BLOCK), 0, (hipStream_t)stream, d_x, rows, cols,
                       row_stride, d_thr, d_out);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

// libfx: the stage that produces the fixed-size point sets the PCA path reads
// (reference src/preprocessing/prepare_all.py: center_and_scale :199-217,
// farthest_point_sampling :250-293).  float64 arithmetic in NumPy's operation
// order, so that centroids, scaled coordinates and sampled indices equal the
// reference's bit for bit:
//   * np.mean(points, axis=0) adds the rows one after the other (no pairwise
//     tree along axis 0): one lane per (cloud, column) walks down its cloud;
//   * np.linalg.norm(v, axis=1) = sqrt((x*x + y*y) + z*z);
//   * np.argmax returns the first of equal maxima.
#include "fx_common.h"

#include <vector>

namespace {

constexpr int FPS_THREADS = 1024;
constexpr int FPS_WAVES = FPS_THREADS / FX_WAVE;

struct CloudArgs {
    const void *pts;         // [total][3] float32 or float64
    int wide;                // 1: float64 input
    const int64_t *offsets;  // [B + 1], device
    int64_t B;
    double *centroid;        // [B][3] scratch
    unsigned long long *maxbits;  // [B] scratch, bit pattern of the largest norm (non-negative doubles order like integers)
    float *out_pts;          // [total][3]
    float *out_centroid;     // [B][3]
    double *out_maxdist;     // [B] or null
    int64_t total;
};

__device__ __forceinline__ double cloud_load(const CloudArgs &A, int64_t i, int c)
{
    return A.wide ? ((const double *)A.pts)[i * 3 + c] : (double)((const float *)A.pts)[i * 3 + c];
}

// Column sums of a cloud with the rows added in order, as NumPy does.  A wave
// per cloud: all lanes stream the cloud through LDS in tiles (coalesced loads,
// the next tile in flight), lanes 0..2 each add one column's values one after
// the other.
constexpr int SUM_TILE = 128;            // rows per tile
constexpr int SUM_TILE3 = SUM_TILE * 3;  // values per tile
constexpr int SUM_VPL = SUM_TILE3 / FX_WAVE;

__global__ __launch_bounds__(FX_BLOCK) void cloud_colsum_kernel(CloudArgs A)
{
    __shared__ double s_tile[FX_WAVES_PER_BLOCK][2][SUM_TILE3];
    const int lane = fx_lane(), wave = fx_wave();
    const int64_t b = (int64_t)blockIdx.x * FX_WAVES_PER_BLOCK + wave;
    if (b >= A.B)
        return;
    const int64_t lo = A.offsets[b], hi = A.offsets[b + 1];
    const int64_t m = hi - lo, n3 = m * 3;
    const int64_t base = lo * 3;
    auto load = [&](int64_t tile, double (&v)[SUM_VPL]) {
#pragma unroll
        for (int j = 0; j < SUM_VPL; j++) {
            const int64_t e = tile * SUM_TILE3 + j * FX_WAVE + lane;
            v[j] = e < n3 ? (A.wide ? ((const double *)A.pts)[base + e] : (double)((const float *)A.pts)[base + e]) : 0.0;
        }
    };
    auto store = [&](int buf, const double (&v)[SUM_VPL]) {
#pragma unroll
        for (int j = 0; j < SUM_VPL; j++)
            s_tile[wave][buf][j * FX_WAVE + lane] = v[j];
    };
    const int64_t tiles = (m + SUM_TILE - 1) / SUM_TILE;
    double v[SUM_VPL];
    double s = 0.0;
    if (tiles > 0) {
        load(0, v);
        store(0, v);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int64_t k = 0; k < tiles; k++) {
        if (k + 1 < tiles)
            load(k + 1, v);
        if (lane < 3) {
            const int64_t left = m - k * SUM_TILE;
            const int rows = (int)(left < SUM_TILE ? left : SUM_TILE);
            const double *p = &s_tile[wave][k & 1][lane];
            int r = 0;
            for (; r + 16 <= rows; r += 16) {
                double w[16];
#pragma unroll
                for (int j = 0; j < 16; j++)
                    w[j] = p[(r + j) * 3];
#pragma unroll
                for (int j = 0; j < 16; j++)
                    s += w[j];
            }
            for (; r < rows; r++)
                s += p[r * 3];
        }
        if (k + 1 < tiles)
            store((int)((k + 1) & 1), v);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane < 3)
        A.centroid[b * 3 + lane] = m > 0 ? s / (double)m : 0.0;
    if (lane == 0)
        A.maxbits[b] = 0ull;
}

__device__ __forceinline__ int64_t cloud_of_point(const int64_t *offsets, int64_t B, int64_t i)
{
    int64_t lo = 0, hi = B;  // offsets[lo] <= i < offsets[hi]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= i)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

// Largest distance from the centroid per cloud.
__global__ __launch_bounds__(FX_BLOCK) void cloud_maxnorm_kernel(CloudArgs A)
{
    const int64_t i = (int64_t)blockIdx.x * FX_BLOCK + threadIdx.x;
    if (i >= A.total)
        return;
    const int64_t b = cloud_of_point(A.offsets, A.B, i);
    const double x = cloud_load(A, i, 0) - A.centroid[b * 3 + 0];
    const double y = cloud_load(A, i, 1) - A.centroid[b * 3 + 1];
    const double z = cloud_load(A, i, 2) - A.centroid[b * 3 + 2];
    double d = sqrt((x * x + y * y) + z * z);
    // one atomic per wave when the whole wave lies in one cloud (the usual case)
    const int64_t b0 = __shfl(b, 0, FX_WAVE);
    const bool full = __ballot(1) == ~0ull;
    if (full && __all(b == b0)) {
        d = fx_wave_max_f64(d);
        if (fx_lane() == 0)
            atomicMax(&A.maxbits[b], (unsigned long long)__double_as_longlong(d));
    } else {
        atomicMax(&A.maxbits[b], (unsigned long long)__double_as_longlong(d));
    }
}

__global__ __launch_bounds__(FX_BLOCK) void cloud_scale_kernel(CloudArgs A)
{
    const int64_t g = (int64_t)blockIdx.x * FX_BLOCK + threadIdx.x;
    if (g < A.B * 3)
        A.out_centroid[g] = (float)A.centroid[g];
    if (g < A.B && A.out_maxdist)
        A.out_maxdist[g] = __longlong_as_double((long long)A.maxbits[g]);
    if (g >= A.total)
        return;
    const int64_t b = cloud_of_point(A.offsets, A.B, g);
    const double md = __longlong_as_double((long long)A.maxbits[b]);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double v = cloud_load(A, g, c) - A.centroid[b * 3 + c];
        A.out_pts[g * 3 + c] = (float)(md > 0.0 ? v / md : v);
    }
}

// ----------------------------------------------------------------- FPS
struct FpsArgs {
    const float *pts;         // [total][3]
    const int64_t *offsets;   // [B + 1], device
    const int32_t *ids;       // clouds of this launch
    const int64_t *start;     // [B] first index per cloud
    int n_samples;
    int64_t *idx;             // [B][n_samples]
    float *sampled;           // [B][n_samples][3] or null
    double *dist;             // [total] scratch, used by clouds that do not fit in registers
};

// sqrt(d2) can only lower `dist` if d2 < dist^2; the product is within a few
// ulps of the squared distance `dist` came from, so beyond this margin the
// square root need not be taken (rounding is monotone).
__device__ __forceinline__ bool fps_may_lower(double d2, double dist)
{
    return !(d2 > dist * dist * (1.0 + 0x1p-50));
}

// A block per cloud.  PPT > 0: the cloud has at most FPS_THREADS * PPT points
// and a thread keeps its points and their distances to the chosen set in
// registers; PPT == 0: any size, distances in scratch.
template <int PPT>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(FpsArgs A)
{
    __shared__ double s_best[2][FPS_WAVES];
    __shared__ int s_idx[2][FPS_WAVES];
    const int b = A.ids[blockIdx.x];
    const int t = threadIdx.x, lane = fx_lane(), wave = fx_wave();
    const int64_t lo = A.offsets[b];
    const int m = (int)(A.offsets[b + 1] - lo);
    const float *P = A.pts + lo * 3;
    double *D = A.dist + lo;
    int64_t *out_idx = A.idx + (int64_t)b * A.n_samples;
    float *out_pts = A.sampled ? A.sampled + (int64_t)b * A.n_samples * 3 : nullptr;

    constexpr bool REG = PPT > 0;
    float px[REG ? PPT : 1], py[REG ? PPT : 1], pz[REG ? PPT : 1];
    double dist[REG ? PPT : 1];
    if (REG) {
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const int i = t + k * FPS_THREADS;
            px[k] = i < m ? P[i * 3 + 0] : 0.f;
            py[k] = i < m ? P[i * 3 + 1] : 0.f;
            pz[k] = i < m ? P[i * 3 + 2] : 0.f;
            dist[k] = 0.0;
        }
    }

    int cur = (int)A.start[b];
    for (int it = 0; it < A.n_samples; it++) {
        const double cx = (double)P[cur * 3 + 0], cy = (double)P[cur * 3 + 1], cz = (double)P[cur * 3 + 2];
        if (t == 0) {
            out_idx[it] = cur;
            if (out_pts) {
                out_pts[it * 3 + 0] = (float)cx;
                out_pts[it * 3 + 1] = (float)cy;
                out_pts[it * 3 + 2] = (float)cz;
            }
        }
        if (it == A.n_samples - 1)
            break;
        // distance of every point to the chosen set, and the farthest one (the first among equals)
        double bd = -1.0;
        int bi = 0x7fffffff;
        if (REG) {
#pragma unroll
            for (int k = 0; k < PPT; k++) {
                const int i = t + k * FPS_THREADS;
                const double dx = (double)px[k] - cx, dy = (double)py[k] - cy, dz = (double)pz[k] - cz;
                const double d2 = (dx * dx + dy * dy) + dz * dz;
                if (it == 0)
                    dist[k] = sqrt(d2);
                else if (fps_may_lower(d2, dist[k]))
                    dist[k] = fmin(dist[k], sqrt(d2));
                if (i < m && dist[k] > bd) {
                    bd = dist[k];
                    bi = i;
                }
            }
        } else {
            for (int i = t; i < m; i += FPS_THREADS) {
                const double dx = (double)P[i * 3 + 0] - cx, dy = (double)P[i * 3 + 1] - cy,
                             dz = (double)P[i * 3 + 2] - cz;
                const double d2 = (dx * dx + dy * dy) + dz * dz;
                double d;
                if (it == 0) {
                    d = sqrt(d2);
                    D[i] = d;
                } else {
                    d = D[i];
                    if (fps_may_lower(d2, d)) {
                        const double nd = sqrt(d2);
                        if (nd < d) {
                            d = nd;
                            D[i] = d;
                        }
                    }
                }
                if (d > bd) {
                    bd = d;
                    bi = i;
                }
            }
        }
#pragma unroll
        for (int o = FX_WAVE / 2; o > 0; o >>= 1) {
            const double od = __shfl_xor(bd, o, FX_WAVE);
            const int oi = __shfl_xor(bi, o, FX_WAVE);
            if (od > bd || (od == bd && oi < bi)) {
                bd = od;
                bi = oi;
            }
        }
        const int slot = it & 1;  // two sets of slots: one barrier per round is enough
        if (lane == 0) {
            s_best[slot][wave] = bd;
            s_idx[slot][wave] = bi;
        }
        __syncthreads();
        bd = s_best[slot][0];
        bi = s_idx[slot][0];
#pragma unroll
        for (int w = 1; w < FPS_WAVES; w++) {
            const double od = s_best[slot][w];
            const int oi = s_idx[slot][w];
            if (od > bd || (od == bd && oi < bi)) {
                bd = od;
                bi = oi;
            }
        }
        cur = bi;
    }
}

inline size_t fps_align(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" int fx_center_scale(const void *d_pts, int wide, const int64_t *h_offsets, int64_t B, float *d_out,
                               float *d_centroid, double *d_maxdist, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    FX_REQUIRE(B >= 0, "fx_center_scale: B = %lld", (long long)B);
    if (B == 0)
        return FX_OK;
    FX_REQUIRE(d_pts && h_offsets && d_out && d_centroid, "fx_center_scale: null pointer");
    FX_REQUIRE(h_offsets[0] == 0, "fx_center_scale: offsets[0] = %lld", (long long)h_offsets[0]);
    for (int64_t b = 0; b < B; b++)
        FX_REQUIRE(h_offsets[b + 1] >= h_offsets[b], "fx_center_scale: offsets decrease at cloud %lld", (long long)b);
    const int64_t total = h_offsets[B];
    const size_t off_bytes = fps_align((size_t)(B + 1) * sizeof(int64_t));
    const size_t cen_bytes = fps_align((size_t)B * 3 * sizeof(double));
    const size_t max_bytes = fps_align((size_t)B * sizeof(unsigned long long));
    FxCallGuard guard((hipStream_t)stream);
    unsigned char *base = (unsigned char *)guard.scratch(off_bytes + cen_bytes + max_bytes, 5);
    if (!base)
        return FX_ENOMEM;
    CloudArgs A;
    A.pts = d_pts;
    A.wide = wide ? 1 : 0;
    A.offsets = (const int64_t *)base;
    A.B = B;
    A.centroid = (double *)(base + off_bytes);
    A.maxbits = (unsigned long long *)(base + off_bytes + cen_bytes);
    A.out_pts = d_out;
    A.out_centroid = d_centroid;
    A.out_maxdist = d_maxdist;
    A.total = total;
    FX_HIP_CHECK(hipMemcpyAsync(base, h_offsets, (size_t)(B + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    FxProfileScope prof(stream);
    hipLaunchKernelGGL(cloud_colsum_kernel, dim3((unsigned)((B + FX_WAVES_PER_BLOCK - 1) / FX_WAVES_PER_BLOCK)),
                       dim3(FX_BLOCK), 0, stream, A);
    if (total > 0)
        hipLaunchKernelGGL(cloud_maxnorm_kernel, dim3((unsigned)((total + FX_BLOCK - 1) / FX_BLOCK)), dim3(FX_BLOCK),
                           0, stream, A);
    const int64_t span = total > B * 3 ? total : B * 3;
    hipLaunchKernelGGL(cloud_scale_kernel, dim3((unsigned)((span + FX_BLOCK - 1) / FX_BLOCK)), dim3(FX_BLOCK), 0,
                       stream, A);
    FX_HIP_CHECK(hipGetLastError());
    // the staging copy of the offsets must not be overwritten by a later call before it has been read
    FX_HIP_CHECK(hipStreamSynchronize(stream));
    return FX_OK;
}

extern "C" int fx_fps_f32(const float *d_pts, const int64_t *h_offsets, int64_t B, int n_samples,
                          const int64_t *h_start, int64_t *d_idx, float *d_sampled, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    FX_REQUIRE(B >= 0 && B < (1ll << 31), "fx_fps_f32: B = %lld", (long long)B);
    if (B == 0)
        return FX_OK;
    FX_REQUIRE(d_pts && h_offsets && h_start && d_idx, "fx_fps_f32: null pointer");
    FX_REQUIRE(n_samples >= 1, "fx_fps_f32: n_samples = %d", n_samples);
    FX_REQUIRE(h_offsets[0] == 0, "fx_fps_f32: offsets[0] = %lld", (long long)h_offsets[0]);
    // size classes: 1, 2, 4, 8, 16 points per thread in registers, then the scratch path
    std::vector<int32_t> cls[6];
    for (int64_t b = 0; b < B; b++) {
        const int64_t m = h_offsets[b + 1] - h_offsets[b];
        FX_REQUIRE(m > n_samples && m < (1ll << 31) / 3,
                   "fx_fps_f32: cloud %lld has %lld points; sampling %d needs more than that many (and fewer than 2^31 / 3)",
                   (long long)b, (long long)m, n_samples);
        FX_REQUIRE(h_start[b] >= 0 && h_start[b] < m, "fx_fps_f32: start index %lld of cloud %lld outside 0..%lld",
                   (long long)h_start[b], (long long)b, (long long)m - 1);
        int c = 0;
        while (c < 5 && m > (int64_t)FPS_THREADS << c)
            c++;
        cls[c].push_back((int32_t)b);
    }
    const int64_t total = h_offsets[B];
    const size_t off_bytes = fps_align((size_t)(B + 1) * sizeof(int64_t));
    const size_t start_bytes = fps_align((size_t)B * sizeof(int64_t));
    const size_t ids_bytes = fps_align((size_t)B * sizeof(int32_t));
    const size_t dist_bytes = cls[5].empty() ? 0 : fps_align((size_t)total * sizeof(double));
    FxCallGuard guard((hipStream_t)stream);
    unsigned char *base = (unsigned char *)guard.scratch(off_bytes + start_bytes + ids_bytes + dist_bytes, 5);
    if (!base)
        return FX_ENOMEM;
    std::vector<int32_t> ids;
    for (int c = 0; c < 6; c++)
        ids.insert(ids.end(), cls[c].begin(), cls[c].end());
    FX_HIP_CHECK(hipMemcpyAsync(base, h_offsets, (size_t)(B + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    FX_HIP_CHECK(hipMemcpyAsync(base + off_bytes, h_start, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, stream));
    FX_HIP_CHECK(hipMemcpyAsync(base + off_bytes + start_bytes, ids.data(), (size_t)B * sizeof(int32_t),
                                hipMemcpyHostToDevice, stream));
    FpsArgs A;
    A.pts = d_pts;
    A.offsets = (const int64_t *)base;
    A.start = (const int64_t *)(base + off_bytes);
    A.n_samples = n_samples;
    A.idx = d_idx;
    A.sampled = d_sampled;
    A.dist = (double *)(base + off_bytes + start_bytes + ids_bytes);
    FxProfileScope prof(stream);
    const int32_t *d_ids = (const int32_t *)(base + off_bytes + start_bytes);
    size_t at = 0;
    for (int c = 0; c < 6; c++) {
        if (cls[c].empty())
            continue;
        A.ids = d_ids + at;
        at += cls[c].size();
        const dim3 grid((unsigned)cls[c].size()), block(FPS_THREADS);
        switch (c) {
        case 0: hipLaunchKernelGGL(fps_kernel<1>, grid, block, 0, stream, A); break;
        case 1: hipLaunchKernelGGL(fps_kernel<2>, grid, block, 0, stream, A); break;
        case 2: hipLaunchKernelGGL(fps_kernel<4>, grid, block, 0, stream, A); break;
        case 3: hipLaunchKernelGGL(fps_kernel<8>, grid, block, 0, stream, A); break;
        case 4: hipLaunchKernelGGL(fps_kernel<16>, grid, block, 0, stream, A); break;
        default: hipLaunchKernelGGL(fps_kernel<0>, grid, block, 0, stream, A); break;
        }
    }
    FX_HIP_CHECK(hipGetLastError());
    // the host vectors staged above go out of scope with this call
    FX_HIP_CHECK(hipStreamSynchronize(stream));
    return FX_OK;
}

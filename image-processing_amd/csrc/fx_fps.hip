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

#include <algorithm>
#include <string.h>
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

// Column sums of a cloud with the rows added in order, as NumPy does: the chain
// of additions is the algorithm, so a cloud costs at least its row count times
// the latency of a dependent float64 addition.  A block per cloud keeps that
// chain fed: all four waves stream the cloud into LDS in tiles (coalesced loads,
// the next tile requested before the current one is summed), and three lanes
// of the first wave add one column each, value after value, reading the LDS
// tile (widened to float64 and transposed, so that one read brings two
// values) sixteen rows ahead of the additions.
constexpr int SUM_TILE = 1024;           // rows per tile
constexpr int SUM_TILE3 = SUM_TILE * 3;  // values per tile
constexpr int SUM_VPT = SUM_TILE3 / FX_BLOCK;
constexpr int SUM_AHEAD = 16;

template <typename T>
__device__ __forceinline__ double cloud_colsum_block(const T *p, int64_t m, double *s_tile /* [2][3][SUM_TILE] */)
{
    const int t = threadIdx.x;
    const int64_t n3 = m * 3;
    const int64_t tiles = (m + SUM_TILE - 1) / SUM_TILE;
    T v[SUM_VPT];
    auto load = [&](int64_t tile) {
#pragma unroll
        for (int j = 0; j < SUM_VPT; j++) {
            const int64_t e = tile * SUM_TILE3 + j * FX_BLOCK + t;
            v[j] = e < n3 ? p[e] : (T)0;
        }
    };
    // widened and transposed on the way in: a column's values lie side by side
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < SUM_VPT; j++) {
            const int e = j * FX_BLOCK + t;
            s_tile[buf * SUM_TILE3 + (e % 3) * SUM_TILE + e / 3] = (double)v[j];
        }
    };
    double s = 0.0;
    if (tiles > 0) {
        load(0);
        store(0);
    }
    __syncthreads();
    for (int64_t k = 0; k < tiles; k++) {
        if (k + 1 < tiles)
            load(k + 1);
        if (t < 3) {
            const int64_t left = m - k * SUM_TILE;
            const int rows = (int)(left < SUM_TILE ? left : SUM_TILE);
            const double *q = s_tile + (k & 1) * SUM_TILE3 + t * SUM_TILE;
            const int chunks = rows / SUM_AHEAD;
            double2 w0[SUM_AHEAD / 2], w1[SUM_AHEAD / 2];
            auto fetch = [&](double2 (&w)[SUM_AHEAD / 2], int c) {
#pragma unroll
                for (int j = 0; j < SUM_AHEAD / 2; j++)
                    w[j] = *(const double2 *)&q[c * SUM_AHEAD + 2 * j];
            };
            auto add = [&](const double2 (&w)[SUM_AHEAD / 2]) {
#pragma unroll
                for (int j = 0; j < SUM_AHEAD / 2; j++) {
                    s += w[j].x;
                    s += w[j].y;
                }
            };
            if (chunks > 0)
                fetch(w0, 0);
            int c = 0;
            for (; c + 2 <= chunks; c += 2) {
                fetch(w1, c + 1);
                add(w0);
                if (c + 2 < chunks)
                    fetch(w0, c + 2);
                add(w1);
            }
            if (c < chunks)
                add(w0);
            for (int r = chunks * SUM_AHEAD; r < rows; r++)
                s += q[r];
        }
        if (k + 1 < tiles)
            store((int)((k + 1) & 1));
        __syncthreads();
    }
    return s;
}

__global__ __launch_bounds__(FX_BLOCK) void cloud_colsum_kernel(CloudArgs A)
{
    __shared__ double s_tile[2 * SUM_TILE3];
    const int64_t b = blockIdx.x;
    const int64_t lo = A.offsets[b], m = A.offsets[b + 1] - lo;
    const double s = A.wide ? cloud_colsum_block((const double *)A.pts + lo * 3, m, s_tile)
                            : cloud_colsum_block((const float *)A.pts + lo * 3, m, s_tile);
    if (threadIdx.x < 3)
        A.centroid[b * 3 + threadIdx.x] = m > 0 ? s / (double)m : 0.0;
    if (threadIdx.x == 0)
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
};

// sqrt(d2) can only lower `dist` if d2 < dist^2; the product is within a few
// ulps of the squared distance `dist` came from, so beyond this margin the
// square root need not be taken (rounding is monotone).
__device__ __forceinline__ bool fps_may_lower(double d2, double dist)
{
    return !(d2 > dist * dist * (1.0 + 0x1p-50));
}

// A block per cloud of at most FPS_THREADS * PPT points: a thread keeps its
// points and their distances to the chosen set in registers.
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
    int64_t *out_idx = A.idx + (int64_t)b * A.n_samples;
    float *out_pts = A.sampled ? A.sampled + (int64_t)b * A.n_samples * 3 : nullptr;

    float px[PPT], py[PPT], pz[PPT];
    double dist[PPT];
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int i = t + k * FPS_THREADS;
        px[k] = i < m ? P[i * 3 + 0] : 0.f;
        py[k] = i < m ? P[i * 3 + 1] : 0.f;
        pz[k] = i < m ? P[i * 3 + 2] : 0.f;
        dist[k] = 0.0;
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

// ------------------------------------------------- FPS of large clouds
// Clouds that do not fit in a block's registers.  The samples are those of the
// plain algorithm; only the work is different.  A new sample can lower the
// distance-to-the-chosen-set of points near it and of no others, so the cloud
// is first sorted into space-filling-curve order (a counting sort on the
// Morton code of a grid cell; the order within a cell is arbitrary) and cut into
// groups of consecutive points with their bounding boxes.  Per round a group is
// visited only if its box comes closer to the new sample than the largest
// distance in the group; every other group keeps its distances and its maximum.
// Rounding is monotone, so the distance from the sample to the box, evaluated
// with the operations of the point distance, never exceeds the computed
// distance of a point inside the box: skipping is exact.
//
// The kernel keeps squared distances (the minimum of square roots is the square
// root of the minimum); np.argmax's "first of equal maxima" is decided on the
// square roots, which can be equal for squared distances that differ in the
// last bits, so whenever another candidate lies within that window the winner
// is chosen by an exact pass over the candidates' square roots.
constexpr int FPS_MAXG = 2048;           // groups per cloud (two per thread)
#ifndef FPS_PRUNE_ABOVE
#define FPS_PRUNE_ABOVE 4096             // clouds with more points take this kernel
#endif
constexpr int FPS_BIG_LDS_PER_GROUP = 8 + 24 + 12 + 4 + 4;

struct FpsBigDesc {
    int64_t soff;   // first element of the cloud in `sorted` / `mind2`
    int64_t coff;   // first counter of the cloud in `cells`
    int gb;         // grid bits per axis
    int gshift;     // log2(points per group)
};

struct FpsBigArgs {
    const float *pts;
    const int64_t *offsets;
    const int32_t *ids;
    const int64_t *start;
    int n_samples;
    int64_t *idx;
    float *sampled;
    const FpsBigDesc *desc;   // [blocks of this launch]
    float4 *sorted;           // x, y, z, bit pattern of the original index
    double *mind2;            // squared distance to the chosen set, in sorted order
    unsigned int *cells;      // grid counters of the clouds whose grid does not fit in LDS
    int maxg;                 // pitch of the per-group LDS arrays
};

__device__ __forceinline__ unsigned int fps_spread3(unsigned int v)
{
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__device__ __forceinline__ unsigned int fps_wave_min_u32(unsigned int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int w = (unsigned int)__shfl_xor((int)v, o, FX_WAVE);
        v = w < v ? w : v;
    }
    return v;
}

__global__ __launch_bounds__(FPS_THREADS) void fps_pruned_kernel(FpsBigArgs A)
{
    extern __shared__ __align__(16) unsigned char s_dyn[];
    const int G = A.maxg;
    double *s_gmax = (double *)s_dyn;                       // [G] largest squared distance of a group
    float *s_box = (float *)(s_gmax + G);                   // [6][G] lo x, y, z, hi x, y, z
    float *s_gxyz = s_box + 6 * G;                          // [3][G] coordinates of the group's farthest point
    unsigned int *s_gidx = (unsigned int *)(s_gxyz + 3 * G);  // [G] its original index; bit 31: decide by square roots
    unsigned int *s_list = s_gidx + G;                      // [G] groups to visit
    __shared__ double s_red[FPS_WAVES];
    __shared__ float s_fred[6][FPS_WAVES];
    __shared__ unsigned int s_scan[FPS_WAVES];
    __shared__ double s_c[3];
    __shared__ unsigned int s_cidx;
    __shared__ unsigned int s_cnt[2];
    __shared__ unsigned long long s_win;

    const int b = A.ids[blockIdx.x];
    const FpsBigDesc D = A.desc[blockIdx.x];
    const int t = threadIdx.x, lane = fx_lane(), wave = fx_wave();
    const int64_t lo_row = A.offsets[b];
    const int m = (int)(A.offsets[b + 1] - lo_row);
    const float *P = A.pts + lo_row * 3;
    float4 *S = A.sorted + D.soff;
    double *M2 = A.mind2 + D.soff;
    unsigned int *cid = (unsigned int *)M2;   // cell of a point, until the distances are initialised
    const int ncell = 1 << (3 * D.gb);
    const int gs = 1 << D.gshift;
    const int ng = (m + gs - 1) >> D.gshift;
    int64_t *out_idx = A.idx + (int64_t)b * A.n_samples;
    float *out_pts = A.sampled ? A.sampled + (int64_t)b * A.n_samples * 3 : nullptr;
    const int start = (int)A.start[b];
    const double inf = __longlong_as_double(0x7ff0000000000000ll);

    // bounds of the cloud; clear the grid
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    constexpr int UN = 4;   // independent loads in flight per thread in the passes over the cloud
    {
        int i = t;
        for (; i + (UN - 1) * FPS_THREADS < m; i += UN * FPS_THREADS) {
            float v[UN][3];
#pragma unroll
            for (int u = 0; u < UN; u++)
#pragma unroll
                for (int c = 0; c < 3; c++)
                    v[u][c] = P[(i + u * FPS_THREADS) * 3 + c];
#pragma unroll
            for (int u = 0; u < UN; u++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    lo[c] = fminf(lo[c], v[u][c]);
                    hi[c] = fmaxf(hi[c], v[u][c]);
                }
        }
        for (; i < m; i += FPS_THREADS)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float v = P[i * 3 + c];
                lo[c] = fminf(lo[c], v);
                hi[c] = fmaxf(hi[c], v);
            }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        lo[c] = fx_wave_min_f32_dpp(lo[c]);
        hi[c] = fx_wave_max_f32_dpp(hi[c]);
        if (lane == 0) {
            s_fred[c][wave] = lo[c];
            s_fred[3 + c][wave] = hi[c];
        }
    }
    if (t == 0) {
        s_cnt[0] = 0u;
        s_cnt[1] = 0u;
        s_win = ~0ull;
    }
    __syncthreads();
    float scale[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        for (int w = 0; w < FPS_WAVES; w++) {
            lo[c] = fminf(lo[c], s_fred[c][w]);
            hi[c] = fmaxf(hi[c], s_fred[3 + c][w]);
        }
        const float ext = hi[c] - lo[c];
        scale[c] = ext > 0.f && ext < __builtin_inff() ? (float)(1 << D.gb) / ext : 0.f;
    }
    // Counting sort by grid cell.  The counters live in LDS (the space the group tables take later) unless
    // the grid is too fine for it.
    const int gmaxc = (1 << D.gb) - 1;
    auto cell_of = [&](float x, float y, float z) {
        const float v[3] = {x, y, z};
        unsigned int code = 0;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            int q = (int)((v[c] - lo[c]) * scale[c]);
            q = q < 0 ? 0 : (q > gmaxc ? gmaxc : q);
            code |= fps_spread3((unsigned int)q) << c;
        }
        return code;
    };
    auto sort_cloud = [&](unsigned int *cells) {
    for (int c = t; c < ncell; c += FPS_THREADS)
        cells[c] = 0u;
    __syncthreads();
    {
        int i = t;
        for (; i + (UN - 1) * FPS_THREADS < m; i += UN * FPS_THREADS) {
            float v[UN][3];
#pragma unroll
            for (int u = 0; u < UN; u++)
#pragma unroll
                for (int c = 0; c < 3; c++)
                    v[u][c] = P[(i + u * FPS_THREADS) * 3 + c];
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const unsigned int code = cell_of(v[u][0], v[u][1], v[u][2]);
                cid[i + u * FPS_THREADS] = code;
                atomicAdd(&cells[code], 1u);
            }
        }
        for (; i < m; i += FPS_THREADS) {
            const unsigned int code = cell_of(P[i * 3 + 0], P[i * 3 + 1], P[i * 3 + 2]);
            cid[i] = code;
            atomicAdd(&cells[code], 1u);
        }
    }
    __syncthreads();
    // exclusive prefix sum over the cells (a contiguous run per thread)
    {
        const int per = ncell / FPS_THREADS;   // 4, 32 or 256
        unsigned int own = 0;
        for (int k = 0; k < per; k += 4) {
            const uint4 c4 = *(const uint4 *)&cells[t * per + k];
            own += (c4.x + c4.y) + (c4.z + c4.w);
        }
        unsigned int inc = own;
#pragma unroll
        for (int o = 1; o < FX_WAVE; o <<= 1) {
            const unsigned int v = (unsigned int)__shfl_up((int)inc, o, FX_WAVE);
            if (lane >= o)
                inc += v;
        }
        if (lane == FX_WAVE - 1)
            s_scan[wave] = inc;
        __syncthreads();
        unsigned int before = inc - own;
        for (int w = 0; w < wave; w++)
            before += s_scan[w];
        for (int k = 0; k < per; k += 4) {
            const uint4 c4 = *(const uint4 *)&cells[t * per + k];
            *(uint4 *)&cells[t * per + k] = make_uint4(before, before + c4.x, before + c4.x + c4.y, before + c4.x + c4.y + c4.z);
            before += (c4.x + c4.y) + (c4.z + c4.w);
        }
    }
    __syncthreads();
    // scatter into cell order
    {
        int i = t;
        for (; i + (UN - 1) * FPS_THREADS < m; i += UN * FPS_THREADS) {
            float v[UN][3];
            unsigned int pos[UN];
#pragma unroll
            for (int u = 0; u < UN; u++) {
#pragma unroll
                for (int c = 0; c < 3; c++)
                    v[u][c] = P[(i + u * FPS_THREADS) * 3 + c];
                pos[u] = atomicAdd(&cells[cid[i + u * FPS_THREADS]], 1u);
            }
#pragma unroll
            for (int u = 0; u < UN; u++)
                S[pos[u]] = make_float4(v[u][0], v[u][1], v[u][2], __int_as_float(i + u * FPS_THREADS));
        }
        for (; i < m; i += FPS_THREADS) {
            const unsigned int pos = atomicAdd(&cells[cid[i]], 1u);
            S[pos] = make_float4(P[i * 3 + 0], P[i * 3 + 1], P[i * 3 + 2], __int_as_float(i));
        }
    }
    __syncthreads();   // the counters' space is free again
    };
    if (D.gb <= 5)
        sort_cloud((unsigned int *)s_dyn);
    else
        sort_cloud(A.cells + D.coff);
    if (t == 0) {
        s_c[0] = (double)P[start * 3 + 0];
        s_c[1] = (double)P[start * 3 + 1];
        s_c[2] = (double)P[start * 3 + 2];
        s_cidx = (unsigned int)start;
    }
    __syncthreads();
    // distances start at infinity; boxes of the groups
    for (int i = t; i < m; i += FPS_THREADS)
        M2[i] = inf;
    auto put_box = [&](int g, float (&l3)[3], float (&h3)[3]) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            l3[c] = fx_wave_min_f32_dpp(l3[c]);
            h3[c] = fx_wave_max_f32_dpp(h3[c]);
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                s_box[c * G + g] = l3[c];
                s_box[(3 + c) * G + g] = h3[c];
            }
            s_gmax[g] = inf;
        }
    };
    if (D.gshift == 6) {
        for (int g0 = wave; g0 < ng; g0 += FPS_WAVES * UN) {
            float4 q[UN];
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const int p = ((g0 + u * FPS_WAVES) << 6) + lane;
                q[u] = S[p < m ? p : m - 1];
            }
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const int g = g0 + u * FPS_WAVES;
                if (g >= ng)
                    break;
                const bool ok = (g << 6) + lane < m;
                float l3[3] = {ok ? q[u].x : __builtin_inff(), ok ? q[u].y : __builtin_inff(), ok ? q[u].z : __builtin_inff()};
                float h3[3] = {ok ? q[u].x : -__builtin_inff(), ok ? q[u].y : -__builtin_inff(), ok ? q[u].z : -__builtin_inff()};
                put_box(g, l3, h3);
            }
        }
    } else {
        for (int g = wave; g < ng; g += FPS_WAVES) {
            float l3[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
            float h3[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
            for (int p = (g << D.gshift) + lane; p < ((g + 1) << D.gshift) && p < m; p += FX_WAVE) {
                const float4 q = S[p];
                l3[0] = fminf(l3[0], q.x); h3[0] = fmaxf(h3[0], q.x);
                l3[1] = fminf(l3[1], q.y); h3[1] = fmaxf(h3[1], q.y);
                l3[2] = fminf(l3[2], q.z); h3[2] = fmaxf(h3[2], q.z);
            }
            put_box(g, l3, h3);
        }
    }
    __syncthreads();

    for (int it = 0;; it++) {
        const double cx = s_c[0], cy = s_c[1], cz = s_c[2];
        if (t == 0) {
            out_idx[it] = (int64_t)s_cidx;
            if (out_pts) {
                out_pts[it * 3 + 0] = (float)cx;
                out_pts[it * 3 + 1] = (float)cy;
                out_pts[it * 3 + 2] = (float)cz;
            }
            s_cnt[1] = 0u;
            s_win = ~0ull;
        }
        if (it == A.n_samples - 1)
            break;
        // groups whose box comes closer to the new sample than their largest distance
#pragma unroll
        for (int k = 0; k < FPS_MAXG / FPS_THREADS; k++) {
            const int g = t + k * FPS_THREADS;
            bool act = false;
            if (g < ng) {
                const double ax = fmax(fmax((double)s_box[0 * G + g] - cx, cx - (double)s_box[3 * G + g]), 0.0);
                const double ay = fmax(fmax((double)s_box[1 * G + g] - cy, cy - (double)s_box[4 * G + g]), 0.0);
                const double az = fmax(fmax((double)s_box[2 * G + g] - cz, cz - (double)s_box[5 * G + g]), 0.0);
                const double lb = (ax * ax + ay * ay) + az * az;
                act = !(lb >= s_gmax[g]);
            }
            const unsigned long long bal = __ballot(act);
            if (bal) {
                unsigned int base = 0;
                if (lane == 0)
                    base = atomicAdd(&s_cnt[0], (unsigned int)__popcll(bal));
                base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
                if (act)
                    s_list[base + __builtin_amdgcn_mbcnt_hi((unsigned int)(bal >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((unsigned int)bal, 0u))] = (unsigned int)g;
            }
        }
        __syncthreads();
        const int nl = (int)s_cnt[0];
        // the wave's result for one group: largest value, first original index with it, whether another
        // value lies within the window in which square roots can coincide
        auto finish = [&](int g, double b1, double b2, unsigned int bi, float wx, float wy, float wz) {
            const double mx = fx_wave_max_f64_dpp(b1);
            const double thr = mx - mx * 0x1p-49;
            const bool cand = b1 == mx;
            const unsigned long long anyd = __ballot(cand ? b2 >= thr : b1 >= thr);
            unsigned long long own = __ballot(cand);
            unsigned int wi = bi;
            if (__popcll(own) != 1) {   // equal maxima: the first original index among them
                wi = fps_wave_min_u32(cand ? bi : ~0u);
                own = __ballot(cand && bi == wi);
            }
            const int src = own ? __builtin_ctzll(own) : 0;
            wi = (unsigned int)__builtin_amdgcn_readlane((int)wi, src);
            wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wx), src));
            wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wy), src));
            wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wz), src));
            if (lane == 0) {
                s_gmax[g] = mx;
                s_gidx[g] = (wi & 0x7fffffffu) | (anyd || !own ? 0x80000000u : 0u);
                s_gxyz[0 * G + g] = wx;
                s_gxyz[1 * G + g] = wy;
                s_gxyz[2 * G + g] = wz;
            }
        };
        if (D.gshift == 6) {
            // a point per lane: the loads of four groups are issued together
            constexpr int NB = 4;
            for (int e0 = wave; e0 < nl; e0 += FPS_WAVES * NB) {
                float4 q[NB];
                double md[NB];
                int g[NB];
                bool ok[NB];
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const int e = e0 + j * FPS_WAVES;
                    g[j] = e < nl ? (int)s_list[e] : 0;
                    const int p = (g[j] << 6) + lane;
                    ok[j] = e < nl && p < m;
                    q[j] = ok[j] ? S[p] : make_float4(0.f, 0.f, 0.f, 0.f);
                    md[j] = ok[j] ? M2[p] : -1.0;
                }
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    if (e0 + j * FPS_WAVES >= nl)
                        break;
                    const double dx = (double)q[j].x - cx, dy = (double)q[j].y - cy, dz = (double)q[j].z - cz;
                    const double d2 = (dx * dx + dy * dy) + dz * dz;
                    if (ok[j] && d2 < md[j]) {
                        md[j] = d2;
                        M2[(g[j] << 6) + lane] = d2;
                    }
                    finish(g[j], md[j], -1.0, ok[j] ? (unsigned int)__float_as_int(q[j].w) : ~0u, q[j].x, q[j].y, q[j].z);
                }
            }
        } else {
            for (int e = wave; e < nl; e += FPS_WAVES) {
                const int g = (int)s_list[e];
                double b1 = -1.0, b2 = -1.0;   // the lane's largest and second largest distinct values
                unsigned int bi = ~0u;
                float wx = 0.f, wy = 0.f, wz = 0.f;
                const int pend = ((g + 1) << D.gshift) < m ? ((g + 1) << D.gshift) : m;
                for (int p = (g << D.gshift) + lane; p < pend; p += FX_WAVE) {
                    const float4 q = S[p];
                    double md = M2[p];
                    const double dx = (double)q.x - cx, dy = (double)q.y - cy, dz = (double)q.z - cz;
                    const double d2 = (dx * dx + dy * dy) + dz * dz;
                    if (d2 < md) {
                        md = d2;
                        M2[p] = md;
                    }
                    const unsigned int orig = (unsigned int)__float_as_int(q.w);
                    if (md > b1) {
                        b2 = b1;
                        b1 = md;
                        bi = orig;
                        wx = q.x; wy = q.y; wz = q.z;
                    } else if (md == b1) {
                        if (orig < bi) {
                            bi = orig;
                            wx = q.x; wy = q.y; wz = q.z;
                        }
                    } else if (md > b2) {
                        b2 = md;
                    }
                }
                finish(g, b1, b2, bi, wx, wy, wz);
            }
        }
        __syncthreads();
        if (t == 0)
            s_cnt[0] = 0u;
        // largest distance over the groups
        double v[FPS_MAXG / FPS_THREADS];
        double mine = -1.0;
#pragma unroll
        for (int k = 0; k < FPS_MAXG / FPS_THREADS; k++) {
            const int g = t + k * FPS_THREADS;
            v[k] = g < ng ? s_gmax[g] : -1.0;
            mine = fmax(mine, v[k]);
        }
        mine = fx_wave_max_f64_dpp(mine);
        if (lane == 0)
            s_red[wave] = mine;
        __syncthreads();
        double top = s_red[0];
#pragma unroll
        for (int w = 1; w < FPS_WAVES; w++)
            top = fmax(top, s_red[w]);
        const double thr = top - top * 0x1p-49;
#pragma unroll
        for (int k = 0; k < FPS_MAXG / FPS_THREADS; k++)
            if (v[k] >= thr)
                s_list[atomicAdd(&s_cnt[1], 1u)] = (unsigned int)(t + k * FPS_THREADS);
        __syncthreads();
        const int nn = (int)s_cnt[1];
        const unsigned int only = nn == 1 ? s_gidx[s_list[0]] : 0x80000000u;
        if (!(only >> 31)) {
            // one candidate group and no second candidate inside it
            if (t == 0) {
                const int g = (int)s_list[0];
                s_c[0] = (double)s_gxyz[0 * G + g];
                s_c[1] = (double)s_gxyz[1 * G + g];
                s_c[2] = (double)s_gxyz[2 * G + g];
                s_cidx = only;
            }
        } else {
            // first index among the points whose distance (the square root) equals the largest one
            const double root = sqrt(top);
            unsigned long long key = ~0ull;
            for (int e = wave; e < nn; e += FPS_WAVES) {
                const int g = (int)s_list[e];
                const int pend = ((g + 1) << D.gshift) < m ? ((g + 1) << D.gshift) : m;
                for (int p = (g << D.gshift) + lane; p < pend; p += FX_WAVE) {
                    const double md = M2[p];
                    if (md >= thr && sqrt(md) == root) {
                        const unsigned long long k2 =
                            ((unsigned long long)(unsigned int)__float_as_int(S[p].w) << 32) | (unsigned int)p;
                        key = k2 < key ? k2 : key;
                    }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long w = (unsigned long long)__shfl_xor((long long)key, o, FX_WAVE);
                key = w < key ? w : key;
            }
            if (lane == 0 && key != ~0ull)
                atomicMin(&s_win, key);
            __syncthreads();
            if (t == 0 && s_win != ~0ull) {
                const float4 q = S[(unsigned int)s_win];
                s_c[0] = (double)q.x;
                s_c[1] = (double)q.y;
                s_c[2] = (double)q.z;
                s_cidx = (unsigned int)(s_win >> 32);
            }
        }
        __syncthreads();
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
    {
        // the offsets go through the stream's page-locked block, so that nothing has to wait for the copy
        void *pin = guard.staging((size_t)(B + 1) * sizeof(int64_t));
        if (!pin)
            return FX_ENOMEM;
        memcpy(pin, h_offsets, (size_t)(B + 1) * sizeof(int64_t));
        FX_HIP_CHECK(hipMemcpyAsync(base, pin, (size_t)(B + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream));
        if (guard.staged() != FX_OK)
            return FX_EHIP;
    }
    FxProfileScope prof(stream);
    FX_REQUIRE(B < (1ll << 31), "fx_center_scale: B = %lld", (long long)B);
    hipLaunchKernelGGL(cloud_colsum_kernel, dim3((unsigned)B), dim3(FX_BLOCK), 0, stream, A);
    if (total > 0)
        hipLaunchKernelGGL(cloud_maxnorm_kernel, dim3((unsigned)((total + FX_BLOCK - 1) / FX_BLOCK)), dim3(FX_BLOCK),
                           0, stream, A);
    const int64_t span = total > B * 3 ? total : B * 3;
    hipLaunchKernelGGL(cloud_scale_kernel, dim3((unsigned)((span + FX_BLOCK - 1) / FX_BLOCK)), dim3(FX_BLOCK), 0,
                       stream, A);
    FX_HIP_CHECK(hipGetLastError());
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
    // size classes: 1, 2, 4 points per thread in registers (up to 4096 points), then the cell-sorted kernel
    std::vector<int32_t> cls[6];
    for (int64_t b = 0; b < B; b++) {
        const int64_t m = h_offsets[b + 1] - h_offsets[b];
        FX_REQUIRE(m > n_samples && m < (1ll << 31) / 3,
                   "fx_fps_f32: cloud %lld has %lld points; sampling %d needs more than that many (and fewer than 2^31 / 3)",
                   (long long)b, (long long)m, n_samples);
        FX_REQUIRE(h_start[b] >= 0 && h_start[b] < m, "fx_fps_f32: start index %lld of cloud %lld outside 0..%lld",
                   (long long)h_start[b], (long long)b, (long long)m - 1);
        int c = 0;
        while (c < 2 && m > (int64_t)FPS_THREADS << c)
            c++;
        if (m > FPS_PRUNE_ABOVE)
            c = 5;
        cls[c].push_back((int32_t)b);
    }
    // clouds beyond the register classes: cell-sorted copy, squared distances, grid counters
    std::vector<FpsBigDesc> desc(cls[5].size());
    int64_t big_points = 0, big_cells = 0;
    size_t prep_lds = 0;
    int maxg = 64;
    for (size_t k = 0; k < cls[5].size(); k++) {
        const int64_t m = h_offsets[cls[5][k] + 1] - h_offsets[cls[5][k]];
        FpsBigDesc &d = desc[k];
        d.soff = big_points;
        d.coff = big_cells;
        d.gb = m < (1 << 15) ? 4 : (m < (1 << 19) ? 5 : 6);
        d.gshift = 6;
        while (((m + (1ll << d.gshift) - 1) >> d.gshift) > FPS_MAXG)
            d.gshift++;
        const int ng = (int)((m + (1ll << d.gshift) - 1) >> d.gshift);
        maxg = ng > maxg ? ng : maxg;
        big_points += (m + 15) & ~15ll;
        if (d.gb > 5)
            big_cells += 1ll << (3 * d.gb);
        else
            prep_lds = std::max(prep_lds, sizeof(unsigned int) << (3 * d.gb));
    }
    maxg = (maxg + 63) & ~63;
    const size_t off_bytes = fps_align((size_t)(B + 1) * sizeof(int64_t));
    const size_t start_bytes = fps_align((size_t)B * sizeof(int64_t));
    const size_t ids_bytes = fps_align((size_t)B * sizeof(int32_t));
    const size_t desc_bytes = fps_align(desc.size() * sizeof(FpsBigDesc));
    const size_t sorted_bytes = fps_align((size_t)big_points * sizeof(float4));
    const size_t mind_bytes = fps_align((size_t)big_points * sizeof(double));
    const size_t cell_bytes = fps_align((size_t)big_cells * sizeof(unsigned int));
    FxCallGuard guard((hipStream_t)stream);
    unsigned char *base = (unsigned char *)guard.scratch(
        off_bytes + start_bytes + ids_bytes + desc_bytes + sorted_bytes + mind_bytes + cell_bytes, 5);
    if (!base)
        return FX_ENOMEM;
    std::vector<int32_t> ids;
    for (int c = 0; c < 6; c++)
        ids.insert(ids.end(), cls[c].begin(), cls[c].end());
    unsigned char *big = base + off_bytes + start_bytes + ids_bytes;
    {
        // offsets | start indices | cloud list | descriptors of the large clouds, laid out as in the scratch block,
        // through the stream's page-locked block: one copy, and nothing waits for it
        const size_t head = off_bytes + start_bytes + ids_bytes + desc_bytes;
        unsigned char *pin = (unsigned char *)guard.staging(head);
        if (!pin)
            return FX_ENOMEM;
        memcpy(pin, h_offsets, (size_t)(B + 1) * sizeof(int64_t));
        memcpy(pin + off_bytes, h_start, (size_t)B * sizeof(int64_t));
        memcpy(pin + off_bytes + start_bytes, ids.data(), (size_t)B * sizeof(int32_t));
        if (!desc.empty())
            memcpy(pin + off_bytes + start_bytes + ids_bytes, desc.data(), desc.size() * sizeof(FpsBigDesc));
        FX_HIP_CHECK(hipMemcpyAsync(base, pin, head, hipMemcpyHostToDevice, stream));
        if (guard.staged() != FX_OK)
            return FX_EHIP;
    }
    FpsArgs A;
    A.pts = d_pts;
    A.offsets = (const int64_t *)base;
    A.start = (const int64_t *)(base + off_bytes);
    A.n_samples = n_samples;
    A.idx = d_idx;
    A.sampled = d_sampled;
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
        default: {
            FpsBigArgs G;
            G.pts = d_pts;
            G.offsets = A.offsets;
            G.ids = A.ids;
            G.start = A.start;
            G.n_samples = n_samples;
            G.idx = d_idx;
            G.sampled = d_sampled;
            G.desc = (const FpsBigDesc *)big;
            G.sorted = (float4 *)(big + desc_bytes);
            G.mind2 = (double *)(big + desc_bytes + sorted_bytes);
            G.cells = (unsigned int *)(big + desc_bytes + sorted_bytes + mind_bytes);
            G.maxg = maxg;
            const size_t lds = std::max((size_t)maxg * FPS_BIG_LDS_PER_GROUP, prep_lds);
            if (lds > 48 * 1024)
                FX_HIP_CHECK(hipFuncSetAttribute((const void *)fps_pruned_kernel,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(fps_pruned_kernel, grid, block, lds, stream, G);
            break;
        }
        }
    }
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

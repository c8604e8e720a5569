// Multi-view renderer of point clouds: the reference's renderer without
// Open3D / pyrender (src/preprocessing/render_views.py:294-366,
// render_fallback_matplotlib), i.e. what matplotlib's Agg backend draws for
//     ax.scatter(xs, ys, s=point_size, marker="o", linewidths=0)
// on a resolution x resolution canvas with equal aspect and limits
// [min - 1, max + 1]:
//   * a point lands on the pixel (floor(px + 0.5), floor(res - py + 0.5)) of
//     its display coordinates (Agg snaps marker positions to whole pixels);
//   * every marker is the same small stamp of coverage values (rasterised once
//     by Agg; the table of stamps per point size is host data);
//   * markers are blended one after the other, in point order, with Agg's
//     integer "plain" RGBA blend onto the opaque white figure, so a pixel's
//     value depends on the order of the markers that touch it.
// One block per (cloud, view): the canvas lives in LDS, the block projects and
// snaps the points in parallel, and every wave applies, in point order, the
// stamps that fall into its band of canvas rows, a lane per stamp cell (LDS
// operations of a wave complete in order).
#include "fx_common.h"

namespace {

constexpr int RENDER_MAX_K = 7;        // stamps up to 7 x 7 cells (49 lanes of the stamping wave)
constexpr int RENDER_CHUNK = 2048;     // snapped positions kept in LDS at a time
constexpr int RENDER_MAX_RES = 4096;
constexpr int RENDER_CANVAS_LDS = 144 * 1024;   // canvas bytes a block keeps in LDS: views above 384 x 384 are cut into bands of rows
#ifndef RENDER_WAVES_N
#define RENDER_WAVES_N 8
#endif
constexpr int RENDER_WAVES = RENDER_WAVES_N;  // waves per block (a power of two): the stamping runs RENDER_WAVES-fold parallel
constexpr int RENDER_THREADS = RENDER_WAVES * FX_WAVE;

struct RenderArgs {
    const float *pts;         // [total][3]
    const int64_t *offsets;   // [B + 1], device
    int64_t B;
    int V;
    const double *cam;        // [B][V][9]: right, true_up, look-at
    const double *scale;      // [B]: (resolution * 0.35) / bbox diagonal
    int res;
    int K;
    int marker;               // marker gray value: 0 on a white background, 255 on a black one
    unsigned char stamp[RENDER_MAX_K * RENDER_MAX_K];
    uint8_t *out;             // [B][V][res][res]
    int band_rows;            // canvas rows per block (a multiple of 8, or res)
    int nbands;               // blocks per view
};

__device__ __forceinline__ double block_minmax(double v, bool is_max, double *red)
{
    v = is_max ? fx_wave_max_f64_dpp(v) : -fx_wave_max_f64_dpp(-v);
    __syncthreads();
    if (fx_lane() == 0)
        red[fx_wave()] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int w = 1; w < RENDER_WAVES; w++)
        r = is_max ? fmax(r, red[w]) : fmin(r, red[w]);
    return r;
}

// Agg's fixed_blender_rgba_plain::blend_pix for one channel of an opaque
// pixel p, source value c, coverage-derived alpha a in 1..255:
//   (((c << 8) - 255 p) a + ((255 p) << 8)) / (((a + 255) << 8) - 255 a)
__device__ __forceinline__ uint32_t agg_blend(uint32_t p, uint32_t c, uint32_t a, uint32_t den, float inv_den)
{
    const uint32_t r = p * 255u;
    const uint32_t num = ((c << 8) - r) * a + (r << 8);  // unsigned wrap-around cancels, the true value is below 2^24
    uint32_t q = (uint32_t)((float)num * inv_den);
    int32_t rem = (int32_t)(num - q * den);
    if (rem < 0)
        q -= 1;
    else if (rem >= (int32_t)den)
        q += 1;
    return q;
}

__global__ __launch_bounds__(RENDER_THREADS) void render_views_kernel(RenderArgs A)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double s_red[RENDER_WAVES];
    __shared__ uint32_t s_snap[RENDER_CHUNK];
    __shared__ uint32_t s_list[RENDER_WAVES][FX_WAVE];   // per wave: its points of the current batch
    uint8_t *canvas = lds_raw;
    const int t = threadIdx.x;
    const int64_t bv = blockIdx.x / A.nbands;
    const int band = (int)(blockIdx.x - bv * A.nbands);
    const int64_t b = bv / A.V;
    const int res = A.res;
    // the block's band of canvas rows [r0, r1): bands are disjoint, so each keeps the point order on its own
    const int r0 = band * A.band_rows, r1 = r0 + A.band_rows < res ? r0 + A.band_rows : res;
    const int npx = (r1 - r0) * res;
    const int64_t lo = A.offsets[b], n = A.offsets[b + 1] - lo;
    const float *pts = A.pts + lo * 3;
    const double *cam = A.cam + bv * 9;
    const double rx = cam[0], ry = cam[1], rz = cam[2], ux = cam[3], uy = cam[4], uz = cam[5];
    const double cx = cam[6], cy = cam[7], cz = cam[8];
    const double scale = A.scale[b];

    for (int i = t * 4; i < npx; i += RENDER_THREADS * 4) {
        if (i + 4 <= npx)
            *reinterpret_cast<uint32_t *>(canvas + i) = 0xFFFFFFFFu;
        else
            for (int k = i; k < npx; k++)
                canvas[k] = 255;
    }

    // display-plane coordinates of a point, as the reference forms them
    auto plane = [&](int64_t i, double &X, double &Y) {
        const double dx = (double)pts[i * 3] - cx, dy = (double)pts[i * 3 + 1] - cy, dz = (double)pts[i * 3 + 2] - cz;
        X = ((dx * rx + dy * ry) + dz * rz) * scale;
        Y = ((dx * ux + dy * uy) + dz * uz) * scale;
    };
    const double inf = __builtin_inf();
    double xmin = inf, xmax = -inf, ymin = inf, ymax = -inf;
    for (int64_t i = t; i < n; i += RENDER_THREADS) {
        double X, Y;
        plane(i, X, Y);
        xmin = fmin(xmin, X), xmax = fmax(xmax, X);
        ymin = fmin(ymin, Y), ymax = fmax(ymax, Y);
    }
    xmin = block_minmax(xmin, false, s_red);
    xmax = block_minmax(xmax, true, s_red);
    ymin = block_minmax(ymin, false, s_red);
    ymax = block_minmax(ymax, true, s_red);
    // axis limits, equal aspect inside the square figure, box centred
    const double x0 = xmin - 1.0, x1 = xmax + 1.0, y0 = ymin - 1.0, y1 = ymax + 1.0;
    const double W = x1 - x0, H = y1 - y0;
    const double dres = (double)res;
    const double sx = dres / W, sy = dres / H;
    const double s = sx < sy ? sx : sy;
    const double offx = (dres - W * s) / 2.0, offy = (dres - H * s) / 2.0;

    // Stamping: the canvas rows are dealt to the waves in stripes of 8 rows
    // (stripe s to wave s % RENDER_WAVES: an object in the middle of the view loads all
    // waves alike); a wave walks the points in order and applies the stamp
    // cells that fall into its stripes, a lane per cell.  A pixel belongs to one
    // wave, and the LDS operations of a wave complete in order, so the
    // read-modify-write sequences on a pixel keep the point order without any
    // barrier.  Markers that share no pixel commute, so a wave takes as many
    // consecutive points of its own as its lanes hold stamps (seven 3 x 3 stamps)
    // in one step whenever none of them overlaps an earlier one of the step, and
    // one after the other otherwise.
    const int K = A.K, half = K / 2, KK = K * K;
    const int lane = fx_lane(), wave = fx_wave();
    // stamps per step: as many as the lanes hold, but no more than KK + 1 -- lane (grp, cell) checks stamp grp
    // against the earlier stamp number `cell`, and cell only runs to KK - 1 (a 1 x 1 marker: two per step)
    const int G = FX_WAVE / KK < KK + 1 ? FX_WAVE / KK : KK + 1;
    const int grp = lane / KK, cell = lane - grp * KK;
    const bool lane_used = grp < G;
    const int cell_dy = cell / K - half, cell_dx = cell % K - half;
    const uint32_t alpha = lane_used ? A.stamp[cell] : 0u;
    const uint32_t den = ((alpha + 255u) << 8) - alpha * 255u;
    const float inv_den = 1.0f / (float)den;
    const uint32_t marker = (uint32_t)A.marker;
    auto owner = [](int row) -> int { return (row >> 3) & (RENDER_WAVES - 1); };  // of rows >= -1024
    uint32_t *my_list = s_list[wave];
    auto stamp_cell = [&](uint32_t rc) {
        const int row = (int)(rc >> 16) - 1024 + cell_dy;
        const int col = (int)(rc & 0xFFFFu) - 1024 + cell_dx;
        if (alpha != 0u && row >= r0 && row < r1 && owner(row) == wave && col >= 0 && col < res) {
            const int at = (row - r0) * res + col;
            const uint32_t p = canvas[at];
            canvas[at] = (uint8_t)(alpha == 255u ? marker : agg_blend(p, marker, alpha, den, inv_den));
        }
    };

    for (int64_t c0 = 0; c0 < n; c0 += RENDER_CHUNK) {
        const int m = (int)(n - c0 < RENDER_CHUNK ? n - c0 : RENDER_CHUNK);
        __syncthreads();
        for (int i = t; i < m; i += RENDER_THREADS) {
            double X, Y;
            plane(c0 + i, X, Y);
            const double px = offx + (X - x0) * s;
            const double py = offy + (Y - y0) * s;
            const int col = (int)floor(px + 0.5);
            const int row = (int)floor((dres - py) + 0.5);
            // stored with a bias so that positions a little outside the canvas stay representable
            s_snap[i] = ((uint32_t)(row + 1024) << 16) | (uint32_t)(col + 1024);
        }
        __syncthreads();
        for (int base = 0; base < m; base += FX_WAVE) {
            const int i = base + lane;
            const uint32_t mine = i < m ? s_snap[i] : 0u;
            const int my_row = (int)(mine >> 16) - 1024;
            // a stamp of at most 7 rows touches one or two stripes
            const bool take = i < m && my_row + half >= r0 && my_row - half < r1 &&
                              (owner(my_row - half + 1024) == wave || owner(my_row + half + 1024) == wave);
            unsigned long long todo = __ballot(take);
            if (G < 2) {
                while (todo) {
                    const int j = __ffsll((long long)todo) - 1;
                    todo &= todo - 1;
                    stamp_cell((uint32_t)__builtin_amdgcn_readlane((int)mine, j));
                }
                continue;
            }
            // the wave's points of this batch, in order
            const int cnt = __popcll(todo);
            if (take)
                my_list[__builtin_amdgcn_mbcnt_hi((unsigned int)(todo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)todo, 0u))] = mine;
            // the list is written by the lanes that hold a point and read by others of the same wave
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int g0 = 0; g0 < cnt; g0 += G) {
                const bool live = lane_used && g0 + grp < cnt;
                const uint32_t rc = live ? my_list[g0 + grp] : 0u;
                // lane (grp, cell) compares stamp grp with the earlier stamp number `cell` of the step
                const uint32_t other = (uint32_t)__shfl((int)rc, cell * KK, FX_WAVE);
                const int dr = (int)(rc >> 16) - (int)(other >> 16), dc = (int)(rc & 0xFFFFu) - (int)(other & 0xFFFFu);
                const bool clash = live && cell < grp && (unsigned)(dr + K - 1) < (unsigned)(2 * K - 1) &&
                                   (unsigned)(dc + K - 1) < (unsigned)(2 * K - 1);
                if (!__ballot(clash)) {
                    if (live)
                        stamp_cell(rc);
                } else {
                    for (int e = 0; e < G && g0 + e < cnt; e++)
                        if (live && grp == e)
                            stamp_cell(rc);
                }
            }
        }
    }
    __syncthreads();
    uint8_t *out = A.out + bv * (int64_t)res * res + (int64_t)r0 * res;
    if ((npx & 15) == 0 && ((uintptr_t)out & 15) == 0) {
        for (int i = t * 16; i < npx; i += RENDER_THREADS * 16)
            *reinterpret_cast<uint4 *>(out + i) = *reinterpret_cast<const uint4 *>(canvas + i);
    } else {
        for (int i = t; i < npx; i += RENDER_THREADS)
            out[i] = canvas[i];
    }
}

// Per cloud: minimum and maximum of every coordinate and the mean as
// np.mean(axis=0) forms it (rows added one after the other in float64, then
// one division).  One wave per cloud.
__global__ __launch_bounds__(FX_BLOCK) void cloud_bounds_kernel(const float *pts_all, const int64_t *offsets, int64_t B,
                                                              double *bounds)
{
    const int lane = fx_lane();
    const int64_t b = (int64_t)blockIdx.x * FX_WAVES_PER_BLOCK + fx_wave();
    if (b >= B)
        return;
    const int64_t lo = offsets[b], n = offsets[b + 1] - lo;
    const float *pts = pts_all + lo * 3;
    const float inf = __builtin_inff();
    float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
    for (int64_t i = lane; i < n; i += FX_WAVE) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float v = pts[i * 3 + c];
            mn[c] = fminf(mn[c], v);
            mx[c] = fmaxf(mx[c], v);
        }
    }
    double *o = bounds + b * 9;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float lo_c = -fx_wave_max_f32_dpp(-mn[c]), hi_c = fx_wave_max_f32_dpp(mx[c]);
        if (lane == 0) {
            o[c] = (double)lo_c;
            o[3 + c] = (double)hi_c;
        }
    }
    if (lane < 3) {
        double sum = 0.0;
        int64_t i = 0;
        for (; i + 8 <= n; i += 8) {
            double w[8];
#pragma unroll
            for (int j = 0; j < 8; j++)
                w[j] = (double)pts[(i + j) * 3 + lane];
#pragma unroll
            for (int j = 0; j < 8; j++)
                sum += w[j];
        }
        for (; i < n; i++)
            sum += (double)pts[i * 3 + lane];
        o[6 + lane] = sum / (double)n;
    }
}

constexpr int RENDER_MAX_VIEWS = 64;

struct CamArgs {
    const double *bounds;  // [B][9]: min, max, mean
    int64_t B;
    int V;
    double radius_factor;
    double res_scale;      // resolution * 0.35
    double trig[RENDER_MAX_VIEWS][4];  // per view: cos(elevation), sin(elevation), cos(azimuth), sin(azimuth)
    double *cam;           // [B][V][9]
    double *scale;         // [B]
};

struct Vec3 {
    double x, y, z;
};
__device__ __forceinline__ double norm3(const Vec3 &v) { return sqrt((v.x * v.x + v.y * v.y) + v.z * v.z); }
__device__ __forceinline__ Vec3 cross3(const Vec3 &a, const Vec3 &b)
{
    return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Camera poses (reference compute_camera_poses :93-164) and the in-plane basis
// of the renderer (:306-320) for every (cloud, view), one thread each, in the
// float64 operation order of the host mirror (render_views.camera_table).
__global__ __launch_bounds__(FX_BLOCK) void camera_table_kernel(CamArgs A)
{
    const int64_t i = (int64_t)blockIdx.x * FX_BLOCK + threadIdx.x;
    if (i >= A.B * A.V)
        return;
    const int64_t b = i / A.V;
    const int v = (int)(i - b * A.V);
    const double *bd = A.bounds + b * 9;
    const Vec3 ext{bd[3] - bd[0], bd[4] - bd[1], bd[5] - bd[2]};
    const Vec3 cen{bd[6], bd[7], bd[8]};
    const double diag = norm3(ext);
    const double radius = diag > 1e-9 ? diag * A.radius_factor : A.radius_factor;
    const double cel = A.trig[v][0], sel = A.trig[v][1], caz = A.trig[v][2], saz = A.trig[v][3];
    const Vec3 pos{(radius * cel) * caz + cen.x, (radius * cel) * saz + cen.y, radius * sel + cen.z};
    Vec3 fwd{cen.x - pos.x, cen.y - pos.y, cen.z - pos.z};
    if (norm3(fwd) < 1e-9)
        fwd = Vec3{0.0, 0.0, 1.0};
    {
        const double n = norm3(fwd);
        fwd = Vec3{fwd.x / n, fwd.y / n, fwd.z / n};
    }
    const Vec3 up = fabs(fwd.z) > 0.999 ? Vec3{0.0, 1.0, 0.0} : Vec3{0.0, 0.0, 1.0};
    // the renderer's basis from (position, look-at, up)
    Vec3 f{cen.x - pos.x, cen.y - pos.y, cen.z - pos.z};
    if (norm3(f) < 1e-12)
        f = Vec3{0.0, 0.0, 1.0};
    {
        const double n = norm3(f);
        f = Vec3{f.x / n, f.y / n, f.z / n};
    }
    Vec3 right = cross3(f, up);
    if (norm3(right) < 1e-9)
        right = cross3(f, Vec3{0.0, 1.0, 0.0});
    {
        const double n = norm3(right) + 1e-12;
        right = Vec3{right.x / n, right.y / n, right.z / n};
    }
    Vec3 tup = cross3(right, f);
    {
        const double n = norm3(tup) + 1e-12;
        tup = Vec3{tup.x / n, tup.y / n, tup.z / n};
    }
    double *o = A.cam + i * 9;
    o[0] = right.x, o[1] = right.y, o[2] = right.z;
    o[3] = tup.x, o[4] = tup.y, o[5] = tup.z;
    o[6] = cen.x, o[7] = cen.y, o[8] = cen.z;
    if (v == 0)
        A.scale[b] = A.res_scale / (diag < 1e-9 ? 1.0 : diag);
}

}  // namespace

extern "C" int fx_camera_table_f64(const double *d_bounds, int64_t B, int V, const double *h_trig, double radius_factor,
                                   int resolution, double *d_cam, double *d_scale, void *stream)
{
    FX_REQUIRE(B >= 0 && V >= 1 && V <= RENDER_MAX_VIEWS, "fx_camera_table_f64: %lld clouds x %d views (at most %d views)",
               (long long)B, V, RENDER_MAX_VIEWS);
    if (B == 0)
        return FX_OK;
    FX_REQUIRE(d_bounds && h_trig && d_cam && d_scale, "fx_camera_table_f64: NULL argument");
    CamArgs A;
    A.bounds = d_bounds;
    A.B = B;
    A.V = V;
    A.radius_factor = radius_factor;
    A.res_scale = (double)resolution * 0.35;
    for (int v = 0; v < RENDER_MAX_VIEWS; v++)
        for (int k = 0; k < 4; k++)
            A.trig[v][k] = v < V ? h_trig[v * 4 + k] : 0.0;
    A.cam = d_cam;
    A.scale = d_scale;
    const int64_t total = B * V;
    hipLaunchKernelGGL(camera_table_kernel, dim3((unsigned)((total + FX_BLOCK - 1) / FX_BLOCK)), dim3(FX_BLOCK), 0, (hipStream_t)stream, A);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

extern "C" int fx_cloud_bounds_f32(const float *d_pts, const int64_t *d_offsets, int64_t B, double *d_bounds, void *stream)
{
    FX_REQUIRE(B >= 0, "fx_cloud_bounds_f32: B = %lld", (long long)B);
    if (B == 0)
        return FX_OK;
    FX_REQUIRE(d_pts && d_offsets && d_bounds, "fx_cloud_bounds_f32: NULL argument");
    hipLaunchKernelGGL(cloud_bounds_kernel, dim3((unsigned)((B + FX_WAVES_PER_BLOCK - 1) / FX_WAVES_PER_BLOCK)), dim3(FX_BLOCK), 0,
                       (hipStream_t)stream, d_pts, d_offsets, B, d_bounds);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

extern "C" int fx_render_views_u8(const float *d_pts, const int64_t *d_offsets, int64_t B, int V, const double *d_cam,
                                  const double *d_scale, int resolution, const uint8_t *stamp, int K, int marker_value,
                                  uint8_t *d_out, void *stream)
{
    FX_REQUIRE(B >= 0 && V >= 1, "fx_render_views_u8: %lld clouds x %d views", (long long)B, V);
    if (B == 0)
        return FX_OK;
    FX_REQUIRE(d_pts && d_offsets && d_cam && d_scale && stamp && d_out, "fx_render_views_u8: NULL argument");
    FX_REQUIRE(resolution >= 8 && resolution <= RENDER_MAX_RES, "fx_render_views_u8: resolution %d outside 8..%d", resolution,
               RENDER_MAX_RES);
    FX_REQUIRE(K >= 1 && K <= RENDER_MAX_K && (K & 1), "fx_render_views_u8: stamp side %d (odd, at most %d)", K, RENDER_MAX_K);
    FX_REQUIRE(marker_value == 0 || marker_value == 255, "fx_render_views_u8: marker value %d (0 or 255)", marker_value);
    FX_REQUIRE(B * V <= 0x7fffffffLL, "fx_render_views_u8: too many views for one launch");
    RenderArgs A;
    A.pts = d_pts;
    A.offsets = d_offsets;
    A.B = B;
    A.V = V;
    A.cam = d_cam;
    A.scale = d_scale;
    A.res = resolution;
    A.K = K;
    A.marker = marker_value;
    for (int i = 0; i < RENDER_MAX_K * RENDER_MAX_K; i++)
        A.stamp[i] = i < K * K ? stamp[i] : 0;
    A.out = d_out;
    A.band_rows = resolution;
    if ((size_t)resolution * resolution > (size_t)RENDER_CANVAS_LDS)
        A.band_rows = (RENDER_CANVAS_LDS / resolution) & ~7;   // whole stripes of eight rows
    A.nbands = (resolution + A.band_rows - 1) / A.band_rows;
    FX_REQUIRE(B * V * A.nbands <= 0x7fffffffLL, "fx_render_views_u8: too many views for one launch");
    const size_t lds = ((size_t)A.band_rows * resolution + 15) & ~(size_t)15;
    if (lds > 48 * 1024)
        FX_HIP_CHECK(hipFuncSetAttribute((const void *)render_views_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    FxProfileScope prof((hipStream_t)stream);
    hipLaunchKernelGGL(render_views_kernel, dim3((unsigned)(B * V * A.nbands)), dim3(RENDER_THREADS), lds, (hipStream_t)stream, A);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

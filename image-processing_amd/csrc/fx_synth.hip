// Synthetic inputs: binary shape images and Gaussian point clouds, generated
// on the device from a counter-based hash so that any shard of a data set can
// be produced on any rank.  The shape rasteriser is integer-only; the host
// twin in the Python package (synth.py) produces identical bytes.
#include "fx_common.h"

namespace {

__host__ __device__ __forceinline__ u64 mix64(u64 z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Hash of (seed, item index, slot).
__host__ __device__ __forceinline__ u64 draw(u64 seed, u64 index, u64 slot)
{
    return mix64(mix64(seed ^ (index * 0xD1342543DE82EF95ull)) + slot * 0x9E3779B97F4A7C15ull);
}

// Uniform integer in [lo, lo + span) from the high bits of a draw.
__host__ __device__ __forceinline__ i64 draw_range(u64 h, i64 lo, i64 span) { return lo + (i64)((h >> 20) % (u64)span); }

constexpr int MAX_ELL = 6;
constexpr int MAX_VERT = 8;

// 16 directions, round(1024 * (cos, sin)(2 pi k / 16))
__constant__ int kDirX[16] = {1024, 946, 724, 392, 0, -392, -724, -946, -1024, -946, -724, -392, 0, 392, 724, 946};
__constant__ int kDirY[16] = {0, 392, 724, 946, 1024, 946, 724, 392, 0, -392, -724, -946, -1024, -946, -724, -392};

struct Shape {
    int n_ell;
    i64 ecx[MAX_ELL], ecy[MAX_ELL], ea2[MAX_ELL], eb2[MAX_ELL], eux[MAX_ELL], euy[MAX_ELL], erad[MAX_ELL];
    int n_vert;
    i64 vx[MAX_VERT], vy[MAX_VERT];
    i64 pminx, pmaxx, pminy, pmaxy;
};

__device__ void make_shape(Shape &S, u64 seed, u64 index, int H, int W)
{
    const i64 m = W < H ? W : H;
    S.n_ell = 3 + (int)((draw(seed, index, 0) >> 20) % 4);
    for (int e = 0; e < MAX_ELL; e++) {
        const u64 base = 16 + 8 * e;
        S.ecx[e] = draw_range(draw(seed, index, base + 0), W / 4, W / 2 > 0 ? W / 2 : 1);
        S.ecy[e] = draw_range(draw(seed, index, base + 1), H / 4, H / 2 > 0 ? H / 2 : 1);
        const i64 lo = m / 25 > 1 ? m / 25 : 1, span = m / 8 > 1 ? m / 8 : 1;
        const i64 a = draw_range(draw(seed, index, base + 2), lo, span);
        const i64 b = draw_range(draw(seed, index, base + 3), lo, span);
        S.ea2[e] = a * a;
        S.eb2[e] = b * b;
        S.eux[e] = draw_range(draw(seed, index, base + 4), 1, 16);
        S.euy[e] = draw_range(draw(seed, index, base + 5), -16, 33);
        S.erad[e] = a > b ? a : b;
    }
    S.n_vert = 5 + (int)((draw(seed, index, 1) >> 20) % 4);
    const i64 pcx = draw_range(draw(seed, index, 2), (3 * (i64)W) / 10, (2 * (i64)W) / 5 > 0 ? (2 * (i64)W) / 5 : 1);
    const i64 pcy = draw_range(draw(seed, index, 3), (3 * (i64)H) / 10, (2 * (i64)H) / 5 > 0 ? (2 * (i64)H) / 5 : 1);
    S.pminx = S.pminy = (i64)1 << 40;
    S.pmaxx = S.pmaxy = -((i64)1 << 40);
    for (int k = 0; k < MAX_VERT; k++) {
        if (k < S.n_vert) {
            const int dir = (k * 16) / S.n_vert;
            const i64 lo = m / 20 > 1 ? m / 20 : 1, span = m / 6 > 1 ? m / 6 : 1;
            const i64 r = draw_range(draw(seed, index, 80 + k), lo, span);
            S.vx[k] = pcx + ((kDirX[dir] * r) >> 10);
            S.vy[k] = pcy + ((kDirY[dir] * r) >> 10);
            S.pminx = S.vx[k] < S.pminx ? S.vx[k] : S.pminx;
            S.pmaxx = S.vx[k] > S.pmaxx ? S.vx[k] : S.pmaxx;
            S.pminy = S.vy[k] < S.pminy ? S.vy[k] : S.pminy;
            S.pmaxy = S.vy[k] > S.pmaxy ? S.vy[k] : S.pmaxy;
        } else {
            S.vx[k] = S.vy[k] = 0;
        }
    }
}

__device__ __forceinline__ bool inside_shape(const Shape &S, i64 x, i64 y)
{
    for (int e = 0; e < S.n_ell; e++) {
        const i64 dx = x - S.ecx[e], dy = y - S.ecy[e];
        const i64 adx = dx < 0 ? -dx : dx, ady = dy < 0 ? -dy : dy;
        if (adx > S.erad[e] || ady > S.erad[e])
            continue;
        const i64 du = S.eux[e] * dx + S.euy[e] * dy;
        const i64 dv = -S.euy[e] * dx + S.eux[e] * dy;
        const i64 n2 = S.eux[e] * S.eux[e] + S.euy[e] * S.euy[e];
        if (du * du * S.eb2[e] + dv * dv * S.ea2[e] <= S.ea2[e] * S.eb2[e] * n2)
            return true;
    }
    if (x < S.pminx || x > S.pmaxx || y < S.pminy || y > S.pmaxy)
        return false;
    bool in = false;
    for (int k = 0; k < S.n_vert; k++) {
        const int k2 = k + 1 == S.n_vert ? 0 : k + 1;
        const i64 x1 = S.vx[k], y1 = S.vy[k], x2 = S.vx[k2], y2 = S.vy[k2];
        if ((y1 > y) != (y2 > y)) {
            const i64 val = (y - y1) * (x2 - x1) - (x - x1) * (y2 - y1);
            if (val != 0 && ((val > 0) == (y2 > y1)))
                in = !in;
        }
    }
    return in;
}

// One block per (image, band of rows); each thread rasterises 4-pixel groups.
__global__ __launch_bounds__(256) void synth_shapes_kernel(uint8_t *img, int64_t B, int H, int W, int64_t pitch,
                                                          int64_t stride, u64 seed, int64_t first_index, int bands)
{
    __shared__ Shape S;
    const int64_t b = blockIdx.x / bands;
    const int band = blockIdx.x % bands;
    if (threadIdx.x == 0)
        make_shape(S, seed, (u64)(first_index + b), H, W);
    __syncthreads();
    const int rows = (H + bands - 1) / bands;
    const int ya = band * rows, yb = (ya + rows) < H ? (ya + rows) : H;
    uint8_t *base = img + b * stride;
    const int gpr = (W + 3) / 4;  // 4-pixel groups per row
    const int64_t total = (int64_t)(yb - ya) * gpr;
    const bool vec = (pitch % 4 == 0) && (stride % 4 == 0) && ((uintptr_t)img % 4 == 0);
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const int y = ya + (int)(i / gpr);
        const int x = (int)(i % gpr) * 4;
        uint32_t word = 0;
        for (int k = 0; k < 4; k++)
            if (x + k < W && inside_shape(S, x + k, y))
                word |= 0xFFu << (8 * k);
        uint8_t *p = base + (int64_t)y * pitch + x;
        if (vec && x + 4 <= W) {
            *reinterpret_cast<uint32_t *>(p) = word;
        } else {
            for (int k = 0; k < 4 && x + k < W; k++)
                p[k] = (uint8_t)(word >> (8 * k));
        }
    }
}

__device__ __forceinline__ float unit_float(u64 h) { return ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f); }

__device__ __forceinline__ void normal_pair(u64 h1, u64 h2, float &a, float &b)
{
    const float r = sqrtf(-2.0f * logf(unit_float(h1)));
    const float phi = 6.283185307179586f * unit_float(h2);
    a = r * cosf(phi);
    b = r * sinf(phi);
}

// One block per set: anisotropic Gaussian cloud, rotated by a random unit
// quaternion.
__global__ __launch_bounds__(256) void synth_points_kernel(float *pts, int64_t B, int64_t N, u64 seed, int64_t first_index)
{
    const int64_t b = blockIdx.x;
    const u64 index = (u64)(first_index + b);
    float q[4], s[3];
    {
        float n0, n1, n2, n3;
        normal_pair(draw(seed, index, 0), draw(seed, index, 1), n0, n1);
        normal_pair(draw(seed, index, 2), draw(seed, index, 3), n2, n3);
        const float inv = rsqrtf(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3 + 1e-30f);
        q[0] = n0 * inv;
        q[1] = n1 * inv;
        q[2] = n2 * inv;
        q[3] = n3 * inv;
        for (int k = 0; k < 3; k++)
            s[k] = 0.05f * expf(unit_float(draw(seed, index, 4 + k)) * 2.995732273553991f);  // ln 20
    }
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float r00 = 1 - 2 * (y * y + z * z), r01 = 2 * (x * y - w * z), r02 = 2 * (x * z + w * y);
    const float r10 = 2 * (x * y + w * z), r11 = 1 - 2 * (x * x + z * z), r12 = 2 * (y * z - w * x);
    const float r20 = 2 * (x * z - w * y), r21 = 2 * (y * z + w * x), r22 = 1 - 2 * (x * x + y * y);
    float *out = pts + b * N * 3;
    for (int64_t i = threadIdx.x; i < N; i += 256) {
        float g0, g1, g2, g3;
        normal_pair(draw(seed, index, 16 + 4 * (u64)i), draw(seed, index, 17 + 4 * (u64)i), g0, g1);
        normal_pair(draw(seed, index, 18 + 4 * (u64)i), draw(seed, index, 19 + 4 * (u64)i), g2, g3);
        g0 *= s[0];
        g1 *= s[1];
        g2 *= s[2];
        out[3 * i + 0] = r00 * g0 + r01 * g1 + r02 * g2;
        out[3 * i + 1] = r10 * g0 + r11 * g1 + r12 * g2;
        out[3 * i + 2] = r20 * g0 + r21 * g1 + r22 * g2;
    }
}

}  // namespace

extern "C" int fx_synth_shapes_u8(uint8_t *d_img, int64_t B, int H, int W, int64_t pitch_bytes,
                                  int64_t img_stride_bytes, uint64_t seed, int64_t first_index, void *stream)
{
    FX_REQUIRE(B >= 0, "fx_synth_shapes_u8: B = %lld", (long long)B);
    if (B == 0)
        return FX_OK;
    FX_REQUIRE(d_img != nullptr, "fx_synth_shapes_u8: d_img is NULL");
    FX_REQUIRE(H >= 8 && W >= 8 && H <= 16384 && W <= 16384, "fx_synth_shapes_u8: image %d x %d outside 8..16384", H, W);
    FX_REQUIRE(pitch_bytes >= W, "fx_synth_shapes_u8: pitch %lld < width %d", (long long)pitch_bytes, W);
    FX_REQUIRE(B == 1 || img_stride_bytes >= pitch_bytes * (int64_t)(H - 1) + W, "fx_synth_shapes_u8: stride too small");
    // enough blocks per image to fill the chip when B is small
    int bands = 1;
    while (bands < 64 && B * bands < 2048 && (H / (bands * 2)) >= 8)
        bands *= 2;
    FX_REQUIRE(B * bands <= 0x7fffffffLL, "fx_synth_shapes_u8: B too large for one launch");
    hipLaunchKernelGGL(synth_shapes_kernel, dim3((unsigned)(B * bands)), dim3(256), 0, (hipStream_t)stream, d_img, B, H,
                       W, pitch_bytes, img_stride_bytes, (u64)seed, first_index, bands);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

extern "C" int fx_synth_points_f32(float *d_pts, int64_t B, int64_t N, uint64_t seed, int64_t first_index, void *stream)
{
    FX_REQUIRE(B >= 0 && N >= 0, "fx_synth_points_f32: B = %lld, N = %lld", (long long)B, (long long)N);
    if (B == 0 || N == 0)
        return FX_OK;
    FX_REQUIRE(d_pts != nullptr, "fx_synth_points_f32: d_pts is NULL");
    FX_REQUIRE(B <= 0x7fffffffLL, "fx_synth_points_f32: B too large for one launch");
    hipLaunchKernelGGL(synth_points_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, d_pts, B, N, (u64)seed,
                       first_index);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

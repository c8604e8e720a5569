// Hu-moment kernels: fused binarisation + ten raw spatial moments in one
// coalesced pass over the image, central/normalised moments, the seven Hu
// invariants, the log transform, and (optionally) the PCA descriptor of the
// contour pixels out of an LDS-resident bit mask.
//
// Replaces the arithmetic of reference src/features/extract_2d_hu.py:443-478
// (cv2.moments / cv2.HuMoments / log transform) and :641-644 (per-sample
// mean/std); the contour descriptor applies
// src/features/extract_3d_pca_features.py:59-145 to contour pixels.
//
// Bandwidth-bound integer reduction: 16 B per lane loads, v_dot4_u32_u8 for
// the in-chunk power sums, 32-bit per-lane partials flushed into 64-bit ones,
// wave shuffles + LDS for the block reduction.  No MFMA.
#include "fx_common.h"
#include <atomic>
#include "fx_eig.h"
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace {

// ------------------------------------------------------------------ geometry
// A block of 256 threads owns one image.  The image is cut into 16-pixel
// chunks; thread t owns the column strip tx = t % cpr and visits rows
// ty, ty + R, ty + 2R, ... with ty = t / cpr and R = 256 / cpr rows per pass,
// so its x offset is fixed and the in-chunk sums can stay in strip-local
// coordinates until the very end.
struct HuArgs {
    const uint8_t *img;
    int64_t B;
    int H, W;
    int64_t pitch, stride;
    int thresh, on_value;
    double *moments24;
    double *hu7;
    float *hulog7;
    int n_bins;
    float *cpca;
    double *ceig;
    int cpr;   // 16-pixel chunks per row (last one may be partial)
    int rpp;   // rows per pass = 256 / cpr
    int wpr;   // 32-bit mask words per row = ceil(cpr / 2)
    int wrpp;  // mask rows per pass = 256 / wpr
    // band kernel: each wave holds bpw bands of cpr lanes; a band is hb
    // consecutive rows; qcap entries fit in the contour queue
    int bpw, nbands, hb, qcap;
    int wave_lds;  // bytes of LDS per wave: its queue part, at least the 4 KiB reduction slab
    // tiled mode (images larger than one block handles): the band kernel runs on
    // TILE_W x TILE_H tiles, tiles_x * tiles_y blocks per image, sums stay tile-local
    int tiles_x, tiles_y;
    // images whose band-kernel contour queue overflowed; the mask kernel redoes them
    const uint32_t *redo_count;
    const uint32_t *redo_list;
    int only_redo;  // mask kernel: work through redo_list instead of all B images
    // row-band mode: the images are bands of taller images that start at row
    // y_offset; the exact raw moments in the tall image's coordinates are
    // written as three 48-bit limbs each instead of being finalised
    int64_t *partial;  // [B][10][3] or null
    int64_t y_offset;
};

constexpr int MODE_GRAY = 0;     // weight = pixel value
constexpr int MODE_BIT7 = 1;     // weight = on_value * (pixel > 127)
constexpr int MODE_THRESH = 2;   // weight = on_value * (pixel > thresh)

__host__ __device__ constexpr uint32_t pack4(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    return a | (b << 8) | (c << 16) | (d << 24);
}
// byte j of dword d of a chunk is pixel i = 4d + j of the chunk
__host__ __device__ constexpr uint32_t K1(int d) { return pack4(4 * d, 4 * d + 1, 4 * d + 2, 4 * d + 3); }
__host__ __device__ constexpr uint32_t sq(int i) { return (uint32_t)(i * i); }
__host__ __device__ constexpr uint32_t cu(int i) { return (uint32_t)(i * i * i); }
__host__ __device__ constexpr uint32_t K2(int d)
{
    return pack4(sq(4 * d), sq(4 * d + 1), sq(4 * d + 2), sq(4 * d + 3));
}
__host__ __device__ constexpr uint32_t K3L(int d)
{
    return pack4(cu(4 * d) & 255, cu(4 * d + 1) & 255, cu(4 * d + 2) & 255, cu(4 * d + 3) & 255);
}
__host__ __device__ constexpr uint32_t K3H(int d)
{
    return pack4(cu(4 * d) >> 8, cu(4 * d + 1) >> 8, cu(4 * d + 2) >> 8, cu(4 * d + 3) >> 8);
}

__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot4(a, b, c, false);
}

// 0/1 per byte: byte > t, with k = (255 - t) * 0x00010001.
__device__ __forceinline__ uint32_t gt_bytes(uint32_t a, uint32_t k)
{
    uint32_t e = (a & 0x00FF00FFu) + k;
    uint32_t o = ((a >> 8) & 0x00FF00FFu) + k;
    return ((e >> 8) & 0x00010001u) | (o & 0x01000100u);
}

// Per-lane 32-bit partial sums over the visited rows j = 0, 1, ... of the strip
// (up to 32 rows with gray weights, 128 with 0/1 weights):
// a[k][q] = sum_j j^q * s_k(j), s_k = sum_i i^k * w_i over the chunk.
struct Acc32 {
    uint32_t a00, a10, a20, a30, a01, a11, a21, a02, a12, a03;
    __device__ __forceinline__ void clear() { a00 = a10 = a20 = a30 = a01 = a11 = a21 = a02 = a12 = a03 = 0; }
};

struct Tot64 {
    u64 t00, t10, t20, t30, t01, t11, t21, t02, t12, t03;
    __device__ __forceinline__ void clear() { t00 = t10 = t20 = t30 = t01 = t11 = t21 = t02 = t12 = t03 = 0; }
};

// Power sums over the set bits of one byte of a 16-pixel foreground mask, for
// the kernels whose weights are 0/1: entry [m] for the low byte (pixels 0..7),
// entry [256 + m] for the high byte (pixels 8..15).  x = count | sum i << 5 |
// sum i^2 << 12 (fields wide enough for the sum of a low and a high entry),
// y = sum i^3.
struct BandTabEntry {
    uint32_t x, y;
};
struct BandTab {
    BandTabEntry e[512];
};
constexpr BandTab make_band_tab()
{
    BandTab t{};
    for (int half = 0; half < 2; half++) {
        for (int m = 0; m < 256; m++) {
            uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            for (int k = 0; k < 8; k++) {
                if (m & (1 << k)) {
                    const uint32_t i = (uint32_t)(8 * half + k);
                    s0 += 1;
                    s1 += i;
                    s2 += i * i;
                    s3 += i * i * i;
                }
            }
            t.e[half * 256 + m].x = s0 | (s1 << 5) | (s2 << 12);
            t.e[half * 256 + m].y = s3;
        }
    }
    return t;
}
__device__ const BandTab g_band_tab = make_band_tab();
constexpr int BAND_TAB_BYTES = (int)sizeof(BandTab);  // 4 KiB

// acc += a * b for a, b below 2^24: one full-rate instruction.  (Written as C, hipcc proves the factors
// small, drops the 24-bit multiply for a plain one and emits v_mul_lo_u32, which issues at a quarter of
// the rate, plus an add.)  Only for factors that ordinary VALU instructions produced: hipcc does not
// see into the string, so it inserts none of the wait states a v_dot4 result would need.
__device__ __forceinline__ void mad24(uint32_t &acc, uint32_t a, uint32_t b)
{
    asm("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc) : "s"(a), "v"(b));
}

// TAB: the moment sums of a 0/1-weighted chunk come from two lookups in `tab`
// (an LDS copy of g_band_tab at a fixed offset) instead of twenty dot products.
template <int MODE, bool TAB = false>
__device__ __forceinline__ void chunk_sums(const uint4 &v, uint32_t kthr, uint32_t j, Acc32 &a,
                                           uint32_t &bits16, const unsigned char *tab = nullptr)
{
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t src[4];
    uint32_t fg[4];  // foreground of the contour mask: 0/1 per byte, or 0/128 (SH = 7)
    // pixel > 127 is bit 7 of the byte: with 0/1 weights the bit stays where it is and the mask
    // comes out of the dot products times 128
    constexpr int SH = (MODE == MODE_BIT7 && TAB) ? 7 : 0;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        if (MODE == MODE_THRESH) {
            fg[d] = gt_bytes(w[d], kthr);
            src[d] = fg[d];
        } else if (SH) {
            fg[d] = w[d] & 0x80808080u;
            src[d] = 0;  // the table supplies the sums
        } else {
            fg[d] = (w[d] >> 7) & 0x01010101u;
            src[d] = (MODE == MODE_BIT7) ? fg[d] : w[d];
        }
    }
    // byte k of a dword weighs 2^k (low nibble) or 2^(k+4): the dot products are the mask bits
    const uint32_t lo8 = dot4(fg[1], 0x80402010u, dot4(fg[0], 0x08040201u, 0u));  // << SH
    const uint32_t hi8 = dot4(fg[3], 0x80402010u, dot4(fg[2], 0x08040201u, 0u));  // << SH
    bits16 = SH ? ((lo8 >> SH) | (hi8 << (8 - SH))) : (lo8 | (hi8 << 8));

    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (TAB && MODE != MODE_GRAY) {
        static_assert(sizeof(BandTabEntry) == 8, "table entries are addressed as mask * 8");
        const uint32_t ol = SH > 3 ? lo8 >> (SH - 3) : lo8 << (3 - SH);
        const uint32_t oh = SH > 3 ? hi8 >> (SH - 3) : hi8 << (3 - SH);
        const uint2 L = *reinterpret_cast<const uint2 *>(tab + ol);
        const uint2 U = *reinterpret_cast<const uint2 *>(tab + 256 * sizeof(BandTabEntry) + oh);
        const uint32_t packed = L.x + U.x;
        s0 = packed & 31u;
        s1 = (packed >> 5) & 127u;
        s2 = packed >> 12;
        s3 = L.y + U.y;
        const uint32_t j2 = j * j, j3 = j2 * j;  // j < 128, wave-uniform
        a.a00 += s0;
        a.a10 += s1;
        a.a20 += s2;
        a.a30 += s3;
        mad24(a.a01, j, s0);
        mad24(a.a11, j, s1);
        mad24(a.a21, j, s2);
        mad24(a.a02, j2, s0);
        mad24(a.a12, j2, s1);
        mad24(a.a03, j3, s0);
        return;
    } else {
        uint32_t s3l = 0, s3h = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            s0 = dot4(src[d], 0x01010101u, s0);
            s1 = dot4(src[d], K1(d), s1);
            s2 = dot4(src[d], K2(d), s2);
            s3l = dot4(src[d], K3L(d), s3l);
            s3h = dot4(src[d], K3H(d), s3h);
        }
        s3 = s3l + (s3h << 8);
    }
    uint32_t j2 = j * j, j3 = j2 * j;  // j < 128, wave-uniform
    a.a00 += s0;
    a.a10 += s1;
    a.a20 += s2;
    a.a30 += s3;
    a.a01 += __umul24(j, s0);
    a.a11 += __umul24(j, s1);
    a.a21 += __umul24(j, s2);
    a.a02 += __umul24(j2, s0);
    a.a12 += __umul24(j2, s1);
    a.a03 += __umul24(j3, s0);
}

// Rows visited are y = y0 + R*j; fold the strip-local 32-bit sums into the
// 64-bit sums over y (x still strip-local).
__device__ __forceinline__ void flush(const Acc32 &a, u64 y0, u64 R, Tot64 &t)
{
    u64 y02 = y0 * y0, R2 = R * R;
    t.t00 += a.a00;
    t.t10 += a.a10;
    t.t20 += a.a20;
    t.t30 += a.a30;
    t.t01 += y0 * a.a00 + R * a.a01;
    t.t11 += y0 * a.a10 + R * a.a11;
    t.t21 += y0 * a.a20 + R * a.a21;
    t.t02 += y02 * a.a00 + 2 * y0 * R * a.a01 + R2 * a.a02;
    t.t12 += y02 * a.a10 + 2 * y0 * R * a.a11 + R2 * a.a12;
    t.t03 += y02 * y0 * a.a00 + 3 * y02 * R * a.a01 + 3 * y0 * R2 * a.a02 + R2 * R * a.a03;
}

// The same for the band kernel (every row visited, y0 < 2^14): the factors are given as 32-bit values, so
// that a product is one v_mad_u64_u32 instead of the four multiplications of a 64 x 64-bit product.
__device__ __forceinline__ void flush_band(const Acc32 &a, uint32_t y0, Tot64 &t)
{
    const u64 y1 = y0, y2 = (u64)(y0 * y0), y2x3 = (u64)(3u * y0 * y0), y1x2 = (u64)(2u * y0), y1x3 = (u64)(3u * y0);
    const u64 y3 = y2 * y1;   // < 2^42
    t.t00 += a.a00;
    t.t10 += a.a10;
    t.t20 += a.a20;
    t.t30 += a.a30;
    t.t01 += y1 * (u64)a.a00 + a.a01;
    t.t11 += y1 * (u64)a.a10 + a.a11;
    t.t21 += y1 * (u64)a.a20 + a.a21;
    t.t02 += y2 * (u64)a.a00 + y1x2 * (u64)a.a01 + a.a02;
    t.t12 += y2 * (u64)a.a10 + y1x2 * (u64)a.a11 + a.a12;
    t.t03 += y3 * (u64)a.a00 + y2x3 * (u64)a.a01 + y1x3 * (u64)a.a02 + a.a03;
}

// Sum NV 64-bit values over the block; every thread gets the totals.  red is
// LDS with room for FX_WAVES_PER_BLOCK * NV u64.
template <int NV>
__device__ __forceinline__ void block_sum_u64(u64 (&v)[NV], u64 *red)
{
#pragma unroll
    for (int k = 0; k < NV; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            v[k] += (u64)__shfl_down((unsigned long long)v[k], o, FX_WAVE);
    }
    if (fx_lane() == 0) {
#pragma unroll
        for (int k = 0; k < NV; k++)
            red[fx_wave() * NV + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) {
        u64 s = 0;
#pragma unroll
        for (int w = 0; w < FX_WAVES_PER_BLOCK; w++)
            s += red[w * NV + k];
        v[k] = s;
    }
    __syncthreads();
}

__device__ __forceinline__ double block_max_f64(double v, double *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = fmax(v, __shfl_down(v, o, FX_WAVE));
    if (fx_lane() == 0)
        red[fx_wave()] = v;
    __syncthreads();
    double m = red[0];
#pragma unroll
    for (int w = 1; w < FX_WAVES_PER_BLOCK; w++)
        m = fmax(m, red[w]);
    __syncthreads();
    return m;
}

// OpenCV completeMomentState + HuMoments + the reference's log transform, in
// IEEE double without contraction, operation order kept.
__device__ void finalize_moments(const double (&raw)[10], double *moments24, double *hu7, float *hulog7)
{
    const double m00 = raw[0], m10 = raw[1], m01 = raw[2], m20 = raw[3], m11 = raw[4];
    const double m02 = raw[5], m30 = raw[6], m21 = raw[7], m12 = raw[8], m03 = raw[9];
    double cx = 0.0, cy = 0.0, inv_m00 = 0.0;
    if (fabs(m00) > 2.220446049250313e-16) {
        inv_m00 = 1.0 / m00;
        cx = m10 * inv_m00;
        cy = m01 * inv_m00;
    }
    double mu20 = m20 - m10 * cx;
    double mu11 = m11 - m10 * cy;
    double mu02 = m02 - m01 * cy;
    double mu30 = m30 - cx * (3 * mu20 + cx * m10);
    double mu11x2 = mu11 + mu11;
    double mu21 = m21 - cx * (mu11x2 + cx * m01) - cy * mu20;
    double mu12 = m12 - cy * (mu11x2 + cy * m10) - cx * mu02;
    double mu03 = m03 - cy * (3 * mu02 + cy * m01);

    double inv_sqrt_m00 = sqrt(fabs(inv_m00));
    double s2 = inv_m00 * inv_m00, s3 = s2 * inv_sqrt_m00;
    double nu20 = mu20 * s2, nu11 = mu11 * s2, nu02 = mu02 * s2;
    double nu30 = mu30 * s3, nu21 = mu21 * s3, nu12 = mu12 * s3, nu03 = mu03 * s3;

    if (moments24) {
#pragma unroll
        for (int k = 0; k < 10; k++)
            moments24[k] = raw[k];
        moments24[10] = mu20;
        moments24[11] = mu11;
        moments24[12] = mu02;
        moments24[13] = mu30;
        moments24[14] = mu21;
        moments24[15] = mu12;
        moments24[16] = mu03;
        moments24[17] = nu20;
        moments24[18] = nu11;
        moments24[19] = nu02;
        moments24[20] = nu30;
        moments24[21] = nu21;
        moments24[22] = nu12;
        moments24[23] = nu03;
    }

    double hu[7];
    double t0 = nu30 + nu12;
    double t1 = nu21 + nu03;
    double q0 = t0 * t0, q1 = t1 * t1;
    double n4 = 4 * nu11;
    double s = nu20 + nu02;
    double d = nu20 - nu02;
    hu[0] = s;
    hu[1] = d * d + n4 * nu11;
    hu[3] = q0 + q1;
    hu[5] = d * (q0 - q1) + n4 * t0 * t1;
    t0 *= q0 - 3 * q1;
    t1 *= 3 * q0 - q1;
    q0 = nu30 - 3 * nu12;
    q1 = 3 * nu21 - nu03;
    hu[2] = q0 * q0 + q1 * q1;
    hu[4] = q0 * t0 + q1 * t1;
    hu[6] = q1 * t0 - q0 * t1;

#pragma unroll
    for (int i = 0; i < 7; i++) {
        if (hu7)
            hu7[i] = hu[i];
        if (hulog7) {
            double h = hu[i];
            double v = 0.0;
            if (h > 0.0)
                v = -log10(h + 1e-300);
            else if (h < 0.0)
                v = log10(-h + 1e-300);
            if (isinf(v) || isnan(v))
                v = 0.0;
            hulog7[i] = (float)v;
        }
    }
}

// Contour words: bit i of word (y, wx) is pixel (32*wx + i, y).  mask holds
// H + 2 rows (a zero row above and below) of wpr words.
__device__ __forceinline__ uint32_t contour_word(const uint32_t *mask, int wpr, int y, int wx)
{
    const uint32_t *row = mask + (y + 1) * wpr;
    uint32_t c = row[wx];
    uint32_t up = row[wx - wpr];
    uint32_t dn = row[wx + wpr];
    uint32_t lbit = wx > 0 ? (row[wx - 1] >> 31) : 0u;
    uint32_t rbit = wx < wpr - 1 ? (row[wx + 1] & 1u) : 0u;
    uint32_t left = (c << 1) | lbit;
    uint32_t right = (c >> 1) | (rbit << 31);
    return c & ~(up & dn & left & right);
}

constexpr int LDS_RED_BYTES = 1024;   // block reduction scratch (u64 / double)
constexpr int LDS_BCAST_BYTES = 128;  // doubles broadcast from thread 0
constexpr int LDS_HIST_BYTES = 256;   // 64 x u32 histogram
constexpr int LDS_HEAD_BYTES = LDS_RED_BYTES + LDS_BCAST_BYTES + LDS_HIST_BYTES;

template <int MODE, bool CONTOUR, bool ALIGNED>
__device__ __forceinline__ void hu_image_body(const HuArgs &A, const int64_t b, unsigned char *lds_raw)
{
    u64 *red = reinterpret_cast<u64 *>(lds_raw);
    double *bcast = reinterpret_cast<double *>(lds_raw + LDS_RED_BYTES);
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds_raw + LDS_RED_BYTES + LDS_BCAST_BYTES);
    uint32_t *mask = reinterpret_cast<uint32_t *>(lds_raw + LDS_HEAD_BYTES);

    const int t = threadIdx.x;
    const uint8_t *img = A.img + b * A.stride;
    const int H = A.H, W = A.W;

    if (CONTOUR) {
        const int words = (H + 2) * A.wpr;
        for (int i = t; i < words; i += FX_BLOCK)
            mask[i] = 0;
        if (t < 64)
            hist[t] = 0;
        __syncthreads();
    }

    // ---- pass 1: one read of the image -----------------------------------
    const int ty = t / A.cpr, tx = t - ty * A.cpr;
    const int R = A.rpp;
    const int x0 = tx * 16;
    const uint32_t kthr = (uint32_t)(255 - (A.thresh < 0 ? 127 : A.thresh)) * 0x00010001u;
    Tot64 T;
    T.clear();

    if (ty < R) {
        const bool full = ALIGNED && (x0 + 16 <= W);
        uint16_t *mask16 = reinterpret_cast<uint16_t *>(mask);
        auto load_chunk = [&](int y) -> uint4 {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (y < H) {
                const uint8_t *p = img + (int64_t)y * A.pitch + x0;
                if (full) {
                    v = fx_load_once16(p);
                } else {
                    uint32_t w[4] = {0, 0, 0, 0};
                    int n = W - x0;
                    n = n > 16 ? 16 : n;
                    for (int k = 0; k < n; k++)
                        w[k >> 2] |= (uint32_t)p[k] << (8 * (k & 3));
                    v = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
            return v;
        };
        auto put_bits = [&](int y, uint32_t bits) {
            if (CONTOUR && y < H)
                mask16[((y + 1) * A.wpr) * 2 + tx] = (uint16_t)bits;
        };

        for (int y = ty; y < H;) {
            Acc32 a;
            a.clear();
            const int y0 = y;
#pragma unroll 1
            for (int j = 0; j < 32 && y < H; j += 4, y += 4 * R) {
                uint4 v0 = load_chunk(y);
                uint4 v1 = load_chunk(y + R);
                uint4 v2 = load_chunk(y + 2 * R);
                uint4 v3 = load_chunk(y + 3 * R);
                uint32_t b0, b1, b2, b3;
                chunk_sums<MODE>(v0, kthr, j, a, b0);
                chunk_sums<MODE>(v1, kthr, j + 1, a, b1);
                chunk_sums<MODE>(v2, kthr, j + 2, a, b2);
                chunk_sums<MODE>(v3, kthr, j + 3, a, b3);
                put_bits(y, b0);
                put_bits(y + R, b1);
                put_bits(y + 2 * R, b2);
                put_bits(y + 3 * R, b3);
            }
            flush(a, (u64)y0, (u64)R, T);
        }
    }

    // strip-local x -> image x
    u64 M[10];
    {
        const u64 X = (u64)x0, X2 = X * X, X3 = X2 * X;
        M[0] = T.t00;
        M[1] = X * T.t00 + T.t10;
        M[2] = T.t01;
        M[3] = X2 * T.t00 + 2 * X * T.t10 + T.t20;
        M[4] = X * T.t01 + T.t11;
        M[5] = T.t02;
        M[6] = X3 * T.t00 + 3 * X2 * T.t10 + 3 * X * T.t20 + T.t30;
        M[7] = X2 * T.t01 + 2 * X * T.t11 + T.t21;
        M[8] = X * T.t02 + T.t12;
        M[9] = T.t03;
    }
    block_sum_u64<10>(M, red);  // includes the barrier that completes the mask

    if (t == 0) {
        const u64 scale = (MODE == MODE_GRAY) ? 1ull : (u64)A.on_value;
        double raw[10];
#pragma unroll
        for (int k = 0; k < 10; k++)
            raw[k] = (double)(M[k] * scale);
        finalize_moments(raw, A.moments24 ? A.moments24 + b * 24 : nullptr,
                         A.hu7 ? A.hu7 + b * 7 : nullptr, A.hulog7 ? A.hulog7 + b * 7 : nullptr);
    }

    if (!CONTOUR)
        return;

    // ---- contour descriptor out of the LDS bit mask ------------------------
    const int nb = A.n_bins;
    const int wy = t / A.wpr, wx = t - wy * A.wpr;
    const int WR = A.wrpp;
    const bool wactive = wy < WR;
    const int X0 = wx * 32;

    // sums over contour pixels: n, sum x, sum y, sum x^2, sum xy, sum y^2
    u64 S[6] = {0, 0, 0, 0, 0, 0};
    if (wactive) {
        u64 c0 = 0, c1 = 0, cy = 0, c2 = 0, cxy = 0, cyy = 0;  // x word-local
        for (int y = wy; y < H; y += WR) {
            uint32_t bw = contour_word(mask, A.wpr, y, wx);
            if (bw == 0)
                continue;
            uint32_t p0 = __popc(bw & 0xAAAAAAAAu), p1 = __popc(bw & 0xCCCCCCCCu);
            uint32_t p2 = __popc(bw & 0xF0F0F0F0u), p3 = __popc(bw & 0xFF00FF00u);
            uint32_t p4 = __popc(bw & 0xFFFF0000u);
            uint32_t r0 = __popc(bw);
            uint32_t r1 = p0 + 2 * p1 + 4 * p2 + 8 * p3 + 16 * p4;
            // sum i^2 = sum_k 4^k p_k + sum_{j<k} 2^(j+k+1) popc(b & M_j & M_k)
            uint32_t r2 = p0 + 4 * p1 + 16 * p2 + 64 * p3 + 256 * p4;
            r2 += 4 * __popc(bw & 0xAAAAAAAAu & 0xCCCCCCCCu);
            r2 += 8 * __popc(bw & 0xAAAAAAAAu & 0xF0F0F0F0u);
            r2 += 16 * __popc(bw & 0xAAAAAAAAu & 0xFF00FF00u);
            r2 += 32 * __popc(bw & 0xAAAAAAAAu & 0xFFFF0000u);
            r2 += 16 * __popc(bw & 0xCCCCCCCCu & 0xF0F0F0F0u);
            r2 += 32 * __popc(bw & 0xCCCCCCCCu & 0xFF00FF00u);
            r2 += 64 * __popc(bw & 0xCCCCCCCCu & 0xFFFF0000u);
            r2 += 64 * __popc(bw & 0xF0F0F0F0u & 0xFF00FF00u);
            r2 += 128 * __popc(bw & 0xF0F0F0F0u & 0xFFFF0000u);
            r2 += 256 * __popc(bw & 0xFF00FF00u & 0xFFFF0000u);
            const u64 Y = (u64)y;
            c0 += r0;
            c1 += r1;
            c2 += r2;
            cy += Y * r0;
            cxy += Y * r1;
            cyy += Y * Y * r0;
        }
        const u64 X = (u64)X0;
        S[0] = c0;
        S[1] = X * c0 + c1;
        S[2] = cy;
        S[3] = X * X * c0 + 2 * X * c1 + c2;
        S[4] = X * cy + cxy;
        S[5] = cyy;
    }
    block_sum_u64<6>(S, red);

    const u64 n = S[0];
    const int64_t row = b * (9 + nb);
    if (n < 3) {
        const float nanv = __builtin_nanf("");
        for (int i = t; i < 9 + nb; i += FX_BLOCK)
            A.cpca[row + i] = nanv;
        if (A.ceig && t < 6)
            A.ceig[b * 6 + t] = __builtin_nan("");
        return;
    }

    if (t == 0) {
        // covariance (ddof = 1) from exact integer sums: n*Sxx - Sx^2 etc.
        const double dn = (double)n;
        const double inv = 1.0 / (dn * (double)(n - 1));
        const i64 sx = (i64)S[1], sy = (i64)S[2];
        const double cxx = (double)((i64)(n * S[3]) - sx * sx) * inv;
        const double cxy = (double)((i64)(n * S[4]) - sx * sy) * inv;
        const double cyy = (double)((i64)(n * S[5]) - sy * sy) * inv;
        double lam[3], v1[3], head[9];
        fx_eig3(cxx, cxy, 0.0, cyy, 0.0, 0.0, lam, v1);
        fx_pca_head(lam, head);
        for (int i = 0; i < 9; i++)
            A.cpca[row + i] = (float)head[i];
        if (A.ceig) {
            for (int i = 0; i < 3; i++) {
                A.ceig[b * 6 + i] = lam[i];
                A.ceig[b * 6 + 3 + i] = v1[i];
            }
        }
        bcast[0] = (double)S[1] / dn;  // centroid
        bcast[1] = (double)S[2] / dn;
        bcast[2] = v1[0];
        bcast[3] = v1[1];
    }
    __syncthreads();
    const double ccx = bcast[0], ccy = bcast[1], vx = bcast[2], vy = bcast[3];

    // |projection on PC1| is monotone in x along a row, so the largest value
    // in a word is at its lowest or highest set bit.
    double top = 0.0;
    if (wactive) {
        for (int y = wy; y < H; y += WR) {
            uint32_t bw = contour_word(mask, A.wpr, y, wx);
            if (bw == 0)
                continue;
            const double dy = (double)y - ccy;
            const int lo = __ffs((int)bw) - 1, hi = 31 - __clz((int)bw);
            const double plo = fabs(fma(dy, vy, ((double)(X0 + lo) - ccx) * vx));
            const double phi = fabs(fma(dy, vy, ((double)(X0 + hi) - ccx) * vx));
            top = fmax(top, fmax(plo, phi));
        }
    }
    top = block_max_f64(top, reinterpret_cast<double *>(red));

    if (top > 0.0) {
        const double step = top / (double)nb;
        const double inv_step = (double)nb * fx_rcp_n1(top);  // first guess of a bin only: fx_hist_bin settles it against `step`
        if (wactive) {
            for (int y = wy; y < H; y += WR) {
                uint32_t bw = contour_word(mask, A.wpr, y, wx);
                const double dy = (double)y - ccy;
                while (bw) {
                    const int i = __ffs((int)bw) - 1;
                    bw &= bw - 1;
                    const double p = fabs(fma(dy, vy, ((double)(X0 + i) - ccx) * vx));
                    atomicAdd(&hist[fx_hist_bin(p, step, inv_step, nb)], 1u);
                }
            }
        }
        __syncthreads();
        if (t < nb)
            A.cpca[row + 9 + t] = (float)((double)hist[t] / (double)n);
    } else if (t < nb) {
        A.cpca[row + 9 + t] = t == 0 ? 1.0f : 0.0f;
    }
}


// One block per image; in redo mode a small grid walks the list of images whose
// band-kernel contour queue overflowed (the list is filled by hu_finalize_kernel).
template <int MODE, bool CONTOUR, bool ALIGNED>
__global__ __launch_bounds__(FX_BLOCK) void hu_image_kernel(HuArgs A)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int64_t items = A.only_redo ? (int64_t)A.redo_count[0] : A.B;
    for (int64_t it = blockIdx.x; it < items; it += gridDim.x) {
        hu_image_body<MODE, CONTOUR, ALIGNED>(A, A.only_redo ? (int64_t)A.redo_list[it] : it, lds_raw);
        __syncthreads();
    }
}

// ------------------------------------------------------------ band kernel
// Organisation for images up to 1024 x 1024.  A row of the image (cpr chunks)
// sits in cpr adjacent lanes of one wave; each group of cpr lanes owns a band
// of hb consecutive rows and walks down it.  A lane therefore has the rows
// above and below in its own registers and the chunks left and right in the
// neighbouring lanes, so contour pixels are found while the image streams in.
// Only contour segments (16-pixel masks with at least one contour pixel, about
// 0.5 % of the chunks of a shape image) leave the registers: they go to an LDS
// queue that never leaves the block.  Once the image has streamed through, the
// block sums its moments, and for the contour descriptor every wave walks its
// own part of the queue twice (largest |projection| on the first principal
// axis, then the histogram): nothing but 16 sums, 4 counters and the
// histogram is written to memory.
//
// hu_finalize_kernel follows (one lane per image: central / normalised
// moments, Hu, log, the descriptor's nine leading scalars -- serial float64
// code that would idle 255 of 256 threads inside the streaming kernel).
constexpr int BAND_LDS_COUNTERS = LDS_HEAD_BYTES;          // 16 B: segments queued by each wave
constexpr int BAND_LDS_EDGES = BAND_LDS_COUNTERS + 16;     // 4 x 64 histogram counters of the contour tail
// the lookup table of the 0/1-weighted modes at a fixed offset (the lookups then address it with an
// immediate), behind it the four reduction slabs of the contour variant, then the per-wave regions:
// the wave's part of the segment queue (contour variant) or the 4 KiB slab of the wave-level
// reduction (moments only)
constexpr int BAND_LDS_TAB = BAND_LDS_EDGES + 4 * FX_BLOCK;
constexpr int BAND_SLAB_BYTES = 4096;
#ifndef BAND_VPP_CONTOUR
#define BAND_VPP_CONTOUR 4
#endif
constexpr int BAND_CSLAB_BYTES = BAND_VPP_CONTOUR * FX_WAVE * 8;  // slab of the contour variant
constexpr int BAND_LDS_CSLAB = BAND_LDS_TAB + BAND_TAB_BYTES;
constexpr int band_lds_wave(bool contour) { return BAND_LDS_CSLAB + (contour ? FX_WAVES_PER_BLOCK * BAND_CSLAB_BYTES : 0); }
constexpr int BAND_LDS_FIXED = BAND_LDS_CSLAB;  // bytes in front of the slabs and the per-wave regions
constexpr int TILE_W = 1024, TILE_H = 256;                 // tiled mode: one block per tile, 64 rows per wave
constexpr int BAND_SUMS = 16;                              // u64 per image: 10 raw moments + 6 contour sums

struct BandScratch {
    u64 *sums;         // [B][16]
    uint32_t *qcount;  // [B][4] segments each wave found (above qcap / 4: the mask kernel redoes the image)
    uint32_t *redo_count;  // [1] zeroed before the launches
    uint32_t *redo_list;   // [B] images for the mask kernel
};

__device__ __forceinline__ bool band_overflowed(const uint32_t *qcount, int64_t b, int qcap)
{
    const uint32_t wcap = (uint32_t)(qcap >> 2);
    const uint32_t *c = qcount + b * FX_WAVES_PER_BLOCK;
    return c[0] > wcap || c[1] > wcap || c[2] > wcap || c[3] > wcap;
}

// Power sums over the set bits of a 16-pixel segment: count, sum i, sum i^2.
__device__ __forceinline__ void segment_sums(uint32_t bits, uint32_t &r0, uint32_t &r1, uint32_t &r2)
{
    const uint32_t p0 = __popc(bits & 0xAAAAu), p1 = __popc(bits & 0xCCCCu);
    const uint32_t p2 = __popc(bits & 0xF0F0u), p3 = __popc(bits & 0xFF00u);
    r0 = __popc(bits);
    r1 = p0 + 2 * p1 + 4 * p2 + 8 * p3;
    // i^2 = sum_k 4^k b_k + sum_{j<k} 2^(j+k+1) b_j b_k over the bits b of i
    r2 = p0 + 4 * p1 + 16 * p2 + 64 * p3;
    r2 += 4 * __popc(bits & 0xAAAAu & 0xCCCCu) + 8 * __popc(bits & 0xAAAAu & 0xF0F0u);
    r2 += 16 * __popc(bits & 0xAAAAu & 0xFF00u) + 16 * __popc(bits & 0xCCCCu & 0xF0F0u);
    r2 += 32 * __popc(bits & 0xCCCCu & 0xFF00u) + 64 * __popc(bits & 0xF0F0u & 0xFF00u);
}

// Tail of the streaming kernel, contour variant: |projection on PC1| maximum
// and histogram of the image's contour pixels, each wave over the segments it
// queued in LDS.  Along a segment the signed projection is monotone in x, so
// its largest magnitude is at an end, and a segment whose two ends fall into
// the same bin on the same side of zero belongs to that bin as a whole.
// Float32 only sorts out the cases that are beyond doubt: everything within
// `eps` of a decision boundary is decided in float64, so the results are those
// of the float64 formula.  Every thread of the block calls this after the
// barrier that completed `red`; the branch conditions below are block-uniform.
__device__ __forceinline__ void band_contour_hist(const HuArgs &A, const int64_t b, unsigned char *lds_raw,
                                                  uint32_t *wq, const int qn)
{
    const u64 *red = reinterpret_cast<const u64 *>(lds_raw);
    double *bcast = reinterpret_cast<double *>(lds_raw + LDS_RED_BYTES);
    const uint32_t *cnt = reinterpret_cast<const uint32_t *>(lds_raw + BAND_LDS_COUNTERS);
    const int t = threadIdx.x, lane = fx_lane(), wave = fx_wave();
    const uint32_t wcap = (uint32_t)(A.qcap >> 2);
    if (cnt[0] > wcap || cnt[1] > wcap || cnt[2] > wcap || cnt[3] > wcap)
        return;  // the mask kernel redoes the image
    const u64 n = red[10] + red[BAND_SUMS + 10] + red[2 * BAND_SUMS + 10] + red[3 * BAND_SUMS + 10];
    if (n < 3)
        return;  // hu_finalize_kernel writes the NaN row
    // Centroid and first principal axis as hu_finalize_kernel derives them.  The streaming loop leaves
    // the vector ALU nearly full for small images, so the serial float64 part runs on one wave and the
    // other three fetch its four results from LDS instead of repeating it.
    uint32_t *hist_all = reinterpret_cast<uint32_t *>(lds_raw + BAND_LDS_EDGES);
    uint32_t *hist = hist_all + wave * FX_WAVE;
    hist[lane] = 0;
    if (wave == 0) {
        u64 cs[6];
        cs[0] = n;
#pragma unroll
        for (int k = 1; k < 6; k++)
            cs[k] = red[10 + k] + red[BAND_SUMS + 10 + k] + red[2 * BAND_SUMS + 10 + k] + red[3 * BAND_SUMS + 10 + k];
        const double dn = (double)n;
        // the common factor of the covariance does not turn its axes: a reciprocal good to an ulp or two will do
        const double inv = fx_rcp_n1(dn * (double)(n - 1));
        const i64 sx = (i64)cs[1], sy = (i64)cs[2];
        const double cxx = (double)((i64)(n * cs[3]) - sx * sx) * inv;
        const double cxy = (double)((i64)(n * cs[4]) - sx * sy) * inv;
        const double cyy = (double)((i64)(n * cs[5]) - sy * sy) * inv;
        double ax, ay;
        fx_eig2_axis(cxx, cxy, cyy, ax, ay);
        // the centroid is the correctly rounded quotient (NumPy's mean of integer coordinates): pixels of
        // symmetric shapes sit exactly on bin edges, and there the last bit of the centroid decides
        const double rn = fx_rcp_n2(dn);
        if (lane == 0) {
            bcast[4] = fx_div_r((double)sx, dn, rn);
            bcast[5] = fx_div_r((double)sy, dn, rn);
            bcast[6] = ax;
            bcast[7] = ay;
        }
    }
    __syncthreads();
    const double ccx = bcast[4], ccy = bcast[5], vx = bcast[6], vy = bcast[7];
    const float cx32 = (float)ccx, cy32 = (float)ccy, vx32 = (float)vx, vy32 = (float)vy;

    auto sproj = [&](uint32_t X, int i, double dyv) -> double {
        return fma(dyv, vy, ((double)(int)(X + i) - ccx) * vx);
    };
    auto larger = [&](uint32_t e, double best) -> double {
        const uint32_t bits = e & 0xFFFFu, pos = e >> 16;
        const uint32_t X = (pos & 63u) * 16u;
        const double dy = (double)(int)(pos >> 6) - ccy;
        const double slo = sproj(X, __ffs((int)bits) - 1, dy), shi = sproj(X, 31 - __clz((int)bits), dy);
        return fmax(best, fmax(fabs(slo), fabs(shi)));
    };
    auto ends32 = [&](uint32_t e, float &slo, float &shi) {
        const uint32_t bits = e & 0xFFFFu, pos = e >> 16;
        const float xb = (float)(int)((pos & 63u) * 16u) - cx32;
        const float ty = ((float)(int)(pos >> 6) - cy32) * vy32;
        const float lo = (float)(__ffs((int)bits) - 1), hi = (float)(31 - __clz((int)bits));
        slo = fmaf(xb + lo, vx32, ty);
        shi = fmaf(xb + hi, vx32, ty);
    };
    // Segments whose bin float32 cannot settle are collected (ballot-compacted,
    // so that the slower code runs with full lanes) at the front of the queue
    // itself: the list never gets ahead of the reading position, and every lane
    // has read its entry of a round before any lane writes.
    uint32_t *list = wq;
    uint32_t nl = 0;  // wave-uniform
    auto defer = [&](bool want, uint32_t e) {
        const unsigned long long m = __ballot(want);
        if (m) {
            const uint32_t at = nl + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (want)
                list[at] = e;
            nl += (uint32_t)__popcll(m);
        }
    };
    auto list_ready = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // ---- largest |projection| ---------------------------------------------------
    // both ends of every segment in float64 (a float32 pass that picks the candidates first costs more
    // than it saves: the vector ALU runs float64 at the same rate)
    double top = 0.0;
    for (int i = lane; i < qn; i += FX_WAVE)
        top = larger(wq[i], top);
    // (one LDS atomic maximum per lane instead of the cross-lane steps and the exchange: 256 atomics on one
    // address take four times as long as this whole pass)
    top = fx_wave_max_f64_dpp(top);
    if (lane == 0)
        bcast[wave] = top;
    __syncthreads();
    top = fmax(fmax(bcast[0], bcast[1]), fmax(bcast[2], bcast[3]));

    const int nb = A.n_bins;
    float *out = A.cpca + b * (9 + nb) + 9;
    if (!(top > 0.0)) {
        if (t < nb)
            out[t] = t == 0 ? 1.0f : 0.0f;
        return;
    }
    // ---- histogram ----------------------------------------------------------------
    const double step = top / (double)nb;
    const double inv_step = (double)nb * fx_rcp_n1(top);  // first guess of a bin only: fx_hist_bin settles it against `step`
    const float eps = 4e-7f * ((float)top * 1.000001f + fabsf(cx32) + fabsf(cy32) + 1024.0f);
    // float32 bin of |s|, or -1 when |s| / step is within `margin` of an integer
    const float inv32 = (float)inv_step;
    const float margin = 2.0f * eps * inv32 + 1e-5f * (float)nb;
    auto bin32 = [&](float v) -> int {
        const float q = fabsf(v) * inv32;
        const float f = floorf(q);
        const float frac = q - f;
        if (frac < margin || frac > 1.0f - margin || q >= (float)nb)
            return -1;
        return (int)f;
    };
    // A segment whose ends lie in adjacent bins on one side of zero: |projection| rises by |vx| per pixel
    // from the end in the lower bin, so it reaches the edge between the two bins at the distance
    // z = (edge - q0) / (|vx| / step) from that end, and the pixels nearer than z are below the edge.
    // When a pixel's distance is within float32's reach of z (m2) the segment goes to the list.
    const float adq = fabsf(vx32) * inv32;  // bins per pixel along a row
    const float inv_adq = adq > 1e-4f ? __builtin_amdgcn_rcpf(adq) : 0.0f;  // 0: rows nearly perpendicular to the axis are not split this way
    const float m2 = margin * inv_adq + 1e-5f;
    for (int base = 0; base < qn; base += FX_WAVE) {
        const int i = base + lane;
        const uint32_t e = i < qn ? wq[i] : 0u;
        bool settled = false;
        if (e) {
            const uint32_t bits = e & 0xFFFFu;
            float a, c;
            ends32(e, a, c);
            const int ba = bin32(a), bc = bin32(c);
            const bool same_side = (a > eps && c > eps) || (a < -eps && c < -eps);
            const bool sure = ba >= 0 && bc >= 0 && same_side;
            const bool up = bc > ba;  // |projection| grows with x
            const int b0 = up ? ba : bc, b1 = up ? bc : ba;
            const int lo = __ffs((int)bits) - 1, hi = 31 - __clz((int)bits);
            const float z = ((float)b1 - fabsf(up ? a : c) * inv32) * inv_adq;
            const float zf = floorf(z);
            const float fz = z - zf;
            const int below = min(max((int)zf + 1, 0), hi - lo + 1);  // pixels at distances 0 .. floor(z) from the lower end
            const uint32_t near_mask = up ? (1u << (lo + below)) - 1u : ~((1u << (hi + 1 - below)) - 1u);
            const uint32_t all = (uint32_t)__popc(bits);
            const uint32_t cum = b1 == b0 ? all : (uint32_t)__popc(bits & near_mask);
            settled = sure && (b1 == b0 || (b1 == b0 + 1 && fz >= m2 && fz <= 1.0f - m2 && inv_adq > 0.0f));
            if (settled) {
                atomicAdd(&hist[b0], cum);
                if (cum != all)
                    atomicAdd(&hist[b1], all - cum);
            }
        }
        defer(e != 0u && !settled, e);
    }
    list_ready();
    // one pixel at a time: float32 where it is certain, float64 otherwise
    for (uint32_t i = lane; i < nl; i += FX_WAVE) {
        const uint32_t e = list[i];
        const uint32_t pos = e >> 16;
        const uint32_t X = (pos & 63u) * 16u;
        const float xb = (float)(int)X - cx32;
        const float t32 = ((float)(int)(pos >> 6) - cy32) * vy32;
        const double dy = (double)(int)(pos >> 6) - ccy;
        uint32_t rest = e & 0xFFFFu;
        while (rest) {
            const int k = __ffs((int)rest) - 1;
            rest &= rest - 1;
            int bin = bin32(fmaf(xb + (float)k, vx32, t32));
            if (bin < 0)
                bin = fx_hist_bin(fabs(sproj(X, k, dy)), step, inv_step, nb);
            atomicAdd(&hist[bin], 1u);
        }
    }
    __syncthreads();
    if (t < nb) {
        const uint32_t h = hist_all[t] + hist_all[FX_WAVE + t] + hist_all[2 * FX_WAVE + t] + hist_all[3 * FX_WAVE + t];
        out[t] = (float)((double)h / (double)n);
    }
}

#ifndef FX_BAND_WAVES
#define FX_BAND_WAVES 4
#endif
// ALIGNED: 0 rows are read byte by byte; 1 and 2 every chunk is a whole, 16-byte aligned vector, read with
// a global load (1) or a buffer load (2).
template <int MODE, bool CONTOUR, int ALIGNED>
__global__ __launch_bounds__(FX_BLOCK, FX_BAND_WAVES) void hu_band_kernel(HuArgs A, BandScratch G)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    u64 *red = reinterpret_cast<u64 *>(lds_raw);
    const int wave_bytes = A.wave_lds;  // >= BAND_SLAB_BYTES

    const int t = threadIdx.x;
    const int lane = fx_lane(), wave = fx_wave();
    const int64_t b = blockIdx.x;  // image, or (image, tile) in tiled mode
    const uint8_t *img = A.img + b * A.stride;
    int H = A.H, W = A.W;
    if (A.tiles_x > 0) {
        const int tiles = A.tiles_x * A.tiles_y;
        const int64_t image = b / tiles;
        const int tile = (int)(b - image * tiles);
        const int tyi = tile / A.tiles_x, txi = tile - tyi * A.tiles_x;
        img = A.img + image * A.stride + (int64_t)tyi * TILE_H * A.pitch + (int64_t)txi * TILE_W;
        H = min(TILE_H, A.H - tyi * TILE_H);
        W = min(TILE_W, A.W - txi * TILE_W);
    }
    const int cpr = A.cpr, hb = A.hb;

    // 0/1 weights: the moment sums of a chunk are looked up by its mask bytes
    constexpr bool TAB = MODE != MODE_GRAY;
    const unsigned char *tab = lds_raw + BAND_LDS_TAB;
    if (TAB) {
        uint2 *dst = reinterpret_cast<uint2 *>(lds_raw + BAND_LDS_TAB);
        const uint2 *srct = reinterpret_cast<const uint2 *>(g_band_tab.e);
        dst[t] = srct[t];
        dst[t + FX_BLOCK] = srct[t + FX_BLOCK];
        __syncthreads();
    }

    const int bl = lane / cpr, tx = lane - bl * cpr;
    const int x0 = tx * 16;
    const bool active = bl < A.bpw && x0 < W;
    // Moments only (SWEEP): the block sweeps the image from top to bottom, 16 * bpw consecutive rows a step --
    // a wave-load takes bpw consecutive rows, a wave its four loads in flight, the four waves one after the
    // other -- so the block reads at one place (16 KB of a 512-pixel-wide image per step).  A read-only kernel
    // with this pattern streams 7.1-7.2 TB/s on this device; with bands per lane group, eight places of 512
    // bytes at once per block, 6.6 TB/s (tools/ubench/stream_read.hip).  A lane's rows are then
    // ya + (uniform row offset), which is all the moment sums ask for; the contour detection, which needs the
    // rows above and below a row, gets them through LDS (below).  The 1024 x 256 tiles of larger images keep
    // the bands (a band of hb consecutive rows per group of cpr lanes): their rows are a pitch apart whichever
    // way a block walks them, 64 x 8192 x 8192 take 0.64 ms with bands and 0.70 ms swept.
    const bool SWEEP = CONTOUR || A.tiles_x == 0;   // (block-uniform; tiles carry no contour descriptor)
    const int band = wave * A.bpw + bl;
    const int ya = SWEEP ? wave * 4 * A.bpw + bl : band * hb;
    const uint32_t kthr = (uint32_t)(255 - (A.thresh < 0 ? 127 : A.thresh)) * 0x00010001u;

    // ALIGNED == 2: buffer loads.  The descriptor of a row (scalar registers, built by the scalar unit)
    // starts at the row's offset in the image or tile and ends where the image ends; a lane adds the
    // offset of its band's first row and its chunk.  Rows below the image, lanes without a chunk and
    // rows past the band (no records) come back as zeros from the range check: no address arithmetic,
    // no compare and no select on the vector unit, which the small images (bands of a few rows, the
    // vector unit nearly full) are bound by: 224 x 224 +8 %.  Bands of 64 rows and more stream 1-2 %
    // faster with global loads (ALIGNED == 1: rows outside the band or the image read a clamped address
    // and are zeroed afterwards); either way the four loads of a group issue back to back.
    typedef uint32_t band_u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t pitch32 = (uint32_t)A.pitch;
    const uint32_t span = (uint32_t)(H - 1) * pitch32 + (uint32_t)W;  // bytes of the image (tile) from its first pixel
    const uint32_t voff = active ? (uint32_t)ya * pitch32 + (uint32_t)x0 : 0x7FFFFFFFu;
    auto load_chunk = [&](int r) -> uint4 {
        if (ALIGNED == 2) {
            const uint32_t roff = (uint32_t)r * pitch32;
            // signed and through max(): an unsigned saturating subtraction exists only on the vector
            // unit, and a descriptor word in a vector register wraps every load in a loop over lanes
            const int left = max((int)span - (int)roff, 0);
            const int records = __builtin_amdgcn_readfirstlane(SWEEP || r < hb ? left : 0);
            const __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(img) + roff, 0, records, 0x00020000);
            const band_u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 2 /* nt */);
            return make_uint4(q.x, q.y, q.z, q.w);
        }
        const int y = ya + r;
        const bool valid = active && (SWEEP || r < hb) && y < H;
        if (ALIGNED == 1) {
            const int yc = valid ? y : 0;
            const int xc = active ? x0 : 0;
            uint4 v = fx_load_once16(img + (int64_t)yc * A.pitch + xc);
            v.x = valid ? v.x : 0u;
            v.y = valid ? v.y : 0u;
            v.z = valid ? v.z : 0u;
            v.w = valid ? v.w : 0u;
            return v;
        }
        uint4 v = make_uint4(0, 0, 0, 0);
        if (valid) {
            const uint8_t *p = img + (int64_t)y * A.pitch + x0;
            uint32_t w[4] = {0, 0, 0, 0};
            int n = W - x0;
            n = n > 16 ? 16 : n;
            for (int k = 0; k < n; k++)
                w[k >> 2] |= (uint32_t)p[k] << (8 * (k & 3));
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
        return v;
    };

    // ---- contour segments -----------------------------------------------------
    // Each wave appends to its own quarter of the LDS queue; the slot of a lane
    // comes from a ballot, so no atomic is needed.  A segment is the 16-pixel
    // mask | (row << 6 | chunk) << 16.
    const int wcap = A.qcap >> 2;
    uint32_t *wq = reinterpret_cast<uint32_t *>(lds_raw + band_lds_wave(CONTOUR) + wave * wave_bytes);
    uint32_t wn = 0;  // wave-uniform: segments this wave has queued (may run past wcap: overflow)
    int ybase = ya;   // the lane's part of a queued segment's row (the sweep with contours sets its own)
    auto push = [&](int r, uint32_t bits) {
        const unsigned long long m = __ballot(bits != 0);
        if (m) {
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const uint32_t idx = wn + below;
            const uint32_t after = wn + (uint32_t)__popcll(m);
            // the room is checked for the wave as a whole (scalar): a wave that runs out of it stores
            // nothing more, and its count past the limit sends the image to the mask kernel
            if (after <= (uint32_t)wcap && bits)
                wq[idx] = bits | ((uint32_t)(((ybase + r) << 6) | tx) << 16);
            wn = after;
        }
    };
    const bool has_left = tx > 0, has_right = tx < cpr - 1;

    // Contour pixels (foreground with a 4-neighbour that is background) of four
    // consecutive rows rp .. rp+3 held as two packed pairs P0 = row rp | row
    // rp+1 << 16, P1 likewise; above = row rp-1, below = row rp+4.  Rows 0 and
    // hb-1 of the band are left out here: their outer neighbour row belongs to
    // another band.  Must be called by every lane of the wave.
    auto contour_bits = [&](uint32_t above, uint32_t P0, uint32_t P1, uint32_t below, uint32_t &B0, uint32_t &B1) {
        const uint32_t up0 = (P0 << 16) | above;
        const uint32_t dn0 = (P0 >> 16) | (P1 << 16);
        const uint32_t up1 = (P1 << 16) | (P0 >> 16);
        const uint32_t dn1 = (P1 >> 16) | (below << 16);
        // bit 0 and bit 15 of the four rows in one word: rows rp, rp+1 at bits
        // 0/15 and 16/31, rows rp+2, rp+3 rotated left by two
        const uint32_t K = 0x80018001u;
        const uint32_t p1k = P1 & K;
        const uint32_t E = (P0 & K) | ((p1k << 2) | (p1k >> 30));
        uint32_t EL = __builtin_amdgcn_update_dpp(0u, E, 0x138, 0xf, 0xf, false);  // wave_shr:1: from lane - 1
        uint32_t ER = __builtin_amdgcn_update_dpp(0u, E, 0x130, 0xf, 0xf, false);  // wave_shl:1: from lane + 1
        EL = has_left ? EL : 0u;
        ER = has_right ? ER : 0u;
        const uint32_t lb0 = (EL >> 15) & 0x00010001u;
        const uint32_t lb1 = ((EL >> 17) | (EL << 15)) & 0x00010001u;
        const uint32_t rb0 = (ER << 15) & 0x80008000u;
        const uint32_t rb1 = (ER << 13) & 0x80008000u;
        const uint32_t left0 = ((P0 << 1) & 0xFFFEFFFEu) | lb0, right0 = ((P0 >> 1) & 0x7FFF7FFFu) | rb0;
        const uint32_t left1 = ((P1 << 1) & 0xFFFEFFFEu) | lb1, right1 = ((P1 >> 1) & 0x7FFF7FFFu) | rb1;
        B0 = P0 & ~(up0 & dn0 & left0 & right0);
        B1 = P1 & ~(up1 & dn1 & left1 & right1);
    };
    // the same for the sweep: rows ybase + rp .. + 3 of the lane's group, none left out (a row above the
    // image or below it has no foreground, so it queues nothing)
    auto detect4s = [&](int rp, uint32_t above, uint32_t P0, uint32_t P1, uint32_t below) {
        uint32_t B0, B1;
        contour_bits(above, P0, P1, below, B0, B1);
        push(rp, B0 & 0xFFFFu);
        push(rp + 1, B0 >> 16);
        push(rp + 2, B1 & 0xFFFFu);
        push(rp + 3, B1 >> 16);
    };
    Tot64 T;
    T.clear();

    // Rows per flush of the 32-bit partial sums.  With 0/1 weights (binary
    // modes) they are 255 times smaller than with gray weights and can run over
    // 128 rows: for bands up to that height the 64-bit totals are formed once,
    // after the loop, and are not live inside it.
    constexpr int SUB = (MODE == MODE_GRAY) ? 32 : 128;
    auto run_rows = [&](int jb, Acc32 &a) {
#pragma unroll 1
        for (int j = 0; j < SUB && jb + j < hb; j += 4) {
            const int r = jb + j;
            uint4 v0 = load_chunk(r);
            uint4 v1 = load_chunk(r + 1);
            uint4 v2 = load_chunk(r + 2);
            uint4 v3 = load_chunk(r + 3);
            uint32_t b0, b1, b2, b3;
            chunk_sums<MODE, TAB>(v0, kthr, j, a, b0, tab);
            chunk_sums<MODE, TAB>(v1, kthr, j + 1, a, b1, tab);
            chunk_sums<MODE, TAB>(v2, kthr, j + 2, a, b2, tab);
            chunk_sums<MODE, TAB>(v3, kthr, j + 3, a, b3, tab);
        }
    };
    if (CONTOUR) {
        // The sweep with contours.  A lane's four rows of a step lie rpl rows apart, and the rows above and
        // below a wave's stretch belong to other waves: the row masks of a step go through LDS, a column of
        // 16-bit masks per chunk column, rows one after the other, and come back as windows of six consecutive
        // rows -- lane group g = wave * rpl + bl takes the rows 4 g - 1 .. 4 g + 2 of the step (one row behind,
        // so that the row below the last one is there), with the row above and the row below.  Three stretches
        // of a step's rows per column in turn: a step reads its own and the last two rows of the one before,
        // and writes nothing a wave one block barrier behind may still read.  One block barrier per step; the
        // next step's loads are issued before it.
        const int rpl = A.bpw, step_rows = 16 * rpl;
        const int steps = (H + step_rows - 1) / step_rows;
        const int seg_steps = SUB / step_rows > 1 ? SUB / step_rows : 1;
        const int col_bytes = (3 * step_rows + 2) * 2;   // (an odd number of 32-bit words: the columns spread over the banks)
        unsigned char *ring = lds_raw + BAND_LDS_CSLAB + tx * col_bytes;   // this lane's column (the slabs are free until the block sums)
        const int group = wave * rpl + bl;   // rows 4 group - 1 .. 4 group + 2 of a step
        ybase = 4 * group - 1;
        const int my_slot = wave * 4 * rpl + bl;   // the lane's first row within a step
        uint4 v0 = load_chunk(0), v1 = load_chunk(rpl), v2 = load_chunk(2 * rpl), v3 = load_chunk(3 * rpl);
        if (active && group == 0)   // the two rows above the image, which the first step finds at the end of the third stretch
            *reinterpret_cast<uint32_t *>(reinterpret_cast<uint16_t *>(ring) + 3 * step_rows - 2) = 0u;
        Acc32 a;
        a.clear();
        int seg_base = 0, region = 0;   // region: which third of a column this step writes
        for (int s = 0; s < steps; s++) {
            const int j0 = s * step_rows - seg_base;
            uint32_t b0, b1, b2, b3;
            chunk_sums<MODE, TAB>(v0, kthr, j0, a, b0, tab);
            chunk_sums<MODE, TAB>(v1, kthr, j0 + rpl, a, b1, tab);
            chunk_sums<MODE, TAB>(v2, kthr, j0 + 2 * rpl, a, b2, tab);
            chunk_sums<MODE, TAB>(v3, kthr, j0 + 3 * rpl, a, b3, tab);
            if (s + 1 < steps) {
                const int r0 = (s + 1) * step_rows;
                v0 = load_chunk(r0), v1 = load_chunk(r0 + rpl), v2 = load_chunk(r0 + 2 * rpl), v3 = load_chunk(r0 + 3 * rpl);
            }
            if ((s + 1) % seg_steps == 0 || s + 1 == steps) {
                flush_band(a, (uint32_t)(ya + seg_base), T);
                a.clear();
                seg_base = (s + 1) * step_rows;
            }
            if (active) {
                uint16_t *mine = reinterpret_cast<uint16_t *>(ring) + region * step_rows + my_slot;
                mine[0] = (uint16_t)b0;
                mine[rpl] = (uint16_t)b1;
                mine[2 * rpl] = (uint16_t)b2;
                mine[3 * rpl] = (uint16_t)b3;
            }
            __syncthreads();
            {
                // rows 4 group - 2 .. 4 group + 3 of the step; for group 0 the first two are the last rows of the step before
                const int prev = region == 0 ? 2 : region - 1;
                const int first = group == 0 ? prev * step_rows + step_rows - 2 : region * step_rows + 4 * group - 2;
                const uint16_t *col = reinterpret_cast<const uint16_t *>(ring);
                uint32_t w0 = 0, w1 = 0, w2 = 0;
                if (active) {
                    w0 = *reinterpret_cast<const uint32_t *>(col + first);
                    const uint32_t *rest = reinterpret_cast<const uint32_t *>(col + region * step_rows + 4 * group);
                    w1 = rest[0];
                    w2 = rest[1];
                }
                detect4s(s * step_rows, w0 & 0xFFFFu, __builtin_amdgcn_alignbit(w1, w0, 16), __builtin_amdgcn_alignbit(w2, w1, 16), w2 >> 16);
            }
            region = region == 2 ? 0 : region + 1;
        }
        // the last row of the last step (if it is a row of the image): group 0 finds it, and the row above it,
        // at the end of the stretch the last step wrote (every wave has passed that step's barrier)
        if (steps * step_rows - 1 < H) {
            const int prev = region == 0 ? 2 : region - 1;
            const uint16_t *col = reinterpret_cast<const uint16_t *>(ring);
            const uint32_t w0 = active && group == 0 ? *reinterpret_cast<const uint32_t *>(col + prev * step_rows + step_rows - 2) : 0u;
            detect4s(steps * step_rows, w0 & 0xFFFFu, w0 >> 16, 0u, 0u);
        }
        __syncthreads();   // the columns lie where the block sums' slabs are
    } else if (SWEEP) {
        // 32-bit partial sums over segments of SUB rows (or one step, where a step is more than that): the row
        // offset within the segment is the wave-uniform factor of the weighted sums
        const int rpl = A.bpw, step_rows = 16 * rpl;
        const int seg_steps = SUB / step_rows > 1 ? SUB / step_rows : 1;
        for (int base = 0; base < H; base += seg_steps * step_rows) {
            Acc32 a;
            a.clear();
#pragma unroll 1
            for (int ss = 0; ss < seg_steps && base + ss * step_rows < H; ss++) {
                const int j0 = ss * step_rows, r0 = base + j0;
                uint4 v0 = load_chunk(r0);
                uint4 v1 = load_chunk(r0 + rpl);
                uint4 v2 = load_chunk(r0 + 2 * rpl);
                uint4 v3 = load_chunk(r0 + 3 * rpl);
                uint32_t b0, b1, b2, b3;
                chunk_sums<MODE, TAB>(v0, kthr, j0, a, b0, tab);
                chunk_sums<MODE, TAB>(v1, kthr, j0 + rpl, a, b1, tab);
                chunk_sums<MODE, TAB>(v2, kthr, j0 + 2 * rpl, a, b2, tab);
                chunk_sums<MODE, TAB>(v3, kthr, j0 + 3 * rpl, a, b3, tab);
            }
            flush_band(a, (uint32_t)(ya + base), T);
        }
    } else if (hb <= SUB) {
        Acc32 a;
        a.clear();
        run_rows(0, a);
        flush_band(a, (uint32_t)ya, T);
    } else {
        for (int jb = 0; jb < hb; jb += SUB) {
            Acc32 a;
            a.clear();
            run_rows(jb, a);
            flush_band(a, (uint32_t)(ya + jb), T);
        }
    }

    constexpr int NV = CONTOUR ? BAND_SUMS : 10;
    u64 V[NV];
    int qn = 0;  // segments in this wave's part of the queue
    {
        // x0 <= 1008: every factor of the shift fits 32 bits (a 64 x 32-bit product is three multiplications
        // instead of four, and the compiler cannot see that x0 is not negative)
        const uint32_t xs = (uint32_t)x0;
        const u64 X = xs, X2 = xs * xs, X3 = xs * xs * xs, Xx2 = 2u * xs, Xx3 = 3u * xs, X2x3 = 3u * xs * xs;
        V[0] = T.t00;
        V[1] = X * T.t00 + T.t10;
        V[2] = T.t01;
        V[3] = X2 * T.t00 + Xx2 * T.t10 + T.t20;
        V[4] = X * T.t01 + T.t11;
        V[5] = T.t02;
        V[6] = X3 * T.t00 + X2x3 * T.t10 + Xx3 * T.t20 + T.t30;
        V[7] = X2 * T.t01 + Xx2 * T.t11 + T.t21;
        V[8] = X * T.t02 + T.t12;
        V[9] = T.t03;
    }

    if (CONTOUR) {
        // The wave reads back only its own part of the queue: LDS operations of
        // one wave complete in order, so no block barrier is needed here.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        qn = wn > (uint32_t)wcap ? 0 : (int)wn;  // overflow: the mask kernel redoes the image

        // sums over contour pixels: n, sum x, sum y, sum x^2, sum xy, sum y^2;
        // at most 16 segments per lane, so 32-bit partial sums cannot overflow
        uint32_t s0 = 0, sx = 0, sy = 0, sxx = 0, sxy = 0, syy = 0;
        for (int i = lane; i < qn; i += FX_WAVE) {
            const uint32_t e = wq[i];
            const uint32_t pos = e >> 16;
            const uint32_t y = pos >> 6, X = (pos & 63u) * 16u;
            uint32_t r0, r1, r2;
            segment_sums(e & 0xFFFFu, r0, r1, r2);
            const uint32_t ex = X * r0 + r1;
            s0 += r0;
            sx += ex;
            sy += y * r0;
            sxx += X * X * r0 + 2 * X * r1 + r2;
            sxy += y * ex;
            syy += y * y * r0;
        }
        V[NV - 6] = s0;
        V[NV - 5] = sx;
        V[NV - 4] = sy;
        V[NV - 3] = sxx;
        V[NV - 2] = sxy;
        V[NV - 1] = syy;
        if (lane == 0) {
            G.qcount[b * FX_WAVES_PER_BLOCK + wave] = wn;
            reinterpret_cast<uint32_t *>(lds_raw + BAND_LDS_COUNTERS)[wave] = wn;
        }
    }
    // ---- block sums ---------------------------------------------------------------
    // Wave level: the lanes park VPP values at a time in the wave's LDS slab as
    // [value][lane]; lane l then adds VPP adjacent lanes of one value and a few
    // butterfly steps finish the 64-lane sum.  (With VPP = 8: 16 LDS accesses
    // and 6 cross-lane moves per lane instead of 12 cross-lane moves per value.)
    unsigned char *slab_raw = CONTOUR ? lds_raw + BAND_LDS_CSLAB + wave * BAND_CSLAB_BYTES
                                      : lds_raw + band_lds_wave(false) + wave * wave_bytes;
    u64 *slab = reinterpret_cast<u64 *>(slab_raw);
    // contour variant: 4 values per pass in a 2 KiB slab (LDS per block decides how many blocks a CU holds)
    constexpr int VPP = CONTOUR ? BAND_VPP_CONTOUR : 8;
    constexpr int LPV = FX_WAVE / VPP;  // lanes that share one value after the column sums
    constexpr int PASSES = (NV + VPP - 1) / VPP;
    u64 wsum[PASSES];  // lane l: wave total of value pass * VPP + l / LPV
#pragma unroll
    for (int pass = 0; pass < PASSES; pass++) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < VPP; k++)
            slab[k * FX_WAVE + lane] = (pass * VPP + k < NV) ? V[(pass * VPP + k < NV) ? pass * VPP + k : 0] : 0ull;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u64 acc = 0;
        const u64 *col = slab + (lane / LPV) * FX_WAVE + (lane % LPV) * VPP;
#pragma unroll
        for (int i = 0; i < VPP; i++)
            acc += col[i];
#pragma unroll
        for (int o = 1; o < LPV; o <<= 1)
            acc += (u64)__shfl_xor((unsigned long long)acc, o, FX_WAVE);
        wsum[pass] = acc;
    }
    // Block level: 4 waves x 16 values through `red`.
    if (lane % LPV == 0) {
#pragma unroll
        for (int pass = 0; pass < PASSES; pass++)
            if (pass * VPP + lane / LPV < BAND_SUMS)
                red[wave * BAND_SUMS + pass * VPP + lane / LPV] = wsum[pass];
    }
    __syncthreads();
    if (t < NV) {
        u64 sum = 0;
#pragma unroll
        for (int w = 0; w < FX_WAVES_PER_BLOCK; w++)
            sum += red[w * BAND_SUMS + t];
        G.sums[b * BAND_SUMS + t] = sum;
    }
    if (CONTOUR)
        band_contour_hist(A, b, lds_raw, wq, qn);
}

// One lane per image: everything that is serial float64 arithmetic.
__global__ __launch_bounds__(FX_BLOCK) void hu_finalize_kernel(HuArgs A, BandScratch G, int gray, int contour)
{
    const int64_t b = (int64_t)blockIdx.x * FX_BLOCK + threadIdx.x;
    if (b >= A.B)
        return;
    const u64 *sums = G.sums + b * BAND_SUMS;
    const u64 scale = gray ? 1ull : (u64)A.on_value;
    double raw[10];
#pragma unroll
    for (int k = 0; k < 10; k++)
        raw[k] = (double)(sums[k] * scale);
    finalize_moments(raw, A.moments24 ? A.moments24 + b * 24 : nullptr, A.hu7 ? A.hu7 + b * 7 : nullptr,
                     A.hulog7 ? A.hulog7 + b * 7 : nullptr);
    if (!contour)
        return;
    if (band_overflowed(G.qcount, b, A.qcap)) {
        G.redo_list[atomicAdd(G.redo_count, 1u)] = (uint32_t)b;
        return;
    }
    const int nb = A.n_bins;
    const int64_t row = b * (9 + nb);
    const u64 n = sums[10];
    if (n < 3) {
        for (int i = 0; i < 9 + nb; i++)
            A.cpca[row + i] = __builtin_nanf("");
        if (A.ceig)
            for (int i = 0; i < 6; i++)
                A.ceig[b * 6 + i] = __builtin_nan("");
        return;
    }
    // covariance (ddof = 1) from exact integer sums: n*Sxx - Sx^2 etc.
    const double dn = (double)n;
    const double inv = 1.0 / (dn * (double)(n - 1));
    const i64 sx = (i64)sums[11], sy = (i64)sums[12];
    const double cxx = (double)((i64)(n * sums[13]) - sx * sx) * inv;
    const double cxy = (double)((i64)(n * sums[14]) - sx * sy) * inv;
    const double cyy = (double)((i64)(n * sums[15]) - sy * sy) * inv;
    double lam[3], v1[3], head[9];
    fx_eig3(cxx, cxy, 0.0, cyy, 0.0, 0.0, lam, v1);
    fx_pca_head(lam, head);
    for (int i = 0; i < 9; i++)
        A.cpca[row + i] = (float)head[i];
    if (A.ceig) {
        for (int i = 0; i < 3; i++) {
            A.ceig[b * 6 + i] = lam[i];
            A.ceig[b * 6 + 3 + i] = v1[i];
        }
    }
}

// Tiled mode, one wave per image: the lanes shift their tiles' local sums to
// image coordinates and add them up in exact 128-bit integers (lane l takes
// tiles l, l + 64, ...; a butterfly adds the lanes), each raw moment is rounded
// once to double (for images up to 512 x 512 this is the value OpenCV's
// double accumulation holds exactly; beyond, it is the value that accumulation
// approximates), then lane 0 finalises as usual.
__global__ __launch_bounds__(FX_BLOCK) void hu_tile_combine_kernel(HuArgs A, BandScratch G, int gray)
{
    typedef unsigned __int128 u128;
    __shared__ u64 wave_sums[FX_WAVES_PER_BLOCK][10][2];
    const int64_t b = blockIdx.x;
    const int lane = fx_lane(), wave = fx_wave();
    const int tiles = A.tiles_x * A.tiles_y;
    u128 M[10];
#pragma unroll
    for (int k = 0; k < 10; k++)
        M[k] = 0;
    for (int tile = threadIdx.x; tile < tiles; tile += FX_BLOCK) {   // (the 128-bit products are the kernel's time)
        const u64 *t = G.sums + (b * tiles + tile) * BAND_SUMS;
        const int tyi = tile / A.tiles_x, txi = tile - tyi * A.tiles_x;
        const u128 X = (u128)(txi * TILE_W), Y = (u128)(tyi * TILE_H) + (u128)(A.partial ? A.y_offset : 0);
        const u128 t00 = t[0], t10 = t[1], t01 = t[2], t20 = t[3], t11 = t[4], t02 = t[5];
        const u128 t30 = t[6], t21 = t[7], t12 = t[8], t03 = t[9];
        M[0] += t00;
        M[1] += t10 + X * t00;
        M[2] += t01 + Y * t00;
        M[3] += t20 + 2 * X * t10 + X * X * t00;
        M[4] += t11 + X * t01 + Y * t10 + X * Y * t00;
        M[5] += t02 + 2 * Y * t01 + Y * Y * t00;
        M[6] += t30 + 3 * X * t20 + 3 * X * X * t10 + X * X * X * t00;
        M[7] += t21 + 2 * X * t11 + X * X * t01 + Y * t20 + 2 * X * Y * t10 + X * X * Y * t00;
        M[8] += t12 + 2 * Y * t11 + Y * Y * t10 + X * t02 + 2 * X * Y * t01 + X * Y * Y * t00;
        M[9] += t03 + 3 * Y * t02 + 3 * Y * Y * t01 + Y * Y * Y * t00;
    }
    if (tiles > 1) {
#pragma unroll
        for (int k = 0; k < 10; k++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const u64 lo = (u64)__shfl_xor((unsigned long long)(u64)M[k], o, FX_WAVE);
                const u64 hi = (u64)__shfl_xor((unsigned long long)(u64)(M[k] >> 64), o, FX_WAVE);
                M[k] += ((u128)hi << 64) | lo;
            }
        }
    }
    // the waves' sums through LDS (integer sums: any order gives the same bits)
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 10; k++) {
            wave_sums[wave][k][0] = (u64)M[k];
            wave_sums[wave][k][1] = (u64)(M[k] >> 64);
        }
    }
    __syncthreads();
    if (threadIdx.x != 0)
        return;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        M[k] = 0;
        for (int w = 0; w < FX_WAVES_PER_BLOCK; w++)
            M[k] += ((u128)wave_sums[w][k][1] << 64) | wave_sums[w][k][0];
    }
    const u128 scale = gray ? 1 : (u128)A.on_value;
    if (A.partial) {
        const u64 limb = (1ull << 48) - 1;
#pragma unroll
        for (int k = 0; k < 10; k++) {
            const u128 v = M[k] * scale;
            int64_t *o = A.partial + (b * 10 + k) * 3;
            o[0] = (int64_t)((u64)v & limb);
            o[1] = (int64_t)((u64)(v >> 48) & limb);
            o[2] = (int64_t)(u64)(v >> 96);
        }
        return;
    }
    double raw[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        const u128 v = M[k] * scale;
        raw[k] = fx_u128_to_double((u64)(v >> 64), (u64)v);
    }
    finalize_moments(raw, A.moments24 ? A.moments24 + b * 24 : nullptr, A.hu7 ? A.hu7 + b * 7 : nullptr,
                     A.hulog7 ? A.hulog7 + b * 7 : nullptr);
}

// Raw moments given as sums of 48-bit limbs (possibly added up over several
// bands / ranks, so a limb may exceed 48 bits): one rounding to float64, then
// the usual finalisation.
__global__ __launch_bounds__(FX_BLOCK) void hu_from_partial_kernel(const int64_t *partial, int64_t B, double *moments24,
                                                                 double *hu7, float *hulog7)
{
    typedef unsigned __int128 u128;
    const int64_t b = (int64_t)blockIdx.x * FX_BLOCK + threadIdx.x;
    if (b >= B)
        return;
    double raw[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        const int64_t *l = partial + (b * 10 + k) * 3;
        const u128 v = (u128)(u64)l[0] + ((u128)(u64)l[1] << 48) + ((u128)(u64)l[2] << 96);
        raw[k] = fx_u128_to_double((u64)(v >> 64), (u64)v);
    }
    finalize_moments(raw, moments24 ? moments24 + b * 24 : nullptr, hu7 ? hu7 + b * 7 : nullptr,
                     hulog7 ? hulog7 + b * 7 : nullptr);
}

__global__ void hu_groupstat_kernel(const float *hulog7, const int64_t *offsets, int64_t G, float *out14)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G * 7)
        return;
    const int64_t g = i / 7;
    const int j = (int)(i - g * 7);
    const int64_t a = offsets[g], b = offsets[g + 1];
    const float cnt = (float)(b - a);
    float s = 0.0f;
    for (int64_t r = a; r < b; r++)
        s = s + hulog7[r * 7 + j];
    const float mean = s / cnt;
    float q = 0.0f;
    for (int64_t r = a; r < b; r++) {
        const float d = hulog7[r * 7 + j] - mean;
        q = q + d * d;
    }
    out14[g * 14 + j] = mean;
    out14[g * 14 + 7 + j] = sqrtf(q / cnt);
}

template <int MODE, bool CONTOUR>
int launch_image_kernel(const HuArgs &A, bool aligned, size_t lds, hipStream_t s)
{
    dim3 grid((unsigned)(A.only_redo ? (A.B < 2048 ? A.B : 2048) : A.B)), block(FX_BLOCK);
    if (aligned) {
        auto k = hu_image_kernel<MODE, CONTOUR, true>;
        if (lds > 48 * 1024)
            FX_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, grid, block, lds, s, A);
    } else {
        auto k = hu_image_kernel<MODE, CONTOUR, false>;
        if (lds > 48 * 1024)
            FX_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k, grid, block, lds, s, A);
    }
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

#ifndef FX_SWEEP_HB
#define FX_SWEEP_HB 0   // the moments-only kernel (block sweep) takes the buffer loads
#endif
// How the band kernel reads the rows (its ALIGNED parameter).  The buffer loads address a row as a
// 32-bit offset from the image's (tile's) first pixel; they pay for bands of fewer than 64 rows.
inline int band_aligned(bool aligned, int W, int H, int64_t pitch_bytes, int hb)
{
    if (!aligned || W % 16 != 0)
        return 0;
    const int64_t rows = (H < 1024 ? H : 1024) + 1100;  // a block sees a tile or an image of at most 1024 rows; a step of the sweep may reach up to 1024 rows below it
    return hb < 64 && rows * pitch_bytes < (1ll << 31) ? 2 : 1;
}

template <int MODE, bool CONTOUR>
int launch_band_kernel(const HuArgs &A, const BandScratch &G, int aligned, size_t lds, hipStream_t s)
{
    const int64_t blocks = A.tiles_x > 0 ? A.B * A.tiles_x * A.tiles_y : A.B;
    dim3 grid((unsigned)blocks), block(FX_BLOCK);
    if (aligned == 2)
        hipLaunchKernelGGL((hu_band_kernel<MODE, CONTOUR, 2>), grid, block, lds, s, A, G);
    else if (aligned == 1)
        hipLaunchKernelGGL((hu_band_kernel<MODE, CONTOUR, 1>), grid, block, lds, s, A, G);
    else
        hipLaunchKernelGGL((hu_band_kernel<MODE, CONTOUR, 0>), grid, block, lds, s, A, G);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

int launch_band(const HuArgs &A, const BandScratch &G, int mode, bool contour, int aligned, size_t lds, hipStream_t s)
{
    if (contour)
        return mode == MODE_GRAY   ? launch_band_kernel<MODE_GRAY, true>(A, G, aligned, lds, s)
               : mode == MODE_BIT7 ? launch_band_kernel<MODE_BIT7, true>(A, G, aligned, lds, s)
                                   : launch_band_kernel<MODE_THRESH, true>(A, G, aligned, lds, s);
    return mode == MODE_GRAY   ? launch_band_kernel<MODE_GRAY, false>(A, G, aligned, lds, s)
           : mode == MODE_BIT7 ? launch_band_kernel<MODE_BIT7, false>(A, G, aligned, lds, s)
                               : launch_band_kernel<MODE_THRESH, false>(A, G, aligned, lds, s);
}

int launch_mask(const HuArgs &A, int mode, bool contour, bool aligned, size_t lds, hipStream_t s)
{
    if (contour)
        return mode == MODE_GRAY   ? launch_image_kernel<MODE_GRAY, true>(A, aligned, lds, s)
               : mode == MODE_BIT7 ? launch_image_kernel<MODE_BIT7, true>(A, aligned, lds, s)
                                   : launch_image_kernel<MODE_THRESH, true>(A, aligned, lds, s);
    return mode == MODE_GRAY   ? launch_image_kernel<MODE_GRAY, false>(A, aligned, lds, s)
           : mode == MODE_BIT7 ? launch_image_kernel<MODE_BIT7, false>(A, aligned, lds, s)
                               : launch_image_kernel<MODE_THRESH, false>(A, aligned, lds, s);
}

// Where an entry point's scratch comes from: the library's per-stream buffer
// (grown on demand) or a fixed block that is never (re)allocated.
struct HuScratch {
    FxCallGuard *guard;     // per-stream buffer, or
    unsigned char *fixed;   // a block of fixed_bytes bytes
    size_t fixed_bytes;
    const char *who;
    // `units` = images (or tiles) the launch keeps a record for
    unsigned char *get(size_t bytes, int64_t units) const
    {
        if (guard)
            return (unsigned char *)guard->scratch(bytes, 1);
        if (units > FX_HU_NOALLOC_MAX_IMAGES || bytes > fixed_bytes) {
            fx_set_error("%s: this batch needs records for %lld images or tiles (%zu bytes); the built-in block holds "
                         "%d (%zu bytes): at most %d images up to 1024 x 1024, or %d tiles of 1024 x 256 of larger "
                         "ones, per call", who, (long long)units, bytes, FX_HU_NOALLOC_MAX_IMAGES, fixed_bytes,
                         FX_HU_NOALLOC_MAX_IMAGES, FX_HU_NOALLOC_MAX_IMAGES);
            return nullptr;
        }
        return fixed;
    }
};

int hu_batch_impl(const char *who, const HuScratch &scratch, const uint8_t *d_img, int64_t B, int H, int W,
                  int64_t pitch_bytes, int64_t img_stride_bytes, int thresh, int on_value, double *d_moments24,
                  double *d_hu7, float *d_hulog7, int n_bins, float *d_cpca, double *d_ceig, hipStream_t s);

}  // namespace

extern "C" int fx_hu_batch_u8(const uint8_t *d_img, int64_t B, int H, int W, int64_t pitch_bytes,
                              int64_t img_stride_bytes, int thresh, int on_value, double *d_moments24,
                              double *d_hu7, float *d_hulog7, int n_bins, float *d_cpca, double *d_ceig,
                              void *stream)
{
    FX_REQUIRE(B >= 0, "fx_hu_batch_u8: B = %lld", (long long)B);
    if (B == 0)
        return FX_OK;
    FxCallGuard guard((hipStream_t)stream);
    const HuScratch scratch = {&guard, nullptr, 0, "fx_hu_batch_u8"};
    return hu_batch_impl("fx_hu_batch_u8", scratch, d_img, B, H, W, pitch_bytes, img_stride_bytes, thresh, on_value,
                         d_moments24, d_hu7, d_hulog7, n_bins, d_cpca, d_ceig, (hipStream_t)stream);
}

namespace {

int hu_batch_impl(const char *who, const HuScratch &scratch, const uint8_t *d_img, int64_t B, int H, int W,
                  int64_t pitch_bytes, int64_t img_stride_bytes, int thresh, int on_value, double *d_moments24,
                  double *d_hu7, float *d_hulog7, int n_bins, float *d_cpca, double *d_ceig, hipStream_t s)
{
    FX_REQUIRE(B >= 0, "%s: B = %lld", who, (long long)B);
    if (B == 0)
        return FX_OK;
    FX_REQUIRE(d_img != nullptr, "%s: d_img is NULL", who);
    FX_REQUIRE(H >= 1 && W >= 1 && H <= 16384 && W <= 16384, "%s: image %d x %d outside 1..16384", who, H, W);
    FX_REQUIRE(pitch_bytes >= W, "%s: pitch %lld < width %d", who, (long long)pitch_bytes, W);
    FX_REQUIRE(B == 1 || img_stride_bytes >= pitch_bytes * (int64_t)(H - 1) + W,
               "%s: image stride %lld overlaps images", who, (long long)img_stride_bytes);
    FX_REQUIRE(thresh < 255, "%s: thresh %d must be < 255 (or negative)", who, thresh);
    FX_REQUIRE(thresh < 0 || (on_value >= 1 && on_value <= 255), "%s: on_value %d outside 1..255", who, on_value);
    FX_REQUIRE(d_moments24 || d_hu7 || d_hulog7 || d_cpca, "%s: no output requested", who);
    FX_REQUIRE(!d_cpca || (n_bins >= 1 && n_bins <= 64), "%s: n_bins %d outside 1..64", who, n_bins);
    FX_REQUIRE(d_cpca || !d_ceig, "%s: d_ceig needs d_cpca", who);
    FX_REQUIRE(B <= 0x7fffffffLL, "%s: B too large for one launch", who);

    HuArgs A;
    A.img = d_img;
    A.B = B;
    A.H = H;
    A.W = W;
    A.pitch = pitch_bytes;
    A.stride = img_stride_bytes;
    A.thresh = thresh;
    A.on_value = on_value;
    A.moments24 = d_moments24;
    A.hu7 = d_hu7;
    A.hulog7 = d_hulog7;
    A.n_bins = n_bins;
    A.cpca = d_cpca;
    A.ceig = d_ceig;
    A.cpr = (W + 15) / 16;
    A.wpr = (A.cpr + 1) / 2;

    const bool contour = d_cpca != nullptr;
    const size_t mask_bytes = contour ? (size_t)(H + 2) * A.wpr * 4 : 0;
    A.rpp = A.cpr <= FX_BLOCK ? FX_BLOCK / A.cpr : 0;
    A.wrpp = A.wpr <= FX_BLOCK ? FX_BLOCK / A.wpr : 0;
    A.redo_count = nullptr;
    A.redo_list = nullptr;
    A.only_redo = 0;
    A.partial = nullptr;
    A.y_offset = 0;
    A.tiles_x = A.tiles_y = 0;
    const bool aligned = ((uintptr_t)d_img % 16 == 0) && (pitch_bytes % 16 == 0) && (img_stride_bytes % 16 == 0 || B == 1);
    const size_t mask_lds = LDS_HEAD_BYTES + mask_bytes;
    const int mode = thresh < 0 ? MODE_GRAY : (thresh == 127 ? MODE_BIT7 : MODE_THRESH);

    // Which organisation:
    //   band   images up to 1024 x 1024: a row fits in one wave and a queue
    //          position (row, chunk) in 10 + 6 bits;
    //   mask   contour descriptor of images up to 2048 x 2048 whose bit mask fits in LDS;
    //   tiled  anything larger, Hu only: band kernel per 1024 x 256 tile + 128-bit combine.
    const bool band = A.cpr <= FX_WAVE && H <= 1024;
    const bool mask_ok = W <= 2048 && H <= 2048 && mask_bytes <= 128 * 1024;

    if (!band && contour) {
        if (!mask_ok) {
            fx_set_error("%s: the contour descriptor supports images up to 2048 x 2048 with at most "
                         "128 KiB of bit mask (got %d x %d); Hu moments alone have no such limit", who, H, W);
            return FX_EINVAL;
        }
        FxProfileScope prof(s);  // launch timing (when switched on) brackets the streaming kernel only
        return launch_mask(A, mode, contour, aligned, mask_lds, s);
    }

    if (!band) {
        // tiled: the band kernel sees 1024 x 256 tiles (one 64-row band per wave)
        A.tiles_x = (W + TILE_W - 1) / TILE_W;
        A.tiles_y = (H + TILE_H - 1) / TILE_H;
        const int64_t blocks = B * A.tiles_x * A.tiles_y;
        FX_REQUIRE(blocks <= 0x7fffffffLL, "%s: too many tiles for one launch", who);
        A.cpr = FX_WAVE;
        A.bpw = 1;
        A.nbands = FX_WAVES_PER_BLOCK;
        A.hb = TILE_H / FX_WAVES_PER_BLOCK;
        A.qcap = 0;
        A.wave_lds = BAND_SLAB_BYTES;
        BandScratch G;
        G.sums = (u64 *)scratch.get((size_t)blocks * BAND_SUMS * sizeof(u64), blocks);
        if (!G.sums)
            return scratch.guard ? FX_ENOMEM : FX_EINVAL;
        G.qcount = nullptr;
        G.redo_count = nullptr;
        G.redo_list = nullptr;
        int rc;
        {
            FxProfileScope prof(s);
            rc = launch_band(A, G, mode, false, band_aligned(aligned, W, H, pitch_bytes, A.hb), (size_t)BAND_LDS_FIXED + (size_t)FX_WAVES_PER_BLOCK * A.wave_lds, s);
        }
        if (rc != FX_OK)
            return rc;
        hipLaunchKernelGGL(hu_tile_combine_kernel, dim3((unsigned)B), dim3(FX_BLOCK), 0, s, A, G,
                           mode == MODE_GRAY ? 1 : 0);
        FX_HIP_CHECK(hipGetLastError());
        return FX_OK;
    }

    A.bpw = FX_WAVE / A.cpr;
    if (mode == MODE_GRAY && A.bpw > 16)
        A.bpw = 16;   // (gray weights, images up to 48 pixels wide: a lane's four rows of a step lie bpw rows apart, and with
                      // more than 21 the cube of the row offset times a chunk's 16 * 255 overflows the 32-bit partial sums --
                      // with or without the contour descriptor)
    A.nbands = A.bpw * FX_WAVES_PER_BLOCK;
    A.hb = (H + A.nbands - 1) / A.nbands;
    // every wave could meet a segment in every chunk it sees -- four loads of bpw rows per step of the sweep:
    // cap per wave, four waves
    const int64_t sweep_steps = (H + 16 * A.bpw - 1) / (16 * A.bpw);
    const int64_t wave_segments = sweep_steps * 4 * A.bpw * A.cpr;
    A.qcap = (int)(wave_segments < 1024 ? wave_segments : 1024) * FX_WAVES_PER_BLOCK;

    // per-launch scratch: sums | queue counts | redo list
    const size_t sums_bytes = (size_t)B * BAND_SUMS * sizeof(u64);
    const size_t count_bytes = (size_t)B * FX_WAVES_PER_BLOCK * 4;
    const size_t redo_bytes = contour ? 16 + (((size_t)B * 4 + 15) & ~(size_t)15) : 0;
    unsigned char *base = scratch.get(sums_bytes + count_bytes + redo_bytes, B);
    if (!base)
        return scratch.guard ? FX_ENOMEM : FX_EINVAL;
    BandScratch G;
    G.sums = (u64 *)base;
    G.qcount = (uint32_t *)(base + sums_bytes);
    G.redo_count = (uint32_t *)(base + sums_bytes + count_bytes);
    G.redo_list = G.redo_count + 4;
    const bool may_overflow = contour && (A.qcap >> 2) < wave_segments;
    if (may_overflow)
        FX_HIP_CHECK(hipMemsetAsync(G.redo_count, 0, 16, s));

    A.wave_lds = contour && A.qcap > BAND_SLAB_BYTES ? A.qcap : BAND_SLAB_BYTES;  // qcap / 4 entries of 4 bytes
    // the columns of row masks of the sweep with contours lie in the block sums' slabs (cpr * bpw <= 64: 6.2 KiB at most)
    FX_REQUIRE(!contour || (int64_t)A.cpr * ((3 * 16 * A.bpw + 2) * 2) <= (int64_t)FX_WAVES_PER_BLOCK * BAND_CSLAB_BYTES,
               "%s: internal: row-mask columns of %d x %d bytes exceed the slabs", who, A.cpr, (3 * 16 * A.bpw + 2) * 2);
    size_t band_lds = (size_t)band_lds_wave(contour) + (size_t)FX_WAVES_PER_BLOCK * A.wave_lds;
    // Blocks per CU follow from the LDS a block asks for: 31 KiB lets five in.  Bands of 64 rows and more
    // stream long enough that four blocks per CU keep the memory system busy, and measure 2 % faster than
    // five (fewer concurrent streams); shorter bands want the fifth block to cover the histogram tails.
    const size_t four_per_cu = 33 * 1024;
    if (contour && A.hb >= 64 && band_lds < four_per_cu)
        band_lds = four_per_cu;
    int rc;
    {
        FxProfileScope prof(s);
        rc = launch_band(A, G, mode, contour, band_aligned(aligned, W, H, pitch_bytes, FX_SWEEP_HB), band_lds, s);
    }
    if (rc != FX_OK)
        return rc;
    hipLaunchKernelGGL(hu_finalize_kernel, dim3((unsigned)((B + FX_BLOCK - 1) / FX_BLOCK)), dim3(FX_BLOCK), 0, s, A, G,
                       mode == MODE_GRAY ? 1 : 0, contour ? 1 : 0);
    FX_HIP_CHECK(hipGetLastError());
    if (!contour)
        return FX_OK;
    if (may_overflow) {
        // images whose contour queue overflowed (very ragged masks) go through the mask kernel
        A.redo_count = G.redo_count;
        A.redo_list = G.redo_list;
        A.only_redo = 1;
        return launch_mask(A, mode, contour, aligned, mask_lds, s);
    }
    return FX_OK;
}

// The block behind fx_hu_batch_u8_noalloc: part of the code object, so every
// device the library runs on holds its own copy at its own address from the
// moment the module is loaded until the process ends.  Host code only ever
// uses the per-device address that hipGetSymbolAddress returns.
constexpr size_t HU_NOALLOC_BYTES =
    (size_t)FX_HU_NOALLOC_MAX_IMAGES * (BAND_SUMS * sizeof(u64) + FX_WAVES_PER_BLOCK * 4 + 4) + 256;
static __device__ __align__(256) unsigned char g_hu_noalloc_scratch[HU_NOALLOC_BYTES];

constexpr int HU_NOALLOC_MAX_DEVICES = 16;
std::mutex g_hu_noalloc_mu;
std::atomic<unsigned char *> g_hu_noalloc_ptr[HU_NOALLOC_MAX_DEVICES] = {};   // written once under the mutex

}  // namespace

// Looks up the current device's copy of the built-in block (once per device).
int fx_hu_noalloc_setup(void)
{
    int dev = -1;
    FX_HIP_CHECK(hipGetDevice(&dev));
    FX_REQUIRE(dev >= 0 && dev < HU_NOALLOC_MAX_DEVICES, "fx_hu_batch_u8_noalloc: device %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_hu_noalloc_mu);
    if (g_hu_noalloc_ptr[dev].load(std::memory_order_acquire))
        return FX_OK;
    unsigned char *ptr = nullptr;
    FX_HIP_CHECK(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(g_hu_noalloc_scratch)));
    FX_REQUIRE(ptr != nullptr, "fx_hu_batch_u8_noalloc: the built-in scratch block has no address on device %d", dev);
    g_hu_noalloc_ptr[dev].store(ptr, std::memory_order_release);
    return FX_OK;
}

extern "C" int fx_hu_batch_u8_noalloc(const uint8_t *d_img, int64_t B, int H, int W, int64_t pitch_bytes,
                                      int64_t img_stride_bytes, int thresh, int on_value, double *d_moments24,
                                      double *d_hu7, float *d_hulog7, int n_bins, float *d_cpca, double *d_ceig,
                                      void *stream)
{
    FX_REQUIRE(B >= 0, "fx_hu_batch_u8_noalloc: B = %lld", (long long)B);
    if (B == 0)
        return FX_OK;
    int dev = -1;
    FX_HIP_CHECK(hipGetDevice(&dev));
    FX_REQUIRE(dev >= 0 && dev < HU_NOALLOC_MAX_DEVICES, "fx_hu_batch_u8_noalloc: device %d out of range", dev);
    if (!g_hu_noalloc_ptr[dev].load(std::memory_order_acquire)) {
        // the address lookup is not a stream operation: it must not run while the caller's stream is capturing
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
            FX_REQUIRE(false, "fx_hu_batch_u8_noalloc: fx_init(%d) has to run before the first call under stream capture", dev);
        const int rc = fx_hu_noalloc_setup();
        if (rc != FX_OK)
            return rc;
    }
    const HuScratch scratch = {nullptr, g_hu_noalloc_ptr[dev].load(std::memory_order_acquire), HU_NOALLOC_BYTES,
                               "fx_hu_batch_u8_noalloc"};
    return hu_batch_impl("fx_hu_batch_u8_noalloc", scratch, d_img, B, H, W, pitch_bytes, img_stride_bytes, thresh,
                         on_value, d_moments24, d_hu7, d_hulog7, n_bins, d_cpca, d_ceig, (hipStream_t)stream);
}

extern "C" int fx_hu_groupstat(const float *d_hulog7, const int64_t *d_group_offsets, int64_t G, float *d_out14,
                               void *stream)
{
    FX_REQUIRE(G >= 0, "fx_hu_groupstat: G = %lld", (long long)G);
    if (G == 0)
        return FX_OK;
    FX_REQUIRE(d_hulog7 && d_group_offsets && d_out14, "fx_hu_groupstat: NULL argument");
    const int64_t n = G * 7;
    FX_REQUIRE((n + 255) / 256 <= 0x7fffffffLL, "fx_hu_groupstat: G too large");
    hipLaunchKernelGGL(hu_groupstat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d_hulog7, d_group_offsets, G, d_out14);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

extern "C" int fx_hu_partial_u8(const uint8_t *d_img, int64_t B, int H, int W, int64_t pitch_bytes,
                                int64_t img_stride_bytes, int thresh, int on_value, int64_t y_offset,
                                int64_t *d_partial, void *stream)
{
    FX_REQUIRE(B >= 0, "fx_hu_partial_u8: B = %lld", (long long)B);
    if (B == 0)
        return FX_OK;
    FX_REQUIRE(d_img && d_partial, "fx_hu_partial_u8: null pointer");
    FX_REQUIRE(H >= 1 && W >= 1 && H <= 16384 && W <= 16384, "fx_hu_partial_u8: band %d x %d outside 1..16384", H, W);
    FX_REQUIRE(y_offset >= 0 && y_offset <= (1ll << 20), "fx_hu_partial_u8: y_offset %lld outside 0..2^20",
               (long long)y_offset);
    FX_REQUIRE(pitch_bytes >= W, "fx_hu_partial_u8: pitch %lld < width %d", (long long)pitch_bytes, W);
    FX_REQUIRE(B == 1 || img_stride_bytes >= pitch_bytes * (int64_t)(H - 1) + W,
               "fx_hu_partial_u8: image stride %lld overlaps images", (long long)img_stride_bytes);
    FX_REQUIRE(thresh < 255, "fx_hu_partial_u8: thresh %d must be < 255 (or negative)", thresh);
    FX_REQUIRE(thresh < 0 || (on_value >= 1 && on_value <= 255), "fx_hu_partial_u8: on_value %d outside 1..255", on_value);
    hipStream_t s = (hipStream_t)stream;
    HuArgs A = {};
    A.img = d_img;
    A.B = B;
    A.H = H;
    A.W = W;
    A.pitch = pitch_bytes;
    A.stride = img_stride_bytes;
    A.thresh = thresh;
    A.on_value = on_value;
    A.partial = d_partial;
    A.y_offset = y_offset;
    const bool aligned = ((uintptr_t)d_img % 16 == 0) && (pitch_bytes % 16 == 0) && (img_stride_bytes % 16 == 0 || B == 1);
    const int mode = thresh < 0 ? MODE_GRAY : (thresh == 127 ? MODE_BIT7 : MODE_THRESH);
    // always tiled: the combine step is where the offset is applied, in 128-bit integers
    A.tiles_x = (W + TILE_W - 1) / TILE_W;
    A.tiles_y = (H + TILE_H - 1) / TILE_H;
    const int64_t blocks = B * A.tiles_x * A.tiles_y;
    FX_REQUIRE(blocks <= 0x7fffffffLL, "fx_hu_partial_u8: too many tiles for one launch");
    A.cpr = FX_WAVE;
    A.bpw = 1;
    A.nbands = FX_WAVES_PER_BLOCK;
    A.hb = TILE_H / FX_WAVES_PER_BLOCK;
    A.qcap = 0;
    A.wave_lds = BAND_SLAB_BYTES;
    BandScratch G = {};
    FxCallGuard guard(s);
    G.sums = (u64 *)guard.scratch((size_t)blocks * BAND_SUMS * sizeof(u64), 1);
    if (!G.sums)
        return FX_ENOMEM;
    int rc;
    {
        FxProfileScope prof(s);
        rc = launch_band(A, G, mode, false, band_aligned(aligned, W, H, pitch_bytes, A.hb),
                         (size_t)BAND_LDS_FIXED + (size_t)FX_WAVES_PER_BLOCK * A.wave_lds, s);
    }
    if (rc != FX_OK)
        return rc;
    hipLaunchKernelGGL(hu_tile_combine_kernel, dim3((unsigned)B), dim3(FX_BLOCK), 0, s, A, G,
                       mode == MODE_GRAY ? 1 : 0);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

extern "C" int fx_hu_from_partial(const int64_t *d_partial, int64_t B, double *d_moments24, double *d_hu7,
                                  float *d_hulog7, void *stream)
{
    FX_REQUIRE(B >= 0, "fx_hu_from_partial: B = %lld", (long long)B);
    if (B == 0)
        return FX_OK;
    FX_REQUIRE(d_partial != nullptr, "fx_hu_from_partial: d_partial is NULL");
    FX_REQUIRE(d_moments24 || d_hu7 || d_hulog7, "fx_hu_from_partial: no output requested");
    hipLaunchKernelGGL(hu_from_partial_kernel, dim3((unsigned)((B + FX_BLOCK - 1) / FX_BLOCK)), dim3(FX_BLOCK), 0,
                       (hipStream_t)stream, d_partial, B, d_moments24, d_hu7, d_hulog7);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

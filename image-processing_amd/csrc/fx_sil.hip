// libfx: silhouette preprocessing of gray views, the step in front of the Hu
// moments (reference src/features/extract_2d_hu.py: compute_otsu :129-188,
// find_largest_bbox :191-221, clean_mask_np :246-346, preprocess_image
// :352-440 -- the branch the reference takes when OpenCV is not installed:
// PIL GaussianBlur(radius=1), NumPy Otsu, scipy.ndimage 3x3 closing/opening,
// bounding-box crop, PIL nearest resize, 3x3 opening, 4-connected component
// size filter, centred paste).  Everything is integer or bit work except the
// Otsu criterion and the resize geometry, which are IEEE float64 in the
// reference's operation order.
//
//   sil_blur_kernel   a block per tile of rows: the six 3-tap passes of PIL's
//                     radius-1 Gaussian (three along x in registers, three
//                     along y from an LDS tile), blurred bytes + 256-bin
//                     histogram to scratch.
//   sil_mask_kernel   a block per image: Otsu level, bit mask (32 pixels per
//                     word) in LDS, bit-parallel morphology, crop geometry,
//                     nearest resize, run-based union-find labelling, size
//                     filter, 0/255 canvas.
#include "fx_common.h"

namespace {

constexpr int SIL_HALO = 3;         // three 3-tap passes
constexpr int SIL_MAX_TROWS = 64;   // rows of a blur tile, halo included
constexpr int SIL_MAX_SIDE = 512;
constexpr int SIL_RUNS_LDS = 1024;  // run table kept in LDS up to this many runs
constexpr int SIL_MAX_GRID = 1024;  // mask kernel grid when the occupancy query fails

// PIL's box blur for GaussianBlur(radius=1): three passes of a box of radius
// 0.25, i.e. taps (fw, ww, fw) / 2^24 with ww = uint32(float32(2^24) /
// float32(1.5)) and fw = (2^24 - ww) / 2, rounded to uint8 after every pass,
// edge pixels replicated.
constexpr uint32_t SIL_WW = 11184811u;
constexpr uint32_t SIL_FW = 2796202u;

// Division of small non-negative numbers (below 2^23) by a block-uniform divisor of at most 512 that is not
// known at compile time: a multiplication by ceil(2^32 / d) instead of the twenty-odd instructions of a
// 32-bit division, which the word loops below would otherwise pay once per mask word and pass.
struct SilDiv {
    uint32_t d, magic;
};
__device__ __forceinline__ SilDiv sil_div_make(int d)
{
    SilDiv v;
    v.d = (uint32_t)d;
    v.magic = d > 1 ? 0xFFFFFFFFu / (uint32_t)d + 1u : 0u;
    return v;
}
__device__ __forceinline__ int sil_div(int n, SilDiv v) { return v.d > 1 ? (int)__umulhi((uint32_t)n, v.magic) : n; }

struct SilArgs {
    const uint8_t *img;
    int64_t B;
    int H, W;
    int64_t pitch, stride;
    int res;
    double crop_margin;
    int crop;
    int min_size;
    uint8_t *mask;    // [B][res][res]
    int32_t *info;    // [B][8] or null
    uint8_t *blur;    // [B][H][W]
    uint32_t *hist;   // [B][256]
    uint32_t *runs;   // [grid][2 * run_cap], used when a mask has more than SIL_RUNS_LDS runs
    int run_cap;
    int wpr;          // words per row of the LDS masks
    int rows;         // rows of the LDS masks
    int tr;           // output rows per blur tile
    int mode;         // FX_SIL_PIL or FX_SIL_OPENCV: which branch of the reference is reproduced
    int32_t *level;   // [B] Otsu level, written by sil_otsu_cv_kernel (OpenCV branch)
    int64_t *partial; // [B][10][3] or null: raw moments of the 0 / 255 silhouette as 48-bit limbs
};

__device__ __forceinline__ uint32_t sil_tap(uint32_t l, uint32_t c, uint32_t r)
{
    return (c * SIL_WW + (l + r) * SIL_FW + (1u << 23)) >> 24;
}

// The three passes along a row on 22 pixels held in registers (16 + 3 on each
// side, v[k] is pixel x0 - 3 + k).  The neighbour of the row's first or last
// pixel is the pixel itself.  WHOLE: the row is a whole number of 16-pixel
// chunks, so its first pixel can only be v[3] and its last only v[18]; every
// lane runs the same code, the edge lanes differ by two selects per pass.
template <bool WHOLE>
__device__ __forceinline__ void sil_row_passes(uint32_t (&v)[22], int x0, int W)
{
    const bool first = x0 == 0, last = x0 + 16 == W;
#pragma unroll
    for (int p = 0; p < 3; p++) {
        uint32_t n[22];
#pragma unroll
        for (int k = 1 + p; k <= 20 - p; k++) {
            uint32_t l = v[k - 1], r = v[k + 1];
            if (WHOLE) {
                if (k == 3)
                    l = first ? v[k] : l;
                if (k == 18)
                    r = last ? v[k] : r;
            } else {
                const int x = x0 - 3 + k;
                l = x - 1 < 0 ? v[k] : l;
                r = x + 1 > W - 1 ? v[k] : r;
            }
            n[k] = sil_tap(l, v[k], r);
        }
#pragma unroll
        for (int k = 1 + p; k <= 20 - p; k++)
            v[k] = n[k];
    }
}

template <bool ALIGNED>
__global__ __launch_bounds__(FX_BLOCK) void sil_blur_kernel(SilArgs A)
{
    extern __shared__ __align__(16) uint8_t sil_smem[];
    const int Wp = (A.W + 15) & ~15;
    const int trows = A.tr + 2 * SIL_HALO;
    uint8_t *tile = sil_smem;                                // [trows][Wp]
    uint32_t *hist = (uint32_t *)(sil_smem + trows * Wp);    // [256]
    const int64_t b = blockIdx.y;
    const int r0 = blockIdx.x * A.tr;
    const uint8_t *src = A.img + b * A.stride;
    const int cpr = Wp >> 4;
    hist[threadIdx.x] = 0;
    // tile row i holds image row r0 - SIL_HALO + i; rows [ilo, ihi] exist
    const int ilo = SIL_HALO - r0 > 0 ? SIL_HALO - r0 : 0;
    const int ihi = A.H - 1 - r0 + SIL_HALO < trows - 1 ? A.H - 1 - r0 + SIL_HALO : trows - 1;

    // three passes along x: 16 pixels + 3 on each side in registers
    const SilDiv dv_cpr = sil_div_make(cpr);
    for (int it = threadIdx.x; it < trows * cpr; it += FX_BLOCK) {
        const int i = sil_div(it, dv_cpr), c = it - i * cpr;
        if (i < ilo || i > ihi)
            continue;
        const uint8_t *row = src + (int64_t)(r0 - SIL_HALO + i) * A.pitch;
        const int x0 = c * 16;
        uint32_t v[22];
        if (ALIGNED) {
            const uint4 m = *(const uint4 *)(row + x0);
            const uint32_t l = x0 > 0 ? *(const uint32_t *)(row + x0 - 4) : 0u;
            const uint32_t r = x0 + 16 < A.W ? *(const uint32_t *)(row + x0 + 16) : 0u;
            v[0] = (l >> 8) & 255u;
            v[1] = (l >> 16) & 255u;
            v[2] = l >> 24;
            const uint32_t mw[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
            for (int k = 0; k < 16; k++)
                v[3 + k] = (mw[k >> 2] >> (8 * (k & 3))) & 255u;
            v[19] = r & 255u;
            v[20] = (r >> 8) & 255u;
            v[21] = (r >> 16) & 255u;
        } else {
#pragma unroll
            for (int k = 0; k < 22; k++) {
                int x = x0 - 3 + k;
                x = x < 0 ? 0 : (x > A.W - 1 ? A.W - 1 : x);
                v[k] = row[x];
            }
        }
        sil_row_passes<ALIGNED>(v, x0, A.W);
        uint4 o;
        o.x = v[3] | (v[4] << 8) | (v[5] << 16) | (v[6] << 24);
        o.y = v[7] | (v[8] << 8) | (v[9] << 16) | (v[10] << 24);
        o.z = v[11] | (v[12] << 8) | (v[13] << 16) | (v[14] << 24);
        o.w = v[15] | (v[16] << 8) | (v[17] << 16) | (v[18] << 24);
        *(uint4 *)(tile + i * Wp + x0) = o;
    }
    __syncthreads();

    // three passes along y: a thread per column walks down the tile; the
    // passes run as a pipeline, pass k one row behind pass k - 1.  Row i of
    // the result needs rows i - 3 .. i + 3 of the tile, or fewer at the image's
    // first and last row, where a pixel stands in for its missing neighbour.
    uint8_t *dst = A.blur + b * (int64_t)A.H * A.W;
    const int olo = ilo > SIL_HALO ? ilo : SIL_HALO;                        // rows this tile writes
    const int ohi = ihi < trows - 1 - SIL_HALO ? ihi : trows - 1 - SIL_HALO;
    // steady state: every row a step touches exists and its output row is the tile's own
    const int f0 = (ilo + 3 > olo + 2 ? ilo + 3 : olo + 2), f1 = (ihi - 2 < ohi + 2 ? ihi - 2 : ohi + 2);
    for (int x = threadIdx.x; x < A.W; x += FX_BLOCK) {
        const uint8_t *col = tile + x;
        uint8_t *out = dst + (int64_t)(r0 - SIL_HALO) * A.W + x;
        uint32_t c0 = 0, c1 = col[ilo * Wp], c2 = ilo + 1 <= ihi ? col[(ilo + 1) * Wp] : c1;  // input rows i-1, i, i+1
        uint32_t p1a = 0, p1b = 0, p1c = 0;   // pass 1 at rows i-2, i-1, i
        uint32_t p2a = 0, p2b = 0, p2c = 0;   // pass 2 at rows i-3, i-2, i-1
        uint32_t run_v = 0, run_n = 0;
        // neighbouring rows mostly share a value: one LDS atomic per stretch
        auto emit = [&](int row, uint32_t v) {
            out[(int64_t)row * A.W] = (uint8_t)v;
            if (v != run_v && run_n) {
                atomicAdd(&hist[run_v], run_n);
                run_n = 0;
            }
            run_v = v;
            run_n++;
        };
        auto edge_step = [&](int i) {
            const uint32_t c3 = i + 2 <= ihi ? col[(i + 2) * Wp] : 0u;  // prefetch of row i + 2
            if (i <= ihi) {
                const uint32_t up = i - 1 < ilo ? c1 : c0, dn = i + 1 > ihi ? c1 : c2;
                p1a = p1b;
                p1b = p1c;
                p1c = sil_tap(up, c1, dn);
            }
            if (i - 1 >= ilo && i - 1 <= ihi) {
                // pass 2 at row i - 1 from pass 1 at rows i - 2, i - 1, i
                const uint32_t mid = i <= ihi ? p1b : p1c;
                const uint32_t above = i <= ihi ? p1a : p1b;
                const uint32_t up = i - 2 < ilo ? mid : above, dn = i > ihi ? mid : p1c;
                p2a = p2b;
                p2b = p2c;
                p2c = sil_tap(up, mid, dn);
            }
            if (i - 2 >= olo && i - 2 <= ohi) {
                // pass 3 at row i - 2 from pass 2 at rows i - 3, i - 2, i - 1
                const uint32_t mid = i - 1 <= ihi ? p2b : p2c;
                const uint32_t above = i - 1 <= ihi ? p2a : p2b;
                const uint32_t up = i - 3 < ilo ? mid : above, dn = i - 1 > ihi ? mid : p2c;
                emit(i - 2, sil_tap(up, mid, dn));
            }
            c0 = c1;
            c1 = c2;
            c2 = c3;
        };
        int i = ilo;
        for (; i <= ihi + 2 && i < f0; i++)
            edge_step(i);
#pragma unroll 6
        for (; i <= f1; i++) {
            const uint32_t c3 = col[(i + 2) * Wp];
            p1a = p1b;
            p1b = p1c;
            p1c = sil_tap(c0, c1, c2);
            p2a = p2b;
            p2b = p2c;
            p2c = sil_tap(p1a, p1b, p1c);
            emit(i - 2, sil_tap(p2a, p2b, p2c));
            c0 = c1;
            c1 = c2;
            c2 = c3;
        }
        for (; i <= ihi + 2; i++)
            edge_step(i);
        if (run_n)
            atomicAdd(&hist[run_v], run_n);
    }
    __syncthreads();
    const uint32_t h = hist[threadIdx.x];
    if (h)
        atomicAdd(&A.hist[b * 256 + threadIdx.x], h);
}

// ---------------------------------------------------------------------------
// OpenCV branch, blur: cv2.GaussianBlur(img, (5, 5), 0) on 8-bit data is the
// table kernel [1 4 6 4 1] / 16 in fixed point: exact sums along x (at most
// 255 * 16), then along y with one rounding, (sum + 128) >> 8, borders
// reflected without repeating the edge pixel (BORDER_REFLECT_101).
// ---------------------------------------------------------------------------
constexpr int SIL_CV_HALO = 2;
// mask words (32 blurred pixels each) a thread of the mask kernel requests together when it thresholds a view:
// (seven would make a 224 x 224 view one wait for global memory instead of two; measured: no faster)
constexpr int SIL_TH_UNR = 4;

__device__ __forceinline__ int sil_refl101(int i, int n)
{
    if (n == 1)
        return 0;
    while (i < 0 || i >= n)
        i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

constexpr int SIL_CV_HIST_COPIES = 4, SIL_CV_HIST_PITCH = 257;

// a + e + 4 (b + d) + 6 c + k in four full-rate instructions (two three-operand additions, two shift-and-adds);
// written as a product, 6 c becomes a 32-bit multiply, which issues at a quarter of the rate.
__device__ __forceinline__ uint32_t sil_taps5(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t k)
{
    const uint32_t inner = b + d + c, outer = a + e + k;
    return (inner << 2) + ((c << 1) + outer);
}

template <bool ALIGNED>
__global__ __launch_bounds__(FX_BLOCK) void sil_blur_cv_kernel(SilArgs A)
{
    extern __shared__ __align__(16) uint8_t sil_smem[];
    const int W = A.W, H = A.H;
    const int Wp = (W + 15) & ~15;
    const int cpr = Wp >> 4;
    const int64_t b = blockIdx.y;
    const int r0 = blockIdx.x * A.tr;
    const int nrows = H - r0 < A.tr ? H - r0 : A.tr;           // rows this tile writes
    const int trows = nrows + 2 * SIL_CV_HALO;
    uint16_t *tile = (uint16_t *)sil_smem;                                        // [tr + 4][Wp] sums along x
    // the histogram in SIL_CV_HIST_COPIES copies, lane l counting in copy l % copies: neighbouring pixels mostly share
    // a few gray levels, and lanes adding to the same word serialise; the pitch puts a level's copies into different banks
    uint32_t *hist = (uint32_t *)(sil_smem + (size_t)(A.tr + 2 * SIL_CV_HALO) * Wp * 2);  // [copies][257]
    uint32_t *my_hist = hist + (threadIdx.x % SIL_CV_HIST_COPIES) * SIL_CV_HIST_PITCH;
    const uint8_t *src = A.img + b * A.stride;
    for (int k = threadIdx.x; k < SIL_CV_HIST_COPIES * SIL_CV_HIST_PITCH; k += FX_BLOCK)
        hist[k] = 0;
    // a tile row keeps the first eight sums of every 16-pixel chunk side by side, then the second eight: the two
    // 16-byte stores of the row pass are then contiguous over the lanes
    const int half_row = Wp >> 1;

    // along x: 16 pixels + 2 on each side in registers; tile row i is image row reflect(r0 - 2 + i)
    const SilDiv dv_cpr = sil_div_make(cpr);
    if (ALIGNED) {   // W is a multiple of 16 (so at least 16) and rows are 16-byte aligned
        // A block lives for a few microseconds, most of them waiting: the loads of two rounds of the loop are
        // issued together, so that a block waits for global memory once instead of twice.
        constexpr int XR = 2;
        for (int it0 = threadIdx.x; it0 < trows * cpr; it0 += XR * FX_BLOCK) {
            uint4 mm[XR];
            uint32_t ll[XR], rr[XR];
            int ti[XR], tc[XR];
#pragma unroll
            for (int u = 0; u < XR; u++) {
                const int it = it0 + u * FX_BLOCK;
                mm[u] = make_uint4(0u, 0u, 0u, 0u);
                ll[u] = rr[u] = 0u;
                ti[u] = -1;
                tc[u] = 0;
                if (it < trows * cpr) {
                    const int i = sil_div(it, dv_cpr), c = it - i * cpr;
                    const uint8_t *row = src + (int64_t)sil_refl101(r0 - SIL_CV_HALO + i, H) * A.pitch;
                    const int x0 = c * 16;
                    ti[u] = i;
                    tc[u] = c;
                    mm[u] = *(const uint4 *)(row + x0);
                    if (x0 > 0)
                        ll[u] = *(const uint32_t *)(row + x0 - 4);
                    if (x0 + 16 < W)
                        rr[u] = *(const uint32_t *)(row + x0 + 16);
                }
            }
#pragma unroll
            for (int u = 0; u < XR; u++) {
                if (ti[u] < 0)
                    continue;
                const int i = ti[u], c = tc[u], x0 = c * 16;
                uint4 *dst_lo = (uint4 *)(tile + (size_t)i * Wp + 8 * c);
                uint4 *dst_hi = (uint4 *)(tile + (size_t)i * Wp + half_row + 8 * c);
                // Pairs of neighbouring pixels as two 16-bit fields of a word, P[j] = (pixel j - 2, pixel j - 1)
                // counted from the chunk's first pixel: one byte permute each.  A sum of the five taps is at most
                // 255 * 16, so plain 32-bit arithmetic on such words never carries from one field into the other
                // and two output pixels cost four instructions.
                const uint4 m = mm[u];
                // left word: bytes 2, 3 = pixels x0 - 2, x0 - 1 (reflected at the row's start: pixels 2, 1)
                const uint32_t l = x0 > 0 ? ll[u] : __builtin_amdgcn_perm(0u, m.x, 0x01020c0cu);
                // right word: bytes 0, 1 = pixels x0 + 16, x0 + 17 (reflected at the row's end: pixels 14, 13)
                const uint32_t r = x0 + 16 < W ? rr[u] : __builtin_amdgcn_perm(0u, m.w, 0x0c0c0102u);
                const uint32_t wd[6] = {l, m.x, m.y, m.z, m.w, r};
                uint32_t P[19];
#pragma unroll
                for (int j = 0; j < 19; j++) {
                    // first pixel of the pair is byte (j + 2) & 3 of word (j + 2) >> 2
                    const int w0 = (j + 2) >> 2, b0 = (j + 2) & 3;
                    P[j] = b0 == 0   ? __builtin_amdgcn_perm(0u, wd[w0], 0x0c010c00u)
                           : b0 == 1 ? __builtin_amdgcn_perm(0u, wd[w0], 0x0c020c01u)
                           : b0 == 2 ? __builtin_amdgcn_perm(0u, wd[w0], 0x0c030c02u)
                                     : __builtin_amdgcn_perm(wd[w0 + 1], wd[w0], 0x0c040c03u);
                }
                uint32_t o[8];
#pragma unroll
                for (int k = 0; k < 8; k++)
                    o[k] = sil_taps5(P[2 * k], P[2 * k + 1], P[2 * k + 2], P[2 * k + 3], P[2 * k + 4], 0u);
                *dst_lo = make_uint4(o[0], o[1], o[2], o[3]);
                *dst_hi = make_uint4(o[4], o[5], o[6], o[7]);
            }
        }
    }
    for (int it = threadIdx.x; !ALIGNED && it < trows * cpr; it += FX_BLOCK) {
        const int i = sil_div(it, dv_cpr), c = it - i * cpr;
        const uint8_t *row = src + (int64_t)sil_refl101(r0 - SIL_CV_HALO + i, H) * A.pitch;
        const int x0 = c * 16;
        uint4 *dst_lo = (uint4 *)(tile + (size_t)i * Wp + 8 * c);
        uint4 *dst_hi = (uint4 *)(tile + (size_t)i * Wp + half_row + 8 * c);
        {
            uint32_t v[20];
#pragma unroll
            for (int k = 0; k < 20; k++)
                v[k] = row[sil_refl101(x0 - SIL_CV_HALO + k, W)];
            uint32_t o[8];
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                const uint32_t lo = v[k] + v[k + 4] + 4u * (v[k + 1] + v[k + 3]) + 6u * v[k + 2];
                const uint32_t hi = v[k + 1] + v[k + 5] + 4u * (v[k + 2] + v[k + 4]) + 6u * v[k + 3];
                o[k >> 1] = lo | (hi << 16);
            }
            *dst_lo = make_uint4(o[0], o[1], o[2], o[3]);
            *dst_hi = make_uint4(o[4], o[5], o[6], o[7]);
        }
    }
    __syncthreads();

    // along y: a thread takes eight neighbouring columns (one 16-byte half chunk of the tile: four packed pairs of
    // 16-bit sums; the largest total, 255 * 256 + 128, still fits 16 bits, so both halves of a word are summed
    // at once) and a stretch of rows.  Lane j of a row takes the tile's j-th half chunk, so a wave's reads are
    // contiguous.
    const int octs = Wp >> 3;
    const SilDiv dv_octs = sil_div_make(octs);
    int segs = FX_BLOCK / octs;
    segs = segs < 1 ? 1 : segs;
    const int rps = (nrows + segs - 1) / segs;
    uint8_t *out = A.blur + b * (int64_t)H * W;
    const bool row_qwords = (W & 7) == 0;
    for (int it = threadIdx.x; it < octs * segs; it += FX_BLOCK) {
        const int sg = sil_div(it, dv_octs), j = it - sg * octs, y0 = sg * rps;
        const int y1 = y0 + rps < nrows ? y0 + rps : nrows;
        const int x = j < cpr ? 16 * j : 16 * (j - cpr) + 8;   // first column of the half chunk
        if (y0 >= y1 || x >= W)
            continue;
        const uint16_t *col = tile + 8 * j;
        auto ld = [&](int i) { return *(const uint4 *)(col + (size_t)i * Wp); };
        // Four output rows per round, the four rows they bring into the window read together and ahead of the
        // arithmetic (the histogram's LDS atomics would otherwise keep every read behind them).
        const int last = trows - 1;
        auto ldc = [&](int i) { return ld(i < last ? i : last); };   // (a row past the tile is read for a step that does not run)
        uint4 c[8];
        c[0] = ld(y0), c[1] = ld(y0 + 1), c[2] = ld(y0 + 2), c[3] = ld(y0 + 3);
        uint32_t prev_lo = 0, prev_hi = 0, rep = 0;
        const int live = W - x;  // columns of this half chunk inside the image
        auto flush = [&]() {
            if (!rep)
                return;
            const uint32_t p0 = prev_lo & 255u;
            if (live >= 8 && prev_lo == __builtin_amdgcn_perm(0u, prev_lo, 0u) && prev_hi == prev_lo) {   // eight equal bytes
                atomicAdd(&my_hist[p0], 8u * rep);
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (k < live)
                        atomicAdd(&my_hist[((k < 4 ? prev_lo : prev_hi) >> (8 * (k & 3))) & 255u], rep);
            }
            rep = 0;
        };
        // the high bytes of the four 16-bit fields of (ta, tb): one byte permute
        auto pack = [](uint32_t ta, uint32_t tb) { return __builtin_amdgcn_perm(tb, ta, 0x07050301u); };
        uint8_t *po = out + (int64_t)(r0 + y0) * W + x;
        for (int y = y0; y < y1; y += 4) {
#pragma unroll
            for (int q = 0; q < 4; q++)
                c[4 + q] = ldc(y + 4 + q);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (y + q < y1) {
                    const uint32_t ta = sil_taps5(c[q].x, c[q + 1].x, c[q + 2].x, c[q + 3].x, c[q + 4].x, 0x00800080u);
                    const uint32_t tb = sil_taps5(c[q].y, c[q + 1].y, c[q + 2].y, c[q + 3].y, c[q + 4].y, 0x00800080u);
                    const uint32_t tc = sil_taps5(c[q].z, c[q + 1].z, c[q + 2].z, c[q + 3].z, c[q + 4].z, 0x00800080u);
                    const uint32_t td = sil_taps5(c[q].w, c[q + 1].w, c[q + 2].w, c[q + 3].w, c[q + 4].w, 0x00800080u);
                    const uint32_t o_lo = pack(ta, tb), o_hi = pack(tc, td);
                    if (row_qwords) {
                        *(uint2 *)po = make_uint2(o_lo, o_hi);
                    } else {
                        for (int k = 0; k < 8 && k < live; k++)
                            po[k] = (uint8_t)((k < 4 ? o_lo : o_hi) >> (8 * (k & 3)));
                    }
                    if (o_lo != prev_lo || o_hi != prev_hi) {
                        flush();
                        prev_lo = o_lo;
                        prev_hi = o_hi;
                    }
                    rep++;
                    po += W;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
                c[q] = c[4 + q];
        }
        flush();
    }
    __syncthreads();
    uint32_t h = 0;
#pragma unroll
    for (int k = 0; k < SIL_CV_HIST_COPIES; k++)
        h += hist[k * SIL_CV_HIST_PITCH + threadIdx.x];
    if (h)
        atomicAdd(&A.hist[b * 256 + threadIdx.x], h);
}

// OpenCV's Otsu level (getThreshVal_Otsu_8u): the class mean is carried from
// level to level in double, so the 256 steps are a serial chain of roundings;
// a lane per image runs it, 64 histograms staged through LDS per block.
constexpr int SIL_OTSU_PITCH = 257;
__global__ __launch_bounds__(FX_BLOCK) void sil_otsu_cv_kernel(const uint32_t *hist, int64_t B, double scale, int32_t *level)
{
    extern __shared__ __align__(16) uint8_t sil_smem[];
    uint32_t *h = (uint32_t *)sil_smem;  // [64][257]
    const int64_t b0 = (int64_t)blockIdx.x * FX_WAVE;
    const int n = B - b0 < FX_WAVE ? (int)(B - b0) : FX_WAVE;
    // all four waves stage the 64 histograms (16 bytes per load, every round's loads in flight together); the
    // first wave then runs the chains
    const uint4 *src = (const uint4 *)(hist + b0 * 256);
#pragma unroll 4
    for (int i = threadIdx.x; i < n * 64; i += FX_BLOCK) {
        const uint4 q = src[i];
        uint32_t *dst = h + (i >> 6) * SIL_OTSU_PITCH + 4 * (i & 63);
        dst[0] = q.x, dst[1] = q.y, dst[2] = q.z, dst[3] = q.w;
    }
    __syncthreads();
    if ((int)threadIdx.x >= n)
        return;
    const uint32_t *mine = h + threadIdx.x * SIL_OTSU_PITCH;
    double mu = 0.0;
    for (int i = 0; i < 256; i++)
        mu += (double)i * (double)mine[i];
    mu *= scale;
    const double eps = 1.1920928955078125e-07;  // FLT_EPSILON
    double mu1 = 0.0, q1 = 0.0, max_sigma = 0.0;
    int max_val = 0;
    for (int i = 0; i < 256; i++) {
        const double p_i = (double)mine[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        const double q2 = 1.0 - q1;
        if (fmin(q1, q2) < eps || fmax(q1, q2) > 1.0 - eps)
            continue;
        mu1 = (mu1 + (double)i * p_i) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) {
            max_sigma = sigma;
            max_val = i;
        }
    }
    level[b0 + threadIdx.x] = max_val;
}

// ------------------------------------------------------------ mask kernel
// Word loops of a rows x cols bit mask: words per row, and for the morphology passes the split into
// strips of `rs` rows of one word column per thread.
struct SilGeom {
    int rows, cols, wc, strips, rs;
    SilDiv dv;
};
__device__ __forceinline__ SilGeom sil_geom(int rows, int cols)
{
    SilGeom g;
    g.rows = rows;
    g.cols = cols;
    g.wc = (cols + 31) >> 5;
    g.dv = sil_div_make(g.wc);
    g.strips = FX_BLOCK / g.wc;
    g.strips = g.strips < 1 ? 1 : g.strips;
    g.rs = (rows + g.strips - 1) / g.strips;
    return g;
}

// 3x3 dilation or erosion of a bit mask, scipy.ndimage semantics with
// border_value = 0: whatever lies outside the rows x cols rectangle is
// background for both operations.  Bits at or beyond `cols` stay zero.
template <bool DILATE>
__device__ void sil_morph(const uint32_t *s, uint32_t *d, const SilGeom &G, int wpr)
{
    const int rows = G.rows, cols = G.cols, wc = G.wc;
    const uint32_t lastmask = (cols & 31) ? ((1u << (cols & 31)) - 1u) : ~0u;
    // strips of `rs` rows of one word column per thread: a row's horizontal
    // combination is computed once and used for three output rows
    const int strips = G.strips, rs = G.rs;
    for (int it = threadIdx.x; it < wc * strips; it += FX_BLOCK) {
        const int sy = sil_div(it, G.dv), w = it - sy * wc, y0 = sy * rs;
        if (y0 >= rows)
            continue;
        const int y1 = y0 + rs < rows ? y0 + rs : rows;
        auto hrow = [&](int yy) -> uint32_t {
            if (yy < 0 || yy >= rows)
                return 0u;  // outside is background for both operations
            const uint32_t *r = s + yy * wpr;
            const uint32_t m = r[w];
            const uint32_t p = w > 0 ? r[w - 1] : 0u;
            const uint32_t n = w + 1 < wc ? r[w + 1] : 0u;
            const uint32_t L = (m << 1) | (p >> 31), R = (m >> 1) | (n << 31);
            return DILATE ? (m | L | R) : (m & L & R);
        };
        uint32_t ha = hrow(y0 - 1), hb = hrow(y0);
        for (int y = y0; y < y1; y++) {
            const uint32_t hc = hrow(y + 1);
            uint32_t acc = DILATE ? (ha | hb | hc) : (ha & hb & hc);
            if (w == wc - 1)
                acc &= lastmask;
            d[y * wpr + w] = acc;
            ha = hb;
            hb = hc;
        }
    }
    __syncthreads();
}

// OpenCV morphology on a bit mask: the 5 x 5 MORPH_ELLIPSE element (the centre
// column in rows -2 and +2, five wide in rows -1, 0, +1) or the 3 x 3 one (a
// cross), with cv2's default border: pixels outside the image never decide,
// i.e. they count as foreground for an erosion and as background for a
// dilation.  Bits at or beyond `cols` are zero in the result.
template <bool DILATE, bool FIVE>
__device__ void sil_morph_cv(const uint32_t *s, uint32_t *d, const SilGeom &G, int wpr)
{
    const int rows = G.rows, cols = G.cols, wc = G.wc;
    const uint32_t lastmask = (cols & 31) ? ((1u << (cols & 31)) - 1u) : ~0u;
    const uint32_t fill = DILATE ? 0u : ~0u;
    const int strips = G.strips, rs = G.rs;
    for (int it = threadIdx.x; it < wc * strips; it += FX_BLOCK) {
        const int sy = sil_div(it, G.dv), w = it - sy * wc, y0 = sy * rs;
        if (y0 >= rows)
            continue;
        const int y1 = y0 + rs < rows ? y0 + rs : rows;
        // what the thread's word column decides once: which neighbour words exist, and for an erosion the
        // bits beyond `cols` of the last word (outside the image: foreground)
        const bool has_p = w > 0, has_n = w + 1 < wc;
        const uint32_t fix_m = (!DILATE && w == wc - 1) ? ~lastmask : 0u;
        const uint32_t fix_n = (!DILATE && w + 1 == wc - 1) ? ~lastmask : 0u;
        const uint32_t *col = s + w;
        // the word of row yy (c) and its combination over the element's widest rows (wd)
        auto load = [&](int yy, uint32_t &c, uint32_t &wd) {
            uint32_t m = fill, p = fill, n = fill;
            if ((unsigned)yy < (unsigned)rows) {
                const uint32_t *r = col + yy * wpr;
                m = r[0] | fix_m;
                if (has_p)
                    p = r[-1];
                if (has_n)
                    n = r[1] | fix_n;
            }
            const uint32_t l1 = (m << 1) | (p >> 31), r1 = (m >> 1) | (n << 31);
            c = m;
            if (!FIVE) {
                wd = DILATE ? (m | l1 | r1) : (m & l1 & r1);
            } else {
                const uint32_t l2 = (m << 2) | (p >> 30), r2 = (m >> 2) | (n << 30);
                wd = DILATE ? (m | l1 | r1 | l2 | r2) : (m & l1 & r1 & l2 & r2);
            }
        };
        uint32_t unused;
        if (FIVE) {
            uint32_t cm2, cm1, c0, cp1, wm1, w0, wp1;
            load(y0 - 2, cm2, unused);
            load(y0 - 1, cm1, wm1);
            load(y0, c0, w0);
            load(y0 + 1, cp1, wp1);
            for (int y = y0; y < y1; y++) {
                uint32_t cp2, wp2;
                load(y + 2, cp2, wp2);
                uint32_t acc = DILATE ? (cm2 | wm1 | w0 | wp1 | cp2) : (cm2 & wm1 & w0 & wp1 & cp2);
                if (w == wc - 1)
                    acc &= lastmask;
                d[y * wpr + w] = acc;
                cm2 = cm1;
                cm1 = c0;
                c0 = cp1;
                cp1 = cp2;
                wm1 = w0;
                w0 = wp1;
                wp1 = wp2;
            }
        } else {
            uint32_t cm1, c0, w0;
            load(y0 - 1, cm1, unused);
            load(y0, c0, w0);
            for (int y = y0; y < y1; y++) {
                uint32_t cp1, wp1;
                load(y + 1, cp1, wp1);
                uint32_t acc = DILATE ? (cm1 | w0 | cp1) : (cm1 & w0 & cp1);
                if (w == wc - 1)
                    acc &= lastmask;
                d[y * wpr + w] = acc;
                cm1 = c0;
                c0 = cp1;
                w0 = wp1;
            }
        }
    }
    __syncthreads();
}

// Root of a run.  Parents only ever decrease, so a node may be pointed at its
// grandparent on the way up (path halving) while other lanes unite.
__device__ __forceinline__ uint32_t sil_find(uint32_t *parent, uint32_t a)
{
    volatile uint32_t *vp = parent;
    uint32_t p = vp[a];
    // indices strictly decrease on the way up, so `a` steps are the most a sound table allows: the bound makes
    // sure every wave leaves the loop whatever the table holds
    for (uint32_t left = a; p != a && left; left--) {
        const uint32_t g = vp[p];
        if (g != p)
            atomicMin(&parent[a], g);
        a = p;
        p = g;
    }
    return a;
}

// Lock-free union.  A root is the smallest index of its tree and may be hung
// below any smaller node of the other tree, so the usual case (a run and the
// run above it) is one read and one atomic, without climbing.
__device__ void sil_unite(uint32_t *parent, uint32_t a, uint32_t b)
{
    for (int tries = 0; tries < 64; tries++) {   // a retry means another lane linked `a` first: a handful at most
        a = sil_find(parent, a);
        if (a == b)
            return;
        if (a < b) {  // b may still belong to a's tree: climb from b instead
            const uint32_t t = a;
            a = b;
            b = t;
            continue;
        }
        const uint32_t old = atomicMin(&parent[a], b);
        if (old == a)
            return;
        a = old;  // somebody linked `a` first; join that tree with b as well
    }
}

// bits of m up to and including position `bit`
__device__ __forceinline__ uint32_t sil_le(int bit) { return bit >= 31 ? ~0u : ((2u << bit) - 1u); }

// Four bytes of d compared with k = 128 * kh + kl (1 <= k <= 255) at once:
// bit j of the result is set when byte j >= k.  kl4 = kl * 0x01010101.
__device__ __forceinline__ uint32_t sil_ge4(uint32_t d, uint32_t kl4, bool kh)
{
    const uint32_t t = ((d & 0x7f7f7f7fu) | 0x80808080u) - kl4;  // bit 7 of a byte: low 7 bits >= kl
    const uint32_t m = ((kh ? (d & t) : (d | t)) & 0x80808080u) >> 7;
    return (m | (m >> 7) | (m >> 14) | (m >> 21)) & 0xFu;
}

// Four mask bits to four bytes of 0 / 255.
__device__ __forceinline__ uint32_t sil_spread4(uint32_t nib)
{
    return (__umul24(nib, 0x00204081u) & 0x01010101u) * 255u;
}

// First pixels of the runs of word w of a mask row.
__device__ __forceinline__ uint32_t sil_starts(const uint32_t *row, int w)
{
    const uint32_t m = row[w], p = w > 0 ? row[w - 1] : 0u;
    return m & ~((m << 1) | (p >> 31));
}

// Index of the run that holds the set pixel (x, y); pre[y * wc + w] is the
// number of runs that start before word w of row y in raster order.
__device__ __forceinline__ uint32_t sil_run_at(const uint32_t *m, const uint32_t *pre, int y, int x, int wc, int wpr)
{
    const int w = x >> 5;
    return pre[y * wc + w] + __popc(sil_starts(m + y * wpr, w) & sil_le(x & 31)) - 1u;
}

struct SilRuns {
    uint32_t *parent;  // root of every run after sil_label (the smallest run index of its component)
    uint32_t *val;     // zeroed: per-root accumulator (size, flag or twice the contour area)
    int n;
};

// Connected components over the row runs of a bit mask (rows x wc words, row
// pitch wpr): a run is named by the raster-order rank of its first pixel, runs
// of neighbouring rows that touch (sharing a column, or for CONN8 also
// diagonally) are united lock-free, then every run is pointed at its root.
// Every thread of the block calls this; the result is block-uniform.
// Inlined into the kernel on purpose.  As a called function (what the compiler chooses when left alone) it made
// hipcc 7.2 park a caller-saved register around the call with a copy placed ahead of the `s_or_b64 exec` that
// re-enables the lanes which sat out the loop in front of the call: those lanes kept a stale copy, and from a
// block's second view on they addressed the histogram in LDS through it.  SIL_REPRO_MISCOMPILE rebuilds that
// variant (tools/isa_lint.py reports it; `make lint-selftest` checks that it still does).
#ifdef SIL_REPRO_MISCOMPILE
__attribute__((noinline))
#else
__forceinline__
#endif
__device__ SilRuns sil_label(const uint32_t *m, int rows, int wc, SilDiv dv, int wpr, bool conn8, uint32_t *pre, uint32_t *s_scan,
                             int *s_nruns, uint32_t *lds_runs, uint32_t *glob_runs, int run_cap)
{
    const int t = threadIdx.x;
    const int nw = rows * wc;
    const int per = (nw + FX_BLOCK - 1) / FX_BLOCK;
#ifdef SIL_TIMING
    __shared__ unsigned long long s_lt[8];
#define SIL_LTICK(k) do { if (t == 0 && blockIdx.x == 0) s_lt[k] = __builtin_readcyclecounter(); } while (0)
#else
#define SIL_LTICK(k)
#endif
    SIL_LTICK(0);
    __syncthreads();
    // (a thread's run counts are kept for the second loop when they are few: 224 x 224 views have seven words
    // per thread)
    constexpr int KEEP = 8;
    uint32_t kept[KEEP];
    {
        uint32_t local = 0;
        if (per <= KEEP) {
#pragma unroll
            for (int j = 0; j < KEEP; j++) {
                const int wi = t * per + j;
                kept[j] = 0;
                if (j < per && wi < nw) {
                    const int y = sil_div(wi, dv), w = wi - y * wc;
                    kept[j] = __popc(sil_starts(m + y * wpr, w));
                }
                local += kept[j];
            }
        } else {
            for (int j = 0; j < per; j++) {
                const int wi = t * per + j;
                if (wi < nw) {
                    const int y = sil_div(wi, dv), w = wi - y * wc;
                    local += __popc(sil_starts(m + y * wpr, w));
                }
            }
        }
        // exclusive prefix over the block: wave scan, then the waves in front
        uint32_t inc = local;
#pragma unroll
        for (int o = 1; o < FX_WAVE; o <<= 1) {
            const uint32_t u = __shfl_up(inc, o, FX_WAVE);
            if (fx_lane() >= o)
                inc += u;
        }
        if (fx_lane() == FX_WAVE - 1)
            s_scan[fx_wave()] = inc;
        s_scan[16 + t] = inc - local;
    }
    __syncthreads();
    {
        uint32_t base = s_scan[16 + t], total = 0;
#pragma unroll
        for (int w = 0; w < FX_WAVES_PER_BLOCK; w++) {
            base += w < fx_wave() ? s_scan[w] : 0u;
            total += s_scan[w];
        }
        if (per <= KEEP) {
#pragma unroll
            for (int j = 0; j < KEEP; j++) {
                const int wi = t * per + j;
                if (j < per && wi < nw) {
                    pre[wi] = base;
                    base += kept[j];
                }
            }
        } else {
            for (int j = 0; j < per; j++) {
                const int wi = t * per + j;
                if (wi < nw) {
                    const int y = sil_div(wi, dv), w = wi - y * wc;
                    pre[wi] = base;
                    base += __popc(sil_starts(m + y * wpr, w));
                }
            }
        }
        if (t == 0)
            *s_nruns = (int)total;
    }
    __syncthreads();
    SilRuns R;
    R.n = *s_nruns;
    R.parent = R.n <= SIL_RUNS_LDS ? lds_runs : glob_runs;
    R.val = R.n <= SIL_RUNS_LDS ? lds_runs + SIL_RUNS_LDS : glob_runs + run_cap;
    uint32_t *parent = R.parent;
    for (int r = t; r < R.n; r += FX_BLOCK) {
        parent[r] = (uint32_t)r;
        R.val[r] = 0u;
    }
    __syncthreads();
    // Two rounds.  First every run is hung below the run above its first contact (a plain store: only this
    // thread writes the run's parent), which leaves chains as long as the shape is tall; pointer jumping
    // flattens them.  Then the remaining contacts (a run that touches several runs above) are united on the
    // flat forest, where a root is one step away, and the few new links are flattened again.  Uniting
    // everything at once instead makes single threads climb chains of hundreds of links: a run below a
    // tall shape that joins the background on its left to the background on its right.
    // Round two only has something to do in a word where a run met a second run above, or went on meeting runs
    // after its first word: round one notes those words (one bit per loop step; at most 32 steps at 512 x 512).
    uint32_t again = 0u;
    auto contacts = [&](bool first_round) {
        int step = 0;
        for (int wi = t + wc; wi < nw; wi += FX_BLOCK, step++) {
            if (!first_round && !((again >> (step & 31)) & 1u))
                continue;
            int links = 0;
            const int y = sil_div(wi, dv), w = wi - y * wc;
            const uint32_t *row = m + y * wpr, *above = row - wpr;
            const uint32_t cur = row[w], u = above[w];
            if (!cur)
                continue;
            const uint32_t mp = w > 0 ? row[w - 1] : 0u, up = w > 0 ? above[w - 1] : 0u;
            const uint32_t sm = cur & ~((cur << 1) | (mp >> 31)), su = u & ~((u << 1) | (up >> 31));
            const uint32_t pm = pre[wi], pu = pre[wi - wc];
            auto link = [&](uint32_t a, uint32_t b) {   // run a of this row touches run b of the row above
                if (first_round) {
                    // the first contact of a run in raster order: the lowest position claims the run
                    atomicMin(&R.val[a], b);
                    links++;
                } else if (R.val[a] != b) {
                    sil_unite(parent, a, b);
                }
            };
            const uint32_t o = cur & u, op = mp & up;
            uint32_t os = o & ~((o << 1) | (op >> 31));  // first pixel of every stretch of overlap
            while (os) {
                const int bit = __ffs((int)os) - 1;
                os &= os - 1;
                const uint32_t le = sil_le(bit);
                link(pm + __popc(sm & le) - 1u, pu + __popc(su & le) - 1u);
            }
            if (conn8) {
                const uint32_t mn = w + 1 < wc ? row[w + 1] : 0u, un = w + 1 < wc ? above[w + 1] : 0u;
                const uint32_t mL = (cur << 1) | (mp >> 31), mR = (cur >> 1) | (mn << 31);
                const uint32_t uL = (u << 1) | (up >> 31), uR = (u >> 1) | (un << 31);
                uint32_t dl = cur & ~mL & uL & ~u;  // a run starts here; the row above has the pixel to the left only
                uint32_t dr = cur & ~mR & uR & ~u;  // a run ends here; the row above has the pixel to the right only
                while (dl) {
                    const int bit = __ffs((int)dl) - 1;
                    dl &= dl - 1;
                    link(pm + __popc(sm & sil_le(bit)) - 1u, sil_run_at(m, pre, y - 1, 32 * w + bit - 1, wc, wpr));
                }
                while (dr) {
                    const int bit = __ffs((int)dr) - 1;
                    dr &= dr - 1;
                    link(pm + __popc(sm & sil_le(bit)) - 1u, sil_run_at(m, pre, y - 1, 32 * w + bit + 1, wc, wpr));
                }
            }
            // a single contact of a run that starts in this word is that run's first one (contacts further right, in
            // this word or the next, meet runs with higher indices): nothing is left for round two
            if (first_round && (links >= 2 || (links == 1 && (cur & 1u) && (mp >> 31))))
                again |= 1u << (step & 31);
        }
    };
    // deep: the chains may be as long as the shape is tall (after round one); otherwise a link or two (after round
    // two), which a round or two of pointer jumping settles
    auto flatten = [&](bool deep) {
#ifndef SIL_FLATTEN_JUMPING
        if (deep && R.n <= SIL_RUNS_LDS) {
            // Every run to its root in one sweep by the first wave.  A parent is always a smaller index, so with the
            // runs taken 64 at a time in rising order every parent outside the window is a root already (one look-up);
            // inside the window the lanes jump over each other's pointers in registers, six rounds for 64 runs.  No
            // block barrier per round, and the rounds do not grow with the height of the shape.
            if (t < FX_WAVE) {
                for (int base = 0; base < R.n; base += FX_WAVE) {
                    const int r = base + t;
                    uint32_t p = r < R.n ? parent[r] : (uint32_t)r;
#pragma unroll
                    for (int round = 0; round < 6; round++) {
                        const uint32_t q = (uint32_t)__shfl((int)p, (int)(p - (uint32_t)base) & (FX_WAVE - 1), FX_WAVE);
                        p = p >= (uint32_t)base ? q : p;
                    }
                    if (p < (uint32_t)base)
                        p = parent[p];
                    if (r < R.n)
                        parent[r] = p;
                }
            }
            __threadfence_block();
            __syncthreads();
            return;
        }
#endif
        // every run to its root: pointer jumping, the depth halves per round
        for (int round = 0; round < 48; round++) {   // the depth halves per round: 17 rounds flatten any chain of 2^17 runs
            int changed = 0;
            for (int r = t; r < R.n; r += FX_BLOCK) {
                const uint32_t q = ((volatile uint32_t *)parent)[r];
                const uint32_t g = ((volatile uint32_t *)parent)[q];
                if (g != q) {
                    parent[r] = g;
                    changed = 1;
                }
            }
            __threadfence_block();
            if (!__syncthreads_or(changed))
                break;
        }
    };
    // round one: val[] (still free) collects, per run, the smallest run index it touches above
    for (int r = t; r < R.n; r += FX_BLOCK)
        R.val[r] = 0xFFFFFFFFu;
    __syncthreads();
    SIL_LTICK(1);
    contacts(true);
    __threadfence_block();
    __syncthreads();
    for (int r = t; r < R.n; r += FX_BLOCK)
        if (R.val[r] != 0xFFFFFFFFu)
            parent[r] = R.val[r];
    __threadfence_block();
    __syncthreads();
    SIL_LTICK(2);
    flatten(true);
    SIL_LTICK(3);
    // round two: the other contacts
    contacts(false);
    __threadfence_block();
    __syncthreads();
    SIL_LTICK(4);
    flatten(false);
    SIL_LTICK(5);
    for (int r = t; r < R.n; r += FX_BLOCK)
        R.val[r] = 0u;
    __syncthreads();
#ifdef SIL_TIMING
    if (t == 0 && blockIdx.x == 0)
        printf("label: runs %d  scan %llu contacts1 %llu flatten1 %llu contacts2 %llu flatten2 %llu\n", R.n, s_lt[1] - s_lt[0],
               s_lt[2] - s_lt[1], s_lt[3] - s_lt[2], s_lt[4] - s_lt[3], s_lt[5] - s_lt[4]);
#endif
    return R;
}

// Calls f(segment bits, run index) for every stretch of set bits of word w of row y.
template <typename F>
__device__ __forceinline__ void sil_for_segments(const uint32_t *m, const uint32_t *pre, int y, int w, int wc, int wpr, F f)
{
    uint32_t word = m[y * wpr + w];
    if (!word)
        return;
    const uint32_t sm = sil_starts(m + y * wpr, w);
    const uint32_t pm = pre[y * wc + w];
    while (word) {
        const uint32_t low = word & (0u - word);
        const uint32_t seg = word & ~(word + low);  // the lowest stretch of ones
        const int bit = __ffs((int)word) - 1;
        f(seg, pm + __popc(sm & sil_le(bit)) - 1u);
        word &= ~seg;
    }
}

#ifdef SIL_TIMING
#ifndef SIL_TICK_IMG
#define SIL_TICK_IMG 0
#endif
#define SIL_TICK(n)                                        \
    do {                                                   \
        if (t == 0 && blockIdx.x == 0 && b == SIL_TICK_IMG) \
            s_tick[n] = __builtin_readcyclecounter();      \
    } while (0)
#else
#define SIL_TICK(n)
#endif

enum {
    SV_COUNT = 0, SV_THRESH, SV_INV, SV_MINX, SV_MINY, SV_MAXX, SV_MAXY, SV_CX0, SV_CY0, SV_CW, SV_CH, SV_NW,
    SV_NH, SV_NRUNS, SV_ANYKEEP, SV_NEAR, SV_MAXSZ, SV_HOLES4, SV_REFILL, SV_NROOT_FG, SV_NROOT_CC, SV_N
};

// Block-wide bounding box: per-thread extremes into s_v[SV_MINX .. SV_MAXY].
__device__ __forceinline__ void sil_box_reduce(int mnx, int mny, int mxx, int mxy, int *s_v)
{
#pragma unroll
    for (int o = FX_WAVE / 2; o > 0; o >>= 1) {
        const int a = __shfl_xor(mnx, o, FX_WAVE), b2 = __shfl_xor(mny, o, FX_WAVE);
        const int c = __shfl_xor(mxx, o, FX_WAVE), d2 = __shfl_xor(mxy, o, FX_WAVE);
        mnx = a < mnx ? a : mnx;
        mny = b2 < mny ? b2 : mny;
        mxx = c > mxx ? c : mxx;
        mxy = d2 > mxy ? d2 : mxy;
    }
    if (fx_lane() == 0 && mxx >= 0) {
        atomicMin(&s_v[SV_MINX], mnx);
        atomicMin(&s_v[SV_MINY], mny);
        atomicMax(&s_v[SV_MAXX], mxx);
        atomicMax(&s_v[SV_MAXY], mxy);
    }
}

// CV: the OpenCV branch of the reference (A.mode == FX_SIL_OPENCV), else its PIL / scipy branch.
template <bool CV>
__global__ __launch_bounds__(FX_BLOCK, 5) void sil_mask_kernel(SilArgs A)
{
    extern __shared__ __align__(16) uint8_t sil_smem[];
    __shared__ double s_between[FX_WAVES_PER_BLOCK];
    __shared__ uint32_t s_hist[256];
    __shared__ uint32_t s_scan[16 + 256];
    __shared__ uint32_t s_runs[2 * SIL_RUNS_LDS];  // parent | per-root value of the labelling in progress
    // the resize tables live in the run table's space: they are built after the contour stage's labelling has
    // been read for the last time and read before the final labelling starts
    static_assert((2 * SIL_MAX_SIDE + 32) * sizeof(uint16_t) <= sizeof(s_runs), "resize tables do not fit the run table");
    uint16_t *const s_xtab = reinterpret_cast<uint16_t *>(s_runs);   // [SIL_MAX_SIDE + 32]
    uint16_t *const s_ytab = s_xtab + SIL_MAX_SIDE + 32;              // [SIL_MAX_SIDE]
    __shared__ int s_v[SV_N];
    __shared__ unsigned long long s_best, s_first;
    __shared__ unsigned long long s_mom[10][8];
#ifdef SIL_TIMING
    __shared__ unsigned long long s_tick[24];
#endif

    const int t_block = threadIdx.x;
    const int nw = A.rows * A.wpr;
    uint32_t *mA = (uint32_t *)sil_smem;
    uint32_t *mB = mA + nw;
    uint32_t *pre = mB + nw;
    const int H = A.H, W = A.W, wpr = A.wpr, res = A.res;
    const int wcH = (W + 31) >> 5;
    const uint32_t lastH = (W & 31) ? ((1u << (W & 31)) - 1u) : ~0u;
    const SilGeom GH = sil_geom(H, W);
    const SilDiv dvV = sil_div_make((W + 32) >> 5), dvP = sil_div_make(res >> 4);

    for (int64_t b = blockIdx.x; b < A.B; b += gridDim.x) {
        // The thread index is made opaque once per view: what is derived from it (addresses into the LDS arrays and
        // the scratch buffers) is then computed where it is used instead of being kept -- and, at 96 registers,
        // spilled -- across the whole loop over the views.  This is for speed only (5 %); results do not depend on it.
        int t = t_block;
        unsigned long long zero64 = 0ull;   // (likewise: a 64-bit zero hoisted out of the loop ends up spilled)
#ifndef SIL_REPRO_MISCOMPILE
        asm volatile("" : "+v"(t));
        asm volatile("" : "+v"(zero64));
#endif
        __syncthreads();
        // ---- Otsu level (reference :101-126): float64 in the reference's order
        if (!CV)
            s_hist[t] = A.hist[b * 256 + t];
        if (t < SV_N)
            s_v[t] = t == SV_MINX ? W : t == SV_MINY ? H : (t == SV_MAXX || t == SV_MAXY) ? -1 : 0;
        if (t == 0) {
            s_best = zero64;
            s_first = ~zero64;
        }
        if (t < 80)
            s_mom[t >> 3][t & 7] = zero64;
        __syncthreads();
        SIL_TICK(0);
        constexpr bool cv = CV;
        uint32_t *glob_runs = A.runs + (int64_t)blockIdx.x * 2 * A.run_cap;
        if (cv) {
            if (t == 0)
                s_v[SV_THRESH] = A.level[b];  // OpenCV's Otsu level: a serial recurrence, run by sil_otsu_cv_kernel
        } else {
            // inclusive prefix sums of the counts and of level * count (both below 2^32)
            const int lane = fx_lane(), wave = fx_wave();
            uint32_t wb = s_hist[t], sb = (uint32_t)t * wb;
#pragma unroll
            for (int o = 1; o < FX_WAVE; o <<= 1) {
                const uint32_t uc = __shfl_up(wb, o, FX_WAVE), us = __shfl_up(sb, o, FX_WAVE);
                if (lane >= o) {
                    wb += uc;
                    sb += us;
                }
            }
            if (lane == FX_WAVE - 1) {
                s_scan[wave] = wb;
                s_scan[4 + wave] = sb;
            }
            __syncthreads();
            uint32_t wt = 0;
#pragma unroll
            for (int w = 0; w < FX_WAVES_PER_BLOCK; w++) {
                if (w < wave) {
                    wb += s_scan[w];
                    sb += s_scan[4 + w];
                }
                wt += s_scan[4 + w];
            }
            const uint32_t total = (uint32_t)(H * W);
            double between = -1.0;  // level not considered
            if (wb > 0 && total - wb > 0) {
                const double w_back = (double)wb, w_fore = (double)(total - wb);
                const double m_back = (double)sb / w_back;
                const double m_fore = (double)(wt - sb) / w_fore;
                const double d = m_back - m_fore;
                between = ((w_back * w_fore) * d) * d;
            }
            // largest criterion, the higher level among equals (`>=` in the reference's loop)
            int level = t;
#pragma unroll
            for (int o = FX_WAVE / 2; o > 0; o >>= 1) {
                const double ob = __shfl_xor(between, o, FX_WAVE);
                const int ol = __shfl_xor(level, o, FX_WAVE);
                if (ob > between || (ob == between && ol > level)) {
                    between = ob;
                    level = ol;
                }
            }
            if (lane == 0) {
                s_between[wave] = between;
                s_scan[8 + wave] = (uint32_t)level;
            }
            __syncthreads();
            if (t == 0) {
                double best = -1.0;
                int lv = 0;
#pragma unroll
                for (int w = 0; w < FX_WAVES_PER_BLOCK; w++) {
                    if (s_between[w] >= 0.0 && s_between[w] >= best) {  // waves hold ascending levels
                        best = s_between[w];
                        lv = (int)s_scan[8 + w];
                    }
                }
                s_v[SV_THRESH] = lv;
            }
        }
        __syncthreads();
        const uint32_t thresh = (uint32_t)s_v[SV_THRESH];

        // ---- bit mask of blur > thresh
        const uint8_t *blur = A.blur + b * (int64_t)H * W;
        SIL_TICK(1);
        {
            uint32_t cnt = 0;
            const uint32_t kcmp = thresh + 1u;  // pixel > thresh  <=>  pixel >= kcmp
            const uint32_t kl4 = (kcmp & 127u) * 0x01010101u;
            const bool kh = (kcmp & 128u) != 0;
            if ((W & 15) == 0) {
                for (int w0 = t; w0 < H * wcH; w0 += SIL_TH_UNR * FX_BLOCK) {
                    uint4 q[SIL_TH_UNR][2];
#pragma unroll
                    for (int j = 0; j < SIL_TH_UNR; j++) {
                        const int wi = w0 + j * FX_BLOCK;
                        q[j][0] = q[j][1] = make_uint4(0u, 0u, 0u, 0u);
                        if (wi < H * wcH) {
                            const int y = sil_div(wi, GH.dv), w = wi - y * wcH;
                            const uint8_t *p = blur + (int64_t)y * W + 32 * w;
                            q[j][0] = *(const uint4 *)p;
                            if (32 * w + 16 < W)
                                q[j][1] = *(const uint4 *)(p + 16);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < SIL_TH_UNR; j++) {
                        const int wi = w0 + j * FX_BLOCK;
                        if (wi < H * wcH) {
                            const int y = sil_div(wi, GH.dv), w = wi - y * wcH;
                            uint32_t word = 0;
                            if (kcmp <= 255u) {
#pragma unroll
                                for (int h = 0; h < 2; h++)
                                    word |= (sil_ge4(q[j][h].x, kl4, kh) | (sil_ge4(q[j][h].y, kl4, kh) << 4) |
                                             (sil_ge4(q[j][h].z, kl4, kh) << 8) | (sil_ge4(q[j][h].w, kl4, kh) << 12))
                                            << (16 * h);
                                // a zero-filled second half reads as "not above" unless kcmp == 0, which cannot be
                            }
                            mA[y * wpr + w] = word;
                            cnt += __popc(word);
                        }
                    }
                }
            } else {
                for (int wi = t; wi < H * wcH; wi += FX_BLOCK) {
                    const int y = sil_div(wi, GH.dv), w = wi - y * wcH;
                    const uint8_t *p = blur + (int64_t)y * W + 32 * w;
                    uint32_t word = 0;
                    for (int k = 0; k < 32 && 32 * w + k < W; k++)
                        word |= (p[k] > thresh ? 1u : 0u) << k;
                    mA[y * wpr + w] = word;
                    cnt += __popc(word);
                }
            }
            cnt = (uint32_t)fx_wave_sum_u64(cnt);
            if (fx_lane() != 0)
                cnt = 0;
            if (cnt)
                atomicAdd(&s_v[SV_COUNT], (int)cnt);
        }
        __syncthreads();
        // foreground above half of the image: the silhouette is the other class
        const bool inverted = 2ll * s_v[SV_COUNT] > (long long)H * W;
        if (inverted) {
            for (int wi = t; wi < H * wcH; wi += FX_BLOCK) {
                const int y = sil_div(wi, GH.dv), w = wi - y * wcH;
                uint32_t word = ~mA[y * wpr + w];
                if (w == wcH - 1)
                    word &= lastH;
                mA[y * wpr + w] = word;
            }
        }
        __syncthreads();

        SIL_TICK(2);
        if (!cv) {
            // ---- 3x3 closing then opening (reference :170-186, scipy.ndimage branch)
            sil_morph<true>(mA, mB, GH, wpr);
            sil_morph<false>(mB, mA, GH, wpr);
            sil_morph<false>(mA, mB, GH, wpr);
            sil_morph<true>(mB, mA, GH, wpr);

            SIL_TICK(3);
            // ---- bounding box of the foreground (reference :209-214, branch without cv2)
            int mnx = W, mny = H, mxx = -1, mxy = -1;
            for (int wi = t; wi < H * wcH; wi += FX_BLOCK) {
                const int y = sil_div(wi, GH.dv), w = wi - y * wcH;
                const uint32_t m = mA[y * wpr + w];
                if (m) {
                    const int lo = 32 * w + __ffs((int)m) - 1, hi = 32 * w + 31 - __clz((int)m);
                    mnx = lo < mnx ? lo : mnx;
                    mxx = hi > mxx ? hi : mxx;
                    mny = y < mny ? y : mny;
                    mxy = y > mxy ? y : mxy;
                }
            }
            sil_box_reduce(mnx, mny, mxx, mxy, s_v);
        } else {
            // ---- cv2.morphologyEx with the 5x5 ellipse: close, iterations = 2 (dilate, dilate, erode, erode),
            // then open (erode, dilate) (reference :171-177)
            sil_morph_cv<true, true>(mA, mB, GH, wpr);
            sil_morph_cv<true, true>(mB, mA, GH, wpr);
            sil_morph_cv<false, true>(mA, mB, GH, wpr);
            sil_morph_cv<false, true>(mB, mA, GH, wpr);
            sil_morph_cv<false, true>(mA, mB, GH, wpr);
            sil_morph_cv<true, true>(mB, mA, GH, wpr);

            SIL_TICK(3);
            // ---- box of the external contour with the largest contourArea (reference :198-221).
            // The external contours are the 8-connected components of the mask with its holes filled.  Many
            // masks have no hole, and then the mask's own components are those.  The Euler number E (components
            // minus holes) follows from the 2 x 2 windows around the corners of the pixel grid:
            // 4 E = #(one pixel set) - #(three set) - 2 #(a diagonal pair).  E <= 0 means holes for certain;
            // otherwise the mask is labelled as it is, and it has no hole exactly when it has E components.
            SIL_TICK(14);
            const int wcV = (W + 32) >> 5;  // corners x = 0 .. W
            // First the mask is labelled as it is.  Without holes (as many components as the Euler number) its
            // components are the external contours.  With holes the corner sums below give a component's outer area
            // minus its holes' areas: the component with the largest such value is still the winner if it beats
            // twice the bounding-box area of the box of every other component (an outer border cannot enclose more
            // than its box), and its box is the same with or without the holes.  Only when that test fails are the
            // holes filled (steps 1-3) and everything below is done again on the filled mask.
            const auto contour_pass = [&](const uint32_t *mF, const bool first) -> bool {
            SilRuns fg;
            // (3) 8-connected components: of the mask as it is, or of the filled mask = the external contours
            SIL_TICK(16);
            fg = sil_label(mF, H, wcH, GH.dv, wpr, true, pre, s_scan, &s_v[SV_NRUNS], s_runs, glob_runs, A.run_cap);
            if (first) {
                int roots = 0;
                for (int r = t; r < fg.n; r += FX_BLOCK)
                    roots += fg.parent[r] == (uint32_t)r ? 4 : 0;
#pragma unroll
                for (int o = FX_WAVE / 2; o > 0; o >>= 1)
                    roots += __shfl_xor(roots, o, FX_WAVE);
                if (fx_lane() == 0 && roots) {
                    atomicSub(&s_v[SV_HOLES4], roots);   // 4 (E - C) = -4 holes
                    atomicAdd(&s_v[SV_NROOT_FG], roots >> 2);
                }
                __syncthreads();
                // A single component (the usual silhouette) is the winner whatever its area and holes: its box is
                // the box of the foreground.
                if (s_v[SV_NROOT_FG] == 1) {   // block-uniform
                    int mnx = W, mny = H, mxx = -1, mxy = -1;
                    for (int wi = t; wi < H * wcH; wi += FX_BLOCK) {
                        const int y = sil_div(wi, GH.dv), w = wi - y * wcH;
                        const uint32_t m = mF[y * wpr + w];
                        if (m) {
                            const int lo = 32 * w + __ffs((int)m) - 1, hi = 32 * w + 31 - __clz((int)m);
                            mnx = lo < mnx ? lo : mnx;
                            mxx = hi > mxx ? hi : mxx;
                            mny = y < mny ? y : mny;
                            mxy = y > mxy ? y : mxy;
                        }
                    }
                    sil_box_reduce(mnx, mny, mxx, mxy, s_v);
                    return false;
                }
                // several components: the Euler number, to know whether any of them has a hole
                {
                    int e4 = 0;
                    for (int vi = t; vi < (H + 1) * wcV; vi += FX_BLOCK) {
                        const int vy = sil_div(vi, dvV), w = vi - vy * wcV;
                        const uint32_t *ru = mA + (vy - 1) * wpr, *rd = mA + vy * wpr;
                        const uint32_t up = vy > 0 && w < wcH ? ru[w] : 0u, upp = vy > 0 && w > 0 ? ru[w - 1] : 0u;
                        const uint32_t dn = vy < H && w < wcH ? rd[w] : 0u, dnp = vy < H && w > 0 ? rd[w - 1] : 0u;
                        const uint32_t ca = (up << 1) | (upp >> 31), cb = up, cc = (dn << 1) | (dnp >> 31), cd = dn;
                        const uint32_t odd = ca ^ cb ^ cc ^ cd;
                        const uint32_t three = odd & ((ca & cb) | (cc & cd));
                        const uint32_t diag = (ca & cd & ~(cb | cc)) | (cb & cc & ~(ca | cd));
                        e4 += __popc(odd & ~three) - __popc(three) - 2 * __popc(diag);
                    }
#pragma unroll
                    for (int o = FX_WAVE / 2; o > 0; o >>= 1)
                        e4 += __shfl_xor(e4, o, FX_WAVE);
                    if (fx_lane() == 0 && e4)
                        atomicAdd(&s_v[SV_HOLES4], e4);
                }
                __syncthreads();
            }
            const bool holes = first && s_v[SV_HOLES4] != 0;
            SIL_TICK(15);
            // (4) twice the signed area inside each outer border, as a sum over the corners of the pixel grid:
            // the border follower steps between the two foreground pixels of a straight corner and cuts across
            // the background pixel of a concave corner; saddle corners give two opposite steps that cancel.
            // Summed as (qy - py) * (qx + px) over the steps p -> q, which over a closed border equals the
            // shoelace sum and to which steps along a row contribute nothing.  A step goes up (qy = py - 1)
            // exactly when both pixels left of the corner are set, and down otherwise.
            SIL_TICK(17);
            uint32_t acc_root = 0xFFFFFFFFu, acc = 0u;  // a thread's corners mostly lie on one border
            for (int vi = t; vi < (H + 1) * wcV; vi += FX_BLOCK) {
                const int vy = sil_div(vi, dvV), w = vi - vy * wcV;
                const uint32_t *ru = mF + (vy - 1) * wpr, *rd = mF + vy * wpr;
                const uint32_t up = vy > 0 && w < wcH ? ru[w] : 0u, upp = vy > 0 && w > 0 ? ru[w - 1] : 0u;
                const uint32_t dn = vy < H && w < wcH ? rd[w] : 0u, dnp = vy < H && w > 0 ? rd[w - 1] : 0u;
                // pixels around the corner (vx, vy): a = (vx-1, vy-1), b = (vx, vy-1), c = (vx-1, vy), d = (vx, vy)
                const uint32_t ca = (up << 1) | (upp >> 31), cb = up, cc = (dn << 1) | (dnp >> 31), cd = dn;
                const uint32_t vert = (ca & cc & ~cb & ~cd) | (cb & cd & ~ca & ~cc);
                const uint32_t conc = (ca & cb & cc & ~cd) | (ca & cb & cd & ~cc) | (ca & cc & cd & ~cb) | (cb & cc & cd & ~ca);
                uint32_t act = vert | conc;
                if (!act)
                    continue;
                const uint32_t su = cb & ~ca, sd = cd & ~cc;  // run starts of the two rows within this word
                const uint32_t pu = up ? pre[(vy - 1) * wcH + w] : 0u, pd = dn ? pre[vy * wcH + w] : 0u;
                while (act) {
                    const int bit = __ffs((int)act) - 1;
                    act &= act - 1;
                    const uint32_t le = sil_le(bit);
                    const int vx = 32 * w + bit;
                    // any set pixel at the corner names the component: d, else b, else c (then a and c alone are set)
                    uint32_t run;
                    if ((cd >> bit) & 1u)
                        run = pd + __popc(sd & le) - 1u;
                    else if ((cb >> bit) & 1u)
                        run = pu + __popc(su & le) - 1u;
                    else
                        run = sil_run_at(mF, pre, vy, vx - 1, wcH, wpr);
                    const int mag = 2 * vx - 1, st = (int)((vert >> bit) & 1u);  // qx + px: 2vx - 1 across a corner pixel
                    const int contrib = ((ca & cc) >> bit) & 1u ? st - mag : mag + st;  // 2vx - 2 up the left side, 2vx down the right
                    const uint32_t root = fg.parent[run];
                    if (root != acc_root) {
                        if (acc_root != 0xFFFFFFFFu)
                            atomicAdd(&fg.val[acc_root], acc);
                        acc_root = root;
                        acc = 0u;
                    }
                    acc += (uint32_t)contrib;
                }
            }
            if (acc_root != 0xFFFFFFFFu)
                atomicAdd(&fg.val[acc_root], acc);
            __threadfence_block();
            __syncthreads();
            SIL_TICK(18);
            // (5) the largest area; among equals the contour found last in raster order (cv2.findContours lists
            // the contours in reverse order of discovery and max() keeps the first)
            for (int r = t; r < fg.n; r += FX_BLOCK) {
                if (fg.parent[r] == (uint32_t)r) {
                    const int a2 = (int)fg.val[r];
                    atomicMax(&s_best, ((unsigned long long)(uint32_t)(a2 < 0 ? -a2 : a2) << 32) | (unsigned long long)r);
                }
            }
            __syncthreads();
            const uint32_t best = (uint32_t)(s_best & 0xFFFFFFFFull), best_a2 = (uint32_t)(s_best >> 32);
            // boxes of all components at once (four words per run in the free mask buffer) when the winner has to be
            // confirmed against the others' boxes; a scan for the winner's box alone otherwise
            const bool all_boxes = holes && fg.n > 0 && 4 * fg.n <= nw;
            bool again = holes && !all_boxes;
            if (all_boxes) {
                uint32_t *bx0 = mB, *bx1 = mB + fg.n, *by0 = mB + 2 * fg.n, *by1 = mB + 3 * fg.n;
                for (int r = t; r < fg.n; r += FX_BLOCK) {
                    bx0[r] = by0[r] = 0xFFFFFFFFu;
                    bx1[r] = by1[r] = 0u;
                }
                __syncthreads();
                for (int wi = t; wi < H * wcH; wi += FX_BLOCK) {
                    const int y = sil_div(wi, GH.dv), w = wi - y * wcH;
                    sil_for_segments(mF, pre, y, w, wcH, wpr, [&](uint32_t seg, uint32_t id) {
                        const uint32_t root = fg.parent[id];
                        atomicMin(&bx0[root], (uint32_t)(32 * w + __ffs((int)seg) - 1));
                        atomicMax(&bx1[root], (uint32_t)(32 * w + 31 - __clz((int)seg)));
                        atomicMin(&by0[root], (uint32_t)y);
                        atomicMax(&by1[root], (uint32_t)y);
                    });
                }
                __threadfence_block();
                __syncthreads();
                for (int r = t; r < fg.n; r += FX_BLOCK)
                    if (fg.parent[r] == (uint32_t)r && (uint32_t)r != best &&
                        2u * (bx1[r] - bx0[r]) * (by1[r] - by0[r]) >= best_a2)
                        s_v[SV_REFILL] = 1;
                __syncthreads();
                again = s_v[SV_REFILL] != 0;
                if (!again && t == 0) {
                    s_v[SV_MINX] = (int)bx0[best];
                    s_v[SV_MAXX] = (int)bx1[best];
                    s_v[SV_MINY] = (int)by0[best];
                    s_v[SV_MAXY] = (int)by1[best];
                }
            }
            if (again) {   // block-uniform
                __syncthreads();
                if (t == 0)
                    s_best = zero64;
                __syncthreads();
                return true;
            }
            if (!all_boxes) {
                int mnx = W, mny = H, mxx = -1, mxy = -1;
                if (fg.n > 0) {
                    for (int wi = t; wi < H * wcH; wi += FX_BLOCK) {
                        const int y = sil_div(wi, GH.dv), w = wi - y * wcH;
                        sil_for_segments(mF, pre, y, w, wcH, wpr, [&](uint32_t seg, uint32_t id) {
                            if (fg.parent[id] != best)
                                return;
                            const int lo = 32 * w + __ffs((int)seg) - 1, hi = 32 * w + 31 - __clz((int)seg);
                            mnx = lo < mnx ? lo : mnx;
                            mxx = hi > mxx ? hi : mxx;
                            mny = y < mny ? y : mny;
                            mxy = y > mxy ? y : mxy;
                        });
                    }
                }
                sil_box_reduce(mnx, mny, mxx, mxy, s_v);
            }
            return false;
            };
            if (contour_pass(mA, true)) {
            // (1) background runs, 4-connected; those that reach the image frame are the outside
            for (int wi = t; wi < H * wcH; wi += FX_BLOCK) {
                const int y = sil_div(wi, GH.dv), w = wi - y * wcH;
                mB[y * wpr + w] = ~mA[y * wpr + w] & (w == wcH - 1 ? lastH : ~0u);
            }
            SilRuns bg = sil_label(mB, H, wcH, GH.dv, wpr, false, pre, s_scan, &s_v[SV_NRUNS], s_runs, glob_runs, A.run_cap);
            for (int wi = t; wi < H * wcH; wi += FX_BLOCK) {
                const int y = sil_div(wi, GH.dv), w = wi - y * wcH;
                sil_for_segments(mB, pre, y, w, wcH, wpr, [&](uint32_t seg, uint32_t id) {
                    const bool frame = y == 0 || y == H - 1 || (w == 0 && (seg & 1u)) ||
                                       (w == wcH - 1 && ((seg >> ((W - 1) & 31)) & 1u));
                    if (frame)
                        bg.val[bg.parent[id]] = 1u;
                });
            }
            __threadfence_block();
            __syncthreads();
            // (2) holes filled: components nested inside a hole vanish, outer borders stay as they are.  The
            // complement is taken from mA again because mB is being overwritten.
            for (int wi = t; wi < H * wcH; wi += FX_BLOCK) {
                const int y = sil_div(wi, GH.dv), w = wi - y * wcH;
                const uint32_t m = mA[y * wpr + w];
                uint32_t comp = ~m & (w == wcH - 1 ? lastH : ~0u), holes = 0u;
                if (comp) {
                    const uint32_t cp = w > 0 ? ~mA[y * wpr + w - 1] : 0u;
                    const uint32_t sm = comp & ~((comp << 1) | (cp >> 31));
                    const uint32_t pm = pre[wi];
                    while (comp) {
                        const uint32_t low = comp & (0u - comp);
                        const uint32_t seg = comp & ~(comp + low);
                        const int bit = __ffs((int)comp) - 1;
                        if (!bg.val[bg.parent[pm + __popc(sm & sil_le(bit)) - 1u]])
                            holes |= seg;
                        comp &= ~seg;
                    }
                }
                mB[y * wpr + w] = m | holes;
            }
            (void)contour_pass(mB, false);
            }
        }
        __syncthreads();

        SIL_TICK(4);
        // ---- crop rectangle and target size (reference :352-440)
        if (t == 0) {
            int x0 = 0, y0 = 0, x1 = W, y1 = H;
            if (A.crop) {
                int bx = 0, by = 0, bw = W, bh = H;
                if (s_v[SV_MAXX] >= 0) {
                    bx = s_v[SV_MINX];
                    by = s_v[SV_MINY];
                    bw = s_v[SV_MAXX] - bx + 1;
                    bh = s_v[SV_MAXY] - by + 1;
                }
                const int side = bw > bh ? bw : bh;
                // Python round(): half to even
                const int pad = A.crop_margin <= 1.0 ? (int)rint(A.crop_margin * (double)side)
                                                     : (int)rint(A.crop_margin);
                x0 = bx - pad > 0 ? bx - pad : 0;
                y0 = by - pad > 0 ? by - pad : 0;
                x1 = bx + bw + pad < W ? bx + bw + pad : W;
                y1 = by + bh + pad < H ? by + bh + pad : H;
                if (x1 <= x0 || y1 <= y0) {
                    x0 = 0;
                    y0 = 0;
                    x1 = W;
                    y1 = H;
                }
            }
            const int cw = x1 - x0, ch = y1 - y0;
            const double sx = (double)res / (double)cw, sy = (double)res / (double)ch;
            const double scale = sx < sy ? sx : sy;
            int new_w = (int)rint((double)cw * scale), new_h = (int)rint((double)ch * scale);
            new_w = new_w < 1 ? 1 : (new_w > res ? res : new_w);
            new_h = new_h < 1 ? 1 : (new_h > res ? res : new_h);
            s_v[SV_CX0] = x0;
            s_v[SV_CY0] = y0;
            s_v[SV_CW] = cw;
            s_v[SV_CH] = ch;
            s_v[SV_NW] = new_w;
            s_v[SV_NH] = new_h;
        }
        __syncthreads();
        const int cx0 = s_v[SV_CX0], cy0 = s_v[SV_CY0], cw = s_v[SV_CW], ch = s_v[SV_CH];
        const int new_w = s_v[SV_NW], new_h = s_v[SV_NH];
        SIL_TICK(5);
        // PIL's nearest resize steps through the source with a float64
        // accumulator: xo = a/2, then xo += a per output pixel.
        // Every entry is int(xo).  The accumulated xo differs from (x + 0.5) * a
        // by less than 1e-10, so the product decides unless it lies that close
        // to an integer; then the table is stepped through in order.
        {
            const double ax = (double)cw / (double)new_w, ay = (double)ch / (double)new_h;
            int near = 0;
            for (int x = t; x < ((new_w + 31) & ~31); x += FX_BLOCK) {
                const double v = ((double)x + 0.5) * ax;
                const int xin = (int)v;
                near |= fabs(v - rint(v)) < 1e-9 ? 1 : 0;
                s_xtab[x] = x < new_w && xin < cw ? (uint16_t)xin : (uint16_t)0xFFFF;
            }
            for (int y = t; y < new_h; y += FX_BLOCK) {
                const double v = ((double)y + 0.5) * ay;
                const int yin = (int)v;
                near |= fabs(v - rint(v)) < 1e-9 ? 2 : 0;
                s_ytab[y] = yin < ch ? (uint16_t)yin : (uint16_t)0xFFFF;
            }
            if (near)
                atomicOr(&s_v[SV_NEAR], near);
            __syncthreads();
            const int flags = s_v[SV_NEAR];
            if ((t == 0 && (flags & 1)) || (t == FX_WAVE && (flags & 2))) {
                const int n_out = t == 0 ? new_w : new_h, n_in = t == 0 ? cw : ch;
                uint16_t *tab = t == 0 ? s_xtab : s_ytab;
                const double a = t == 0 ? ax : ay;
                double xo = a * 0.5;
                for (int x = 0; x < n_out; x++) {
                    const int xin = (int)xo;
                    tab[x] = xin < n_in ? (uint16_t)xin : (uint16_t)0xFFFF;
                    xo += a;
                }
            }
        }
        __syncthreads();

        SIL_TICK(6);
        // ---- resized mask (new_h x new_w) into mB
        const SilGeom GS = sil_geom(new_h, new_w);
        const int wcS = GS.wc;
        if (cw <= new_w) {
            // Enlarging (the usual case: a cropped view back to the working size): the source columns of an
            // output word never step by more than one pixel, so they lie inside the 32 pixels that start at the
            // word's first source column.  A thread keeps a word column's 32 offsets in registers and walks
            // down a strip of rows: two mask words and two instructions per output bit, instead of two LDS
            // reads per bit.
            for (int it = t; it < wcS * GS.strips; it += FX_BLOCK) {
                const int sy_ = sil_div(it, GS.dv), w = it - sy_ * wcS, y0 = sy_ * GS.rs;
                if (y0 >= new_h)
                    continue;
                const int y1 = y0 + GS.rs < new_h ? y0 + GS.rs : new_h;
                const uint32_t first = s_xtab[32 * w];  // x = 32 w < new_w: a valid entry
                uint32_t off[32], valid = 0u;
#pragma unroll
                for (int k = 0; k < 32; k++) {
                    const uint32_t e = s_xtab[32 * w + k];
                    off[k] = (e - first) & 31u;
                    valid |= (e != 0xFFFFu ? 1u : 0u) << k;
                }
                const int x0 = cx0 + (int)first, xw = x0 >> 5, xs = x0 & 31;
                for (int y = y0; y < y1; y++) {
                    const uint32_t sy = s_ytab[y];
                    uint32_t word = 0;
                    if (sy != 0xFFFFu) {
                        const uint32_t *r = mA + (cy0 + (int)sy) * wpr;
                        const uint32_t lo = r[xw], hi = xw + 1 < wpr ? r[xw + 1] : 0u;
                        const uint32_t win = xs ? (lo >> xs) | (hi << (32 - xs)) : lo;
#pragma unroll
                        for (int k = 0; k < 32; k++)
                            word |= ((win >> off[k]) & 1u) << k;
                        word &= valid;
                    }
                    mB[y * wpr + w] = word;
                }
            }
        } else {
            for (int wi = t; wi < new_h * wcS; wi += FX_BLOCK) {
                const int y = sil_div(wi, GS.dv), w = wi - y * wcS;
                const uint32_t sy = s_ytab[y];
                uint32_t word = 0;
                if (sy != 0xFFFFu) {
                    const uint32_t *r = mA + (cy0 + (int)sy) * wpr;
                    uint32_t sx[32];
#pragma unroll
                    for (int k = 0; k < 32; k++)
                        sx[k] = s_xtab[32 * w + k];
#pragma unroll
                    for (int k = 0; k < 32; k++) {
                        const int x = cx0 + (int)(sx[k] == 0xFFFFu ? 0u : sx[k]);
                        const uint32_t bit = (r[x >> 5] >> (x & 31)) & 1u;
                        word |= (sx[k] == 0xFFFFu ? 0u : bit) << k;
                    }
                }
                mB[y * wpr + w] = word;
            }
        }
        __syncthreads();

        SIL_TICK(7);
        // ---- opening of the resized mask (reference :246-264): 3x3 box, or cv2's 3x3 ellipse (a cross)
        if (!cv) {
            sil_morph<false>(mB, mA, GS, wpr);
            sil_morph<true>(mA, mB, GS, wpr);
        } else {
            sil_morph_cv<false, false>(mB, mA, GS, wpr);
            sil_morph_cv<true, false>(mA, mB, GS, wpr);
        }

        SIL_TICK(8);
        SIL_TICK(9);
        // ---- connected components over row runs: 4-connected (scipy.ndimage.label) or 8-connected
        // (cv2.connectedComponentsWithStats(connectivity=8)), reference :284-311
        const int nwS = new_h * wcS;
        SilRuns cc = sil_label(mB, new_h, wcS, GS.dv, wpr, cv, pre, s_scan, &s_v[SV_NRUNS], s_runs, glob_runs, A.run_cap);
        SIL_TICK(10);
        // A single component is kept whatever its size (it is the largest when none reaches min_size): the mask
        // stays as it is.
        {
            int roots = 0;
            for (int r = t; r < cc.n; r += FX_BLOCK)
                roots += cc.parent[r] == (uint32_t)r ? 1 : 0;
#pragma unroll
            for (int o = FX_WAVE / 2; o > 0; o >>= 1)
                roots += __shfl_xor(roots, o, FX_WAVE);
            if (fx_lane() == 0 && roots)
                atomicAdd(&s_v[SV_NROOT_CC], roots);
            __syncthreads();
        }
        const bool one_cc = s_v[SV_NROOT_CC] == 1;   // block-uniform
        const uint32_t *mK = one_cc ? mB : mA;       // the kept pixels
        if (!one_cc) {
        // component sizes
        for (int wi = t; wi < nwS; wi += FX_BLOCK) {
            const int y = sil_div(wi, GS.dv), w = wi - y * wcS;
            sil_for_segments(mB, pre, y, w, wcS, wpr, [&](uint32_t seg, uint32_t id) {
                atomicAdd(&cc.val[cc.parent[id]], (uint32_t)__popc(seg));
            });
        }
        __threadfence_block();
        __syncthreads();
        SIL_TICK(11);
        // keep components of at least min_size pixels; when none is that large keep the largest one, among
        // equals the one with the lowest label: first in raster order for scipy, first in the raster order of
        // 2x2 blocks for OpenCV (whose labelling scans such blocks)
        for (int r = t; r < cc.n; r += FX_BLOCK) {
            if (cc.parent[r] == (uint32_t)r) {
                const uint32_t sz = cc.val[r];
                if ((int)sz >= A.min_size)
                    s_v[SV_ANYKEEP] = 1;
                atomicMax(&s_v[SV_MAXSZ], (int)sz);
            }
        }
        __syncthreads();
        const bool any_keep = s_v[SV_ANYKEEP] != 0;
        if (!any_keep && cc.n > 0) {
            const uint32_t maxsz = (uint32_t)s_v[SV_MAXSZ];
            for (int wi = t; wi < nwS; wi += FX_BLOCK) {
                const int y = sil_div(wi, GS.dv), w = wi - y * wcS;
                sil_for_segments(mB, pre, y, w, wcS, wpr, [&](uint32_t seg, uint32_t id) {
                    const uint32_t root = cc.parent[id];
                    if (cc.val[root] != maxsz)
                        return;
                    const uint32_t x = 32u * w + (uint32_t)(__ffs((int)seg) - 1);
                    const uint32_t key = cv ? (((uint32_t)y >> 1) << 16) | (x >> 1) : root;
                    atomicMin(&s_first, ((unsigned long long)key << 32) | (unsigned long long)root);
                });
            }
            __syncthreads();
        }
        const uint32_t best_root = (uint32_t)(s_first & 0xFFFFFFFFull);
        // kept pixels into mA
        for (int wi = t; wi < nwS; wi += FX_BLOCK) {
            const int y = sil_div(wi, GS.dv), w = wi - y * wcS;
            uint32_t kept = 0u;
            sil_for_segments(mB, pre, y, w, wcS, wpr, [&](uint32_t seg, uint32_t id) {
                const uint32_t root = cc.parent[id];
                const bool keep = any_keep ? (int)cc.val[root] >= A.min_size : root == best_root;
                kept |= keep ? seg : 0u;
            });
            mA[y * wpr + w] = kept;
        }
        __syncthreads();

        }
        SIL_TICK(12);
        const int px0 = (res - new_w) / 2, py0 = (res - new_h) / 2;
        // ---- raw moments of the silhouette as the 0 / 255 image it is on the canvas (what cv2.moments sums over
        // that image, reference :449): exact integers from the row runs of the bit mask.  Over a run of columns
        // a .. b the column powers have closed sums; a run that crosses a word is taken piece by piece.
        if (A.partial) {
            // Per piece (at most 32 columns, columns below 512) everything fits 32 bits: s1 = sum x, six times
            // sum x^2 (the division by six waits for the total), sum x^3 = s1 * (T(b) + T(a - 1)) with the
            // triangular numbers T; the products with the row's powers go into 64-bit sums.
            unsigned long long acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int wi = t; wi < nwS; wi += FX_BLOCK) {
                const int y = sil_div(wi, GS.dv), w = wi - y * wcS;
                uint32_t word = mK[y * wpr + w];
                const uint32_t Y = (uint32_t)(y + py0), Y2 = Y * Y;
                const unsigned long long Y3 = (unsigned long long)Y2 * Y;
                uint32_t r0 = 0, r1 = 0;           // the row's pieces in this word: counts and sums of x
                unsigned long long r2 = 0, r3 = 0; // 6 sum x^2, sum x^3
                while (word) {
                    const uint32_t low = word & (0u - word);
                    const uint32_t seg = word & ~(word + low);  // the lowest stretch of ones
                    const uint32_t a = (uint32_t)(px0 + 32 * w + __ffs((int)word) - 1), len = __popc(seg), bq = a + len - 1;
                    word &= ~seg;
                    const uint32_t tb = (bq * (bq + 1)) >> 1, ta = (a * (a - 1)) >> 1;   // T(b), T(a - 1)  (a = 0: 0)
                    const uint32_t s1 = tb - ta;
                    const uint32_t f2b = bq * (bq + 1) * (2 * bq + 1), f2a = a == 0 ? 0u : (a - 1) * a * (2 * a - 1);
                    r0 += len;
                    r1 += s1;
                    r2 += f2b - f2a;
                    r3 += (unsigned long long)s1 * (tb + ta);
                }
                if (r0) {
                    acc[0] += r0;
                    acc[1] += r1;
                    acc[2] += (unsigned long long)Y * r0;
                    acc[3] += r2;
                    acc[4] += (unsigned long long)Y * r1;
                    acc[5] += (unsigned long long)Y2 * r0;
                    acc[6] += r3;
                    acc[7] += Y * r2;
                    acc[8] += (unsigned long long)Y2 * r1;
                    acc[9] += Y3 * r0;
                }
            }
            // eight copies per moment, lane l adding to copy l % 8: 64 lanes on one LDS word would take turns
            if (acc[0]) {
#pragma unroll
                for (int k = 0; k < 10; k++)
                    atomicAdd(&s_mom[k][t & 7], acc[k]);
            }
            __syncthreads();
            if (t < 10) {
                unsigned long long v = 0;
#pragma unroll
                for (int c = 0; c < 8; c++)
                    v += s_mom[t][c];
                if (t == 3 || t == 7)
                    v /= 6ull;   // sums of x^2 were kept six-fold
                v *= 255ull;     // pixel value 255; at most 255 * 512^2 * 511^3 < 2^63
                int64_t *limb = A.partial + (b * 10 + t) * 3;
                limb[0] = (int64_t)(v & ((1ull << 48) - 1ull));
                limb[1] = (int64_t)(v >> 48);
                limb[2] = 0;
            }
        }
        // ---- centred paste on the res x res canvas
        uint8_t *out = A.mask + b * (int64_t)res * res;
        if (!A.mask) {
            // the caller wants the moments only
        } else if ((res & 15) == 0) {
            const int cpr = res >> 4;
            for (int ci = t; ci < res * cpr; ci += FX_BLOCK) {
                const int Y = sil_div(ci, dvP), X0 = (ci - Y * cpr) * 16;
                const int y = Y - py0;
                uint32_t o[4] = {0u, 0u, 0u, 0u};
                const int x = X0 - px0;  // mask column of the chunk's first pixel
                if (y >= 0 && y < new_h && x > -16 && x < new_w) {
                    const uint32_t *r = mK + y * wpr;
                    const int w0 = x >> 5;  // floor, also for negative x
                    const uint32_t lo = w0 >= 0 && w0 < wcS ? r[w0] : 0u;
                    const uint32_t hi = w0 + 1 >= 0 && w0 + 1 < wcS ? r[w0 + 1] : 0u;
                    const uint32_t bits = (uint32_t)(((((u64)hi) << 32) | lo) >> (x & 31)) & 0xFFFFu;
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        o[q] = sil_spread4((bits >> (4 * q)) & 0xFu);
                }
                *(uint4 *)(out + (int64_t)Y * res + X0) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        } else {
            for (int pi = t; pi < res * res; pi += FX_BLOCK) {
                const int Y = pi / res, X = pi - Y * res;
                const int y = Y - py0, x = X - px0;
                uint8_t v = 0;
                if (y >= 0 && y < new_h && x >= 0 && x < new_w && ((mK[y * wpr + (x >> 5)] >> (x & 31)) & 1u))
                    v = 255;
                out[pi] = v;
            }
        }
        SIL_TICK(13);
#ifdef SIL_TIMING
        if (t == 0 && blockIdx.x == 0 && b == SIL_TICK_IMG) {
            printf("sil ticks:");
            for (int i = 1; i <= 13; i++)
                printf(" %d:%llu", i, s_tick[i] - s_tick[i - 1]);
            printf(" total %llu\n", s_tick[13] - s_tick[0]);
            if (cv)
                printf("  contour stage: label + Euler %llu, holes path %llu, corners %llu best+box %llu\n",
                       s_tick[15] - s_tick[14], s_tick[17] - s_tick[15], s_tick[18] - s_tick[17], s_tick[4] - s_tick[18]);
        }
#endif
        if (A.info && t == 0) {
            int32_t *f = A.info + b * 8;
            f[0] = (int32_t)thresh;
            f[1] = inverted ? 1 : 0;
            f[2] = cx0;
            f[3] = cy0;
            f[4] = cx0 + cw;
            f[5] = cy0 + ch;
            f[6] = new_w;
            f[7] = new_h;
        }
    }
}

inline size_t sil_align(size_t v) { return (v + 255) & ~(size_t)255; }

// The silhouette stage behind fx_silhouette_batch_u8 (masks) and fx_silhouette_moments_batch_u8 (raw moments of
// the masks, masks optional).
int sil_batch(const uint8_t *d_img, int64_t B, int H, int W, int64_t pitch, int64_t stride, int resolution,
              double crop_margin, int preprocess_crop, int min_size, int mode, uint8_t *d_mask, int32_t *d_info,
              int64_t *d_partial, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    FX_REQUIRE(mode == FX_SIL_PIL || mode == FX_SIL_OPENCV, "fx_silhouette_batch_u8: mode %d (0 = PIL/scipy branch, "
               "1 = OpenCV branch)", mode);
    FX_REQUIRE(B >= 0, "fx_silhouette_batch_u8: B = %lld", (long long)B);
    if (B == 0)
        return FX_OK;
    FX_REQUIRE(d_img && (d_mask || d_partial), "fx_silhouette_batch_u8: null image pointer, or no output");
    FX_REQUIRE(H >= 1 && W >= 1 && H <= SIL_MAX_SIDE && W <= SIL_MAX_SIDE,
               "fx_silhouette_batch_u8: views of %dx%d; supported up to %dx%d", H, W, SIL_MAX_SIDE, SIL_MAX_SIDE);
    FX_REQUIRE(resolution >= 1 && resolution <= SIL_MAX_SIDE, "fx_silhouette_batch_u8: resolution %d outside 1..%d",
               resolution, SIL_MAX_SIDE);
    FX_REQUIRE(pitch >= W && (B == 1 || stride >= (int64_t)(H - 1) * pitch + W),
               "fx_silhouette_batch_u8: pitch %lld / stride %lld too small for %dx%d", (long long)pitch,
               (long long)stride, H, W);
    FX_REQUIRE(crop_margin >= 0.0 && crop_margin < 1e9, "fx_silhouette_batch_u8: crop_margin %g", crop_margin);
    if (min_size < 0) {
        // int(0.001 * resolution * resolution), at least 8 (reference :262-263)
        const double v = 0.001 * (double)resolution * (double)resolution;
        min_size = (int)v > 8 ? (int)v : 8;
    }

    const int64_t px = (int64_t)H * W;
    const int rows = H > resolution ? H : resolution;
    const int side = W > resolution ? W : resolution;
    const int wpr = (side + 31) / 32;
    // most runs a labelling can meet: the resized mask, and for the OpenCV branch also the view itself
    const int run_cap = mode == FX_SIL_OPENCV ? rows * (side / 2 + 1) : resolution * (resolution / 2 + 1);
    const size_t mask_lds = (size_t)3 * rows * wpr * sizeof(uint32_t);
    // The mask kernel's blocks loop over images: a grid larger than what the device holds at once
    // would run its surplus blocks after the others, on a half-empty device.
    int grid = SIL_MAX_GRID;
    {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            (mode == FX_SIL_OPENCV
                 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sil_mask_kernel<true>, FX_BLOCK, mask_lds)
                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sil_mask_kernel<false>, FX_BLOCK, mask_lds)) ==
                hipSuccess &&
            cus > 0 && per_cu > 0)
            grid = cus * per_cu;
        grid = grid > 4096 ? 4096 : grid;
    }
    // A large batch runs as two halves on two streams: one half's Otsu chains (a few hundred waves, each a
    // serial chain), and the ends of its blur and mask launches, leave most of the device idle, and the other
    // half's kernels fill it.  From three views per resident mask block on (224x224 views: 3 840): measured neutral
    // at 4 000 views, 4 % faster at 5 000 and 8 000, 3-5 % slower at 3 000 and below.
    const int lanes = B >= 3 * (int64_t)grid ? 2 : 1;
    const int64_t per_lane = (B + lanes - 1) / lanes;
    int64_t chunk = (int64_t)(1ll << 31) / px;   // blurred copies of a chunk: at most 2 GiB of scratch per half
    chunk = chunk < 1 ? 1 : (chunk > 65536 ? 65536 : chunk);
    chunk = chunk > per_lane ? per_lane : chunk;

    const size_t blur_bytes = sil_align((size_t)chunk * px);
    const size_t hist_bytes = sil_align((size_t)chunk * 256 * sizeof(uint32_t));
    const int64_t run_blocks = chunk < grid ? chunk : grid;  // blocks a launch really has
    const size_t runs_bytes = sil_align((size_t)run_blocks * 2 * run_cap * sizeof(uint32_t));
    const size_t level_bytes = sil_align((size_t)chunk * sizeof(int32_t));
    FxCallGuard guard(stream);
    const size_t lane_bytes = blur_bytes + hist_bytes + runs_bytes + level_bytes;
    unsigned char *base = (unsigned char *)guard.scratch(lanes * lane_bytes, 3);
    if (!base)
        return FX_ENOMEM;

    SilArgs A;
    A.H = H;
    A.W = W;
    A.pitch = pitch;
    A.stride = stride;
    A.res = resolution;
    A.crop_margin = crop_margin;
    A.crop = preprocess_crop ? 1 : 0;
    A.min_size = min_size;
    A.blur = base;
    A.hist = (uint32_t *)(base + blur_bytes);
    A.runs = (uint32_t *)(base + blur_bytes + hist_bytes);
    A.run_cap = run_cap;
    A.wpr = wpr;
    A.rows = rows;
    A.mode = mode;
    A.level = (int32_t *)(base + blur_bytes + hist_bytes + runs_bytes);

    const bool aligned = (W % 16 == 0) && (pitch % 16 == 0) && (stride % 16 == 0) && (((uintptr_t)d_img) % 16 == 0);
    const int Wp = (W + 15) & ~15;
    // tile height: fewest rounds of 256 (row, 16-pixel chunk) items over the view
    int tr = 16;
    {
        const int cpr = Wp / 16;
        long best = -1;
        const int halo = mode == FX_SIL_OPENCV ? SIL_CV_HALO : SIL_HALO;
        for (int cand = 16; cand <= SIL_MAX_TROWS - 2 * halo; cand++) {
            const long tiles = (H + cand - 1) / cand;
            const long rounds = ((long)(cand + 2 * halo) * cpr + FX_BLOCK - 1) / FX_BLOCK;
            const long cost = tiles * rounds * 1000 + tiles * (cand + 2 * halo);
            if (best < 0 || cost < best) {
                best = cost;
                tr = cand;
            }
        }
    }
    A.tr = tr;
    const size_t blur_lds = mode == FX_SIL_OPENCV ? (size_t)(tr + 2 * SIL_CV_HALO) * Wp * 2 + (size_t)SIL_CV_HIST_COPIES * SIL_CV_HIST_PITCH * sizeof(uint32_t)
                                                  : (size_t)(tr + 2 * SIL_HALO) * Wp + 256 * sizeof(uint32_t);
    const size_t otsu_lds = (size_t)FX_WAVE * SIL_OTSU_PITCH * sizeof(uint32_t);
    if (mode == FX_SIL_OPENCV) {
        if (blur_lds > 48 * 1024) {
            FX_HIP_CHECK(hipFuncSetAttribute((const void *)sil_blur_cv_kernel<true>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)blur_lds));
            FX_HIP_CHECK(hipFuncSetAttribute((const void *)sil_blur_cv_kernel<false>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)blur_lds));
        }
        FX_HIP_CHECK(hipFuncSetAttribute((const void *)sil_otsu_cv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)otsu_lds));
    }
    if (mask_lds > 48 * 1024)  // beyond the default limit of dynamic LDS; set per call: the current device may differ
        FX_HIP_CHECK(hipFuncSetAttribute(mode == FX_SIL_OPENCV ? (const void *)sil_mask_kernel<true>
                                                               : (const void *)sil_mask_kernel<false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)mask_lds));

    hipStream_t lane_stream[2] = {stream, nullptr};
    if (lanes == 2 && !(lane_stream[1] = guard.fork()))
        return FX_EHIP;
    // chunk k of both halves is enqueued before chunk k + 1 of either: the halves start together
    const int rc = [&]() -> int {
    for (int64_t c0 = 0; c0 < per_lane; c0 += chunk) {
        for (int lane = 0; lane < lanes; lane++) {
            const int64_t b0 = lane * per_lane + c0;
            const int64_t lane_end = (lane + 1) * per_lane < B ? (lane + 1) * per_lane : B;
            if (b0 >= lane_end)
                continue;
            const int64_t nb = lane_end - b0 < chunk ? lane_end - b0 : chunk;
            const hipStream_t s = lane_stream[lane];
            unsigned char *lb = base + lane * lane_bytes;
            A.blur = lb;
            A.hist = (uint32_t *)(lb + blur_bytes);
            A.runs = (uint32_t *)(lb + blur_bytes + hist_bytes);
            A.level = (int32_t *)(lb + blur_bytes + hist_bytes + runs_bytes);
            A.img = d_img + b0 * stride;
            A.B = nb;
            A.mask = d_mask ? d_mask + b0 * (int64_t)resolution * resolution : nullptr;
            A.partial = d_partial ? d_partial + b0 * 30 : nullptr;
            A.info = d_info ? d_info + b0 * 8 : nullptr;
            FX_HIP_CHECK(hipMemsetAsync(A.hist, 0, (size_t)nb * 256 * sizeof(uint32_t), s));
            const dim3 bgrid((unsigned)((H + tr - 1) / tr), (unsigned)nb);
            if (mode == FX_SIL_OPENCV) {
                if (aligned)
                    hipLaunchKernelGGL(sil_blur_cv_kernel<true>, bgrid, dim3(FX_BLOCK), blur_lds, s, A);
                else
                    hipLaunchKernelGGL(sil_blur_cv_kernel<false>, bgrid, dim3(FX_BLOCK), blur_lds, s, A);
                hipLaunchKernelGGL(sil_otsu_cv_kernel, dim3((unsigned)((nb + FX_WAVE - 1) / FX_WAVE)), dim3(FX_BLOCK),
                                   otsu_lds, s, A.hist, nb, 1.0 / (double)px, A.level);
            } else if (aligned) {
                hipLaunchKernelGGL(sil_blur_kernel<true>, bgrid, dim3(FX_BLOCK), blur_lds, s, A);
            } else {
                hipLaunchKernelGGL(sil_blur_kernel<false>, bgrid, dim3(FX_BLOCK), blur_lds, s, A);
            }
            FX_HIP_CHECK(hipGetLastError());
            const int g = (int)(nb < grid ? nb : grid);  // persistent blocks: as many as are resident at once
            {
                // the longest of the kernels; each half's last chunk is what stays
                FxProfileScope prof(s, nb == B ? 0 : nb, lane);
                if (mode == FX_SIL_OPENCV)
                    hipLaunchKernelGGL(sil_mask_kernel<true>, dim3(g), dim3(FX_BLOCK), mask_lds, s, A);
                else
                    hipLaunchKernelGGL(sil_mask_kernel<false>, dim3(g), dim3(FX_BLOCK), mask_lds, s, A);
            }
            FX_HIP_CHECK(hipGetLastError());
        }
    }
    return FX_OK;
    }();
    // also after a failed launch: what the second stream did get must not outlive the call's place in `stream`
    if (lanes == 2 && guard.join() != FX_OK && rc == FX_OK)
        return FX_EHIP;
    return rc;
}

}  // namespace

extern "C" int fx_silhouette_batch_u8(const uint8_t *d_img, int64_t B, int H, int W, int64_t pitch, int64_t stride,
                                      int resolution, double crop_margin, int preprocess_crop, int min_size,
                                      int mode, uint8_t *d_mask, int32_t *d_info, void *stream)
{
    FX_REQUIRE(B <= 0 || d_mask, "fx_silhouette_batch_u8: null image or mask pointer");
    return sil_batch(d_img, B, H, W, pitch, stride, resolution, crop_margin, preprocess_crop, min_size, mode, d_mask,
                     d_info, nullptr, stream);
}

extern "C" int fx_silhouette_moments_batch_u8(const uint8_t *d_img, int64_t B, int H, int W, int64_t pitch,
                                              int64_t stride, int resolution, double crop_margin, int preprocess_crop,
                                              int min_size, int mode, uint8_t *d_mask, int32_t *d_info,
                                              int64_t *d_partial, void *stream)
{
    FX_REQUIRE(B <= 0 || d_partial, "fx_silhouette_moments_batch_u8: null moments pointer");
    return sil_batch(d_img, B, H, W, pitch, stride, resolution, crop_margin, preprocess_crop, min_size, mode, d_mask,
                     d_info, d_partial, stream);
}

// ---------------------------------------------------------------------------
// Gray half of preprocess_image: pad_and_resize (reference :224-240) =
// PIL ImageOps.contain (bicubic) of the cropped view + centred paste on the
// background colour.  PIL's 8-bit resampler is restated: float64 bicubic
// weights over a support of 2 * max(scale, 1) source pixels, normalised, turned
// into 22-bit fixed point with rounding, a pass along x into a uint8 image,
// then a pass along y, each (2^21 + sum(pixel * k)) >> 22 clamped to 0..255.
// ---------------------------------------------------------------------------
namespace {

constexpr int FIT_PRECISION_BITS = 32 - 8 - 2;
constexpr int FIT_UNROLL = 8;
constexpr int FIT_MAX_COEFFS = 4 * SIL_MAX_SIDE + 5 * SIL_MAX_SIDE;  // outSize * ksize never exceeds this

struct FitArgs {
    const uint8_t *img;
    int64_t B;
    int H, W;
    int64_t pitch, stride;
    const int32_t *info;  // [B][8] as written by the silhouette call (columns 2..5 = crop), or null
    int res;
    int bg;
    uint8_t *out;   // [B][res][res]
    uint8_t *temp;  // [grid][SIL_MAX_SIDE * SIL_MAX_SIDE]
};

__device__ __forceinline__ double fit_bicubic(double x)
{
    if (x < 0.0)
        x = -x;
    if (x < 1.0)
        return (1.5 * x - 2.5) * x * x + 1;
    if (x < 2.0)
        return (((x - 5) * x + 8) * x - 4) * -0.5;
    return 0.0;
}

// Coefficients of one axis: k[xx * ksize + j] (fixed point), lo[xx], cnt[xx].
__device__ void fit_coeffs(int in_size, int out_size, int ksize, double scale, double filterscale, int32_t *k,
                           uint16_t *lo, uint16_t *cnt)
{
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    for (int xx = threadIdx.x; xx < out_size; xx += FX_BLOCK) {
        const double center = 0.0 + ((double)xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        xmin = xmin < 0 ? 0 : xmin;
        int xmax = (int)(center + support + 0.5);
        xmax = xmax > in_size ? in_size : xmax;
        xmax -= xmin;
        int32_t *kk = k + xx * ksize;
        double ww = 0.0;
        for (int x = 0; x < xmax; x++)
            ww += fit_bicubic(((double)(x + xmin) - center + 0.5) * ss);
        for (int x = 0; x < xmax; x++) {
            double w = fit_bicubic(((double)(x + xmin) - center + 0.5) * ss);
            if (ww != 0.0)
                w /= ww;
            kk[x] = w < 0 ? (int)(-0.5 + w * (double)(1 << FIT_PRECISION_BITS))
                          : (int)(0.5 + w * (double)(1 << FIT_PRECISION_BITS));
        }
        for (int x = xmax; x < ksize; x++)
            kk[x] = 0;  // the fixed-length tap loops run over all ksize entries
        lo[xx] = (uint16_t)xmin;
        cnt[xx] = (uint16_t)xmax;
    }
}

__device__ __forceinline__ uint8_t fit_clip8(int v)
{
    v >>= FIT_PRECISION_BITS;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ __launch_bounds__(FX_BLOCK) void gray_fit_kernel(FitArgs A)
{
    __shared__ int32_t s_kx[FIT_MAX_COEFFS], s_ky[FIT_MAX_COEFFS];
    __shared__ uint16_t s_xlo[SIL_MAX_SIDE], s_xcnt[SIL_MAX_SIDE], s_ylo[SIL_MAX_SIDE], s_ycnt[SIL_MAX_SIDE];
    const int t = threadIdx.x, res = A.res;
    uint8_t *temp = A.temp + (int64_t)blockIdx.x * SIL_MAX_SIDE * SIL_MAX_SIDE;

    for (int64_t b = blockIdx.x; b < A.B; b += gridDim.x) {
        __syncthreads();
        int x0 = 0, y0 = 0, x1 = A.W, y1 = A.H;
        if (A.info) {
            const int32_t *f = A.info + b * 8;
            x0 = f[2];
            y0 = f[3];
            x1 = f[4];
            y1 = f[5];
        }
        const int cw = x1 - x0, ch = y1 - y0;
        // ImageOps.contain: the longer side becomes res, the other round(ratio * res) (half to even)
        int nw = res, nh = res;
        {
            const double im_ratio = (double)cw / (double)ch;
            if (im_ratio != 1.0) {
                if (im_ratio > 1.0)
                    nh = (int)rint((double)ch / (double)cw * (double)res);
                else
                    nw = (int)rint((double)cw / (double)ch * (double)res);
            }
            nw = nw < 1 ? 1 : nw;
            nh = nh < 1 ? 1 : nh;
        }
        const bool horiz = nw != cw, vert = nh != ch;
        const double sx = (double)((float)cw - 0.0f) / (double)nw, sy = (double)((float)ch - 0.0f) / (double)nh;
        const double fsx = sx < 1.0 ? 1.0 : sx, fsy = sy < 1.0 ? 1.0 : sy;
        const int kxs = (int)ceil(2.0 * fsx) * 2 + 1, kys = (int)ceil(2.0 * fsy) * 2 + 1;
        if (horiz)
            fit_coeffs(cw, nw, kxs, sx, fsx, s_kx, s_xlo, s_xcnt);
        if (vert)
            fit_coeffs(ch, nh, kys, sy, fsy, s_ky, s_ylo, s_ycnt);
        __syncthreads();

        const uint8_t *src = A.img + b * A.stride + (int64_t)y0 * A.pitch + x0;
        // Image.resize reduces an image that is more than a hundred times as tall as it is wide along y first and
        // along x afterwards (PIL/Image.py, "self.size[1] > self.size[0] * 100 and size[1] < self.size[1]"): the
        // uint8 rounding between the passes makes the order visible.  Such a crop is at most five pixels wide here;
        // a pixel and a tap at a time will do.
        if (horiz && vert && ch > 100 * cw && nh < ch) {
            for (int i = t; i < nh * cw; i += FX_BLOCK) {   // pass along y: temp[yy][x], nh rows of cw pixels
                const int yy = i / cw, x = i - yy * cw;
                const int32_t *k = s_ky + yy * kys;
                const int n = s_ycnt[yy], lo = s_ylo[yy];
                int acc = 1 << (FIT_PRECISION_BITS - 1);
                for (int j = 0; j < n; j++)
                    acc += __mul24((int)src[(int64_t)(lo + j) * A.pitch + x], k[j]);
                temp[i] = fit_clip8(acc);
            }
            __threadfence_block();
            __syncthreads();
            const int qx0 = (res - nw) / 2, qy0 = (res - nh) / 2;
            uint8_t *o = A.out + b * (int64_t)res * res;
            for (int i = t; i < res * res; i += FX_BLOCK) {   // pass along x and paste
                const int Y = i / res, X = i - Y * res;
                const int yy = Y - qy0, xx = X - qx0;
                int v = A.bg;
                if (yy >= 0 && yy < nh && xx >= 0 && xx < nw) {
                    const int32_t *k = s_kx + xx * kxs;
                    const int n = s_xcnt[xx], lo = s_xlo[xx];
                    int acc = 1 << (FIT_PRECISION_BITS - 1);
                    for (int j = 0; j < n; j++)
                        acc += __mul24((int)temp[yy * cw + lo + j], k[j]);
                    v = fit_clip8(acc);
                }
                o[i] = (uint8_t)v;
            }
            continue;
        }
        // pass along x: temp[y][xx], ch rows of nw pixels
        // The common case (no reduction: five taps per axis) takes FIT_UNROLL pixels
        // per thread and step with every load issued before the first use: one
        // pixel and one tap at a time, a thread waits a memory latency per tap.
        if (horiz && kxs == 5) {
            const int total = ch * nw;
            for (int i0 = t; i0 < total; i0 += FX_BLOCK * FIT_UNROLL) {
                int v[FIT_UNROLL][5], kofs[FIT_UNROLL];
#pragma unroll
                for (int u = 0; u < FIT_UNROLL; u++) {
                    const int i = i0 + u * FX_BLOCK;
                    const int ic = i < total ? i : 0;
                    const int y = ic / nw, xx = ic - y * nw;
                    const uint8_t *p = src + (int64_t)y * A.pitch + s_xlo[xx];
                    const int n = s_xcnt[xx];
                    kofs[u] = xx * 5;
#pragma unroll
                    for (int j = 0; j < 5; j++)
                        v[u][j] = p[j < n ? j : n - 1];  // taps beyond n have a zero coefficient
                }
#pragma unroll
                for (int u = 0; u < FIT_UNROLL; u++) {
                    int acc = 1 << (FIT_PRECISION_BITS - 1);
#pragma unroll
                    for (int j = 0; j < 5; j++)
                        acc += __mul24(v[u][j], s_kx[kofs[u] + j]);
                    if (i0 + u * FX_BLOCK < total)
                        temp[i0 + u * FX_BLOCK] = fit_clip8(acc);
                }
            }
            __threadfence_block();
            __syncthreads();
        } else if (horiz) {
            for (int i = t; i < ch * nw; i += FX_BLOCK) {
                const int y = i / nw, xx = i - y * nw;
                const uint8_t *p = src + (int64_t)y * A.pitch + s_xlo[xx];
                const int32_t *k = s_kx + xx * kxs;
                const int n = s_xcnt[xx];
                int acc = 1 << (FIT_PRECISION_BITS - 1);
                for (int j = 0; j < n; j++)
                    acc += __mul24((int)p[j], k[j]);
                temp[i] = fit_clip8(acc);
            }
            __threadfence_block();
            __syncthreads();
        }
        // pass along y and paste
        const int px0 = (res - nw) / 2, py0 = (res - nh) / 2;
        uint8_t *out = A.out + b * (int64_t)res * res;
        if (!horiz && !vert) {
            // The crop already has the target size (e.g. uncropped views at the working
            // resolution): a copy with a background frame.  A thread takes 16 pixels of
            // a row so that its 16 loads are in flight together; one pixel per step
            // would cost a memory latency per pixel.
            const int cpr = (res + 15) >> 4;
            for (int it = t; it < res * cpr; it += FX_BLOCK) {
                const int Y = it / cpr, X0 = (it - Y * cpr) << 4;
                const int yy = Y - py0;
                const bool row_in = yy >= 0 && yy < nh;
                const uint8_t *srow = src + (int64_t)(row_in ? yy : 0) * A.pitch;
                uint32_t v[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int xx = X0 + k - px0;
                    v[k] = srow[xx < 0 ? 0 : (xx > nw - 1 ? nw - 1 : xx)];
                }
                uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int xx = X0 + k - px0;
                    const uint32_t px = (row_in && xx >= 0 && xx < nw) ? v[k] : (uint32_t)A.bg;
                    w[k >> 2] |= px << (8 * (k & 3));
                }
                uint8_t *o = out + Y * res + X0;
                if (X0 + 16 <= res && ((uintptr_t)o & 3) == 0) {
                    uint32_t *o4 = reinterpret_cast<uint32_t *>(o);
                    o4[0] = w[0], o4[1] = w[1], o4[2] = w[2], o4[3] = w[3];
                } else {
                    for (int k = 0; k < 16 && X0 + k < res; k++)
                        o[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
                }
            }
            continue;
        }
        const bool fast_y = !vert || kys == 5;
        for (int i0 = t; fast_y && i0 < res * res; i0 += FX_BLOCK * FIT_UNROLL) {
            int v[FIT_UNROLL][5], kofs[FIT_UNROLL];
            bool inside[FIT_UNROLL];
#pragma unroll
            for (int u = 0; u < FIT_UNROLL; u++) {
                const int i = i0 + u * FX_BLOCK;
                const int ic = i < res * res ? i : 0;
                const int Y = ic / res, X = ic - Y * res;
                int yy = Y - py0, xx = X - px0;
                inside[u] = yy >= 0 && yy < nh && xx >= 0 && xx < nw;
                yy = inside[u] ? yy : 0;
                xx = inside[u] ? xx : 0;
                const int n = vert ? (int)s_ycnt[yy] : 1, lo = vert ? (int)s_ylo[yy] : yy;
                kofs[u] = yy * 5;
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const int r = lo + (j < n ? j : n - 1);
                    v[u][j] = (!vert && j > 0) ? 0 : horiz ? (int)temp[r * nw + xx] : (int)src[(int64_t)r * A.pitch + xx];
                }
            }
#pragma unroll
            for (int u = 0; u < FIT_UNROLL; u++) {
                int val = v[u][0];
                if (vert) {
                    int acc = 1 << (FIT_PRECISION_BITS - 1);
#pragma unroll
                    for (int j = 0; j < 5; j++)
                        acc += __mul24(v[u][j], s_ky[kofs[u] + j]);
                    val = fit_clip8(acc);
                }
                if (i0 + u * FX_BLOCK < res * res)
                    out[i0 + u * FX_BLOCK] = (uint8_t)(inside[u] ? val : A.bg);
            }
        }
        for (int i = t; !fast_y && i < res * res; i += FX_BLOCK) {
            const int Y = i / res, X = i - Y * res;
            const int yy = Y - py0, xx = X - px0;
            int v = A.bg;
            if (yy >= 0 && yy < nh && xx >= 0 && xx < nw) {
                if (vert) {
                    const int32_t *k = s_ky + yy * kys;
                    const int n = s_ycnt[yy], lo = s_ylo[yy];
                    int acc = 1 << (FIT_PRECISION_BITS - 1);
                    if (horiz) {
                        for (int j = 0; j < n; j++)
                            acc += __mul24((int)temp[(lo + j) * nw + xx], k[j]);
                    } else {
                        for (int j = 0; j < n; j++)
                            acc += __mul24((int)src[(int64_t)(lo + j) * A.pitch + xx], k[j]);
                    }
                    v = fit_clip8(acc);
                } else {
                    v = horiz ? temp[yy * nw + xx] : src[(int64_t)yy * A.pitch + xx];
                }
            }
            out[i] = (uint8_t)v;
        }
    }
}

}  // namespace

extern "C" int fx_gray_fit_batch_u8(const uint8_t *d_img, int64_t B, int H, int W, int64_t pitch, int64_t stride,
                                    const int32_t *d_info, int resolution, int bg_value, uint8_t *d_out,
                                    void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    FX_REQUIRE(B >= 0, "fx_gray_fit_batch_u8: B = %lld", (long long)B);
    if (B == 0)
        return FX_OK;
    FX_REQUIRE(d_img && d_out, "fx_gray_fit_batch_u8: null image or output pointer");
    FX_REQUIRE(H >= 1 && W >= 1 && H <= SIL_MAX_SIDE && W <= SIL_MAX_SIDE,
               "fx_gray_fit_batch_u8: views of %dx%d; supported up to %dx%d", H, W, SIL_MAX_SIDE, SIL_MAX_SIDE);
    FX_REQUIRE(resolution >= 1 && resolution <= SIL_MAX_SIDE, "fx_gray_fit_batch_u8: resolution %d outside 1..%d",
               resolution, SIL_MAX_SIDE);
    FX_REQUIRE(pitch >= W && (B == 1 || stride >= (int64_t)(H - 1) * pitch + W),
               "fx_gray_fit_batch_u8: pitch %lld / stride %lld too small for %dx%d", (long long)pitch,
               (long long)stride, H, W);
    FX_REQUIRE(bg_value >= 0 && bg_value <= 255, "fx_gray_fit_batch_u8: bg_value %d", bg_value);
    // persistent blocks: as many as the device holds at once (surplus blocks would run afterwards, on a partly empty device)
    int resident = 768;
    {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gray_fit_kernel, FX_BLOCK, 0) == hipSuccess && cus > 0 &&
            per_cu > 0)
            resident = cus * per_cu > 2048 ? 2048 : cus * per_cu;
    }
    const int grid = (int)(B < resident ? B : resident);
    FxCallGuard guard((hipStream_t)stream_);
    uint8_t *temp = (uint8_t *)guard.scratch((size_t)grid * SIL_MAX_SIDE * SIL_MAX_SIDE, 4);
    if (!temp)
        return FX_ENOMEM;
    FitArgs A;
    A.img = d_img;
    A.B = B;
    A.H = H;
    A.W = W;
    A.pitch = pitch;
    A.stride = stride;
    A.info = d_info;
    A.res = resolution;
    A.bg = bg_value;
    A.out = d_out;
    A.temp = temp;
    FxProfileScope prof(stream);
    hipLaunchKernelGGL(gray_fit_kernel, dim3(grid), dim3(FX_BLOCK), 0, stream, A);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

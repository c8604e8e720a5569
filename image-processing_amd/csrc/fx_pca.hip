// Point-set PCA descriptor kernels (reference
// src/features/extract_3d_pca_features.py:59-145, inputs widened from the
// float32 file contents as load_sample :48-56 does).
//
// Sets of up to 2048 points are read from HBM exactly once and stay in
// registers for every pass:
//   * pca_sets_full_kernel    a wave per set (8 / 16 / 32 points per lane): the
//                             aligned 512- and 1024-point batches of the
//                             pipeline, and (RAGGED) every other layout of 257
//                             to 2048 points -- set lists, CSR offsets, 2-D;
//   * pca_sets_quad_kernel    sets of up to 256 points, four to a wave;
//   * pca_sets_redo_kernel    the few sets those hand on (pca_sets_reg_body:
//                             float64 decisions throughout);
//   * pca_sets_wave_kernel    a set staged in the wave's LDS slab (batches of
//                             fewer than three points per set: NaN rows).
// What is per set and serial (mean, covariance, eigen-solve, the nine scalars)
// is computed by one wave per block, a lane per set, or by the 16 lanes of a
// set's row.
//
// Large clouds, and every set of a float64 batch: cut into chunks of 8192
// points that a persistent launch works through twice (sums; projections +
// histogram), see the section on large clouds below; a CSR batch is sorted into
// the lists of the kernels on the device.
#include "fx_common.h"
#include "fx_eig.h"
#include <stdlib.h>
#include <string.h>

namespace {

struct PcaArgs {
    const float *pts;
    const double *pts64;     // float64 input (fx_pca_batch_f64): the points, and pts is NULL
    const int64_t *offsets;  // device, B+1 entries, or NULL
    const int32_t *set_ids;  // device list of the sets this launch handles, or NULL = all
    int64_t n_sets;          // number of sets this launch handles
    int64_t N;               // fixed set size when offsets == NULL
    int dims;
    int n_bins;
    int normalize;
    float *out;
    double *eig;
    int slab_floats;  // LDS floats per wave (wave kernel)
    const int32_t *n_sets_dev;  // when set: the number of entries of set_ids, known on the device only
    int32_t *redo_list;         // pca_sets_full_kernel: sets left to pca_sets_redo_kernel, and their number
    int32_t *redo_count;
    int redo_zeroed;            // host side: *redo_count is zero already (no memset ahead of the launch)
};

constexpr int HIST_MAX = 64;

struct SetView {
    const float *base;
    int64_t n;
    int64_t set;
};

__device__ __forceinline__ SetView resolve_set(const PcaArgs &A, int64_t slot)
{
    SetView v;
    v.set = A.set_ids ? (int64_t)A.set_ids[slot] : slot;
    if (A.offsets) {
        const int64_t a = A.offsets[v.set];
        v.n = A.offsets[v.set + 1] - a;
        v.base = A.pts + a * A.dims;
    } else {
        v.n = A.N;
        v.base = A.pts + v.set * A.N * A.dims;
    }
    return v;
}

__device__ __forceinline__ void write_nan_row(const PcaArgs &A, int64_t set, int lane, int stride)
{
    const int width = 9 + A.n_bins;
    for (int i = lane; i < width; i += stride)
        A.out[set * width + i] = __builtin_nanf("");
    if (A.eig)
        for (int i = lane; i < 6; i += stride)
            A.eig[set * 6 + i] = __builtin_nan("");
}

// Everything after the second-moment sums: covariance scaling exactly as
// np.cov does (multiply by 1/(n-1)), eigen-decomposition, the nine scalars.
template <bool FAST = false>
__device__ __forceinline__ void cov_to_head(const double (&S)[6], int64_t n, double (&head)[9], double (&lam)[3],
                                            double (&v1)[3])
{
    const double inv = 1.0 / (double)(n - 1);
    fx_eig3<FAST>(S[0] * inv, S[1] * inv, S[2] * inv, S[3] * inv, S[4] * inv, S[5] * inv, lam, v1);
    fx_pca_head(lam, head);
}

template <int WPB>
__global__ __launch_bounds__(FX_WAVE *WPB) void pca_sets_wave_kernel(PcaArgs A)
{
    extern __shared__ __align__(16) float lds[];
    const int lane = fx_lane(), wave = fx_wave();
    const int64_t slot = (int64_t)blockIdx.x * WPB + wave;
    if (slot >= (A.n_sets_dev ? (int64_t)*A.n_sets_dev : A.n_sets))
        return;  // whole wave leaves; no block-level barrier is used below
    const SetView sv = resolve_set(A, slot);
    const int64_t set = sv.set;
    const int n = (int)sv.n;
    const int dims = A.dims;
    if (n < 3) {
        write_nan_row(A, set, lane, FX_WAVE);
        return;
    }

    // Element k of the set lives at slab[k]; the slab start is shifted so that
    // 16-byte alignment in global memory and in LDS coincide.
    float *slab = lds + (size_t)wave * A.slab_floats + (((uintptr_t)sv.base >> 2) & 3);
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds + (size_t)WPB * A.slab_floats) + wave * HIST_MAX;

    // ---- stage the set: HBM -> LDS, 16 B per lane -----------------------------
    {
        const int F = n * dims;
        const float *g = sv.base;
        int head = (int)((16 - ((uintptr_t)g & 15)) & 15) >> 2;  // floats up to 16 B alignment
        head = head > F ? F : head;
        if (lane < head)
            slab[lane] = g[lane];
        const int body = (F - head) & ~3;
        const float4 *g4 = reinterpret_cast<const float4 *>(g + head);
        float4 *s4 = reinterpret_cast<float4 *>(slab + head);
        for (int i = lane; i < (body >> 2); i += FX_WAVE)
            s4[i] = g4[i];
        const int tail0 = head + body;
        if (tail0 + lane < F)
            slab[tail0 + lane] = g[tail0 + lane];
        if (lane < HIST_MAX)
            hist[lane] = 0;
    }
    // LDS traffic of one wave is processed in issue order; the fence keeps the
    // compiler from moving the reads above the writes.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    auto px = [&](int i) -> double { return (double)slab[i * dims]; };
    auto py = [&](int i) -> double { return (double)slab[i * dims + 1]; };
    auto pz = [&](int i) -> double { return dims == 3 ? (double)slab[i * dims + 2] : 0.0; };

    // ---- one pass: first and second moments about a pivot -------------------
    // The pivot is the set's first point, so the shifted coordinates are of the
    // size of the spread and S = sum d d^T - (sum d)(sum d)^T / n loses nothing
    // at float64 (the inputs are float32).
    const float kx32 = slab[0], ky32 = slab[1], kz32 = dims == 3 ? slab[2] : 0.0f;
    const double kx = kx32, ky = ky32, kz = kz32;
    double D[3] = {0, 0, 0}, Q[6] = {0, 0, 0, 0, 0, 0};
    for (int i = lane; i < n; i += FX_WAVE) {
        const double dx = px(i) - kx, dy = py(i) - ky, dz = pz(i) - kz;
        D[0] += dx;
        D[1] += dy;
        D[2] += dz;
        Q[0] = fma(dx, dx, Q[0]);
        Q[1] = fma(dx, dy, Q[1]);
        Q[2] = fma(dx, dz, Q[2]);
        Q[3] = fma(dy, dy, Q[3]);
        Q[4] = fma(dy, dz, Q[4]);
        Q[5] = fma(dz, dz, Q[5]);
    }
#pragma unroll
    for (int k = 0; k < 3; k++)
        D[k] = fx_wave_sum_f64(D[k]);
#pragma unroll
    for (int k = 0; k < 6; k++)
        Q[k] = fx_wave_sum_f64(Q[k]);
    const double dn = (double)n;
    const double inv_n = 1.0 / dn;
    const double ox = D[0] * inv_n, oy = D[1] * inv_n, oz = D[2] * inv_n;  // mean - pivot
    const double mx = kx + ox, my = ky + oy, mz = kz + oz;
    double S[6];
    S[0] = fma(-D[0], ox, Q[0]);
    S[1] = fma(-D[0], oy, Q[1]);
    S[2] = fma(-D[0], oz, Q[2]);
    S[3] = fma(-D[1], oy, Q[3]);
    S[4] = fma(-D[1], oz, Q[4]);
    S[5] = fma(-D[2], oz, Q[5]);

    // ---- optional normalisation by the largest centred norm ----------------
    double rmax = 0.0;
    if (A.normalize) {
        double r2 = 0.0;
        for (int i = lane; i < n; i += FX_WAVE) {
            const double dx = px(i) - mx, dy = py(i) - my, dz = pz(i) - mz;
            r2 = fmax(r2, dx * dx + dy * dy + dz * dz);
        }
        rmax = sqrt(fx_wave_max_f64(r2));
    }
    const bool scale = A.normalize && rmax > 0.0;
    if (scale) {
        const double w = 1.0 / (rmax * rmax);
#pragma unroll
        for (int k = 0; k < 6; k++)
            S[k] *= w;
    }

    // every lane holds the same sums and runs the same eigen-solve
    double head[9], lam[3], v1[3];
    cov_to_head<true>(S, n, head, lam, v1);

    // ---- |projection on PC1|: maximum, then histogram ----------------------
    // float64 value of a point (the formula of record)
    auto proj = [&](int i) -> double {
        double dx = px(i) - mx, dy = py(i) - my, dz = pz(i) - mz;
        if (scale) {
            dx /= rmax;
            dy /= rmax;
            dz /= rmax;
        }
        return fabs(fma(dz, v1[2], fma(dy, v1[1], dx * v1[0])));
    };
    // float32 value: pivot-relative coordinates keep its error at the scale of
    // the spread, eps32 bounds it generously.  It settles everything that is not
    // within eps32 of a decision; those few points are decided in float64.
    const float ox32 = (float)ox, oy32 = (float)oy, oz32 = (float)oz;
    const float w32 = scale ? (float)(1.0 / rmax) : 1.0f;
    const float vx32 = (float)v1[0] * w32, vy32 = (float)v1[1] * w32, vz32 = (float)v1[2] * w32;
    auto proj32 = [&](int i) -> float {
        const float dx = (slab[i * dims] - kx32) - ox32, dy = (slab[i * dims + 1] - ky32) - oy32;
        const float dz = dims == 3 ? (slab[i * dims + 2] - kz32) - oz32 : -oz32;
        return fabsf(fmaf(dz, vz32, fmaf(dy, vy32, dx * vx32)));
    };
    float top32 = 0.0f;
    for (int i = lane; i < n; i += FX_WAVE)
        top32 = fmaxf(top32, proj32(i));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        top32 = fmaxf(top32, __shfl_xor(top32, o, FX_WAVE));
    const float eps32 = 2e-6f * (top32 + (fabsf(ox32) + fabsf(oy32) + fabsf(oz32)) * w32) + 1e-37f;
    double top = 0.0;
    for (int i = lane; i < n; i += FX_WAVE)
        if (proj32(i) >= top32 - 2.0f * eps32)  // the exact maximum is among these
            top = fmax(top, proj(i));
    top = fx_wave_max_f64(top);

    const int nb = A.n_bins;
    const int width = 9 + nb;
    if (top > 0.0) {
        const double step = top / (double)nb;
        const double inv_step = (double)nb / top;
        const float inv32 = (float)inv_step;
        const float margin = 2.0f * eps32 * inv32 + 1e-5f * (float)nb;
        for (int i = lane; i < n; i += FX_WAVE) {
            const float q = proj32(i) * inv32;
            const float f = floorf(q);
            const float frac = q - f;
            int bin;
            if (frac < margin || frac > 1.0f - margin || q >= (float)nb)
                bin = fx_hist_bin(proj(i), step, inv_step, nb);
            else
                bin = (int)f;
            atomicAdd(&hist[bin], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < nb)
            A.out[set * width + 9 + lane] = (float)((double)hist[lane] / dn);
    } else if (lane < nb) {
        A.out[set * width + 9 + lane] = lane == 0 ? 1.0f : 0.0f;
    }
    if (lane < 9) {
        double h = head[0];
#pragma unroll
        for (int k = 1; k < 9; k++)
            h = lane == k ? head[k] : h;
        A.out[set * width + lane] = (float)h;
    }
    if (A.eig && lane < 6) {
        double e = lam[0];
        e = lane == 1 ? lam[1] : e;
        e = lane == 2 ? lam[2] : e;
        e = lane == 3 ? v1[0] : e;
        e = lane == 4 ? v1[1] : e;
        e = lane == 5 ? v1[2] : e;
        A.eig[set * 6 + lane] = e;
    }
}

// ------------------------------------------------------- register-resident sets
// Equal-sized sets of up to 64 * PPL points (the pipeline's 1024-point clouds:
// PPL = 16): every lane loads 4-point groups (48 contiguous bytes, three 16-byte
// loads) straight into registers and keeps its points there for all passes, so
// the kernel needs no LDS slab, runs at register-limited occupancy and the
// passes are straight-line code over PPL independent points.  Same arithmetic
// as pca_sets_wave_kernel.
#ifndef FX_PCA_WAVES
#define FX_PCA_WAVES 4
#endif
// Sum of nine float64 values over the lanes of a wave through LDS; every lane
// receives the same nine totals (bit for bit: the lanes go on to run the same
// eigen-solve).  Lane l < 63 adds a ninth of row l / 7; nine lanes add the seven
// parts of a row.  36 LDS operations and 15 additions per lane instead of 54
// additions and 108 cross-lane moves of a butterfly.  `slab` holds
// RED9_WORDS doubles of this wave.
constexpr int RED9_PITCH = 65;  // rows one entry apart in the banks
constexpr int RED9_WORDS = 9 * RED9_PITCH + 64 + 16;
constexpr int RED9_TOT = 9 * RED9_PITCH + 64;   // the nine totals; seven spare doubles follow them
// Leaves the totals in slab[RED9_TOT .. RED9_TOT + 8], visible to the wave.
__device__ __forceinline__ void wave_sum9_slab(const double (&v)[9], double *slab)
{
    const int lane = fx_lane();
#pragma unroll
    for (int k = 0; k < 9; k++)
        slab[k * RED9_PITCH + lane] = v[k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double *part = slab + 9 * RED9_PITCH, *tot = slab + RED9_TOT;
    if (lane < 63) {
        const int row = lane / 7, seg = lane - row * 7;
        const double *r = slab + row * RED9_PITCH + seg * 9;
        double s = r[0];
#pragma unroll
        for (int j = 1; j < 9; j++)
            s += r[j];
        if (seg == 6)
            s += r[9];  // entry 63
        part[lane] = s;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < 9) {
        const double *r = part + lane * 7;
        double s = r[0];
#pragma unroll
        for (int j = 1; j < 7; j++)
            s += r[j];
        tot[lane] = s;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void wave_sum9(double (&v)[9], double *slab)
{
    wave_sum9_slab(v, slab);
#pragma unroll
    for (int k = 0; k < 9; k++)
        v[k] = slab[RED9_TOT + k];
}

#ifndef FX_PCA_NW
#define FX_PCA_NW 4
#endif
// RAGGED: sets of different sizes (CSR offsets, work list) or a fixed size that
// is not a multiple of four points / not 16-byte aligned.  Lane l then holds the
// points l, l + 64, ... and loads them one by one (12 bytes, 768 contiguous
// bytes per load instruction of the wave); 4 % slower than the 16-byte loads of
// the aligned layout, where lane l holds the four-point groups l, l + 64, ...
template <int PPL, bool RAGGED, int NW>
__device__ __forceinline__ void pca_sets_reg_body(const PcaArgs &A, const int64_t blk, const int64_t n_sets)
{
    // The histogram of a wave is kept in HIST_COPIES copies, lane l counting in
    // copy l % HIST_COPIES: most points of a set fall into a few bins, and 64
    // lanes adding to the same few LDS words serialise.  A pitch of 65 words
    // puts the same bin of different copies into different banks.
    constexpr int HIST_COPIES = 8, HIST_PITCH = HIST_MAX + 1;
    __shared__ uint32_t hist_all[NW][HIST_COPIES * HIST_PITCH];
    __shared__ double red_all[NW][RED9_WORDS];
    __shared__ double cov_all[NW][6], eig_all[NW][6];
    const int lane = fx_lane(), wave = fx_wave();
    const int64_t slot = blk * NW + wave;
    // a wave past the last set stays for the block barriers: it works on the first set and writes nothing
    const bool in_range = slot < n_sets;
    const SetView sv = resolve_set(A, in_range ? slot : 0);
    const int64_t set = sv.set;
    const int n = (int)sv.n;
    const bool live = in_range && n >= 3;   // a set of fewer than three points gets a NaN row
    if (RAGGED && in_range && n < 3)
        write_nan_row(A, set, lane, FX_WAVE);
    uint32_t *hist = hist_all[wave];
#pragma unroll
    for (int k = 0; k < HIST_COPIES; k++)
        hist[k * HIST_PITCH + lane] = 0;
    uint32_t *my_hist = hist + (lane % HIST_COPIES) * HIST_PITCH;

    float x[PPL], y[PPL], z[PPL];
    constexpr int G = RAGGED ? PPL : PPL / 4;  // predicates: per point, or per 4-point group
    bool have[G];
    if (RAGGED) {
#pragma unroll
        for (int k = 0; k < PPL; k++) {
            const int p = lane + FX_WAVE * k;
            have[k] = p < n;
            x[k] = y[k] = z[k] = 0.0f;
            if (have[k]) {   // 2-D points are (x, y, 0)
                const float *q = sv.base + A.dims * p;
                x[k] = q[0], y[k] = q[1];
                if (A.dims == 3)
                    z[k] = q[2];
            }
        }
    } else {
        const float4 *g4 = reinterpret_cast<const float4 *>(sv.base);
#pragma unroll
        for (int k = 0; k < PPL / 4; k++) {
            const int grp = lane + FX_WAVE * k;  // points 4*grp .. 4*grp+3
            have[k] = 4 * grp < n;               // n is a multiple of 4
            const int gc = have[k] ? grp : 0;
            const float4 a = g4[3 * gc], b = g4[3 * gc + 1], c = g4[3 * gc + 2];
            x[4 * k] = a.x, y[4 * k] = a.y, z[4 * k] = a.z;
            x[4 * k + 1] = a.w, y[4 * k + 1] = b.x, z[4 * k + 1] = b.y;
            x[4 * k + 2] = b.z, y[4 * k + 2] = b.w, z[4 * k + 2] = c.x;
            x[4 * k + 3] = c.y, y[4 * k + 3] = c.z, z[4 * k + 3] = c.w;
        }
    }
    // pivot = the set's first point (lane 0, group 0)
    const float kx32 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x[0])));
    const float ky32 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(y[0])));
    const float kz32 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(z[0])));
    const double kx = kx32, ky = ky32, kz = kz32;

    double D[3] = {0, 0, 0}, Q[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < PPL; i++) {
        if (have[RAGGED ? i : i >> 2]) {
            const double dx = (double)x[i] - kx, dy = (double)y[i] - ky, dz = (double)z[i] - kz;
            D[0] += dx;
            D[1] += dy;
            D[2] += dz;
            Q[0] = fma(dx, dx, Q[0]);
            Q[1] = fma(dx, dy, Q[1]);
            Q[2] = fma(dx, dz, Q[2]);
            Q[3] = fma(dy, dy, Q[3]);
            Q[4] = fma(dy, dz, Q[4]);
            Q[5] = fma(dz, dz, Q[5]);
        }
    }
    {
        double v9[9] = {D[0], D[1], D[2], Q[0], Q[1], Q[2], Q[3], Q[4], Q[5]};
        wave_sum9(v9, red_all[wave]);
#pragma unroll
        for (int k = 0; k < 3; k++)
            D[k] = v9[k];
#pragma unroll
        for (int k = 0; k < 6; k++)
            Q[k] = v9[3 + k];
    }
    const double dn = (double)n;
    const double inv_n = 1.0 / dn;
    const double ox = D[0] * inv_n, oy = D[1] * inv_n, oz = D[2] * inv_n;  // mean - pivot
    const double mx = kx + ox, my = ky + oy, mz = kz + oz;
    double S[6];
    S[0] = fma(-D[0], ox, Q[0]);
    S[1] = fma(-D[0], oy, Q[1]);
    S[2] = fma(-D[0], oz, Q[2]);
    S[3] = fma(-D[1], oy, Q[3]);
    S[4] = fma(-D[1], oz, Q[4]);
    S[5] = fma(-D[2], oz, Q[5]);

    double rmax = 0.0;
    if (A.normalize) {
        double r2 = 0.0;
#pragma unroll
        for (int i = 0; i < PPL; i++) {
            if (have[RAGGED ? i : i >> 2]) {
                const double dx = (double)x[i] - mx, dy = (double)y[i] - my, dz = (double)z[i] - mz;
                r2 = fmax(r2, dx * dx + dy * dy + dz * dz);
            }
        }
        rmax = sqrt(fx_wave_max_f64_dpp(r2));
    }
    const bool scale = A.normalize && rmax > 0.0;
    if (scale) {
        const double w = 1.0 / (rmax * rmax);
#pragma unroll
        for (int k = 0; k < 6; k++)
            S[k] *= w;
    }

    // The eigen-solve is serial code: one wave of the block runs it for all NW
    // sets of the block at once, a lane per set, instead of every wave running
    // it on 64 identical lanes.  The duty rotates over the waves with the block
    // index, so that it does not pile up on one SIMD.
    double lam[3], v1[3];
    {
        const double inv = 1.0 / (double)(n - 1);  // np.cov multiplies by 1/(n-1)
        if (lane < 6) {
            double c = S[0];
            c = lane == 1 ? S[1] : c;
            c = lane == 2 ? S[2] : c;
            c = lane == 3 ? S[3] : c;
            c = lane == 4 ? S[4] : c;
            c = lane == 5 ? S[5] : c;
            cov_all[wave][lane] = c * inv;
        }
        __syncthreads();
        if (wave == (int)(blk % NW) && lane < NW) {
            const double *c = cov_all[lane];
            double l3[3], w3[3];
            // the direct solve where it applies (well separated, clearly positive eigenvalues), the sweeps otherwise
            if (!fx_eig3_direct(c[0], c[1], c[2], c[3], c[4], c[5], l3, w3))
                fx_eig3<true>(c[0], c[1], c[2], c[3], c[4], c[5], l3, w3);
            double *e = eig_all[lane];
            e[0] = l3[0], e[1] = l3[1], e[2] = l3[2];
            e[3] = w3[0], e[4] = w3[1], e[5] = w3[2];
        }
        __syncthreads();
        const double *e = eig_all[wave];
        lam[0] = e[0], lam[1] = e[1], lam[2] = e[2];
        v1[0] = e[3], v1[1] = e[4], v1[2] = e[5];
    }
    // the nine scalars (fx_pca_head), lane k computing entry k: one division
    // per lane instead of five one after the other
#pragma unroll
    for (int q = 0; q < 3; q++)
        lam[q] = lam[q] < 0.0 ? 0.0 : lam[q];
    const double trace = lam[0] + lam[1] + lam[2];
    double head_mine;
    {
        const double safe = trace > 1e-12 ? trace : 1e-12;
        const int e = lane < 9 ? lane : 0;
        const double num = (e == 0 || e == 3 || e == 6) ? lam[0] : (e == 1 || e == 4 || e == 7) ? lam[1] : (e == 2 || e == 5) ? lam[2] : trace;
        const double den = e < 3 || e == 8 ? 1.0 : e < 6 ? safe : e == 6 ? lam[1] + 1e-12 : lam[2] + 1e-12;
        head_mine = num / den;
    }

    // float64 value of a point (the formula of record)
    auto proj = [&](float px, float py, float pz) -> double {
        double dx = (double)px - mx, dy = (double)py - my, dz = (double)pz - mz;
        if (scale) {
            dx /= rmax;
            dy /= rmax;
            dz /= rmax;
        }
        return fabs(fma(dz, v1[2], fma(dy, v1[1], dx * v1[0])));
    };
    // float32 value, pivot-relative so that its error is at the scale of the
    // spread; eps32 bounds it generously.  It settles every point that is not
    // within eps32 of a decision; those few are decided in float64.
    const float ox32 = (float)ox, oy32 = (float)oy, oz32 = (float)oz;
    const float w32 = scale ? (float)(1.0 / rmax) : 1.0f;
    const float vx32 = (float)v1[0] * w32, vy32 = (float)v1[1] * w32, vz32 = (float)v1[2] * w32;
    float p32[PPL];
    float top32 = 0.0f;
#pragma unroll
    for (int i = 0; i < PPL; i++) {
        const float dx = (x[i] - kx32) - ox32, dy = (y[i] - ky32) - oy32, dz = (z[i] - kz32) - oz32;
        p32[i] = have[RAGGED ? i : i >> 2] ? fabsf(fmaf(dz, vz32, fmaf(dy, vy32, dx * vx32))) : -1.0f;
        top32 = fmaxf(top32, p32[i]);
    }
    top32 = fx_wave_max_f32_dpp(top32);
    const float eps32 = 2e-6f * (top32 + (fabsf(ox32) + fabsf(oy32) + fabsf(oz32)) * w32) + 1e-37f;
    double top = 0.0;
#pragma unroll
    for (int i = 0; i < PPL; i++)
        if (p32[i] >= top32 - 2.0f * eps32)  // the exact maximum is among these
            top = fmax(top, proj(x[i], y[i], z[i]));
    top = fx_wave_max_f64_dpp(top);

    const int nb = A.n_bins;
    const int width = 9 + nb;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (top > 0.0) {
        const double step = top / (double)nb;
        const double inv_step = (double)nb / top;
        const float inv32 = (float)inv_step;
        const float margin = 2.0f * eps32 * inv32 + 1e-5f * (float)nb;
        const float qmax = (float)nb - margin;
#pragma unroll
        for (int i = 0; i < PPL; i++) {
            if (have[RAGGED ? i : i >> 2]) {
                // q within `margin` of an integer (a bin edge), or at the top edge: float64 decides
                const float q = p32[i] * inv32;
                int bin;
                if (fabsf(q - rintf(q)) < margin || q >= qmax)
                    bin = fx_hist_bin(proj(x[i], y[i], z[i]), step, inv_step, nb);
                else
                    bin = (int)q;
                atomicAdd(&my_hist[bin], 1u);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (live && lane < nb) {
            // count / n: both are integers below 2^24, so the float32 quotient is the
            // float64 quotient rounded to float32
            uint32_t h = 0;
#pragma unroll
            for (int k = 0; k < HIST_COPIES; k++)
                h += hist[k * HIST_PITCH + lane];
            A.out[set * width + 9 + lane] = (float)h / (float)n;
        }
    } else if (live && lane < nb) {
        A.out[set * width + 9 + lane] = lane == 0 ? 1.0f : 0.0f;
    }
    if (live && lane < 9)
        A.out[set * width + lane] = (float)head_mine;
    if (live && A.eig && lane < 6) {
        double e = lam[0];
        e = lane == 1 ? lam[1] : e;
        e = lane == 2 ? lam[2] : e;
        e = lane == 3 ? v1[0] : e;
        e = lane == 4 ? v1[1] : e;
        e = lane == 5 ? v1[2] : e;
        A.eig[set * 6 + lane] = e;
    }
}

// The sets pca_sets_full_kernel hands back (A.set_ids, their number in *A.n_sets_dev): a small grid walks the list.
template <int PPL, int NW, bool RAGGED = false>
__global__ __launch_bounds__(FX_WAVE *NW, FX_PCA_WAVES) void pca_sets_redo_kernel(PcaArgs A)
{
    const int64_t n_sets = (int64_t)*A.n_sets_dev;
    for (int64_t blk = blockIdx.x; blk * NW < n_sets; blk += gridDim.x) {
        pca_sets_reg_body<PPL, RAGGED, NW>(A, blk, n_sets);
        __syncthreads();
    }
}

// ------------------------------------------------- full sets: the C4 kernel
// Equal-sized, 16-byte aligned 3-D sets of exactly 64 * PPL points (the
// pipeline's 1024-point clouds): no per-point predicates, 1 / n and
// 1 / (n - 1) from the host.  Same float64 arithmetic as pca_sets_reg_body.
// What differs is how the few points are handled that float32 cannot settle
// (the candidates for the exact maximum, points within rounding of a bin
// edge): instead of a float64 branch per point index (thirty-two inlined
// copies, the widened coordinates kept alive for them) such points are
// appended to a short LDS list and evaluated afterwards, a lane per point;
// and one pass over the points settles maximum and bins together.
// The float32 pre-value of the projection is three FMAs on the raw
// coordinates (v.x - v.mean); its error bound grows with the distance of the
// set from the origin (sets far away send more points to the float64 path,
// results are the same).
// What the solving wave hands back for one set.
struct SetRec {
    double lam[3], v1[3];   // eigenvalues (descending, negative ones clipped to 0), first principal axis
    double mean[3], rmax;   // rmax = 0: no scaling
    float vq[4];            // float32 projection: vq[0..2] . x - vq[3]
    float eps32;            // its error bound
    int solved;
    float head[9];
    float pad;
};
// RAGGED: sets of any size up to 64 * PPL points, 2-D or 3-D, any alignment, through set lists and CSR offsets
// (lane l holds the points l, l + 64, ..., loaded one by one).  The slots past the end of a set hold the set's
// first point, which is the pivot of the sums: such a slot adds exact zeros to them and, being a copy of a point,
// changes no maximum, so the passes stay free of predicates; only the histogram add is masked.
template <int PPL, int NW, bool NORMALIZE, bool RAGGED = false>
// (sets of 512 points leave room for a fifth wave per SIMD: 12 % faster there; at 1024 points it would spill;
// 32 points per lane take 96 registers for the coordinates: two waves)
__global__ __launch_bounds__(FX_WAVE *NW, PPL > 16 ? 2 : (PPL <= 8 && !RAGGED) ? 5 : FX_PCA_WAVES) void pca_sets_full_kernel(PcaArgs A, double inv_n, double inv_nm1, double inv_nb)
{
    constexpr int HIST_COPIES = 8, HIST_PITCH = HIST_MAX + 1;
    __shared__ uint32_t hist_all[NW][HIST_COPIES * HIST_PITCH];
    __shared__ double red_all[NW][RED9_WORDS];
    __shared__ SetRec rec_all[NW];
    static_assert(NW * 16 <= FX_WAVE, "a lane per (set, head scalar)");
    const int lane = fx_lane(), wave = fx_wave();
    constexpr int G = PPL / 4;  // 4-point groups (48 bytes) per lane
    const int64_t slot = (int64_t)blockIdx.x * NW + wave;
    uint32_t *hist = hist_all[wave];
    uint32_t *my_hist = hist + (lane % HIST_COPIES) * HIST_PITCH;
    const int nb = A.n_bins;
    const int width = 9 + nb;
    const int64_t n_sets = RAGGED && A.n_sets_dev ? (int64_t)*A.n_sets_dev : A.n_sets;
    if (RAGGED && (int64_t)blockIdx.x * NW >= n_sets)
        return;  // a surplus block of a launch sized by an upper bound (the whole block leaves, before any barrier)

    {
        // a wave past the last set, or (RAGGED) with a set of fewer than three points, stays for the block
        // barriers and writes no features
        const bool in_range = slot < n_sets;
        float x[PPL], y[PPL], z[PPL];
        int64_t set;
        int n;
        bool live;
        if (!RAGGED) {
            n = FX_WAVE * PPL;
            live = in_range;
            set = live ? slot : n_sets - 1;   // (works on the last set)
            const float4 *g4 = reinterpret_cast<const float4 *>(A.pts + set * (int64_t)(3 * n));
#pragma unroll
            for (int k = 0; k < G; k++) {
                const int grp = lane + FX_WAVE * k;  // points 4 * grp .. 4 * grp + 3
                const float4 a = g4[3 * grp], b = g4[3 * grp + 1], c = g4[3 * grp + 2];
                x[4 * k] = a.x, y[4 * k] = a.y, z[4 * k] = a.z;
                x[4 * k + 1] = a.w, y[4 * k + 1] = b.x, z[4 * k + 1] = b.y;
                x[4 * k + 2] = b.z, y[4 * k + 2] = b.w, z[4 * k + 2] = c.x;
                x[4 * k + 3] = c.y, y[4 * k + 3] = c.z, z[4 * k + 3] = c.w;
            }
        } else {
            const SetView sv = resolve_set(A, in_range ? slot : 0);
            set = sv.set;
            live = in_range && sv.n >= 3 && sv.n <= FX_WAVE * PPL;
            n = live ? (int)sv.n : 0;
            if (in_range && sv.n < 3)
                write_nan_row(A, set, lane, FX_WAVE);
#pragma unroll
            for (int k = 0; k < PPL; k++)
                x[k] = y[k] = z[k] = 0.0f;
            if (live) {   // wave-uniform
                struct P3 {
                    float a, b, c;
                };
                struct P2 {
                    float a, b;
                };
                if (A.dims == 3) {
#pragma unroll
                    for (int k = 0; k < PPL; k++) {
                        const int p = lane + FX_WAVE * k;
                        const P3 v = *reinterpret_cast<const P3 *>(sv.base + 3 * (p < n ? p : 0));
                        x[k] = v.a, y[k] = v.b, z[k] = v.c;
                    }
                } else {   // 2-D points are (x, y, 0)
#pragma unroll
                    for (int k = 0; k < PPL; k++) {
                        const int p = lane + FX_WAVE * k;
                        const P2 v = *reinterpret_cast<const P2 *>(sv.base + 2 * (p < n ? p : 0));
                        x[k] = v.a, y[k] = v.b;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < HIST_COPIES; k++)
            hist[k * HIST_PITCH + lane] = 0;

        // pivot = the set's first point (lane 0, group 0)
        const float kx32 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x[0])));
        const float ky32 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(y[0])));
        const float kz32 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(z[0])));
        const double kx = kx32, ky = ky32, kz = kz32;
        double v9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < PPL; i++) {
            const double dx = (double)x[i] - kx, dy = (double)y[i] - ky, dz = (double)z[i] - kz;
            v9[0] += dx;
            v9[1] += dy;
            v9[2] += dz;
            v9[3] = fma(dx, dx, v9[3]);
            v9[4] = fma(dx, dy, v9[4]);
            v9[5] = fma(dx, dz, v9[5]);
            v9[6] = fma(dy, dy, v9[6]);
            v9[7] = fma(dy, dz, v9[7]);
            v9[8] = fma(dz, dz, v9[8]);
        }
        // Everything that is computed once per set -- mean, covariance, eigen-solve, the nine scalars of the
        // head, the float32 projection coefficients and their error bound -- is done by one wave of the block
        // for all NW sets at once, a lane per set, instead of by every wave on 64 identical lanes (the duty
        // rotates over the waves with the block index).  The sums go to it through the reduction slab, the
        // results come back in a record per set.
        double *slab = red_all[wave];
        wave_sum9_slab(v9, slab);
        if (NORMALIZE) {
            // NORMALIZE is a template argument: as a run-time branch it makes the compiler keep all widened
            // coordinates of the first pass alive for this one (48 more registers, one wave less per SIMD)
            const double inv_cnt = RAGGED ? 1.0 / (double)(live ? n : 3) : inv_n;
            const double mx = kx + slab[RED9_TOT] * inv_cnt, my = ky + slab[RED9_TOT + 1] * inv_cnt, mz = kz + slab[RED9_TOT + 2] * inv_cnt;
            double r2 = 0.0;
#pragma unroll
            for (int i = 0; i < PPL; i++) {
                const double dx = (double)x[i] - mx, dy = (double)y[i] - my, dz = (double)z[i] - mz;
                r2 = fmax(r2, dx * dx + dy * dy + dz * dz);
            }
            const double rmax = sqrt(fx_wave_max_f64_dpp(r2));
            if (lane == 0)
                slab[RED9_TOT + 9] = rmax;
        }
        if (lane == 0) {
            slab[RED9_TOT + 10] = kx;
            slab[RED9_TOT + 11] = ky;
            slab[RED9_TOT + 12] = kz;
            if (RAGGED)
                slab[RED9_TOT + 13] = (double)(live ? n : 3);
        }
        __syncthreads();
        if (wave == (int)(blockIdx.x % NW)) {
            if (lane < NW) {
                const double *t = red_all[lane] + RED9_TOT;
                SetRec *rec = &rec_all[lane];
                const double cnt_inv = RAGGED ? 1.0 / t[13] : inv_n;
                const double ox = t[0] * cnt_inv, oy = t[1] * cnt_inv, oz = t[2] * cnt_inv;  // mean - pivot
                const double mx = t[10] + ox, my = t[11] + oy, mz = t[12] + oz;
                double S[6];
                S[0] = fma(-t[0], ox, t[3]);
                S[1] = fma(-t[0], oy, t[4]);
                S[2] = fma(-t[0], oz, t[5]);
                S[3] = fma(-t[1], oy, t[6]);
                S[4] = fma(-t[1], oz, t[7]);
                S[5] = fma(-t[2], oz, t[8]);
                // every centred point lies within sqrt(sum of squared distances) of the mean
                const float reach32 = sqrtf((float)(S[0] + S[3] + S[5])) * 1.000001f;
                const double rmax = NORMALIZE ? t[9] : 0.0;
                const bool scale = NORMALIZE && rmax > 0.0;
                double wcov = RAGGED ? 1.0 / (t[13] - 1.0) : inv_nm1;  // np.cov multiplies by 1/(n-1)
                if (scale)
                    wcov *= 1.0 / (rmax * rmax);
                double lam[3] = {0.0, 0.0, 0.0}, v1[3] = {0.0, 0.0, 0.0};
                // Matrices the direct solve does not take (eigenvalues too close, or one next to nothing) go
                // to pca_sets_redo_kernel with the set.
                bool solved = fx_eig3_direct(S[0] * wcov, S[1] * wcov, S[2] * wcov, S[3] * wcov, S[4] * wcov, S[5] * wcov, lam, v1);
                if (!solved) {   // (a few lanes of one wave in four: the sweeps cost little here)
                    fx_eig3<true>(S[0] * wcov, S[1] * wcov, S[2] * wcov, S[3] * wcov, S[4] * wcov, S[5] * wcov, lam, v1);
                    solved = true;
                }
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    rec->lam[q] = lam[q] < 0.0 ? 0.0 : lam[q];
                    rec->v1[q] = v1[q];
                }
                rec->mean[0] = mx, rec->mean[1] = my, rec->mean[2] = mz;
                rec->rmax = scale ? rmax : 0.0;
                // float32 value of a projection: v.x - v.mean in three FMAs.  Each of its five roundings is
                // relative to a partial sum of at most |v.mean| + |v| |x| <= w (|mean|_1 + |mean|_2 + reach):
                // eps32 bounds the total generously.  It settles every point that is not within eps32 of a
                // decision; those few are decided in float64.
                const float w32 = scale ? (float)(1.0 / rmax) : 1.0f;
                rec->vq[0] = (float)v1[0] * w32;
                rec->vq[1] = (float)v1[1] * w32;
                rec->vq[2] = (float)v1[2] * w32;
                rec->vq[3] = (float)(fma(v1[2], mz, fma(v1[1], my, v1[0] * mx))) * w32;
                const float m1 = fabsf((float)mx) + fabsf((float)my) + fabsf((float)mz);
                rec->eps32 = 1e-6f * w32 * (2.0f * m1 + reach32) + 1e-37f;
                rec->solved = solved ? 1 : 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // the nine scalars of the head, a lane per (set, scalar): one division for the whole block
            const int hs = lane >> 4, e = lane & 15;
            if (hs < NW && e < 9) {
                const double *lam = rec_all[hs].lam;
                const double trace = lam[0] + lam[1] + lam[2];
                const double safe = trace > 1e-12 ? trace : 1e-12;
                const double num = (e == 0 || e == 3 || e == 6) ? lam[0] : (e == 1 || e == 4 || e == 7) ? lam[1] : (e == 2 || e == 5) ? lam[2] : trace;
                const double den = e < 3 || e == 8 ? 1.0 : e < 6 ? safe : e == 6 ? lam[1] + 1e-12 : lam[2] + 1e-12;
                rec_all[hs].head[e] = (float)(num / den);
            }
        }
        __syncthreads();
        const SetRec *rec = &rec_all[wave];
        if (RAGGED && !live)
            return;   // after the block's last barrier
        if (!rec->solved) {   // wave-uniform; after the block's last barrier
            if (live && lane == 0)
                A.redo_list[atomicAdd(A.redo_count, 1)] = (int32_t)set;
            return;
        }
        const bool scale = NORMALIZE && rec->rmax > 0.0;

        // float64 value of a point (the formula of record)
        auto proj = [&](float px, float py, float pz) -> double {
            double dx = (double)px - rec->mean[0], dy = (double)py - rec->mean[1], dz = (double)pz - rec->mean[2];
            if (scale) {
                const double rmax = rec->rmax;
                dx /= rmax;
                dy /= rmax;
                dz /= rmax;
            }
            return fabs(fma(dz, rec->v1[2], fma(dy, rec->v1[1], dx * rec->v1[0])));
        };
        const float vx32 = rec->vq[0], vy32 = rec->vq[1], vz32 = rec->vq[2], c32 = rec->vq[3], eps32 = rec->eps32;
        float top32 = 0.0f;
#pragma unroll
        for (int i = 0; i < PPL; i++)
            top32 = fmaxf(top32, fabsf(fmaf(z[i], vz32, fmaf(y[i], vy32, fmaf(x[i], vx32, -c32)))));
        top32 = fx_wave_max_nonneg_f32_dpp(top32);
        // Points that float32 cannot settle are appended to a short list in LDS (a wave-uniform branch per point
        // index, taken for about one index in sixteen) and evaluated in float64 afterwards, a lane per listed
        // point: the float64 code exists once and runs without divergence.
        float *slow_list = reinterpret_cast<float *>(red_all[wave]);  // the reduction slab is free again
        constexpr int SLOW_CAP = (RED9_WORDS * 2) / 3 - FX_WAVE;  // points the slab holds, one wave's worth in reserve
        int n_slow = 0;
        auto append = [&](unsigned long long mask, bool mine, float px, float py, float pz) {
            const int pos = n_slow + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (mine) {
                slow_list[3 * pos] = px;
                slow_list[3 * pos + 1] = py;
                slow_list[3 * pos + 2] = pz;
            }
            n_slow += __builtin_popcountll(mask);
        };
        auto list_sync = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        list_sync();
        // One pass settles the bins.  The float32 maximum is within eps32 of the exact one, so bins taken with
        // the scale n_bins / top32 are those of the exact scale for every point whose q = |projection| * scale
        // stays `margin` away from an integer; the others -- the candidates for the exact maximum among them,
        // which all lie within the margin below n_bins -- go to the list.  The exact maximum then comes from the
        // listed points, and they are binned with it in float64.
        bool redo = !(top32 > 8.0f * eps32);   // projections at rounding level: everything in float64, below
        double top = 0.0;
        const float inv32 = (float)nb / top32;
        const float qx = vx32 * inv32, qy = vy32 * inv32, qz = vz32 * inv32, qc = c32 * inv32;
        auto q32 = [&](float px, float py, float pz) -> float { return fabsf(fmaf(pz, qz, fmaf(py, qy, fmaf(px, qx, -qc)))); };
        if (!redo) {
            const float margin = (3.0f * eps32 + 6e-7f * top32) * inv32 + 1e-5f * (float)nb;
#pragma unroll
            for (int i = 0; i < PPL; i++) {
                // every point is counted in the bin float32 gives it (q <= n_bins + margin: the row has a spare
                // entry); a listed point is taken out of that bin again when the list is evaluated
                const float q = q32(x[i], y[i], z[i]);
                const bool have = !RAGGED || lane + FX_WAVE * i < n;
                const bool slow = have && fabsf(q - rintf(q)) < margin;
                if (have)
                    atomicAdd(&my_hist[(int)q], 1u);
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(slow);
                if (mask && !redo) {
                    if (n_slow > SLOW_CAP)
                        redo = true;   // degenerate sets (most points at one |projection|): the list would overflow
                    else
                        append(mask, slow, x[i], y[i], z[i]);
                }
            }
        }
        // Sets the single pass cannot take (projections at rounding level, or so many points at one
        // |projection| that the list overflows) are handed to pca_sets_redo_kernel.
        if (redo) {
            if (live && lane == 0)
                A.redo_list[atomicAdd(A.redo_count, 1)] = (int32_t)set;
            return;   // after the block's last barrier
        }
        list_sync();
        // (a list of up to 64 points, the usual case, is one point per lane: its value is kept for the binning)
        const double mine = lane < n_slow ? proj(slow_list[3 * lane], slow_list[3 * lane + 1], slow_list[3 * lane + 2]) : 0.0;
        top = mine;
        for (int e = lane + FX_WAVE; e < n_slow; e += FX_WAVE)
            top = fmax(top, proj(slow_list[3 * e], slow_list[3 * e + 1], slow_list[3 * e + 2]));
        top = fx_wave_max_nonneg_f64_dpp(top);
        if (!(top > 1e-280) || !(top < 1e280)) {   // the shortcuts below assume a plain double
            if (live && lane == 0)
                A.redo_list[atomicAdd(A.redo_count, 1)] = (int32_t)set;
            return;
        }
        {
            // step = top / n_bins, correctly rounded, from the host's correctly rounded 1 / n_bins: two
            // residual corrections (the first makes the quotient faithful, the second then rounds it
            // correctly -- Markstein's theorem)
            const double dnb = (double)nb;
            double step = top * inv_nb;
            step = fma(fma(-dnb, step, top), inv_nb, step);
            step = fma(fma(-dnb, step, top), inv_nb, step);
            const double inv_step = dnb * __builtin_amdgcn_rcp(top);   // within 1e-7: fx_hist_bin_near corrects the bin
            for (int e = lane; e < n_slow; e += FX_WAVE) {
                const float px = slow_list[3 * e], py = slow_list[3 * e + 1], pz = slow_list[3 * e + 2];
                atomicSub(&my_hist[(int)q32(px, py, pz)], 1u);   // the same arithmetic as in the pass: the same bin
                atomicAdd(&my_hist[fx_hist_bin_near(e == lane ? mine : proj(px, py, pz), step, inv_step, nb)], 1u);
            }
        }
        list_sync();
        if (live && lane < nb) {
            uint32_t h = 0;
#pragma unroll
            for (int k = 0; k < HIST_COPIES; k++)
                h += hist[k * HIST_PITCH + lane];
            A.out[set * width + 9 + lane] = (float)h / (float)n;
        }
        if (live && lane < 9)
            A.out[set * width + lane] = rec->head[lane];
        if (live && A.eig && lane < 6)
            A.eig[set * 6 + lane] = rec->lam[lane];   // lam[3] and v1[3] lie one after the other
    }
}

// ------------------------------------------------ small sets: four per wave
// Sets of up to 256 points (2-D or 3-D, any alignment): a wave takes four sets, 16 lanes each, lane r of a
// set holding its points r, r + 16, ...  What is per set and serial -- mean, covariance, eigen-solve, head,
// float32 coefficients -- is computed by every lane of the set's row (the same instructions serve the four
// sets of the wave at once; the row sums come from an exchange butterfly, which leaves bit-identical totals in
// all 16 lanes, so the lanes of a row agree on everything that follows).  No block barrier: waves are
// independent.  Bins as in pca_sets_full_kernel: one pass with the float32 scale, the points float32 cannot
// settle go to a list of the wave (tagged with their row) and are evaluated in float64 afterwards, a lane per
// listed point; the exact maximum of a row is an LDS atomic maximum over its listed points.  Sets that pass
// cannot take (projections at rounding level, a list that overflows) go to the redo list.
constexpr int QUAD_SETS = 4, QUAD_LANES = FX_WAVE / QUAD_SETS, QUAD_PPL = 16;
constexpr int64_t QUAD_MAX_POINTS = QUAD_LANES * QUAD_PPL;   // with QUAD_PPL point slots per lane; the kernel also exists
                                                             // with 8 and 4 slots, for sets of up to 128 and 64 points
constexpr int QUAD_COPIES = 4, QUAD_PITCH = HIST_MAX + 1, QUAD_LIST_CAP = 96;
struct QuadRec {
    double mean[3], v1[3], rmax;   // rmax = 0: no scaling
    float q[4];                    // float32 bin coordinate: |q[0..2] . x - q[3]|
};

template <int CTRL>
__device__ __forceinline__ double quad_xchg_f64(double v)
{
    return fx_dpp_f64<CTRL, 0xf>(v);
}
// sum / maximum over the 16 lanes of a row; every lane gets the same bits (each step combines a lane's value
// with its partner's, and a + b == b + a)
__device__ __forceinline__ double quad_row_sum(double v)
{
    v += quad_xchg_f64<0xB1>(v);    // lane ^ 1
    v += quad_xchg_f64<0x4E>(v);    // lane ^ 2
    v += quad_xchg_f64<0x141>(v);   // 7 - lane within half a row
    v += quad_xchg_f64<0x140>(v);   // 15 - lane
    return v;
}
__device__ __forceinline__ double quad_row_max(double v)
{
    v = fmax(v, quad_xchg_f64<0xB1>(v));
    v = fmax(v, quad_xchg_f64<0x4E>(v));
    v = fmax(v, quad_xchg_f64<0x141>(v));
    v = fmax(v, quad_xchg_f64<0x140>(v));
    return v;
}
__device__ __forceinline__ uint32_t quad_row_max_u32(uint32_t v)
{
    const auto step = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false));
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false));
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false));
    v = step(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false));
    return v;
}

// The LDS of a wave of the quad kernels.
struct QuadLds {
    uint32_t hist[QUAD_SETS * QUAD_COPIES * QUAD_PITCH];
    float4 list[QUAD_LIST_CAP + FX_WAVE];
    QuadRec rec[QUAD_SETS];
    unsigned long long top[QUAD_SETS];
};

// One wave's work: the four sets from entry `first` of the list A.set_ids (n_sets entries; NULL: the sets
// themselves), with PPL point slots per lane.
template <int PPL>
__device__ __forceinline__ void pca_sets_quad_body(const PcaArgs &A, QuadLds &W, const int64_t n_sets, const int64_t first, double inv_nb)
{
    const int lane = fx_lane();
    const int row = lane / QUAD_LANES, r = lane % QUAD_LANES;
    const int64_t slot = first + row;
    const bool in_range = slot < n_sets;
    const SetView sv = resolve_set(A, in_range ? slot : first);
    const int64_t set = sv.set;
    const int nb = A.n_bins, width = 9 + nb, dims = A.dims;
    const bool small = sv.n <= QUAD_LANES * PPL;
    const bool live = in_range && sv.n >= 3 && small;
    const int n = live ? (int)sv.n : 0;
    if (in_range && sv.n < 3)
        write_nan_row(A, set, r, QUAD_LANES);
    auto to_redo = [&](bool which) {   // lane 0 of every row for which `which` holds hands its set on
        if (which && r == 0)
            A.redo_list[atomicAdd(A.redo_count, 1)] = (int32_t)set;
    };
    to_redo(in_range && !small);   // (the dispatcher sends no such set here)

    uint32_t *hist = W.hist;
    for (int k = lane; k < QUAD_SETS * QUAD_COPIES * QUAD_PITCH; k += FX_WAVE)
        hist[k] = 0;
    uint32_t *my_hist = hist + (row * QUAD_COPIES + (r % QUAD_COPIES)) * QUAD_PITCH;

    // Lane r holds the points r, r + 16, ...; the slots past the end of the set hold the set's first point, which
    // is the pivot of the sums: such a slot adds exact zeros to them, and a copy of a point changes no maximum.
    // (No predicates until the histogram.)
    float x[PPL], y[PPL], z[PPL];
    float kx32 = 0.0f, ky32 = 0.0f, kz32 = 0.0f;
#pragma unroll
    for (int k = 0; k < PPL; k++)
        x[k] = y[k] = z[k] = 0.0f;
    if (live) {
        struct P3 {
            float a, b, c;
        };
        struct P2 {
            float a, b;
        };
        if (dims == 3) {
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                const int p = r + QUAD_LANES * k;
                const P3 v = *reinterpret_cast<const P3 *>(sv.base + 3 * (p < n ? p : 0));
                x[k] = v.a, y[k] = v.b, z[k] = v.c;
            }
            const P3 v = *reinterpret_cast<const P3 *>(sv.base);
            kx32 = v.a, ky32 = v.b, kz32 = v.c;
        } else {   // 2-D points are (x, y, 0)
#pragma unroll
            for (int k = 0; k < PPL; k++) {
                const int p = r + QUAD_LANES * k;
                const P2 v = *reinterpret_cast<const P2 *>(sv.base + 2 * (p < n ? p : 0));
                x[k] = v.a, y[k] = v.b;
            }
            const P2 v = *reinterpret_cast<const P2 *>(sv.base);
            kx32 = v.a, ky32 = v.b;
        }
    }
    const double kx = kx32, ky = ky32, kz = kz32;
    double v9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const double dx = (double)x[k] - kx, dy = (double)y[k] - ky, dz = (double)z[k] - kz;
        v9[0] += dx;
        v9[1] += dy;
        v9[2] += dz;
        v9[3] = fma(dx, dx, v9[3]);
        v9[4] = fma(dx, dy, v9[4]);
        v9[5] = fma(dx, dz, v9[5]);
        v9[6] = fma(dy, dy, v9[6]);
        v9[7] = fma(dy, dz, v9[7]);
        v9[8] = fma(dz, dz, v9[8]);
    }
#pragma unroll
    for (int k = 0; k < 9; k++)
        v9[k] = quad_row_sum(v9[k]);
    const double dn = live ? (double)n : 3.0;
    const double inv_n = 1.0 / dn;
    const double ox = v9[0] * inv_n, oy = v9[1] * inv_n, oz = v9[2] * inv_n;   // mean - pivot
    const double mx = kx + ox, my = ky + oy, mz = kz + oz;
    double S[6];
    S[0] = fma(-v9[0], ox, v9[3]);
    S[1] = fma(-v9[0], oy, v9[4]);
    S[2] = fma(-v9[0], oz, v9[5]);
    S[3] = fma(-v9[1], oy, v9[6]);
    S[4] = fma(-v9[1], oz, v9[7]);
    S[5] = fma(-v9[2], oz, v9[8]);
    // every centred point lies within sqrt(sum of squared distances) of the mean
    const float reach32 = sqrtf((float)(S[0] + S[3] + S[5])) * 1.000001f;
    double rmax = 0.0;
    if (A.normalize) {
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < PPL; k++) {
            const double dx = (double)x[k] - mx, dy = (double)y[k] - my, dz = (double)z[k] - mz;
            r2 = fmax(r2, dx * dx + dy * dy + dz * dz);
        }
        rmax = sqrt(quad_row_max(r2));
    }
    const bool scale = A.normalize && rmax > 0.0;
    double wcov = 1.0 / (dn - 1.0);   // np.cov multiplies by 1/(n-1)
    if (scale)
        wcov *= 1.0 / (rmax * rmax);
    double lam[3] = {0.0, 0.0, 0.0}, v1[3] = {1.0, 0.0, 0.0};
    // the direct solve where it applies (well separated, clearly positive eigenvalues), the sweeps otherwise
    if (!fx_eig3_direct(S[0] * wcov, S[1] * wcov, S[2] * wcov, S[3] * wcov, S[4] * wcov, S[5] * wcov, lam, v1))
        fx_eig3<true>(S[0] * wcov, S[1] * wcov, S[2] * wcov, S[3] * wcov, S[4] * wcov, S[5] * wcov, lam, v1);
#pragma unroll
    for (int q = 0; q < 3; q++)
        lam[q] = lam[q] < 0.0 ? 0.0 : lam[q];
    {
        // the nine scalars of the head, lane r of the row computing entry r: one division
        const double trace = lam[0] + lam[1] + lam[2];
        const double safe = trace > 1e-12 ? trace : 1e-12;
        const int e = r < 9 ? r : 0;
        const double num = (e == 0 || e == 3 || e == 6) ? lam[0] : (e == 1 || e == 4 || e == 7) ? lam[1] : (e == 2 || e == 5) ? lam[2] : trace;
        const double den = e < 3 || e == 8 ? 1.0 : e < 6 ? safe : e == 6 ? lam[1] + 1e-12 : lam[2] + 1e-12;
        if (live && r < 9)
            A.out[set * width + r] = (float)(num / den);
        if (live && A.eig && r < 6) {
            double ev = lam[0];
            ev = r == 1 ? lam[1] : ev;
            ev = r == 2 ? lam[2] : ev;
            ev = r == 3 ? v1[0] : ev;
            ev = r == 4 ? v1[1] : ev;
            ev = r == 5 ? v1[2] : ev;
            A.eig[set * 6 + r] = ev;
        }
    }
    // float32 value of a projection: v.x - v.mean in three FMAs; eps32 bounds its error (see pca_sets_full_kernel)
    const float w32 = scale ? (float)(1.0 / rmax) : 1.0f;
    const float vx32 = (float)v1[0] * w32, vy32 = (float)v1[1] * w32, vz32 = (float)v1[2] * w32;
    const float c32 = (float)(fma(v1[2], mz, fma(v1[1], my, v1[0] * mx))) * w32;
    const float m1 = fabsf((float)mx) + fabsf((float)my) + fabsf((float)mz);
    const float eps32 = 1e-6f * w32 * (2.0f * m1 + reach32) + 1e-37f;
    float top32 = 0.0f;
#pragma unroll
    for (int k = 0; k < PPL; k++)
        top32 = fmaxf(top32, fabsf(fmaf(z[k], vz32, fmaf(y[k], vy32, fmaf(x[k], vx32, -c32)))));
    top32 = __int_as_float((int)quad_row_max_u32((uint32_t)__float_as_int(top32)));
    // rows whose projections are at rounding level are redone in float64 throughout
    bool act = live && top32 > 8.0f * eps32;
    to_redo(live && !act);
    const float inv32 = act ? (float)nb / top32 : 0.0f;
    const float qx = vx32 * inv32, qy = vy32 * inv32, qz = vz32 * inv32, qc = c32 * inv32;
    const float margin = (3.0f * eps32 + 6e-7f * top32) * inv32 + 1e-5f * (float)nb;
    if (r == 0) {
        QuadRec *rec = &W.rec[row];
        rec->mean[0] = mx, rec->mean[1] = my, rec->mean[2] = mz;
        rec->v1[0] = v1[0], rec->v1[1] = v1[1], rec->v1[2] = v1[2];
        rec->rmax = scale ? rmax : 0.0;
        rec->q[0] = qx, rec->q[1] = qy, rec->q[2] = qz, rec->q[3] = qc;
        W.top[row] = 0ull;
    }
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    wave_sync();
    // one pass settles the bins of every point float32 can settle; the others are listed with their row
    float4 *list = W.list;
    int n_slow = 0;
    bool overflow = false;
    const int n_act = act ? n : 0;
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const bool hv = r + QUAD_LANES * k < n_act;
        const float q = fabsf(fmaf(z[k], qz, fmaf(y[k], qy, fmaf(x[k], qx, -qc))));
        const bool slow = hv && fabsf(q - rintf(q)) < margin;
        if (hv)
            atomicAdd(&my_hist[(int)q], 1u);   // q <= n_bins + margin: the row has a spare entry
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(slow);
        if (mask && !overflow) {
            if (n_slow > QUAD_LIST_CAP) {
                overflow = true;
            } else {
                const int pos = n_slow + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                if (slow)
                    list[pos] = make_float4(x[k], y[k], z[k], __int_as_float(row));
                n_slow += __builtin_popcountll(mask);
            }
        }
    }
    if (overflow) {   // wave-uniform: degenerate sets (most points at one |projection|)
        to_redo(act);
        return;
    }
    wave_sync();
    // float64 value of a listed point (the formula of record), with the record of the point's row
    auto proj = [&](const QuadRec *rec, float px, float py, float pz) -> double {
        double dx = (double)px - rec->mean[0], dy = (double)py - rec->mean[1], dz = (double)pz - rec->mean[2];
        if (rec->rmax > 0.0) {
            const double rm = rec->rmax;
            dx /= rm;
            dy /= rm;
            dz /= rm;
        }
        return fabs(fma(dz, rec->v1[2], fma(dy, rec->v1[1], dx * rec->v1[0])));
    };
    for (int e = lane; e < n_slow; e += FX_WAVE) {
        const float4 pt = list[e];
        const int er = __float_as_int(pt.w);
        const double pv = proj(&W.rec[er], pt.x, pt.y, pt.z);
        atomicMax(&W.top[er], (unsigned long long)__double_as_longlong(pv));   // pv >= 0: the bits order like the values
    }
    wave_sync();
    {
        const double mytop = __longlong_as_double((long long)W.top[row]);
        const bool sane = mytop > 1e-280 && mytop < 1e280;   // the shortcuts below assume a plain double
        to_redo(act && !sane);
        act = act && sane;
    }
    for (int e = lane; e < n_slow; e += FX_WAVE) {
        const float4 pt = list[e];
        const int er = __float_as_int(pt.w);
        const QuadRec *rec = &W.rec[er];
        const double top = __longlong_as_double((long long)W.top[er]);
        if (!(top > 1e-280 && top < 1e280))
            continue;
        // step = top / n_bins, correctly rounded, from the host's correctly rounded 1 / n_bins (two residual
        // corrections: the first makes the quotient faithful, the second then rounds it correctly)
        const double dnb = (double)nb;
        double step = top * inv_nb;
        step = fma(fma(-dnb, step, top), inv_nb, step);
        step = fma(fma(-dnb, step, top), inv_nb, step);
        const double inv_step = dnb * __builtin_amdgcn_rcp(top);
        uint32_t *h = hist + (er * QUAD_COPIES + (lane % QUAD_COPIES)) * QUAD_PITCH;
        const float q = fabsf(fmaf(pt.z, rec->q[2], fmaf(pt.y, rec->q[1], fmaf(pt.x, rec->q[0], -rec->q[3]))));
        atomicSub(&h[(int)q], 1u);   // the same arithmetic as in the pass: the same bin
        atomicAdd(&h[fx_hist_bin_near(proj(rec, pt.x, pt.y, pt.z), step, inv_step, nb)], 1u);
    }
    wave_sync();
    if (act) {
        for (int i = r; i < nb; i += QUAD_LANES) {
            uint32_t h = 0;
#pragma unroll
            for (int c = 0; c < QUAD_COPIES; c++)
                h += hist[(row * QUAD_COPIES + c) * QUAD_PITCH + i];
            A.out[set * width + 9 + i] = (float)h / (float)n;
        }
    }
}


// A wave per four sets of one list.
template <int NW, int PPL>
__global__ __launch_bounds__(FX_WAVE *NW, PPL <= 4 ? 5 : PPL <= 8 ? 4 : 3) void pca_sets_quad_kernel(PcaArgs A, double inv_nb)
{
    __shared__ QuadLds lds[NW];
    const int64_t n_sets = A.n_sets_dev ? (int64_t)*A.n_sets_dev : A.n_sets;
    const int64_t first = ((int64_t)blockIdx.x * NW + fx_wave()) * QUAD_SETS;
    if (first >= n_sets)
        return;   // the whole wave; there is no block barrier
    pca_sets_quad_body<PPL>(A, lds[fx_wave()], n_sets, first, inv_nb);
}

// The three lists of small sets that the classification of a ragged batch builds (up to 256, 128 and 64 points,
// lengths in counts[0..2]) in one launch: the waves of the grid are dealt to the lists in that order, four
// entries each, so the launch has no more waves than one over a single list of all small sets would have (plus
// one per list for its last, partly filled wave), and a set of 40 points costs a quarter of the point slots.
struct QuadLists {
    const int32_t *ids[3];
    const int32_t *counts;
};
template <int NW>
__global__ __launch_bounds__(FX_WAVE *NW, 3) void pca_sets_quad_mixed_kernel(PcaArgs A, QuadLists Q, double inv_nb)
{
    __shared__ QuadLds lds[NW];
    const int wave = __builtin_amdgcn_readfirstlane(fx_wave());
    const int64_t w = (int64_t)blockIdx.x * NW + wave;
    const int c0 = Q.counts[0], c1 = Q.counts[1], c2 = Q.counts[2];
    const int64_t w0 = (c0 + QUAD_SETS - 1) / QUAD_SETS, w1 = (c1 + QUAD_SETS - 1) / QUAD_SETS, w2 = (c2 + QUAD_SETS - 1) / QUAD_SETS;
    if (w < w0) {
        A.set_ids = Q.ids[0];
        pca_sets_quad_body<16>(A, lds[wave], c0, w * QUAD_SETS, inv_nb);
    } else if (w < w0 + w1) {
        A.set_ids = Q.ids[1];
        pca_sets_quad_body<8>(A, lds[wave], c1, (w - w0) * QUAD_SETS, inv_nb);
    } else if (w < w0 + w1 + w2) {
        A.set_ids = Q.ids[2];
        pca_sets_quad_body<4>(A, lds[wave], c2, (w - w0 - w1) * QUAD_SETS, inv_nb);
    }
}

// ---------------------------------------------------------------- large sets
// ---------------------------------------------------------------------------
// Large clouds (more points than a wave keeps in registers; with float64 input: every set).
//
// A cloud is cut into chunks of CLOUD_CHUNK points, a block per chunk and pass, so a 4M-point cloud spreads over
// the whole chip and a batch of unequal clouds balances at chunk granularity.  One launch per step:
//   sum      sums of d = p - pivot and of d d^T over the chunk (pivot = the cloud's first point, as in the set
//            kernels: the mean need not be known before the second moments are summed), and the chunk's bounding box
//   state    once per cloud: chunk records combined in a fixed order -> mean, covariance -> eigen-solve -> the
//            nine scalars and the first principal axis; from the box, a bound U on |projection|
//   proj     |projection on the axis| of every point: its maximum over the cloud, and for every point a 16-bit
//            code floor(65536 |projection| / U), written to scratch
//   hist     bins between 0 and that maximum from the codes: a code pins its point's projection down to U / 65536,
//            which decides the bin for all but the few points next to a bin edge -- those are read and projected
//            again
//   finish   the row's histogram part from the counts
// (with `normalize` a pass for the largest distance from the mean comes between sum and the eigen-solve).
// A point is read twice, 12 bytes each time, and its code written and read once, 2 bytes: the third pass over the
// points that the bins would otherwise need -- their range is only known when every projection has been seen -- is
// a pass over a sixth of the bytes.  The passes walk the chunk list in alternating directions: what a pass read
// last, a good part of the 256 MiB Infinity Cache, is what the next one reads first.
// ---------------------------------------------------------------------------
#ifndef FX_CLOUD_CHUNK
#define FX_CLOUD_CHUNK 8192
#endif
#ifndef FX_CLOUD_THREADS
#define FX_CLOUD_THREADS 256
#endif
#ifndef FX_CLOUD_BATCH
#define FX_CLOUD_BATCH 4
#endif
constexpr int CLOUD_CHUNK = FX_CLOUD_CHUNK;
constexpr int CLOUD_THREADS = FX_CLOUD_THREADS;
constexpr int CLOUD_WAVES = CLOUD_THREADS / FX_WAVE;
constexpr int CLOUD_BATCH = FX_CLOUD_BATCH;   // groups of four points a thread has in flight
// per-chunk record, doubles: 0-2 sums of d, 3-8 sums of d d^T, 9-11 the smallest and 12-14 the largest coordinates
constexpr int PART = 16;
// per-cloud record, doubles: 0-2 mean, 3 rmax (0 = no scaling), 4-6 first principal axis, 7 the bound U on
// |projection| (after scaling), 8-13 second moments about the mean, 14-19 bounding box (smallest, largest)
constexpr int STATE = 20;
constexpr int CLOUD_HIST_COPIES = 4, CLOUD_HIST_PITCH = HIST_MAX + 1;   // LDS histograms per wave (lane mod 4), bank-skewed
constexpr double CLOUD_CODES = 65536.0;
static_assert(CLOUD_CHUNK % (8 * CLOUD_THREADS) == 0, "a chunk's codes are moved 16 bytes per thread and round");

struct CloudCtl {
    u64 top_bits;   // largest |projection| (the bits of that double >= 0, which order as it does)
    u64 r2_bits;    // normalize: largest squared distance from the mean
    u64 hist[HIST_MAX];
};

// What a pass needs to know about a chunk, in one 32-byte read.
struct __align__(16) CloudChunk {
    int64_t base;      // the cloud's first point, counted in points from the start of the batch
    int64_t i0;        // the chunk's first point within the cloud
    int32_t count;     // points in the chunk
    int32_t cloud;     // position in cloud_ids of the chunk's cloud
    int32_t pad[2];
};

struct CloudWork {
    const int32_t *cloud_ids;    // [n_clouds] set index of every cloud
    const int32_t *cloud_first;  // [n_clouds + 1] first chunk of every cloud
    const CloudChunk *chunks;    // [n_chunks]
    double *part;                // [n_chunks][PART]
    CloudCtl *ctl;               // [n_clouds], zero at the start of a call
    double *state;               // [n_clouds][STATE]
    uint16_t *codes;             // [n_chunks][CLOUD_CHUNK]
    const int32_t *n_chunks_dev; // list lengths (device memory)
    const int32_t *n_clouds_dev;
};

template <typename T>
struct ChunkView {
    const T *g;       // first point of the cloud
    int64_t n;        // points in the cloud
    int64_t i0, i1;   // this chunk's point range
    int64_t set;
    int cloud, first, nch;
};

template <typename T>
__device__ __forceinline__ const T *cloud_points(const PcaArgs &A)
{
    return sizeof(T) == 8 ? reinterpret_cast<const T *>(A.pts64) : reinterpret_cast<const T *>(A.pts);
}

template <typename T>
__device__ __forceinline__ ChunkView<T> cloud_view(const PcaArgs &A, const CloudWork &Wk, int cloud)
{
    ChunkView<T> v;
    v.cloud = cloud;
    v.set = Wk.cloud_ids[cloud];
    v.first = Wk.cloud_first[cloud];
    v.nch = Wk.cloud_first[cloud + 1] - v.first;
    if (A.offsets) {
        const int64_t a = A.offsets[v.set];
        v.n = A.offsets[v.set + 1] - a;
        v.g = cloud_points<T>(A) + a * A.dims;
    } else {
        v.n = A.N;
        v.g = cloud_points<T>(A) + v.set * A.N * A.dims;
    }
    v.i0 = 0;
    v.i1 = v.n;
    return v;
}

// (what the passes over the chunks use: g, i0, i1, cloud)
template <typename T>
__device__ __forceinline__ ChunkView<T> chunk_view(const PcaArgs &A, const CloudWork &Wk, int c)
{
    const CloudChunk d = Wk.chunks[c];
    ChunkView<T> v;
    v.g = cloud_points<T>(A) + d.base * A.dims;
    v.n = 0;
    v.i0 = d.i0;
    v.i1 = d.i0 + d.count;
    v.set = 0;
    v.cloud = d.cloud;
    v.first = 0;
    v.nch = 0;
    return v;
}

struct CloudPoint {
    double x, y, z;
};
template <typename T>
__device__ __forceinline__ CloudPoint cloud_point(const T *g, int64_t i, int dims)
{
    CloudPoint p;
    p.x = (double)g[i * dims];
    p.y = (double)g[i * dims + 1];
    p.z = dims == 3 ? (double)g[i * dims + 2] : 0.0;
    return p;
}

// Calls f(k, x, y, z) for every point of the chunk, k = its index within the chunk.  float32 3-D points, the
// usual case: where the chunk's floats are 16-byte aligned a thread takes four consecutive points with three
// 16-byte loads, several such groups at a time; the up to three points before the first aligned one and
// after the last whole group, and every other layout, are read one by one.
template <typename T, typename F>
__device__ __forceinline__ void cloud_for_points(const PcaArgs &A, const ChunkView<T> &v, F f)
{
    const int dims = A.dims;
    const int t = threadIdx.x;
    if (sizeof(T) == 4 && dims == 3) {
        const float *g = reinterpret_cast<const float *>(v.g);
        int64_t ib = v.i0;
        for (int k = 0; k < 4 && ib < v.i1 && ((uintptr_t)(g + ib * 3) & 15u); k++)
            ib++;
        const int ng = (int)((v.i1 - ib) / 4);
        const int64_t ie = ib + 4 * (int64_t)ng;
        for (int64_t i = v.i0 + t; i < ib; i += CLOUD_THREADS) {
            const CloudPoint p = cloud_point(g, i, 3);
            f((int)(i - v.i0), p.x, p.y, p.z);
        }
        const float4 *g4 = reinterpret_cast<const float4 *>(g + ib * 3);
        // CLOUD_BATCH groups at a time: all their loads are out before the arithmetic of the first starts (a
        // round trip to memory costs far more than a group's arithmetic)
#pragma unroll 1
        for (int g0 = t; g0 < ng; g0 += CLOUD_BATCH * CLOUD_THREADS) {
            float4 a[CLOUD_BATCH], b[CLOUD_BATCH], c[CLOUD_BATCH];
#pragma unroll
            for (int j = 0; j < CLOUD_BATCH; j++) {
                const int gi = g0 + j * CLOUD_THREADS < ng ? g0 + j * CLOUD_THREADS : g0;   // (beyond the end: group g0 again, unused)
                a[j] = g4[3 * gi];
                b[j] = g4[3 * gi + 1];
                c[j] = g4[3 * gi + 2];
            }
#pragma unroll
            for (int j = 0; j < CLOUD_BATCH; j++) {
                const int gi = g0 + j * CLOUD_THREADS;
                if (gi < ng) {
                    const int k = (int)(ib - v.i0) + 4 * gi;
                    f(k, (double)a[j].x, (double)a[j].y, (double)a[j].z);
                    f(k + 1, (double)a[j].w, (double)b[j].x, (double)b[j].y);
                    f(k + 2, (double)b[j].z, (double)b[j].w, (double)c[j].x);
                    f(k + 3, (double)c[j].y, (double)c[j].z, (double)c[j].w);
                }
            }
        }
        for (int64_t i = ie + t; i < v.i1; i += CLOUD_THREADS) {
            const CloudPoint p = cloud_point(g, i, 3);
            f((int)(i - v.i0), p.x, p.y, p.z);
        }
    } else {
        for (int64_t i = v.i0 + t; i < v.i1; i += CLOUD_THREADS) {
            const CloudPoint p = cloud_point(v.g, i, dims);
            f((int)(i - v.i0), p.x, p.y, p.z);
        }
    }
}

// Block-wide sum / maximum in a fixed order: waves by butterfly, then the waves' results one after the other.
template <bool MAX>
__device__ __forceinline__ double cloud_block_reduce(double v, double *red)
{
    v = MAX ? fx_wave_max_f64_dpp(v) : fx_wave_sum_f64_dpp(v);
    if (fx_lane() == 0)
        red[fx_wave()] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int w = 1; w < CLOUD_WAVES; w++)
        s = MAX ? fmax(s, red[w]) : s + red[w];
    __syncthreads();
    return s;
}

// ---- step bodies (every thread of a block runs them)

// sum: the chunk's record
template <typename T>
__device__ __forceinline__ void cloud_sum_step(const PcaArgs &A, const CloudWork &Wk, const ChunkView<T> &v, int chunk,
                                               double *red)
{
    const CloudPoint k = cloud_point(v.g, 0, A.dims);
    double a9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double lo[3] = {k.x, k.y, k.z}, hi[3] = {k.x, k.y, k.z};   // (the pivot lies in the cloud's box anyway)
    cloud_for_points(A, v, [&](int, double px, double py, double pz) {
        const double dx = px - k.x, dy = py - k.y, dz = pz - k.z;
        a9[0] += dx;
        a9[1] += dy;
        a9[2] += dz;
        a9[3] = fma(dx, dx, a9[3]);
        a9[4] = fma(dx, dy, a9[4]);
        a9[5] = fma(dx, dz, a9[5]);
        a9[6] = fma(dy, dy, a9[6]);
        a9[7] = fma(dy, dz, a9[7]);
        a9[8] = fma(dz, dz, a9[8]);
        lo[0] = fmin(lo[0], px), hi[0] = fmax(hi[0], px);
        lo[1] = fmin(lo[1], py), hi[1] = fmax(hi[1], py);
        lo[2] = fmin(lo[2], pz), hi[2] = fmax(hi[2], pz);
    });
    // fifteen block results behind one pair of barriers: red[q][wave]
#pragma unroll
    for (int q = 0; q < 15; q++) {
        const double w = q < 9 ? fx_wave_sum_f64_dpp(a9[q]) : q < 12 ? -fx_wave_max_f64_dpp(-lo[q - 9]) : fx_wave_max_f64_dpp(hi[q - 12]);
        if (fx_lane() == 0)
            red[q * CLOUD_WAVES + fx_wave()] = w;
    }
    __syncthreads();
    if (threadIdx.x < 15) {
        const int q = threadIdx.x;
        double s = red[q * CLOUD_WAVES];
#pragma unroll
        for (int w = 1; w < CLOUD_WAVES; w++) {
            const double r = red[q * CLOUD_WAVES + w];
            s = q < 9 ? s + r : q < 12 ? fmin(s, r) : fmax(s, r);
        }
        Wk.part[(int64_t)chunk * PART + q] = s;
    }
    __syncthreads();
}

// state: the chunk records of a cloud combined (record i goes to slot i mod 256, a slot adds its records in
// order, the slots are added as cloud_block_reduce does), mean and second moments about it; then, unless the
// scale is still to come (normalize, first call), the eigen-solve, the head of the row and the bound on the
// projections.
// (Out of line: one lane per cloud runs it, and inlined its registers would be the streaming loops' registers.)
__device__ __attribute__((noinline)) void cloud_state_solve(float *out_row, double *eig_row, double *st, int64_t n, double rmax)
{
    double S[6];
    const bool scale = rmax > 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const double m = st[8 + k];
        S[k] = scale ? m / (rmax * rmax) : m;   // moments of the points divided by rmax
    }
    double head[9], lam[3], v1[3];
    cov_to_head(S, n, head, lam, v1);
    // |sum_i (p_i - m_i) a_i| <= sum_i max(|lo_i - m_i|, |hi_i - m_i|) |a_i|, a little more for the rounding of
    // either side
    double bound = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++)
        bound += fmax(fabs(st[14 + k] - st[k]), fabs(st[17 + k] - st[k])) * fabs(v1[k]);
    if (scale)
        bound /= rmax;
    bound = bound * (1.0 + 1e-9) + 1e-300;
    st[3] = scale ? rmax : 0.0;
    st[4] = v1[0];
    st[5] = v1[1];
    st[6] = v1[2];
    st[7] = bound;
    for (int k = 0; k < 9; k++)
        out_row[k] = (float)head[k];
    if (eig_row) {
        for (int k = 0; k < 3; k++) {
            eig_row[k] = lam[k];
            eig_row[3 + k] = v1[k];
        }
    }
}

template <typename T>
__device__ __forceinline__ void cloud_state_step(const PcaArgs &A, const CloudWork &Wk, const ChunkView<T> &v, bool scaled,
                                                 double *red)
{
    double *st = Wk.state + (int64_t)v.cloud * STATE;
    if (!scaled) {
        double t9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        const double inf = __longlong_as_double(0x7ff0000000000000LL);
        double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
        if (threadIdx.x < 256)
            for (int i = threadIdx.x; i < v.nch; i += 256) {
                const double *rec = Wk.part + (int64_t)(v.first + i) * PART;
                double r[15];
#pragma unroll
                for (int q = 0; q < 15; q++)   // (the loads of a record are out together)
                    r[q] = rec[q];
#pragma unroll
                for (int q = 0; q < 9; q++)
                    t9[q] += r[q];
#pragma unroll
                for (int q = 0; q < 3; q++)
                    lo[q] = fmin(lo[q], r[9 + q]), hi[q] = fmax(hi[q], r[12 + q]);
            }
#pragma unroll
        for (int q = 0; q < 15; q++) {
            const double w = q < 9 ? fx_wave_sum_f64_dpp(t9[q]) : q < 12 ? -fx_wave_max_f64_dpp(-lo[q - 9]) : fx_wave_max_f64_dpp(hi[q - 12]);
            if (fx_lane() == 0)
                red[q * CLOUD_WAVES + fx_wave()] = w;
        }
        __syncthreads();
        double box[6];
#pragma unroll
        for (int q = 0; q < 15; q++) {
            double s = red[q * CLOUD_WAVES];
#pragma unroll
            for (int w = 1; w < CLOUD_WAVES; w++) {
                const double r = red[q * CLOUD_WAVES + w];
                s = q < 9 ? s + r : q < 12 ? fmin(s, r) : fmax(s, r);
            }
            if (q < 9)
                t9[q] = s;
            else
                box[q - 9] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const CloudPoint k = cloud_point(v.g, 0, A.dims);
            const double dn = (double)v.n;
            const double ox = t9[0] / dn, oy = t9[1] / dn, oz = t9[2] / dn;   // mean - pivot
            st[0] = k.x + ox;
            st[1] = k.y + oy;
            st[2] = k.z + oz;
            st[3] = 0.0;
            st[8] = fma(-t9[0], ox, t9[3]);
            st[9] = fma(-t9[0], oy, t9[4]);
            st[10] = fma(-t9[0], oz, t9[5]);
            st[11] = fma(-t9[1], oy, t9[6]);
            st[12] = fma(-t9[1], oz, t9[7]);
            st[13] = fma(-t9[2], oz, t9[8]);
#pragma unroll
            for (int q = 0; q < 6; q++)
                st[14 + q] = box[q];
        }
        if (A.normalize)
            return;
    }
    if (threadIdx.x == 0) {
        double rmax = 0.0;
        if (scaled)
            rmax = sqrt(__longlong_as_double((long long)Wk.ctl[v.cloud].r2_bits));
        cloud_state_solve(A.out + v.set * (9 + A.n_bins), A.eig ? A.eig + v.set * 6 : (double *)nullptr, st, v.n, rmax);
    }
}

__device__ __forceinline__ double cloud_projection(double px, double py, double pz, const double *st)
{
    double dx = px - st[0], dy = py - st[1], dz = pz - st[2];
    if (st[3] > 0.0) {
        dx /= st[3];
        dy /= st[3];
        dz /= st[3];
    }
    return fabs(fma(dz, st[6], fma(dy, st[5], dx * st[4])));
}

// normalize: the largest squared distance from the mean over the chunk, into the cloud's record
template <typename T>
__device__ __forceinline__ void cloud_rmax_step(const PcaArgs &A, const CloudWork &Wk, const ChunkView<T> &v, double *red)
{
    const double *st = Wk.state + (int64_t)v.cloud * STATE;
    const double mx = st[0], my = st[1], mz = st[2];
    double r2 = 0.0;
    cloud_for_points(A, v, [&](int, double px, double py, double pz) {
        const double dx = px - mx, dy = py - my, dz = pz - mz;
        r2 = fmax(r2, dx * dx + dy * dy + dz * dz);
    });
    r2 = cloud_block_reduce<true>(r2, red);
    if (threadIdx.x == 0)
        atomicMax(&Wk.ctl[v.cloud].r2_bits, (u64)__double_as_longlong(r2));
}

// proj: |projection| of the chunk's points -- the chunk's maximum into the cloud's record, the points' codes
// (staged in LDS, so that they leave in 16-byte stores) into the chunk's stretch of the code array
template <typename T>
__device__ __forceinline__ void cloud_proj_step(const PcaArgs &A, const CloudWork &Wk, const ChunkView<T> &v, int chunk,
                                                const double *st, uint16_t *stage, double *red)
{
    const double to_code = CLOUD_CODES / st[7];
    double top = 0.0;
    cloud_for_points(A, v, [&](int k, double px, double py, double pz) {
        const double pr = cloud_projection(px, py, pz, st);
        const int c = (int)(pr * to_code);   // (pr <= the bound: at most 65536, up to rounding)
        stage[k] = (uint16_t)(c > 65535 ? 65535 : c);
        top = fmax(top, pr);
    });
    top = cloud_block_reduce<true>(top, red);   // (its barriers also order the staging before the copy)
    if (threadIdx.x == 0)
        atomicMax(&Wk.ctl[v.cloud].top_bits, (u64)__double_as_longlong(top));
    const uint4 *from = reinterpret_cast<const uint4 *>(stage);
    uint4 *to = reinterpret_cast<uint4 *>(Wk.codes + (int64_t)chunk * CLOUD_CHUNK);
    const int words = ((int)(v.i1 - v.i0) + 7) / 8;
    for (int k = threadIdx.x; k < words; k += CLOUD_THREADS)
        to[k] = from[k];
    __syncthreads();
}

// hist: bins of the chunk's projections against the cloud's maximum, from the codes.  Code c says that the
// projection lies in [c, c + 1) U / 65536 (give or take the rounding of one multiplication); where both ends of
// that stretch, widened by a thousandth of itself, fall into the same bin, that is the point's bin -- and the
// exact projection is at least that margin away from any bin edge, so the rule of fx_hist_bin gives the same.
// The other points (two codes or so per bin edge) are read and projected again.  Counting: up to eight bins, a
// thread counts its (few) points in eight 8-bit fields of a register and the wave adds its lanes' fields up --
// most points share a few bins, and lanes adding to one LDS word take turns; more bins: LDS, a few copies per
// wave.
constexpr int CLOUD_CODE_ROUNDS = CLOUD_CHUNK / (8 * CLOUD_THREADS);
struct CloudCodes {
    uint4 w[CLOUD_CODE_ROUNDS];
};
// a thread's share of the chunk's codes (all of the chunk's stretch of the array: the caller need not know yet
// how many points the chunk has)
__device__ __forceinline__ CloudCodes cloud_load_codes(const CloudWork &Wk, int chunk)
{
    const uint4 *codes = reinterpret_cast<const uint4 *>(Wk.codes + (int64_t)chunk * CLOUD_CHUNK);
    CloudCodes c;
#pragma unroll
    for (int r = 0; r < CLOUD_CODE_ROUNDS; r++)
        c.w[r] = codes[r * CLOUD_THREADS + threadIdx.x];
    return c;
}

template <typename T>
__device__ __forceinline__ void cloud_hist_step(const PcaArgs &A, const CloudWork &Wk, const ChunkView<T> &v, const CloudCodes &codes,
                                                const double *st, double top, uint32_t (*hist)[CLOUD_HIST_PITCH])
{
    static_assert(CLOUD_CHUNK / CLOUD_THREADS <= 255, "a thread's points must fit an 8-bit count");
    const int nb = A.n_bins;
    for (int k = threadIdx.x; k < CLOUD_WAVES * CLOUD_HIST_COPIES * CLOUD_HIST_PITCH; k += CLOUD_THREADS)
        (&hist[0][0])[k] = 0u;
    __syncthreads();
    if (top > 0.0) {
        const double step = top / (double)nb;
        const double inv_step = (double)nb / top;
        const double code_to_bin = inv_step * (st[7] / CLOUD_CODES);
        uint32_t *mine = hist[fx_wave() * CLOUD_HIST_COPIES + (fx_lane() & (CLOUD_HIST_COPIES - 1))];
        const bool packed = nb <= 8;
        u64 fields = 0ull;
        const int cnt = (int)(v.i1 - v.i0);
        const auto count = [&](int bin) {
            if (packed)
                fields += 1ull << (8 * bin);
            else
                atomicAdd(&mine[bin], 1u);
        };
        static_assert(8 * CLOUD_CODE_ROUNDS <= 32, "one bit per code of a thread");
        uint32_t again = 0u;   // this thread's points that are to be projected again
#pragma unroll
        for (int r = 0; r < CLOUD_CODE_ROUNDS; r++) {
            const int at = r * CLOUD_THREADS + threadIdx.x;
            const uint32_t four[4] = {codes.w[r].x, codes.w[r].y, codes.w[r].z, codes.w[r].w};
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (8 * at + e < cnt) {
                    const double c = (double)((four[e >> 1] >> (16 * (e & 1))) & 0xffffu);
                    const int b_lo = (int)((c - 0.001) * code_to_bin), b_hi = (int)((c + 1.001) * code_to_bin);
                    if (b_lo != b_hi || b_hi > nb - 1)
                        again |= 1u << (8 * r + e);
                    else
                        count(b_lo);
                }
            }
        }
        while (again) {
            const int bit = __builtin_ctz(again);
            again &= again - 1u;
            const int k = 8 * ((bit >> 3) * CLOUD_THREADS + threadIdx.x) + (bit & 7);
            const CloudPoint p = cloud_point(v.g, v.i0 + k, A.dims);
            count(fx_hist_bin(cloud_projection(p.x, p.y, p.z, st), step, inv_step, nb));
        }
        if (packed) {
            // even and odd fields apart, 16 bits each: a wave's sum of a field stays below 2^16
            const u64 even = fx_wave_sum_u64(fields & 0x00ff00ff00ff00ffull);
            const u64 odd = fx_wave_sum_u64((fields >> 8) & 0x00ff00ff00ff00ffull);
            if (fx_lane() < nb) {
                const int b = fx_lane();   // (lanes 0-7: word b of copy (lane mod 4) of their wave)
                mine[b] = (uint32_t)(((b & 1) ? odd : even) >> (16 * (b >> 1))) & 0xffffu;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < nb) {
        uint32_t c = 0;
#pragma unroll
        for (int w = 0; w < CLOUD_WAVES * CLOUD_HIST_COPIES; w++)
            c += hist[w][threadIdx.x];
        if (c)
            atomicAdd(&Wk.ctl[v.cloud].hist[threadIdx.x], (u64)c);
    }
    __syncthreads();
}

// the row's histogram part from the cloud's counts; `top`: the cloud's largest |projection|
__device__ __forceinline__ void cloud_write_hist(const PcaArgs &A, const CloudCtl *ctl, int64_t set, int64_t n, double top)
{
    const int nb = A.n_bins;
    if ((int)threadIdx.x >= nb)
        return;
    float *out = A.out + set * (9 + nb) + 9;
    if (!(top > 0.0)) {
        out[threadIdx.x] = threadIdx.x == 0 ? 1.0f : 0.0f;
        return;
    }
    out[threadIdx.x] = (float)((double)ctl->hist[threadIdx.x] / (double)n);
}

// ---- the launches: a fixed grid whose blocks walk the chunk (or cloud) list.  `back`: from the last chunk to
// the first.
template <typename T>
__global__ __launch_bounds__(CLOUD_THREADS, 4) void cloud_sum_kernel(PcaArgs A, CloudWork Wk)
{
    __shared__ double red[15 * CLOUD_WAVES];
    const int n = *Wk.n_chunks_dev;
    for (int chunk = blockIdx.x; chunk < n; chunk += gridDim.x)
        cloud_sum_step(A, Wk, chunk_view<T>(A, Wk, chunk), chunk, red);
}

// stage 0: after the sums; stage 1 (normalize only): after the distances
template <typename T>
__global__ __launch_bounds__(CLOUD_THREADS) void cloud_state_kernel(PcaArgs A, CloudWork Wk, int stage)
{
    __shared__ double red[15 * CLOUD_WAVES];
    const int n = *Wk.n_clouds_dev;
    for (int cloud = blockIdx.x; cloud < n; cloud += gridDim.x) {
        const ChunkView<T> v = cloud_view<T>(A, Wk, cloud);
        if (v.nch > 0)
            cloud_state_step(A, Wk, v, stage == 1, red);
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(CLOUD_THREADS, 4) void cloud_rmax_kernel(PcaArgs A, CloudWork Wk, int back)
{
    __shared__ double red[CLOUD_WAVES];
    const int n = *Wk.n_chunks_dev;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        const int chunk = back ? n - 1 - k : k;
        cloud_rmax_step(A, Wk, chunk_view<T>(A, Wk, chunk), red);
    }
}

template <typename T>
__global__ __launch_bounds__(CLOUD_THREADS, 4) void cloud_proj_kernel(PcaArgs A, CloudWork Wk, int back)
{
    __shared__ double red[CLOUD_WAVES];
    __shared__ __align__(16) uint16_t stage[CLOUD_CHUNK];
    const int n = *Wk.n_chunks_dev;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        const int chunk = back ? n - 1 - k : k;
        const ChunkView<T> v = chunk_view<T>(A, Wk, chunk);
        double st[8];
#pragma unroll
        for (int q = 0; q < 8; q++)
            st[q] = Wk.state[(int64_t)v.cloud * STATE + q];
        cloud_proj_step(A, Wk, v, chunk, st, stage, red);
    }
}

template <typename T>
__global__ __launch_bounds__(CLOUD_THREADS, 8) void cloud_hist_kernel(PcaArgs A, CloudWork Wk, int back)
{
    __shared__ uint32_t hist[CLOUD_WAVES * CLOUD_HIST_COPIES][CLOUD_HIST_PITCH];
    const int n = *Wk.n_chunks_dev;
    for (int k = blockIdx.x; k < n; k += gridDim.x) {
        const int chunk = back ? n - 1 - k : k;
        const CloudCodes codes = cloud_load_codes(Wk, chunk);   // (on their way while the chunk is looked up)
        const ChunkView<T> v = chunk_view<T>(A, Wk, chunk);
        double st[8];
#pragma unroll
        for (int q = 0; q < 8; q++)
            st[q] = Wk.state[(int64_t)v.cloud * STATE + q];
        const double top = __longlong_as_double((long long)Wk.ctl[v.cloud].top_bits);
        cloud_hist_step(A, Wk, v, codes, st, top, hist);
    }
}

template <typename T>
__global__ __launch_bounds__(CLOUD_THREADS) void cloud_finish_kernel(PcaArgs A, CloudWork Wk)
{
    const int n = *Wk.n_clouds_dev;
    for (int cloud = blockIdx.x; cloud < n; cloud += gridDim.x) {
        const ChunkView<T> v = cloud_view<T>(A, Wk, cloud);
        if (v.nch > 0)   // block-uniform
            cloud_write_hist(A, Wk.ctl + cloud, v.set, v.n, __longlong_as_double((long long)Wk.ctl[cloud].top_bits));
    }
}

constexpr int64_t WAVE_SET_MAX_POINTS = 2048;  // sets up to this size take the LDS path

template <int WPB>
int launch_wave_kernel(PcaArgs A, int64_t max_points, hipStream_t s)
{
    A.slab_floats = (int)(((max_points * A.dims + 3) / 4) * 4 + 4);
    const size_t lds = (size_t)WPB * A.slab_floats * 4 + (size_t)WPB * HIST_MAX * 4;
    auto k = pca_sets_wave_kernel<WPB>;
    if (lds > 48 * 1024)
        FX_HIP_CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t blocks = (A.n_sets + WPB - 1) / WPB;
    FX_REQUIRE(blocks <= 0x7fffffffLL, "fx_pca_batch_f32: too many sets for one launch");
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(FX_WAVE * WPB), lds, s, A);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

constexpr int64_t REG_SET_MAX_POINTS = 1024;  // 3-D sets up to this size are kept in registers

template <int PPL>
int launch_full_sets(const PcaArgs &A, hipStream_t s)
{
    constexpr int NW = FX_PCA_NW;
    auto k = A.normalize ? pca_sets_full_kernel<PPL, NW, true> : pca_sets_full_kernel<PPL, NW, false>;
    const int64_t blocks = (A.n_sets + NW - 1) / NW;
    FX_REQUIRE(blocks <= 0x7fffffffLL, "fx_pca_batch_f32: too many sets for one launch");
    const int64_t n = (int64_t)FX_WAVE * PPL;
    FX_REQUIRE(A.redo_list && A.redo_count, "fx_pca_batch_f32: no list for the sets to redo");
    if (!A.redo_zeroed)
        FX_HIP_CHECK(hipMemsetAsync(A.redo_count, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(FX_WAVE * NW), 0, s, A, 1.0 / (double)n, 1.0 / (double)(n - 1),
                       1.0 / (double)A.n_bins);
    PcaArgs R = A;
    R.set_ids = A.redo_list;
    R.n_sets_dev = A.redo_count;
    hipLaunchKernelGGL((pca_sets_redo_kernel<PPL, NW>), dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(FX_WAVE * NW), 0, s, R);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

// Sets of up to 64 * PPL points of any layout (set lists, CSR offsets, 2-D, unaligned): the ragged variant of the
// full-set kernel; A.redo_list / A.redo_count take the sets left over.
template <int PPL>
int launch_ragged_sets(const PcaArgs &A, hipStream_t s)
{
    constexpr int NW = FX_PCA_NW;
    auto k = A.normalize ? pca_sets_full_kernel<PPL, NW, true, true> : pca_sets_full_kernel<PPL, NW, false, true>;
    const int64_t blocks = (A.n_sets + NW - 1) / NW;
    FX_REQUIRE(blocks <= 0x7fffffffLL, "fx_pca_batch_f32: too many sets for one launch");
    FX_REQUIRE(A.redo_list && A.redo_count, "fx_pca_batch_f32: no list for the sets to redo");
    if (!A.redo_zeroed)
        FX_HIP_CHECK(hipMemsetAsync(A.redo_count, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(FX_WAVE * NW), 0, s, A, 0.0, 0.0, 1.0 / (double)A.n_bins);
    PcaArgs R = A;
    R.set_ids = A.redo_list;
    R.n_sets_dev = A.redo_count;
    hipLaunchKernelGGL((pca_sets_redo_kernel<PPL, NW, true>), dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(FX_WAVE * NW), 0, s, R);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

// Sets of up to max_points <= QUAD_MAX_POINTS points, four per wave; A.redo_list / A.redo_count take the sets left
// over.
int launch_quad_sets(const PcaArgs &A, int64_t max_points, hipStream_t s)
{
    constexpr int NW = FX_PCA_NW;
    const int64_t waves = (A.n_sets + QUAD_SETS - 1) / QUAD_SETS;
    const int64_t blocks = (waves + NW - 1) / NW;
    FX_REQUIRE(blocks <= 0x7fffffffLL, "fx_pca_batch_f32: too many sets for one launch");
    FX_REQUIRE(A.redo_list && A.redo_count, "fx_pca_batch_f32: no list for the sets to redo");
    if (!A.redo_zeroed)
        FX_HIP_CHECK(hipMemsetAsync(A.redo_count, 0, sizeof(int32_t), s));
    auto k = max_points <= QUAD_LANES * 4 ? pca_sets_quad_kernel<NW, 4>
             : max_points <= QUAD_LANES * 8 ? pca_sets_quad_kernel<NW, 8> : pca_sets_quad_kernel<NW, 16>;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(FX_WAVE * NW), 0, s, A, 1.0 / (double)A.n_bins);
    PcaArgs R = A;
    R.set_ids = A.redo_list;
    R.n_sets_dev = A.redo_count;
    hipLaunchKernelGGL((pca_sets_redo_kernel<4, NW, true>), dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(FX_WAVE * NW), 0, s, R);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

// The three lists of small sets of a ragged batch (Q; at most n_sets entries together) in one launch.
int launch_quad_lists(const PcaArgs &A, const QuadLists &Q, hipStream_t s)
{
    constexpr int NW = FX_PCA_NW;
    const int64_t waves = (A.n_sets + QUAD_SETS - 1) / QUAD_SETS + 3;   // (a partly filled wave per list)
    const int64_t blocks = (waves + NW - 1) / NW;
    FX_REQUIRE(blocks <= 0x7fffffffLL, "fx_pca_batch_f32: too many sets for one launch");
    FX_REQUIRE(A.redo_list && A.redo_count, "fx_pca_batch_f32: no list for the sets to redo");
    if (!A.redo_zeroed)
        FX_HIP_CHECK(hipMemsetAsync(A.redo_count, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL((pca_sets_quad_mixed_kernel<NW>), dim3((unsigned)blocks), dim3(FX_WAVE * NW), 0, s, A, Q, 1.0 / (double)A.n_bins);
    PcaArgs R = A;
    R.set_ids = A.redo_list;
    R.n_sets_dev = A.redo_count;
    hipLaunchKernelGGL((pca_sets_redo_kernel<4, NW, true>), dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(FX_WAVE * NW), 0, s, R);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

int launch_reg_sets(const PcaArgs &A, int64_t max_points, bool ragged, hipStream_t s)
{
    if (max_points <= QUAD_MAX_POINTS)
        return launch_quad_sets(A, max_points, s);
    if (!ragged && !A.offsets && !A.set_ids) {
        if (max_points == 1024)
            return launch_full_sets<16>(A, s);
        if (max_points == 512)
            return launch_full_sets<8>(A, s);
    }
    return max_points <= 512 ? launch_ragged_sets<8>(A, s) : launch_ragged_sets<16>(A, s);
}

int launch_small_sets(const PcaArgs &A, int64_t max_points, hipStream_t s)
{
    if (max_points > REG_SET_MAX_POINTS && max_points <= 2 * REG_SET_MAX_POINTS) {
        // sets of 1025..2048 points: the register kernel with 32 points per lane
        return launch_ragged_sets<32>(A, s);
    }
    // keep each block's LDS near 48 KiB so that three blocks share a CU
    if (max_points <= 1024)
        return launch_wave_kernel<4>(A, max_points, s);
    return launch_wave_kernel<2>(A, max_points, s);
}

// ------------------------------------------------ ragged batches, lists on the device
// A CSR batch is sorted into three work lists without the host ever seeing the
// offsets: sets for the register kernel, sets for the LDS kernel, large clouds
// (with their chunk lists).  The list lengths stay in device memory; the
// launches that consume them are sized by upper bounds and their surplus blocks
// leave at once (set kernels) or their blocks walk the list (cloud chain).
// Work lists of a batch: the sets of a CSR batch sorted by the kernel that takes them (built on the device by
// pca_classify_kernel), and for the large clouds the chunk lists.
struct RaggedLists {
    int32_t *counts;       // [0] register-path sets, [1] LDS-path sets, [2] large clouds, [3] chunks, [4] - [6] small
                           // sets (up to 256, 128, 64 points);
                           // [16] - [18] sets that the three set launches leave to their second kernel
    int32_t *reg_ids;      // [B]
    int32_t *lds_ids;      // [B]
    int32_t *quad_ids[3];  // [B] each: sets of up to 256, 128 and 64 points (four per wave)
    int32_t *cloud_ids;    // [max_clouds]
    int32_t *cloud_first;  // [max_clouds + 1]
    CloudChunk *chunks;    // [max_chunks]
    CloudCtl *ctl;         // [max_clouds]
    int64_t max_clouds, max_chunks;
    int64_t max_reg, max_lds;   // entries the launches over the register and the LDS list are sized for
    int64_t total;         // points behind the caller's pointer (CSR batches), or 0
};

__device__ __forceinline__ int64_t pca_set_size(const PcaArgs &A, int64_t set)
{
    return A.offsets ? A.offsets[set + 1] - A.offsets[set] : A.N;
}

constexpr int CLASSIFY_THREADS = 1024, CLASSIFY_WAVES = CLASSIFY_THREADS / FX_WAVE;
// mode 1: every set goes to the cloud list (float64 input).  (Batches of few sets are classified by
// pca_few_prepare_kernel.)
__global__ __launch_bounds__(CLASSIFY_THREADS) void pca_classify_kernel(PcaArgs A, RaggedLists L, int64_t B, int64_t reg_max,
                                                                       int mode)
{
    // one atomic per block and list (thousands of waves adding to the same few counters take turns at the L2);
    // within a block the sets keep their order.  Lists 0 - 2: register path, LDS path, large clouds; 3 - 5: small
    // sets of up to 256, 128, 64 points.
    constexpr int LISTS = 6;
    __shared__ int wave_cnt[CLASSIFY_WAVES][LISTS];
    __shared__ int block_base[LISTS];
    const int64_t b = (int64_t)blockIdx.x * CLASSIFY_THREADS + threadIdx.x;
    int cls = -1;
    bool bad = false;
    if (b < B) {
        // a set of fewer than three points (or with decreasing offsets) gets its NaN row from the set kernels;
        // one that reaches beyond the caller's point count gets it here: nothing behind that count is read
        const int64_t n = pca_set_size(A, b);
        bad = A.offsets && L.total > 0 && (A.offsets[b] < 0 || A.offsets[b + 1] > L.total);
        if (!bad)
            cls = mode == 1 ? 2
                  : n <= QUAD_LANES * 4 ? 5
                  : n <= QUAD_LANES * 8 ? 4
                  : n <= QUAD_MAX_POINTS ? 3
                  : n <= reg_max ? 0
                  : n <= WAVE_SET_MAX_POINTS ? 1 : 2;
    }
    const int lane = fx_lane(), wave = fx_wave();
    unsigned long long mine = 0ull;   // the lanes of this wave in the same list as this lane
#pragma unroll
    for (int c = 0; c < LISTS; c++) {
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(cls == c);
        if (lane == 0)
            wave_cnt[wave][c] = (int)__builtin_popcountll(mask);
        if (cls == c)
            mine = mask;
    }
    __syncthreads();
    if (threadIdx.x < LISTS) {
        const int c = threadIdx.x;
        int total = 0;
        for (int w = 0; w < CLASSIFY_WAVES; w++)
            total += wave_cnt[w][c];
        block_base[c] = total ? atomicAdd(&L.counts[c < 3 ? c : c + 1], total) : 0;
    }
    __syncthreads();
    if (cls >= 0) {
        int pos = block_base[cls] + (int)__builtin_popcountll(mine & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; w++)
            pos += wave_cnt[w][cls];
        int32_t *list = cls == 0 ? L.reg_ids : cls == 1 ? L.lds_ids : cls == 2 ? L.cloud_ids : L.quad_ids[cls - 3];
        const int64_t room = cls == 0 ? L.max_reg : cls == 1 ? L.max_lds : cls == 2 ? L.max_clouds : B;
        if (pos < room)
            list[pos] = (int32_t)b;
        else
            bad = true;   // more sets of this size than the point count allows: the offsets overlap
    }
    if (bad)
        write_nan_row(A, b, 0, 1);
}

// One block.  Exclusive prefix of the clouds' chunk counts -> cloud_first, total -> counts[3]; clouds of fewer
// than three points, and clouds behind the point at which the chunk list would overflow (offsets that overlap),
// get no chunks (pca_chunk_fill_kernel writes their NaN rows).
__device__ __forceinline__ void pca_cloud_scan_body(const PcaArgs &A, const RaggedLists &L, long long *wave_tot, int &carry_s, int &valid_s)
{
    const int t = threadIdx.x, lane = fx_lane(), wave = fx_wave();
    int n_clouds = L.counts[2];
    n_clouds = n_clouds < L.max_clouds ? n_clouds : (int)L.max_clouds;
    if (t == 0)
        carry_s = 0, valid_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n_clouds; i0 += 1024) {
        const int i = i0 + t;
        int nch = 0;
        if (i < n_clouds) {
            const int64_t n = pca_set_size(A, L.cloud_ids[i]);
            const int64_t c = n < 3 ? 0 : (n + CLOUD_CHUNK - 1) / CLOUD_CHUNK;
            nch = c > L.max_chunks ? (int)L.max_chunks + 1 : (int)c;
        }
        long long inc = nch;
#pragma unroll
        for (int o = 1; o < FX_WAVE; o <<= 1) {
            const long long u = __shfl_up(inc, o, FX_WAVE);
            if (lane >= o)
                inc += u;
        }
        if (lane == FX_WAVE - 1)
            wave_tot[wave] = inc;
        __syncthreads();
        long long before = carry_s;
        for (int w = 0; w < wave; w++)
            before += wave_tot[w];
        if (i < n_clouds) {
            const bool fits = before + inc <= L.max_chunks;
            L.cloud_first[i] = fits ? (int)(before + inc - nch) : -1;
            if (fits)
                atomicMax(&valid_s, (int)(before + inc));
        }
        __syncthreads();
        if (t == 1023) {
            const long long c = before + inc;
            carry_s = c > L.max_chunks ? (int)L.max_chunks + 1 : (int)c;   // (stays beyond the limit once it is)
        }
        __syncthreads();
    }
    const int total = valid_s;
    for (int i = t; i < n_clouds; i += 1024)
        if (L.cloud_first[i] < 0)
            L.cloud_first[i] = total;
    if (t == 0) {
        L.cloud_first[n_clouds] = total;
        L.counts[3] = total;
        L.counts[2] = n_clouds;
    }
}
__global__ __launch_bounds__(1024) void pca_cloud_scan_kernel(PcaArgs A, RaggedLists L)
{
    __shared__ long long wave_tot[16];
    __shared__ int carry_s, valid_s;
    pca_cloud_scan_body(A, L, wave_tot, carry_s, valid_s);
}

// A wave per cloud: its chunks' descriptors, its control record cleared; a cloud without chunks gets
// its NaN row here.
__device__ __forceinline__ void pca_chunk_fill_body(const PcaArgs &A, const RaggedLists &L, int first_wave, int n_waves)
{
    const int n_clouds = L.counts[2];
    const int lane = fx_lane();
    for (int i = first_wave; i < n_clouds; i += n_waves) {
        const int a = L.cloud_first[i], b = L.cloud_first[i + 1];
        if (a == b) {
            write_nan_row(A, L.cloud_ids[i], lane, FX_WAVE);
            continue;
        }
        const int64_t set = L.cloud_ids[i];
        const int64_t n = pca_set_size(A, set);
        CloudChunk d;
        d.base = A.offsets ? A.offsets[set] : set * A.N;
        d.cloud = i;
        d.pad[0] = d.pad[1] = 0;
        for (int c = a + lane; c < b; c += FX_WAVE) {
            d.i0 = (int64_t)(c - a) * CLOUD_CHUNK;
            d.count = (int32_t)(n - d.i0 < CLOUD_CHUNK ? n - d.i0 : CLOUD_CHUNK);
            L.chunks[c] = d;
        }
        u64 *z = reinterpret_cast<u64 *>(L.ctl + i);
        for (int k = lane; k < (int)(sizeof(CloudCtl) / sizeof(u64)); k += FX_WAVE)
            z[k] = 0ull;
    }
}
__global__ __launch_bounds__(FX_BLOCK) void pca_chunk_fill_kernel(PcaArgs A, RaggedLists L)
{
    pca_chunk_fill_body(A, L, blockIdx.x * FX_WAVES_PER_BLOCK + fx_wave(), gridDim.x * FX_WAVES_PER_BLOCK);
}

// Batches of few sets (up to FEW_SETS): one block does what memset, pca_classify_kernel (mode 2),
// pca_cloud_scan_kernel and pca_chunk_fill_kernel do for large batches -- four launches of 4 - 5 us each on a
// stream whose whole call takes 0.2 ms.  The steps read what the step before wrote to global memory: all of it
// by this one block, between block barriers.
__global__ __launch_bounds__(1024) void pca_few_prepare_kernel(PcaArgs A, RaggedLists L, int64_t B, int n_counts)
{
    __shared__ long long wave_tot[16];
    __shared__ int carry_s, valid_s;
    __shared__ int n_reg_s, n_cloud_s;
    const int t = threadIdx.x;
    for (int k = t; k < n_counts; k += 1024)
        L.counts[k] = 0;
    if (t == 0)
        n_reg_s = 0, n_cloud_s = 0;
    __syncthreads();
    // sets in order within each list, as the classification kernel leaves them
    for (int64_t b0 = 0; b0 < B; b0 += 1024) {
        const int64_t b = b0 + t;
        int cls = -1;
        bool bad = false;
        if (b < B) {
            const int64_t n = pca_set_size(A, b);
            bad = A.offsets && L.total > 0 && (A.offsets[b] < 0 || A.offsets[b + 1] > L.total);
            if (!bad)
                cls = n <= WAVE_SET_MAX_POINTS ? 0 : 2;
        }
        const unsigned long long m_reg = __builtin_amdgcn_ballot_w64(cls == 0), m_cloud = __builtin_amdgcn_ballot_w64(cls == 2);
        const int wave = fx_wave(), lane = fx_lane();
        if (lane == 0) {
            wave_tot[wave] = (long long)__builtin_popcountll(m_reg) | ((long long)__builtin_popcountll(m_cloud) << 32);
        }
        __syncthreads();
        if (cls >= 0) {
            int pos = cls == 0 ? n_reg_s : n_cloud_s;
            for (int w = 0; w < wave; w++)
                pos += cls == 0 ? (int)(wave_tot[w] & 0xffffffffll) : (int)(wave_tot[w] >> 32);
            pos += (int)__builtin_popcountll((cls == 0 ? m_reg : m_cloud) & ((1ull << lane) - 1ull));
            int32_t *list = cls == 0 ? L.reg_ids : L.cloud_ids;
            const int64_t room = cls == 0 ? L.max_reg : L.max_clouds;
            if (pos < room)
                list[pos] = (int32_t)b;
            else
                bad = true;   // more sets of this size than the point count allows: the offsets overlap
        }
        if (bad)
            write_nan_row(A, b, 0, 1);
        __syncthreads();
        if (t == 0) {
            int r = 0, c = 0;
            for (int w = 0; w < 16; w++)
                r += (int)(wave_tot[w] & 0xffffffffll), c += (int)(wave_tot[w] >> 32);
            n_reg_s += r, n_cloud_s += c;
        }
        __syncthreads();
    }
    if (t == 0) {
        L.counts[0] = n_reg_s;
        L.counts[2] = n_cloud_s;
        carry_s = 0, valid_s = 0;
    }
    __threadfence_block();
    __syncthreads();
    if (L.max_clouds == 0 || L.max_chunks == 0)
        return;
    pca_cloud_scan_body(A, L, wave_tot, carry_s, valid_s);
    __threadfence_block();
    __syncthreads();
    pca_chunk_fill_body(A, L, fx_wave(), 16);
}

__global__ __launch_bounds__(FX_BLOCK) void pca_iota_kernel(int32_t *ids, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * FX_BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * FX_BLOCK)
        ids[i] = (int32_t)i;
}

// Scratch of the cloud path: lists, queue, records behind `base` (NULL: only the size is wanted).  Returns the
// bytes used.
size_t cloud_layout(unsigned char *base, int64_t max_clouds, int64_t max_chunks, RaggedLists &L, CloudWork &Wk)
{
    size_t at = 0;
    auto take = [&](size_t bytes) {
        unsigned char *p = base ? base + at : nullptr;
        at += (bytes + 255) & ~(size_t)255;
        return p;
    };
    L.cloud_ids = (int32_t *)take((size_t)(max_clouds + 1) * sizeof(int32_t));
    L.cloud_first = (int32_t *)take((size_t)(max_clouds + 1) * sizeof(int32_t));
    L.chunks = (CloudChunk *)take((size_t)max_chunks * sizeof(CloudChunk));
    L.ctl = (CloudCtl *)take((size_t)max_clouds * sizeof(CloudCtl));
    Wk.part = (double *)take((size_t)max_chunks * PART * sizeof(double));
    Wk.state = (double *)take((size_t)max_clouds * STATE * sizeof(double));
    Wk.codes = (uint16_t *)take((size_t)max_chunks * CLOUD_CHUNK * sizeof(uint16_t));
    L.max_clouds = max_clouds;
    L.max_chunks = max_chunks;
    Wk.cloud_ids = L.cloud_ids;
    Wk.cloud_first = L.cloud_first;
    Wk.chunks = L.chunks;
    Wk.ctl = L.ctl;
    Wk.n_clouds_dev = L.counts + 2;
    Wk.n_chunks_dev = L.counts + 3;
    return at;
}

// Enqueues everything behind the cloud list: L.cloud_ids[0 .. counts[2]) hold the clouds (device memory).
// prepared: pca_few_prepare_kernel has built the chunk list already.
template <typename T>
int run_clouds(const PcaArgs &A, RaggedLists &L, CloudWork &Wk, hipStream_t s, bool prepared = false)
{
    const int64_t max_clouds = L.max_clouds, max_chunks = L.max_chunks;
    if (max_clouds == 0 || max_chunks == 0)
        return FX_OK;
    if (!prepared) {
        hipLaunchKernelGGL(pca_cloud_scan_kernel, dim3(1), dim3(1024), 0, s, A, L);
        hipLaunchKernelGGL(pca_chunk_fill_kernel, dim3((unsigned)(max_clouds < 4096 ? (max_clouds + FX_WAVES_PER_BLOCK - 1) / FX_WAVES_PER_BLOCK : 1024)),
                           dim3(FX_BLOCK), 0, s, A, L);
    }
    const dim3 cgrid((unsigned)(max_chunks < 4096 ? max_chunks : 4096)), block(CLOUD_THREADS);
    const dim3 sgrid((unsigned)(max_clouds < 1024 ? max_clouds : 1024));
    int back = 1;   // (the passes over the points alternate in direction)
    hipLaunchKernelGGL(cloud_sum_kernel<T>, cgrid, block, 0, s, A, Wk);
    hipLaunchKernelGGL(cloud_state_kernel<T>, sgrid, block, 0, s, A, Wk, 0);
    if (A.normalize) {
        hipLaunchKernelGGL(cloud_rmax_kernel<T>, cgrid, block, 0, s, A, Wk, back);
        hipLaunchKernelGGL(cloud_state_kernel<T>, sgrid, block, 0, s, A, Wk, 1);
        back ^= 1;
    }
    hipLaunchKernelGGL(cloud_proj_kernel<T>, cgrid, block, 0, s, A, Wk, back);
    hipLaunchKernelGGL(cloud_hist_kernel<T>, cgrid, block, 0, s, A, Wk, back);   // (the codes of the chunks just projected first)
    hipLaunchKernelGGL(cloud_finish_kernel<T>, sgrid, block, 0, s, A, Wk);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

// The cloud path for clouds the host knows (set index + point count, host arrays).
template <typename T>
int launch_large_clouds(const PcaArgs &A, const int32_t *h_sets, const int64_t *h_sizes, int64_t n_clouds,
                        FxCallGuard &guard)
{
    hipStream_t s = guard.stream;
    int64_t n_chunks = 0;
    for (int64_t k = 0; k < n_clouds; k++)
        n_chunks += (h_sizes[k] + CLOUD_CHUNK - 1) / CLOUD_CHUNK;
    FX_REQUIRE(n_chunks <= 0x1fffffffLL && n_clouds <= 0x7fffffffLL, "fx_pca_batch_f32: too many chunks for one launch");
    RaggedLists L = {};
    CloudWork Wk = {};
    const size_t counts_bytes = 256;
    L.counts = nullptr;
    const size_t need = counts_bytes + cloud_layout(nullptr, n_clouds, n_chunks, L, Wk);
    unsigned char *base = (unsigned char *)guard.scratch(need, 2);
    if (!base)
        return FX_ENOMEM;
    L.counts = (int32_t *)base;
    cloud_layout(base + counts_bytes, n_clouds, n_chunks, L, Wk);
    L.total = 0;
    FX_HIP_CHECK(hipMemsetAsync(L.counts, 0, counts_bytes, s));
    const int32_t nc = (int32_t)n_clouds;
    hipError_t e = hipMemcpyAsync(L.cloud_ids, h_sets, (size_t)n_clouds * sizeof(int32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess)
        e = hipMemcpyAsync(L.counts + 2, &nc, sizeof(nc), hipMemcpyHostToDevice, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);  // the host lists are the caller's (pageable) arrays
    if (e != hipSuccess) {
        fx_set_error("fx_pca_batch_f32: uploading the cloud list failed: %s", hipGetErrorString(e));
        return FX_EHIP;
    }
    return run_clouds<T>(A, L, Wk, s);
}

constexpr int64_t FEW_SETS = 2048;   // batches of up to this many sets take one launch for all their sets below the
                                     // large clouds

// Ragged batch without a host round trip: `total` is the caller's count of points behind d_pts (an upper bound
// of offsets[B]), from which the sizes of the lists and of the per-chunk records are bounded.  all_clouds: every
// set takes the cloud path (float64 input).
template <typename T>
int launch_ragged_on_device(const PcaArgs &A, int64_t B, int64_t total, bool all_clouds, FxCallGuard &guard)
{
    hipStream_t s = guard.stream;
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // a large cloud has more than WAVE_SET_MAX_POINTS points, so there are at most total / (that + 1) of them,
    // and sum ceil(n / chunk) <= total / chunk + number of clouds
    int64_t max_clouds = all_clouds ? B : total / (WAVE_SET_MAX_POINTS + 1);
    max_clouds = max_clouds > B ? B : max_clouds;
    const int64_t max_chunks = max_clouds ? total / CLOUD_CHUNK + max_clouds : 0;
    FX_REQUIRE(max_chunks <= 0x1fffffffLL, "fx_pca_batch: too many chunks for one launch");
    const size_t counts_bytes = 256;
    const size_t ids_bytes = all_clouds ? 0 : align((size_t)B * sizeof(int32_t));
    const size_t redo_bytes = all_clouds ? 0 : align(((size_t)B + 64) * sizeof(int32_t));
    RaggedLists L = {};
    CloudWork Wk = {};
    const size_t cloud_bytes = cloud_layout(nullptr, max_clouds, max_chunks, L, Wk);
    unsigned char *base = (unsigned char *)guard.scratch(counts_bytes + 5 * ids_bytes + redo_bytes + cloud_bytes, 2);
    if (!base)
        return FX_ENOMEM;
    unsigned char *p = base;
    L.counts = (int32_t *)p, p += counts_bytes;
    L.reg_ids = (int32_t *)p, p += ids_bytes;
    L.lds_ids = (int32_t *)p, p += ids_bytes;
    for (int c = 0; c < 3; c++)
        L.quad_ids[c] = (int32_t *)p, p += ids_bytes;
    int32_t *redo = (int32_t *)p;
    p += redo_bytes;
    cloud_layout(p, max_clouds, max_chunks, L, Wk);
    L.total = A.offsets ? total : 0;
    // The launches over the register and the LDS list are sized by what the point count allows: a set of the
    // former has more than QUAD_MAX_POINTS points, one of the latter more than REG_SET_MAX_POINTS (a block that
    // finds itself behind its list leaves at once, but a quarter of a million of them take 0.15 - 0.3 ms to start
    // and leave).  The classification gives a set that finds its list full a NaN row: the offsets overlap.
    auto bound = [&](int64_t more_than) {
        const int64_t most = total / (more_than + 1);
        return most < B ? (most > 1 ? most : 1) : B;
    };
    const int64_t reg_bound = bound(QUAD_MAX_POINTS), lds_bound = bound(REG_SET_MAX_POINTS);
    // A batch of few sets (the tail configuration: some dozens of clouds) is not worth six launches for its sets
    // of up to WAVE_SET_MAX_POINTS points -- a launch that finds its list empty still costs 9 us of the stream's
    // time, as much as the sets themselves: they all go to one list and to the register kernel with 32 points per
    // lane, which takes any size up to that limit.
    const bool few = !all_clouds && B <= FEW_SETS;
    L.max_reg = few ? B : reg_bound;
    L.max_lds = lds_bound;

    const int64_t reg_max = REG_SET_MAX_POINTS;
    if (few) {
        hipLaunchKernelGGL(pca_few_prepare_kernel, dim3(1), dim3(1024), 0, s, A, L, B, (int)(counts_bytes / sizeof(int32_t)));
    } else {
        FX_HIP_CHECK(hipMemsetAsync(L.counts, 0, counts_bytes, s));
        hipLaunchKernelGGL(pca_classify_kernel, dim3((unsigned)((B + CLASSIFY_THREADS - 1) / CLASSIFY_THREADS)), dim3(CLASSIFY_THREADS), 0, s, A, L, B,
                           reg_max, all_clouds ? 1 : 0);
    }
    FX_HIP_CHECK(hipGetLastError());
    FxProfileScope prof(s);
    int rc = FX_OK;
    if (few) {
        PcaArgs S = A;
        S.set_ids = L.reg_ids;
        S.n_sets = B;
        S.n_sets_dev = L.counts + 0;
        S.redo_count = L.counts + 16;
        S.redo_zeroed = 1;
        S.redo_list = redo + 64;
        rc = launch_small_sets(S, WAVE_SET_MAX_POINTS, s);   // the register kernel with 32 points per lane
        if (rc != FX_OK)
            return rc;
        return run_clouds<T>(A, L, Wk, s, true);
    }
    if (!all_clouds) {
        PcaArgs S = A;
        S.n_sets = B;                 // sizes the grid; the kernel reads the true counts
        S.redo_count = L.counts + 16;   // (counts are zero; the launches run one after the other and share the list)
        S.redo_zeroed = 1;
        S.redo_list = redo + 64;
        QuadLists Q;
        for (int c = 0; c < 3; c++)
            Q.ids[c] = L.quad_ids[c];
        Q.counts = L.counts + 4;
        rc = launch_quad_lists(S, Q, s);
    }
    if (rc == FX_OK && !all_clouds) {
        PcaArgs S = A;
        S.set_ids = L.reg_ids;
        S.n_sets = reg_bound;
        S.n_sets_dev = L.counts + 0;
        S.redo_count = L.counts + 17;
        S.redo_zeroed = 1;
        S.redo_list = redo + 64;
        rc = launch_reg_sets(S, REG_SET_MAX_POINTS, true, s);
    }
    if (rc == FX_OK && !all_clouds) {
        PcaArgs S = A;
        S.set_ids = L.lds_ids;
        S.n_sets = lds_bound;
        S.n_sets_dev = L.counts + 1;
        S.redo_count = L.counts + 18;
        S.redo_zeroed = 1;
        S.redo_list = redo + 64;
        rc = launch_small_sets(S, WAVE_SET_MAX_POINTS, s);
    }
    if (rc != FX_OK)
        return rc;
    return run_clouds<T>(A, L, Wk, s);
}

// A CSR batch without a single point (every set empty, so the caller may have no buffer to point at): NaN rows,
// like every set of fewer than three points.  Anything else without a points pointer is an error.
int pca_batch_without_points(const char *who, const int64_t *d_offsets, int64_t B, int n_bins, float *d_out, double *d_eig,
                             hipStream_t s)
{
    FX_REQUIRE(d_offsets != nullptr && d_out != nullptr, "%s: NULL argument", who);
    FX_REQUIRE(n_bins >= 1 && n_bins <= HIST_MAX, "%s: n_bins %d outside 1..%d", who, n_bins, HIST_MAX);
    int64_t ends[2] = {0, 0};
    FX_HIP_CHECK(hipMemcpyAsync(&ends[0], d_offsets, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    FX_HIP_CHECK(hipMemcpyAsync(&ends[1], d_offsets + B, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    FX_HIP_CHECK(hipStreamSynchronize(s));
    FX_REQUIRE(ends[0] == ends[1], "%s: NULL argument (the offsets speak of %lld points)", who, (long long)(ends[1] - ends[0]));
    FX_HIP_CHECK(hipMemsetAsync(d_out, 0xFF, (size_t)B * (size_t)(9 + n_bins) * sizeof(float), s));   // all bits set: a NaN
    if (d_eig)
        FX_HIP_CHECK(hipMemsetAsync(d_eig, 0xFF, (size_t)B * 6 * sizeof(double), s));
    return FX_OK;
}

}  // namespace

extern "C" int fx_pca_batch_f32(const float *d_pts, const int64_t *d_offsets, int64_t B, int64_t N, int dims,
                                int n_bins, int normalize, float *d_out, double *d_eig, void *stream)
{
    FX_REQUIRE(B >= 0, "fx_pca_batch_f32: B = %lld", (long long)B);
    if (B == 0)
        return FX_OK;
    if (!d_pts && d_offsets)
        return pca_batch_without_points("fx_pca_batch_f32", d_offsets, B, n_bins, d_out, d_eig, (hipStream_t)stream);
    FX_REQUIRE(d_pts && d_out, "fx_pca_batch_f32: NULL argument");
    FX_REQUIRE(dims == 2 || dims == 3, "fx_pca_batch_f32: dims %d must be 2 or 3", dims);
    FX_REQUIRE(n_bins >= 1 && n_bins <= HIST_MAX, "fx_pca_batch_f32: n_bins %d outside 1..%d", n_bins, HIST_MAX);
    FX_REQUIRE(d_offsets || N >= 0, "fx_pca_batch_f32: N = %lld", (long long)N);
    FX_REQUIRE((uintptr_t)d_pts % 4 == 0, "fx_pca_batch_f32: d_pts must be 4-byte aligned");
    FX_REQUIRE(B <= 0x7fffffffLL, "fx_pca_batch_f32: B too large");

    hipStream_t s = (hipStream_t)stream;
    PcaArgs A;
    A.pts = d_pts;
    A.pts64 = nullptr;
    A.offsets = d_offsets;
    A.set_ids = nullptr;
    A.n_sets = B;
    A.N = N;
    A.dims = dims;
    A.n_bins = n_bins;
    A.normalize = normalize;
    A.out = d_out;
    A.eig = d_eig;
    A.slab_floats = 0;
    A.n_sets_dev = nullptr;
    A.redo_list = nullptr;
    A.redo_count = nullptr;
    A.redo_zeroed = 0;
    FxCallGuard guard(s);

    if (!d_offsets) {
        const bool reg_ok = N >= 3 && N <= REG_SET_MAX_POINTS;
        if (N >= 3 && N <= WAVE_SET_MAX_POINTS) {
            // pca_sets_full_kernel's list of sets for pca_sets_redo_kernel
            int32_t *redo = (int32_t *)guard.scratch(((size_t)B + 64) * sizeof(int32_t), 0);
            if (!redo)
                return FX_ENOMEM;
            A.redo_count = redo;
            A.redo_list = redo + 64;
        }
        FxProfileScope prof(s);
        if (reg_ok)
            return launch_reg_sets(A, N, dims != 3 || N % 4 != 0 || (uintptr_t)d_pts % 16 != 0, s);
        if (N <= WAVE_SET_MAX_POINTS)
            return launch_small_sets(A, N < 3 ? 3 : N, s);
        int32_t *sets = (int32_t *)malloc((size_t)B * sizeof(int32_t));
        int64_t *sizes = (int64_t *)malloc((size_t)B * sizeof(int64_t));
        if (!sets || !sizes) {
            free(sets);
            free(sizes);
            fx_set_error("fx_pca_batch_f32: host allocation failed");
            return FX_ENOMEM;
        }
        for (int64_t b = 0; b < B; b++) {
            sets[b] = (int32_t)b;
            sizes[b] = N;
        }
        const int rc = launch_large_clouds<float>(A, sets, sizes, B, guard);
        free(sets);
        free(sizes);
        return rc;
    }

    // Ragged batch.  With the caller's point count the work lists are built on the device and nothing waits.
    if (N > 0)
        return launch_ragged_on_device<float>(A, B, N, false, guard);
    // Without it: read the offsets back once, route small sets to the LDS
    // kernel and large clouds to the streaming kernel.
    int64_t *h_off = (int64_t *)malloc((size_t)(B + 1) * sizeof(int64_t));
    if (!h_off) {
        fx_set_error("fx_pca_batch_f32: host allocation failed");
        return FX_ENOMEM;
    }
    hipError_t e = hipMemcpyAsync(h_off, d_offsets, (size_t)(B + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        free(h_off);
        fx_set_error("fx_pca_batch_f32: reading offsets failed: %s", hipGetErrorString(e));
        return FX_EHIP;
    }
    int32_t *h_ids = (int32_t *)malloc((size_t)B * sizeof(int32_t));
    if (!h_ids) {
        free(h_off);
        fx_set_error("fx_pca_batch_f32: host allocation failed");
        return FX_ENOMEM;
    }
    int64_t *h_large_sizes = (int64_t *)malloc((size_t)B * sizeof(int64_t));
    if (!h_large_sizes) {
        free(h_off);
        free(h_ids);
        fx_set_error("fx_pca_batch_f32: host allocation failed");
        return FX_ENOMEM;
    }
    // work list: sets for the register kernel, then those for the LDS kernel, then the large clouds
    const int64_t reg_max = REG_SET_MAX_POINTS;
    int64_t n_reg = 0, n_small = 0, n_large = 0, max_reg = 3, max_small = 3;
    bool ok = true;
    for (int64_t b = 0; b < B; b++) {
        const int64_t n = h_off[b + 1] - h_off[b];
        if (n < 0)
            ok = false;
        if (n <= reg_max) {
            h_ids[n_reg++] = (int32_t)b;
            max_reg = n > max_reg ? n : max_reg;
        }
    }
    n_small = n_reg;
    for (int64_t b = 0; b < B; b++) {
        const int64_t n = h_off[b + 1] - h_off[b];
        if (n > reg_max && n <= WAVE_SET_MAX_POINTS) {
            h_ids[n_small++] = (int32_t)b;
            max_small = n > max_small ? n : max_small;
        }
    }
    for (int64_t b = 0; b < B; b++) {
        const int64_t n = h_off[b + 1] - h_off[b];
        if (n > WAVE_SET_MAX_POINTS) {
            h_large_sizes[n_large] = n;
            h_ids[n_small + n_large++] = (int32_t)b;
        }
    }
    free(h_off);
    if (!ok) {
        free(h_ids);
        free(h_large_sizes);
        fx_set_error("fx_pca_batch_f32: offsets are not non-decreasing");
        return FX_EINVAL;
    }
    int32_t *d_ids = nullptr;
    if (n_small > 0) {
        d_ids = (int32_t *)guard.scratch(((size_t)n_small + (size_t)B + 64) * sizeof(int32_t), 0);
        if (!d_ids) {
            free(h_ids);
            free(h_large_sizes);
            return FX_ENOMEM;
        }
        e = hipMemcpyAsync(d_ids, h_ids, (size_t)n_small * sizeof(int32_t), hipMemcpyHostToDevice, s);
        if (e == hipSuccess)
            e = hipStreamSynchronize(s);  // h_ids is pageable; it is freed below
        if (e != hipSuccess) {
            free(h_ids);
            free(h_large_sizes);
            fx_set_error("fx_pca_batch_f32: uploading the work list failed: %s", hipGetErrorString(e));
            return FX_EHIP;
        }
    }

    FxProfileScope prof(s);
    int rc = FX_OK;
    if (n_reg > 0) {
        PcaArgs S = A;
        S.set_ids = d_ids;
        S.n_sets = n_reg;
        S.redo_count = d_ids + n_small;   // (the two launches run one after the other: they share the list)
        S.redo_list = d_ids + n_small + 64;
        rc = launch_reg_sets(S, max_reg, true, s);
    }
    if (rc == FX_OK && n_small > n_reg) {
        PcaArgs S = A;
        S.set_ids = d_ids + n_reg;
        S.n_sets = n_small - n_reg;
        S.redo_count = d_ids + n_small;
        S.redo_list = d_ids + n_small + 64;
        rc = launch_small_sets(S, max_small, s);
    }
    if (rc == FX_OK && n_large > 0)
        rc = launch_large_clouds<float>(A, h_ids + n_small, h_large_sizes, n_large, guard);
    free(h_ids);
    free(h_large_sizes);
    return rc;
}

extern "C" int fx_pca_batch_f64(const double *d_pts, const int64_t *d_offsets, int64_t B, int64_t N, int dims,
                                int n_bins, int normalize, float *d_out, double *d_eig, void *stream)
{
    FX_REQUIRE(B >= 0, "fx_pca_batch_f64: B = %lld", (long long)B);
    if (B == 0)
        return FX_OK;
    if (!d_pts && d_offsets)
        return pca_batch_without_points("fx_pca_batch_f64", d_offsets, B, n_bins, d_out, d_eig, (hipStream_t)stream);
    FX_REQUIRE(d_pts && d_out, "fx_pca_batch_f64: NULL argument");
    FX_REQUIRE(dims == 2 || dims == 3, "fx_pca_batch_f64: dims %d must be 2 or 3", dims);
    FX_REQUIRE(n_bins >= 1 && n_bins <= HIST_MAX, "fx_pca_batch_f64: n_bins %d outside 1..%d", n_bins, HIST_MAX);
    FX_REQUIRE(d_offsets || N >= 0, "fx_pca_batch_f64: N = %lld", (long long)N);
    FX_REQUIRE((uintptr_t)d_pts % 8 == 0, "fx_pca_batch_f64: d_pts must be 8-byte aligned");
    FX_REQUIRE(B <= 0x7fffffffLL, "fx_pca_batch_f64: B too large");
    hipStream_t s = (hipStream_t)stream;
    PcaArgs A;
    A.pts = nullptr;
    A.pts64 = d_pts;
    A.offsets = d_offsets;
    A.set_ids = nullptr;
    A.n_sets = B;
    A.N = N;
    A.dims = dims;
    A.n_bins = n_bins;
    A.normalize = normalize;
    A.out = d_out;
    A.eig = d_eig;
    A.slab_floats = 0;
    A.n_sets_dev = nullptr;
    A.redo_list = nullptr;
    A.redo_count = nullptr;
    A.redo_zeroed = 0;
    FxCallGuard guard(s);
    int64_t total = d_offsets ? N : B * N;
    if (d_offsets && N <= 0) {
        // the last offset bounds the lists
        hipError_t e = hipMemcpyAsync(&total, d_offsets + B, sizeof(int64_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess)
            e = hipStreamSynchronize(s);
        if (e != hipSuccess) {
            fx_set_error("fx_pca_batch_f64: reading offsets failed: %s", hipGetErrorString(e));
            return FX_EHIP;
        }
        FX_REQUIRE(total >= 0, "fx_pca_batch_f64: offsets end at %lld", (long long)total);
    }
    return launch_ragged_on_device<double>(A, B, total, true, guard);
}

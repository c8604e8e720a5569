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
// Large clouds: cut into chunks of 16384 points, one launch per pass (sum,
// centred second moments, projection maximum, histogram) with a block per
// chunk; a CSR batch is sorted into the lists of the kernels on the device.
#include "fx_common.h"
#include "fx_eig.h"
#include <stdlib.h>
#include <string.h>

namespace {

struct PcaArgs {
    const float *pts;
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
constexpr int64_t QUAD_MAX_POINTS = QUAD_LANES * QUAD_PPL;
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

template <int NW>
__global__ __launch_bounds__(FX_WAVE *NW, 3) void pca_sets_quad_kernel(PcaArgs A, double inv_nb)
{
    __shared__ uint32_t hist_all[NW][QUAD_SETS * QUAD_COPIES * QUAD_PITCH];
    __shared__ float4 list_all[NW][QUAD_LIST_CAP + FX_WAVE];
    __shared__ QuadRec rec_all[NW][QUAD_SETS];
    __shared__ unsigned long long top_all[NW][QUAD_SETS];
    const int lane = fx_lane(), wave = fx_wave();
    const int row = lane / QUAD_LANES, r = lane % QUAD_LANES;
    const int64_t n_sets = A.n_sets_dev ? (int64_t)*A.n_sets_dev : A.n_sets;
    const int64_t first = ((int64_t)blockIdx.x * NW + wave) * QUAD_SETS;
    if (first >= n_sets)
        return;   // the whole wave; there is no block barrier below
    const int64_t slot = first + row;
    const bool in_range = slot < n_sets;
    const SetView sv = resolve_set(A, in_range ? slot : first);
    const int64_t set = sv.set;
    const int nb = A.n_bins, width = 9 + nb, dims = A.dims;
    const bool small = sv.n <= QUAD_MAX_POINTS;
    const bool live = in_range && sv.n >= 3 && small;
    const int n = live ? (int)sv.n : 0;
    if (in_range && sv.n < 3)
        write_nan_row(A, set, r, QUAD_LANES);
    auto to_redo = [&](bool which) {   // lane 0 of every row for which `which` holds hands its set on
        if (which && r == 0)
            A.redo_list[atomicAdd(A.redo_count, 1)] = (int32_t)set;
    };
    to_redo(in_range && !small);   // (the dispatcher sends no such set here)

    uint32_t *hist = hist_all[wave];
    for (int k = lane; k < QUAD_SETS * QUAD_COPIES * QUAD_PITCH; k += FX_WAVE)
        hist[k] = 0;
    uint32_t *my_hist = hist + (row * QUAD_COPIES + (r % QUAD_COPIES)) * QUAD_PITCH;

    // Lane r holds the points r, r + 16, ...; the slots past the end of the set hold the set's first point, which
    // is the pivot of the sums: such a slot adds exact zeros to them, and a copy of a point changes no maximum.
    // (No predicates until the histogram.)
    float x[QUAD_PPL], y[QUAD_PPL], z[QUAD_PPL];
    float kx32 = 0.0f, ky32 = 0.0f, kz32 = 0.0f;
#pragma unroll
    for (int k = 0; k < QUAD_PPL; k++)
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
            for (int k = 0; k < QUAD_PPL; k++) {
                const int p = r + QUAD_LANES * k;
                const P3 v = *reinterpret_cast<const P3 *>(sv.base + 3 * (p < n ? p : 0));
                x[k] = v.a, y[k] = v.b, z[k] = v.c;
            }
            const P3 v = *reinterpret_cast<const P3 *>(sv.base);
            kx32 = v.a, ky32 = v.b, kz32 = v.c;
        } else {   // 2-D points are (x, y, 0)
#pragma unroll
            for (int k = 0; k < QUAD_PPL; k++) {
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
    for (int k = 0; k < QUAD_PPL; k++) {
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
        for (int k = 0; k < QUAD_PPL; k++) {
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
    for (int k = 0; k < QUAD_PPL; k++)
        top32 = fmaxf(top32, fabsf(fmaf(z[k], vz32, fmaf(y[k], vy32, fmaf(x[k], vx32, -c32)))));
    top32 = __int_as_float((int)quad_row_max_u32((uint32_t)__float_as_int(top32)));
    // rows whose projections are at rounding level are redone in float64 throughout
    bool act = live && top32 > 8.0f * eps32;
    to_redo(live && !act);
    const float inv32 = act ? (float)nb / top32 : 0.0f;
    const float qx = vx32 * inv32, qy = vy32 * inv32, qz = vz32 * inv32, qc = c32 * inv32;
    const float margin = (3.0f * eps32 + 6e-7f * top32) * inv32 + 1e-5f * (float)nb;
    if (r == 0) {
        QuadRec *rec = &rec_all[wave][row];
        rec->mean[0] = mx, rec->mean[1] = my, rec->mean[2] = mz;
        rec->v1[0] = v1[0], rec->v1[1] = v1[1], rec->v1[2] = v1[2];
        rec->rmax = scale ? rmax : 0.0;
        rec->q[0] = qx, rec->q[1] = qy, rec->q[2] = qz, rec->q[3] = qc;
        top_all[wave][row] = 0ull;
    }
    auto wave_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    wave_sync();
    // one pass settles the bins of every point float32 can settle; the others are listed with their row
    float4 *list = list_all[wave];
    int n_slow = 0;
    bool overflow = false;
    const int n_act = act ? n : 0;
#pragma unroll
    for (int k = 0; k < QUAD_PPL; k++) {
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
        const double pv = proj(&rec_all[wave][er], pt.x, pt.y, pt.z);
        atomicMax(&top_all[wave][er], (unsigned long long)__double_as_longlong(pv));   // pv >= 0: the bits order like the values
    }
    wave_sync();
    {
        const double mytop = __longlong_as_double((long long)top_all[wave][row]);
        const bool sane = mytop > 1e-280 && mytop < 1e280;   // the shortcuts below assume a plain double
        to_redo(act && !sane);
        act = act && sane;
    }
    for (int e = lane; e < n_slow; e += FX_WAVE) {
        const float4 pt = list[e];
        const int er = __float_as_int(pt.w);
        const QuadRec *rec = &rec_all[wave][er];
        const double top = __longlong_as_double((long long)top_all[wave][er]);
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

// ---------------------------------------------------------------- large sets
// Clouds that do not fit a wave's LDS slab are cut into chunks of CLOUD_CHUNK
// points; every pass over the data is one launch with one block per chunk, so
// a 4M-point cloud spreads over 256 blocks instead of serialising on one, and
// a batch of unequal clouds is balanced at chunk granularity.  Partial results
// go to a per-chunk record; the next launch's blocks combine the records of
// their own cloud (in a fixed tree order: results do not depend on scheduling).
constexpr int CLOUD_CHUNK = 16384;
constexpr int CLOUD_THREADS = 256;
constexpr int CLOUD_WAVES = CLOUD_THREADS / FX_WAVE;
// per-chunk record, doubles: 0-2 sums of p - pivot, 3 max r^2, 4-9 sums of (p - pivot)(p - pivot)^T, 10 max |projection|
constexpr int PART = 12;
// per-cloud record, doubles: 0-2 mean, 3 rmax (0 = no scaling), 4-6 first principal axis
constexpr int STATE = 8;

struct CloudWork {
    const int32_t *cloud_ids;    // [n_clouds] set index of every large cloud
    const int32_t *cloud_first;  // [n_clouds + 1] first chunk of every cloud
    const int32_t *chunk_cloud;  // [n_chunks] position in cloud_ids of the chunk's cloud
    double *part;                // [n_chunks][PART]
    uint32_t *hpart;             // [n_chunks][HIST_MAX]
    double *state;               // [n_clouds][STATE]
    // device-built work lists (ragged batches classified on the device): the counts live in device memory
    // and the launches use a fixed grid whose blocks loop; NULL = the grid is the list
    const int32_t *n_chunks_dev;
    const int32_t *n_clouds_dev;
};

struct ChunkView {
    const float *g;   // first point of the cloud
    int64_t n;        // points in the cloud
    int64_t i0, i1;   // this chunk's point range
    int64_t set;
    int cloud, first, nch;
};

__device__ __forceinline__ ChunkView chunk_view(const PcaArgs &A, const CloudWork &Wk, int c)
{
    ChunkView v;
    v.cloud = Wk.chunk_cloud[c];
    v.set = Wk.cloud_ids[v.cloud];
    v.first = Wk.cloud_first[v.cloud];
    v.nch = Wk.cloud_first[v.cloud + 1] - v.first;
    if (A.offsets) {
        const int64_t a = A.offsets[v.set];
        v.n = A.offsets[v.set + 1] - a;
        v.g = A.pts + a * A.dims;
    } else {
        v.n = A.N;
        v.g = A.pts + v.set * A.N * A.dims;
    }
    v.i0 = (int64_t)(c - v.first) * CLOUD_CHUNK;
    v.i1 = v.i0 + CLOUD_CHUNK < v.n ? v.i0 + CLOUD_CHUNK : v.n;
    return v;
}

template <bool MAX>
__device__ __forceinline__ double cloud_block_reduce(double v, double *red)
{
    v = MAX ? fx_wave_max_f64(v) : fx_wave_sum_f64(v);
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

// Combines field k of the records of chunks first .. first+nch-1; every thread gets the result.
template <bool MAX>
__device__ __forceinline__ double cloud_combine(const double *part, int first, int nch, int k, double *red)
{
    double v = 0.0;
    for (int i = threadIdx.x; i < nch; i += CLOUD_THREADS) {
        const double p = part[(int64_t)(first + i) * PART + k];
        v = MAX ? fmax(v, p) : v + p;
    }
    return cloud_block_reduce<MAX>(v, red);
}

struct CloudPoint {
    double x, y, z;
};
__device__ __forceinline__ CloudPoint cloud_point(const float *g, int64_t i, int dims)
{
    // (three 4-byte loads per point: a wave's 64 lanes then cover 768 contiguous bytes three times; one 12-byte
    // load per lane was 20 % slower over the whole chain)
    CloudPoint p;
    p.x = (double)g[i * dims];
    p.y = (double)g[i * dims + 1];
    p.z = dims == 3 ? (double)g[i * dims + 2] : 0.0;
    return p;
}

// Calls f(x, y, z) for every point of the chunk.  Where the chunk's floats are 16-byte aligned a thread takes
// four consecutive points with three (2-D: two) 16-byte loads -- a wave then reads 3 KB of contiguous memory per
// round, as the set kernels do -- and only the up to three points before the first aligned one and after the
// last whole group are read one by one.
template <typename F>
__device__ __forceinline__ void cloud_for_points(const PcaArgs &A, const ChunkView &v, F f)
{
    const int dims = A.dims;
    int64_t ib = v.i0;
    for (int k = 0; k < 4 && ib < v.i1 && ((uintptr_t)(v.g + ib * dims) & 15u); k++)
        ib++;
    if ((uintptr_t)(v.g + ib * dims) & 15u)
        ib = v.i1;   // (2-D points at an odd float offset never reach alignment: all one by one)
    const int64_t ng = (v.i1 - ib) / 4, ie = ib + 4 * ng;
    for (int64_t i = v.i0 + threadIdx.x; i < ib; i += CLOUD_THREADS) {
        const CloudPoint p = cloud_point(v.g, i, dims);
        f(p.x, p.y, p.z);
    }
    if (dims == 3) {
        const float4 *g4 = reinterpret_cast<const float4 *>(v.g + ib * 3);
        for (int64_t g = threadIdx.x; g < ng; g += CLOUD_THREADS) {
            const float4 a = g4[3 * g], b = g4[3 * g + 1], c = g4[3 * g + 2];
            f((double)a.x, (double)a.y, (double)a.z);
            f((double)a.w, (double)b.x, (double)b.y);
            f((double)b.z, (double)b.w, (double)c.x);
            f((double)c.y, (double)c.z, (double)c.w);
        }
    } else {
        const float4 *g4 = reinterpret_cast<const float4 *>(v.g + ib * 2);
        for (int64_t g = threadIdx.x; g < ng; g += CLOUD_THREADS) {
            const float4 a = g4[2 * g], b = g4[2 * g + 1];
            f((double)a.x, (double)a.y, 0.0);
            f((double)a.z, (double)a.w, 0.0);
            f((double)b.x, (double)b.y, 0.0);
            f((double)b.z, (double)b.w, 0.0);
        }
    }
    for (int64_t i = ie + threadIdx.x; i < v.i1; i += CLOUD_THREADS) {
        const CloudPoint p = cloud_point(v.g, i, dims);
        f(p.x, p.y, p.z);
    }
}

// First pass: sums of d = p - pivot and of d d^T over the chunk, the pivot being the cloud's first point (as in
// the set kernels: mean = pivot + sum d / n, centred second moments = sum d d^T - (sum d)(sum d)^T / n, so the
// mean does not have to be known before the second moments are summed and one pass over the cloud is saved).
__device__ __forceinline__ void cloud_sum_body(const PcaArgs &A, const CloudWork &Wk, const int chunk, double *red)
{
    const ChunkView v = chunk_view(A, Wk, chunk);
    const CloudPoint k = cloud_point(v.g, 0, A.dims);
    double a9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    cloud_for_points(A, v, [&](double px, double py, double pz) {
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
    });
    double *out = Wk.part + (int64_t)chunk * PART;
#pragma unroll
    for (int q = 0; q < 9; q++) {
        const double tot = cloud_block_reduce<false>(a9[q], red);
        if (threadIdx.x == 0)
            out[q < 3 ? q : q + 1] = tot;   // fields 0-2 and 4-9; field 3 is the largest r^2
    }
}

// What every block of a cloud derives from the combined chunk records: mean, covariance sums about the mean.
struct CloudMoments {
    double mx, my, mz;
    double S[6];
};
__device__ __forceinline__ CloudMoments cloud_moments(const PcaArgs &A, const CloudWork &Wk, const ChunkView &v, double *red)
{
    const CloudPoint k = cloud_point(v.g, 0, A.dims);
    const double dn = (double)v.n;
    double t[9];
#pragma unroll
    for (int q = 0; q < 9; q++)
        t[q] = cloud_combine<false>(Wk.part, v.first, v.nch, q < 3 ? q : q + 1, red);
    const double ox = t[0] / dn, oy = t[1] / dn, oz = t[2] / dn;   // mean - pivot
    CloudMoments m;
    m.mx = k.x + ox, m.my = k.y + oy, m.mz = k.z + oz;
    m.S[0] = fma(-t[0], ox, t[3]);
    m.S[1] = fma(-t[0], oy, t[4]);
    m.S[2] = fma(-t[0], oz, t[5]);
    m.S[3] = fma(-t[1], oy, t[6]);
    m.S[4] = fma(-t[1], oz, t[7]);
    m.S[5] = fma(-t[2], oz, t[8]);
    return m;
}

// normalize only: the largest r^2 of the centred points of the chunk
__device__ __forceinline__ void cloud_rmax_body(const PcaArgs &A, const CloudWork &Wk, const int chunk, double *red)
{
    const ChunkView v = chunk_view(A, Wk, chunk);
    const CloudMoments m = cloud_moments(A, Wk, v, red);
    double r2 = 0.0;
    cloud_for_points(A, v, [&](double px, double py, double pz) {
        const double dx = px - m.mx, dy = py - m.my, dz = pz - m.mz;
        r2 = fmax(r2, dx * dx + dy * dy + dz * dz);
    });
    r2 = cloud_block_reduce<true>(r2, red);
    if (threadIdx.x == 0)
        Wk.part[(int64_t)chunk * PART + 3] = r2;
}

__device__ __forceinline__ double cloud_projection(const CloudPoint &p, const double *st)
{
    double dx = p.x - st[0], dy = p.y - st[1], dz = p.z - st[2];
    if (st[3] > 0.0) {
        dx /= st[3];
        dy /= st[3];
        dz /= st[3];
    }
    return fabs(fma(dz, st[6], fma(dy, st[5], dx * st[4])));
}

// Covariance -> eigen-solve (every block of the cloud repeats it: a few hundred
// instructions) -> largest |projection| of the chunk.  The cloud's first block
// also writes the nine scalars and the axis.
__device__ __forceinline__ void cloud_proj_body(const PcaArgs &A, const CloudWork &Wk, const int chunk, double *red)
{
    const ChunkView v = chunk_view(A, Wk, chunk);
    const CloudMoments m = cloud_moments(A, Wk, v, red);
    double rmax = 0.0;
    if (A.normalize)
        rmax = sqrt(cloud_combine<true>(Wk.part, v.first, v.nch, 3, red));
    const bool scale = A.normalize && rmax > 0.0;
    double S[6];
#pragma unroll
    for (int k = 0; k < 6; k++)
        S[k] = scale ? m.S[k] / (rmax * rmax) : m.S[k];   // moments of the points divided by rmax
    double head[9], lam[3], v1[3];
    cov_to_head(S, v.n, head, lam, v1);
    double st[STATE];
    st[0] = m.mx;
    st[1] = m.my;
    st[2] = m.mz;
    st[3] = scale ? rmax : 0.0;
    st[4] = v1[0];
    st[5] = v1[1];
    st[6] = v1[2];
    double top = 0.0;
    cloud_for_points(A, v, [&](double px, double py, double pz) {
        CloudPoint p;
        p.x = px, p.y = py, p.z = pz;
        top = fmax(top, cloud_projection(p, st));
    });
    top = cloud_block_reduce<true>(top, red);
    if (threadIdx.x == 0) {
        Wk.part[(int64_t)chunk * PART + 10] = top;
        if (chunk == v.first) {
            const int width = 9 + A.n_bins;
            for (int k = 0; k < 9; k++)
                A.out[v.set * width + k] = (float)head[k];
            if (A.eig) {
                for (int k = 0; k < 3; k++) {
                    A.eig[v.set * 6 + k] = lam[k];
                    A.eig[v.set * 6 + 3 + k] = v1[k];
                }
            }
        }
    }
    // the axis goes to the cloud record for the histogram pass; every block of
    // the cloud computed the same values, the first one stores them
    if (threadIdx.x == 0 && chunk == v.first) {
        double *wst = Wk.state + (int64_t)v.cloud * STATE;
#pragma unroll
        for (int k = 0; k < 7; k++)
            wst[k] = st[k];
    }
}

__device__ __forceinline__ void cloud_hist_body(const PcaArgs &A, const CloudWork &Wk, const int chunk, double *red, uint32_t *hist)
{
    const ChunkView v = chunk_view(A, Wk, chunk);
    if (threadIdx.x < HIST_MAX)
        hist[threadIdx.x] = 0;
    const double top = cloud_combine<true>(Wk.part, v.first, v.nch, 10, red);  // its barriers cover the zeroing
    double st[STATE];
    const double *gs = Wk.state + (int64_t)v.cloud * STATE;
#pragma unroll
    for (int k = 0; k < 7; k++)
        st[k] = gs[k];
    const int nb = A.n_bins;
    if (top > 0.0) {
        const double step = top / (double)nb;
        const double inv_step = (double)nb / top;
        cloud_for_points(A, v, [&](double px, double py, double pz) {
            CloudPoint p;
            p.x = px, p.y = py, p.z = pz;
            atomicAdd(&hist[fx_hist_bin(cloud_projection(p, st), step, inv_step, nb)], 1u);
        });
    }
    __syncthreads();
    if (threadIdx.x < HIST_MAX)
        Wk.hpart[(int64_t)chunk * HIST_MAX + threadIdx.x] = hist[threadIdx.x];
}

// One block per cloud: add the chunk histograms (wave w takes chunks w, w + 4,
// ..., lane = bin), normalise, write.  Integer sums, so the order is free.
__device__ __forceinline__ void cloud_finish_body(const PcaArgs &A, const CloudWork &Wk, const int cloud, double *tops,
                                                  u64 (*cnts)[HIST_MAX])
{
    const int lane = fx_lane(), wave = fx_wave();
    const int first = Wk.cloud_first[cloud], nch = Wk.cloud_first[cloud + 1] - first;
    const int64_t set = Wk.cloud_ids[cloud];
    const int64_t n = A.offsets ? A.offsets[set + 1] - A.offsets[set] : A.N;
    double top = 0.0;
    for (int i = threadIdx.x; i < nch; i += FX_BLOCK)
        top = fmax(top, Wk.part[(int64_t)(first + i) * PART + 10]);
    top = fx_wave_max_f64(top);
    u64 cnt = 0;
    for (int i = wave; i < nch; i += FX_WAVES_PER_BLOCK)
        cnt += Wk.hpart[(int64_t)(first + i) * HIST_MAX + lane];
    cnts[wave][lane] = cnt;
    if (lane == 0)
        tops[wave] = top;
    __syncthreads();
    const int nb = A.n_bins;
    if (wave != 0 || lane >= nb)
        return;
    top = fmax(fmax(tops[0], tops[1]), fmax(tops[2], tops[3]));
    float *out = A.out + set * (9 + nb) + 9;
    if (!(top > 0.0)) {
        out[lane] = lane == 0 ? 1.0f : 0.0f;
        return;
    }
    cnt = cnts[0][lane] + cnts[1][lane] + cnts[2][lane] + cnts[3][lane];
    out[lane] = (float)((double)cnt / (double)n);
}

// The launches of the chain: a block per chunk (or per cloud), or, when the lists were built on the device and
// their lengths are not known to the host, a fixed grid whose blocks walk the list.
__device__ __forceinline__ int cloud_list_length(const int32_t *dev_count)
{
    return dev_count ? *dev_count : (int)gridDim.x;
}

__global__ __launch_bounds__(CLOUD_THREADS) void cloud_sum_kernel(PcaArgs A, CloudWork Wk)
{
    __shared__ double red[CLOUD_WAVES];
    const int n = cloud_list_length(Wk.n_chunks_dev);
    for (int chunk = blockIdx.x; chunk < n; chunk += gridDim.x)
        cloud_sum_body(A, Wk, chunk, red);
}

__global__ __launch_bounds__(CLOUD_THREADS) void cloud_rmax_kernel(PcaArgs A, CloudWork Wk)
{
    __shared__ double red[CLOUD_WAVES];
    const int n = cloud_list_length(Wk.n_chunks_dev);
    for (int chunk = blockIdx.x; chunk < n; chunk += gridDim.x)
        cloud_rmax_body(A, Wk, chunk, red);
}

__global__ __launch_bounds__(CLOUD_THREADS) void cloud_proj_kernel(PcaArgs A, CloudWork Wk)
{
    __shared__ double red[CLOUD_WAVES];
    const int n = cloud_list_length(Wk.n_chunks_dev);
    for (int chunk = blockIdx.x; chunk < n; chunk += gridDim.x)
        cloud_proj_body(A, Wk, chunk, red);
}

__global__ __launch_bounds__(CLOUD_THREADS) void cloud_hist_kernel(PcaArgs A, CloudWork Wk)
{
    __shared__ double red[CLOUD_WAVES];
    __shared__ uint32_t hist[HIST_MAX];
    const int n = cloud_list_length(Wk.n_chunks_dev);
    for (int chunk = blockIdx.x; chunk < n; chunk += gridDim.x) {
        cloud_hist_body(A, Wk, chunk, red, hist);
        __syncthreads();  // the histogram has been written out before the next chunk clears it
    }
}

__global__ __launch_bounds__(FX_BLOCK) void cloud_finish_kernel(PcaArgs A, CloudWork Wk)
{
    __shared__ double tops[FX_WAVES_PER_BLOCK];
    __shared__ u64 cnts[FX_WAVES_PER_BLOCK][HIST_MAX];
    const int n = cloud_list_length(Wk.n_clouds_dev);
    for (int cloud = blockIdx.x; cloud < n; cloud += gridDim.x) {
        cloud_finish_body(A, Wk, cloud, tops, cnts);
        __syncthreads();
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
    FX_HIP_CHECK(hipMemsetAsync(A.redo_count, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(FX_WAVE * NW), 0, s, A, 0.0, 0.0, 1.0 / (double)A.n_bins);
    PcaArgs R = A;
    R.set_ids = A.redo_list;
    R.n_sets_dev = A.redo_count;
    hipLaunchKernelGGL((pca_sets_redo_kernel<PPL, NW, true>), dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(FX_WAVE * NW), 0, s, R);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

// Sets of up to QUAD_MAX_POINTS points, four per wave; A.redo_list / A.redo_count take the sets left over.
int launch_quad_sets(const PcaArgs &A, hipStream_t s)
{
    constexpr int NW = FX_PCA_NW;
    const int64_t waves = (A.n_sets + QUAD_SETS - 1) / QUAD_SETS;
    const int64_t blocks = (waves + NW - 1) / NW;
    FX_REQUIRE(blocks <= 0x7fffffffLL, "fx_pca_batch_f32: too many sets for one launch");
    FX_REQUIRE(A.redo_list && A.redo_count, "fx_pca_batch_f32: no list for the sets to redo");
    FX_HIP_CHECK(hipMemsetAsync(A.redo_count, 0, sizeof(int32_t), s));
    hipLaunchKernelGGL((pca_sets_quad_kernel<NW>), dim3((unsigned)blocks), dim3(FX_WAVE * NW), 0, s, A, 1.0 / (double)A.n_bins);
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
        return launch_quad_sets(A, s);
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
struct RaggedLists {
    int32_t *counts;       // [0] register sets, [1] sets of 1025..2048 points, [2] large clouds, [3] chunks, [4] small sets
    int32_t *reg_ids;      // [B]
    int32_t *lds_ids;      // [B]
    int32_t *quad_ids;     // [B] sets of up to QUAD_MAX_POINTS points (four per wave)
    int32_t *cloud_ids;    // [max_clouds]
    int32_t *cloud_first;  // [max_clouds + 1]
    int32_t *chunk_cloud;  // [max_chunks]
    int64_t max_clouds, max_chunks;
};

constexpr int CLASSIFY_THREADS = 1024, CLASSIFY_WAVES = CLASSIFY_THREADS / FX_WAVE;
__global__ __launch_bounds__(CLASSIFY_THREADS) void pca_classify_kernel(PcaArgs A, RaggedLists L, int64_t B, int64_t reg_max)
{
    // one atomic per block and list (thousands of waves adding to the same four counters take turns at the L2);
    // within a block the sets keep their order
    __shared__ int wave_cnt[CLASSIFY_WAVES][4];
    __shared__ int block_base[4];
    const int64_t b = (int64_t)blockIdx.x * CLASSIFY_THREADS + threadIdx.x;
    int cls = -1;
    if (b < B) {
        // a set of fewer than three points (or with decreasing offsets) gets its NaN row from the set kernels
        const int64_t n = A.offsets[b + 1] - A.offsets[b];
        cls = n <= QUAD_MAX_POINTS ? 3 : n <= reg_max ? 0 : n <= WAVE_SET_MAX_POINTS ? 1 : 2;
    }
    const int lane = fx_lane(), wave = fx_wave();
    unsigned long long mine = 0ull;   // the lanes of this wave in the same list as this lane
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(cls == c);
        if (lane == 0)
            wave_cnt[wave][c] = (int)__builtin_popcountll(mask);
        if (cls == c)
            mine = mask;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int c = threadIdx.x;
        int total = 0;
        for (int w = 0; w < CLASSIFY_WAVES; w++)
            total += wave_cnt[w][c];
        block_base[c] = total ? atomicAdd(&L.counts[c == 3 ? 4 : c], total) : 0;
    }
    __syncthreads();
    if (cls >= 0) {
        int pos = block_base[cls] + (int)__builtin_popcountll(mine & ((1ull << lane) - 1ull));
        for (int w = 0; w < wave; w++)
            pos += wave_cnt[w][cls];
        int32_t *list = cls == 0 ? L.reg_ids : cls == 1 ? L.lds_ids : cls == 2 ? L.cloud_ids : L.quad_ids;
        if (cls != 2 || pos < L.max_clouds)
            list[pos] = (int32_t)b;
    }
}

// One block: exclusive prefix of the clouds' chunk counts -> cloud_first, total -> counts[3].
__global__ __launch_bounds__(1024) void pca_cloud_scan_kernel(PcaArgs A, RaggedLists L)
{
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int t = threadIdx.x, lane = fx_lane(), wave = fx_wave();
    int n_clouds = L.counts[2];
    n_clouds = n_clouds < L.max_clouds ? n_clouds : (int)L.max_clouds;
    if (t == 0)
        carry_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n_clouds; i0 += 1024) {
        const int i = i0 + t;
        int nch = 0;
        if (i < n_clouds) {
            const int64_t set = L.cloud_ids[i];
            const int64_t n = A.offsets[set + 1] - A.offsets[set];
            nch = (int)((n + CLOUD_CHUNK - 1) / CLOUD_CHUNK);
        }
        int inc = nch;
#pragma unroll
        for (int o = 1; o < FX_WAVE; o <<= 1) {
            const int u = __shfl_up(inc, o, FX_WAVE);
            if (lane >= o)
                inc += u;
        }
        if (lane == FX_WAVE - 1)
            wave_tot[wave] = inc;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wave; w++)
            before += wave_tot[w];
        if (i < n_clouds)
            L.cloud_first[i] = before + inc - nch;
        __syncthreads();
        if (t == 1023)
            carry_s = before + inc;
        __syncthreads();
    }
    if (t == 0) {
        L.cloud_first[n_clouds] = carry_s;
        L.counts[3] = carry_s < L.max_chunks ? carry_s : (int)L.max_chunks;
        L.counts[2] = n_clouds;
    }
}

__global__ __launch_bounds__(FX_BLOCK) void pca_chunk_fill_kernel(RaggedLists L)
{
    const int n_clouds = L.counts[2];
    for (int i = blockIdx.x * FX_BLOCK + threadIdx.x; i < n_clouds; i += gridDim.x * FX_BLOCK) {
        const int a = L.cloud_first[i], b = L.cloud_first[i + 1];
        for (int c = a; c < b && c < L.max_chunks; c++)
            L.chunk_cloud[c] = i;
    }
}

// Builds the chunk lists for the given large clouds (set index + point count,
// host arrays) and runs the passes.
int launch_large_clouds(const PcaArgs &A, const int32_t *h_sets, const int64_t *h_sizes, int64_t n_clouds,
                        FxCallGuard &guard)
{
    hipStream_t s = guard.stream;
    int64_t n_chunks = 0;
    for (int64_t k = 0; k < n_clouds; k++)
        n_chunks += (h_sizes[k] + CLOUD_CHUNK - 1) / CLOUD_CHUNK;
    FX_REQUIRE(n_chunks <= 0x7fffffffLL && n_clouds <= 0x7fffffffLL, "fx_pca_batch_f32: too many chunks for one launch");
    const size_t ints = (size_t)n_clouds + (size_t)(n_clouds + 1) + (size_t)n_chunks;
    int32_t *h = (int32_t *)malloc(ints * sizeof(int32_t));
    if (!h) {
        fx_set_error("fx_pca_batch_f32: host allocation failed");
        return FX_ENOMEM;
    }
    int32_t *h_ids = h, *h_first = h + n_clouds, *h_cc = h_first + n_clouds + 1;
    int64_t c = 0;
    for (int64_t k = 0; k < n_clouds; k++) {
        h_ids[k] = h_sets[k];
        h_first[k] = (int32_t)c;
        const int64_t nch = (h_sizes[k] + CLOUD_CHUNK - 1) / CLOUD_CHUNK;
        for (int64_t j = 0; j < nch; j++)
            h_cc[c++] = (int32_t)k;
    }
    h_first[n_clouds] = (int32_t)c;

    const size_t list_bytes = (ints * sizeof(int32_t) + 15) & ~(size_t)15;
    const size_t part_bytes = (size_t)n_chunks * PART * sizeof(double);
    const size_t hist_bytes = (size_t)n_chunks * HIST_MAX * sizeof(uint32_t);
    const size_t state_bytes = (size_t)n_clouds * STATE * sizeof(double);
    unsigned char *base = (unsigned char *)guard.scratch(list_bytes + part_bytes + hist_bytes + state_bytes, 2);
    if (!base) {
        free(h);
        return FX_ENOMEM;
    }
    hipError_t e = hipMemcpyAsync(base, h, ints * sizeof(int32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);  // the pageable host list is freed right below
    free(h);
    if (e != hipSuccess) {
        fx_set_error("fx_pca_batch_f32: uploading the chunk list failed: %s", hipGetErrorString(e));
        return FX_EHIP;
    }
    CloudWork Wk;
    Wk.cloud_ids = (const int32_t *)base;
    Wk.cloud_first = Wk.cloud_ids + n_clouds;
    Wk.chunk_cloud = Wk.cloud_first + n_clouds + 1;
    Wk.part = (double *)(base + list_bytes);
    Wk.hpart = (uint32_t *)(base + list_bytes + part_bytes);
    Wk.state = (double *)(base + list_bytes + part_bytes + hist_bytes);
    Wk.n_chunks_dev = nullptr;
    Wk.n_clouds_dev = nullptr;

    const dim3 grid((unsigned)n_chunks), block(CLOUD_THREADS);
    hipLaunchKernelGGL(cloud_sum_kernel, grid, block, 0, s, A, Wk);
    if (A.normalize)
        hipLaunchKernelGGL(cloud_rmax_kernel, grid, block, 0, s, A, Wk);
    hipLaunchKernelGGL(cloud_proj_kernel, grid, block, 0, s, A, Wk);
    hipLaunchKernelGGL(cloud_hist_kernel, grid, block, 0, s, A, Wk);
    hipLaunchKernelGGL(cloud_finish_kernel, dim3((unsigned)n_clouds), dim3(FX_BLOCK), 0, s, A, Wk);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

// Ragged batch without a host round trip: `total` is the caller's count of points behind d_pts (an upper bound
// of offsets[B]), from which the sizes of the lists and of the per-chunk records are bounded.
int launch_ragged_on_device(const PcaArgs &A, int64_t B, int64_t total, FxCallGuard &guard)
{
    hipStream_t s = guard.stream;
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // a large cloud has more than WAVE_SET_MAX_POINTS points, so there are at most total / (that + 1) of them,
    // and sum ceil(n / chunk) <= total / chunk + number of clouds
    int64_t max_clouds = total / (WAVE_SET_MAX_POINTS + 1);
    max_clouds = max_clouds > B ? B : max_clouds;
    const int64_t max_chunks = max_clouds ? total / CLOUD_CHUNK + max_clouds : 0;
    FX_REQUIRE(max_chunks <= 0x7fffffffLL, "fx_pca_batch_f32: too many chunks for one launch");
    const size_t counts_bytes = 256;
    const size_t ids_bytes = align((size_t)B * sizeof(int32_t));
    const size_t cid_bytes = align((size_t)(max_clouds + 1) * sizeof(int32_t));
    const size_t cc_bytes = align((size_t)max_chunks * sizeof(int32_t));
    const size_t part_bytes = align((size_t)max_chunks * PART * sizeof(double));
    const size_t hist_bytes = align((size_t)max_chunks * HIST_MAX * sizeof(uint32_t));
    const size_t state_bytes = align((size_t)max_clouds * STATE * sizeof(double));
    const size_t redo_bytes = align(((size_t)B + 64) * sizeof(int32_t));
    unsigned char *base = (unsigned char *)guard.scratch(
        counts_bytes + 3 * ids_bytes + redo_bytes + 2 * cid_bytes + cc_bytes + part_bytes + hist_bytes + state_bytes, 2);
    if (!base)
        return FX_ENOMEM;
    RaggedLists L;
    unsigned char *p = base;
    L.counts = (int32_t *)p, p += counts_bytes;
    L.reg_ids = (int32_t *)p, p += ids_bytes;
    L.lds_ids = (int32_t *)p, p += ids_bytes;
    L.quad_ids = (int32_t *)p, p += ids_bytes;
    int32_t *redo = (int32_t *)p;
    p += redo_bytes;
    L.cloud_ids = (int32_t *)p, p += cid_bytes;
    L.cloud_first = (int32_t *)p, p += cid_bytes;
    L.chunk_cloud = (int32_t *)p, p += cc_bytes;
    L.max_clouds = max_clouds;
    L.max_chunks = max_chunks;
    CloudWork Wk;
    Wk.cloud_ids = L.cloud_ids;
    Wk.cloud_first = L.cloud_first;
    Wk.chunk_cloud = L.chunk_cloud;
    Wk.part = (double *)p, p += part_bytes;
    Wk.hpart = (uint32_t *)p, p += hist_bytes;
    Wk.state = (double *)p;
    Wk.n_chunks_dev = L.counts + 3;
    Wk.n_clouds_dev = L.counts + 2;

    FX_HIP_CHECK(hipMemsetAsync(L.counts, 0, counts_bytes, s));
    const int64_t reg_max = REG_SET_MAX_POINTS;
    hipLaunchKernelGGL(pca_classify_kernel, dim3((unsigned)((B + CLASSIFY_THREADS - 1) / CLASSIFY_THREADS)), dim3(CLASSIFY_THREADS), 0, s, A, L, B,
                       reg_max);
    FX_HIP_CHECK(hipGetLastError());
    FxProfileScope prof(s);
    int rc = FX_OK;
    {
        PcaArgs S = A;
        S.set_ids = L.quad_ids;
        S.n_sets = B;                 // sizes the grid; the kernel reads the true count
        S.n_sets_dev = L.counts + 4;
        S.redo_count = redo;
        S.redo_list = redo + 64;
        rc = launch_quad_sets(S, s);
    }
    if (rc == FX_OK) {
        PcaArgs S = A;
        S.set_ids = L.reg_ids;
        S.n_sets = B;
        S.n_sets_dev = L.counts + 0;
        S.redo_count = redo;          // (the launches run one after the other: they share the list)
        S.redo_list = redo + 64;
        rc = launch_reg_sets(S, REG_SET_MAX_POINTS, true, s);
    }
    if (rc == FX_OK) {
        PcaArgs S = A;
        S.set_ids = L.lds_ids;
        S.n_sets = B;
        S.n_sets_dev = L.counts + 1;
        S.redo_count = redo;
        S.redo_list = redo + 64;
        rc = launch_small_sets(S, WAVE_SET_MAX_POINTS, s);
    }
    if (rc != FX_OK || max_clouds == 0)
        return rc;
    hipLaunchKernelGGL(pca_cloud_scan_kernel, dim3(1), dim3(1024), 0, s, A, L);
    hipLaunchKernelGGL(pca_chunk_fill_kernel, dim3((unsigned)(max_clouds < 65536 ? (max_clouds + FX_BLOCK - 1) / FX_BLOCK : 256)),
                       dim3(FX_BLOCK), 0, s, L);
    const dim3 grid((unsigned)(max_chunks < 2048 ? max_chunks : 2048)), block(CLOUD_THREADS);
    hipLaunchKernelGGL(cloud_sum_kernel, grid, block, 0, s, A, Wk);
    if (A.normalize)
        hipLaunchKernelGGL(cloud_rmax_kernel, grid, block, 0, s, A, Wk);
    hipLaunchKernelGGL(cloud_proj_kernel, grid, block, 0, s, A, Wk);
    hipLaunchKernelGGL(cloud_hist_kernel, grid, block, 0, s, A, Wk);
    hipLaunchKernelGGL(cloud_finish_kernel, dim3((unsigned)(max_clouds < 1024 ? max_clouds : 1024)), dim3(FX_BLOCK), 0, s, A,
                       Wk);
    FX_HIP_CHECK(hipGetLastError());
    return FX_OK;
}

}  // namespace

extern "C" int fx_pca_batch_f32(const float *d_pts, const int64_t *d_offsets, int64_t B, int64_t N, int dims,
                                int n_bins, int normalize, float *d_out, double *d_eig, void *stream)
{
    FX_REQUIRE(B >= 0, "fx_pca_batch_f32: B = %lld", (long long)B);
    if (B == 0)
        return FX_OK;
    FX_REQUIRE(d_pts && d_out, "fx_pca_batch_f32: NULL argument");
    FX_REQUIRE(dims == 2 || dims == 3, "fx_pca_batch_f32: dims %d must be 2 or 3", dims);
    FX_REQUIRE(n_bins >= 1 && n_bins <= HIST_MAX, "fx_pca_batch_f32: n_bins %d outside 1..%d", n_bins, HIST_MAX);
    FX_REQUIRE(d_offsets || N >= 0, "fx_pca_batch_f32: N = %lld", (long long)N);
    FX_REQUIRE((uintptr_t)d_pts % 4 == 0, "fx_pca_batch_f32: d_pts must be 4-byte aligned");
    FX_REQUIRE(B <= 0x7fffffffLL, "fx_pca_batch_f32: B too large");

    hipStream_t s = (hipStream_t)stream;
    PcaArgs A;
    A.pts = d_pts;
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
        const int rc = launch_large_clouds(A, sets, sizes, B, guard);
        free(sets);
        free(sizes);
        return rc;
    }

    // Ragged batch.  With the caller's point count the work lists are built on the device and nothing waits.
    if (N > 0)
        return launch_ragged_on_device(A, B, N, guard);
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
        rc = launch_large_clouds(A, h_ids + n_small, h_large_sizes, n_large, guard);
    free(h_ids);
    free(h_large_sizes);
    return rc;
}

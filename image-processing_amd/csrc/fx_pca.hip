// Point-set PCA descriptor kernels (reference
// src/features/extract_3d_pca_features.py:59-145, inputs widened from the
// float32 file contents as load_sample :48-56 does).
//
// Small sets (the pipeline's 1024-point clouds): one wavefront per set.  The
// set is copied once from HBM into the wave's LDS slab with coalesced 16 B
// loads; centring, covariance, eigen-decomposition, |projection| maximum and
// histogram all run out of LDS, so HBM is read exactly once.  No block
// barriers: waves are independent.
//
// Large clouds: one block per cloud streaming from global memory in four
// passes (sum, centred second moments, projection maximum, histogram).
#include "fx_common.h"
#include "fx_eig.h"

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
__device__ __forceinline__ void cov_to_head(const double (&S)[6], int64_t n, double (&head)[9], double (&lam)[3],
                                            double (&v1)[3])
{
    const double inv = 1.0 / (double)(n - 1);
    fx_eig3(S[0] * inv, S[1] * inv, S[2] * inv, S[3] * inv, S[4] * inv, S[5] * inv, lam, v1);
    fx_pca_head(lam, head);
}

template <int WPB>
__global__ __launch_bounds__(FX_WAVE *WPB) void pca_sets_wave_kernel(PcaArgs A)
{
    extern __shared__ __align__(16) float lds[];
    const int lane = fx_lane(), wave = fx_wave();
    const int64_t slot = (int64_t)blockIdx.x * WPB + wave;
    if (slot >= A.n_sets)
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

    // ---- mean ---------------------------------------------------------------
    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int i = lane; i < n; i += FX_WAVE) {
        sx += px(i);
        sy += py(i);
        sz += pz(i);
    }
    const double dn = (double)n;
    const double mx = fx_wave_sum_f64(sx) / dn;
    const double my = fx_wave_sum_f64(sy) / dn;
    const double mz = fx_wave_sum_f64(sz) / dn;

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

    // ---- centred second moments --------------------------------------------
    double S[6] = {0, 0, 0, 0, 0, 0};
    for (int i = lane; i < n; i += FX_WAVE) {
        double dx = px(i) - mx, dy = py(i) - my, dz = pz(i) - mz;
        if (scale) {
            dx /= rmax;
            dy /= rmax;
            dz /= rmax;
        }
        S[0] = fma(dx, dx, S[0]);
        S[1] = fma(dx, dy, S[1]);
        S[2] = fma(dx, dz, S[2]);
        S[3] = fma(dy, dy, S[3]);
        S[4] = fma(dy, dz, S[4]);
        S[5] = fma(dz, dz, S[5]);
    }
#pragma unroll
    for (int k = 0; k < 6; k++)
        S[k] = fx_wave_sum_f64(S[k]);

    // every lane holds the same sums and runs the same eigen-solve
    double head[9], lam[3], v1[3];
    cov_to_head(S, n, head, lam, v1);

    // ---- |projection on PC1|: maximum, then histogram ----------------------
    auto proj = [&](int i) -> double {
        double dx = px(i) - mx, dy = py(i) - my, dz = pz(i) - mz;
        if (scale) {
            dx /= rmax;
            dy /= rmax;
            dz /= rmax;
        }
        return fabs(fma(dz, v1[2], fma(dy, v1[1], dx * v1[0])));
    };
    double top = 0.0;
    for (int i = lane; i < n; i += FX_WAVE)
        top = fmax(top, proj(i));
    top = fx_wave_max_f64(top);

    const int nb = A.n_bins;
    const int width = 9 + nb;
    if (top > 0.0) {
        const double step = top / (double)nb;
        const double inv_step = (double)nb / top;
        for (int i = lane; i < n; i += FX_WAVE)
            atomicAdd(&hist[fx_hist_bin(proj(i), step, inv_step, nb)], 1u);
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

// ---------------------------------------------------------------- large sets
constexpr int CLOUD_BLOCK = 512;
constexpr int CLOUD_WAVES = CLOUD_BLOCK / FX_WAVE;

__device__ __forceinline__ double cloud_sum(double v, double *red)
{
    v = fx_wave_sum_f64(v);
    if (fx_lane() == 0)
        red[fx_wave()] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < CLOUD_WAVES; w++)
        s += red[w];
    __syncthreads();
    return s;
}

__device__ __forceinline__ double cloud_max(double v, double *red)
{
    v = fx_wave_max_f64(v);
    if (fx_lane() == 0)
        red[fx_wave()] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int w = 1; w < CLOUD_WAVES; w++)
        s = fmax(s, red[w]);
    __syncthreads();
    return s;
}

__global__ __launch_bounds__(CLOUD_BLOCK) void pca_cloud_block_kernel(PcaArgs A)
{
    __shared__ double red[CLOUD_WAVES];
    __shared__ uint32_t hist[HIST_MAX];
    const int t = threadIdx.x;
    const SetView sv = resolve_set(A, blockIdx.x);
    const int64_t set = sv.set, n = sv.n;
    const int dims = A.dims;
    const float *g = sv.base;
    if (n < 3) {
        write_nan_row(A, set, t, CLOUD_BLOCK);
        return;
    }
    if (t < HIST_MAX)
        hist[t] = 0;

    auto px = [&](int64_t i) -> double { return (double)g[i * dims]; };
    auto py = [&](int64_t i) -> double { return (double)g[i * dims + 1]; };
    auto pz = [&](int64_t i) -> double { return dims == 3 ? (double)g[i * dims + 2] : 0.0; };

    double sx = 0.0, sy = 0.0, sz = 0.0;
    for (int64_t i = t; i < n; i += CLOUD_BLOCK) {
        sx += px(i);
        sy += py(i);
        sz += pz(i);
    }
    const double dn = (double)n;
    const double mx = cloud_sum(sx, red) / dn;
    const double my = cloud_sum(sy, red) / dn;
    const double mz = cloud_sum(sz, red) / dn;

    double rmax = 0.0;
    if (A.normalize) {
        double r2 = 0.0;
        for (int64_t i = t; i < n; i += CLOUD_BLOCK) {
            const double dx = px(i) - mx, dy = py(i) - my, dz = pz(i) - mz;
            r2 = fmax(r2, dx * dx + dy * dy + dz * dz);
        }
        rmax = sqrt(cloud_max(r2, red));
    }
    const bool scale = A.normalize && rmax > 0.0;

    double S[6] = {0, 0, 0, 0, 0, 0};
    for (int64_t i = t; i < n; i += CLOUD_BLOCK) {
        double dx = px(i) - mx, dy = py(i) - my, dz = pz(i) - mz;
        if (scale) {
            dx /= rmax;
            dy /= rmax;
            dz /= rmax;
        }
        S[0] = fma(dx, dx, S[0]);
        S[1] = fma(dx, dy, S[1]);
        S[2] = fma(dx, dz, S[2]);
        S[3] = fma(dy, dy, S[3]);
        S[4] = fma(dy, dz, S[4]);
        S[5] = fma(dz, dz, S[5]);
    }
#pragma unroll
    for (int k = 0; k < 6; k++)
        S[k] = cloud_sum(S[k], red);

    double head[9], lam[3], v1[3];
    cov_to_head(S, n, head, lam, v1);

    auto proj = [&](int64_t i) -> double {
        double dx = px(i) - mx, dy = py(i) - my, dz = pz(i) - mz;
        if (scale) {
            dx /= rmax;
            dy /= rmax;
            dz /= rmax;
        }
        return fabs(fma(dz, v1[2], fma(dy, v1[1], dx * v1[0])));
    };
    double top = 0.0;
    for (int64_t i = t; i < n; i += CLOUD_BLOCK)
        top = fmax(top, proj(i));
    top = cloud_max(top, red);

    const int nb = A.n_bins;
    const int width = 9 + nb;
    if (top > 0.0) {
        const double step = top / (double)nb;
        const double inv_step = (double)nb / top;
        for (int64_t i = t; i < n; i += CLOUD_BLOCK)
            atomicAdd(&hist[fx_hist_bin(proj(i), step, inv_step, nb)], 1u);
        __syncthreads();
        if (t < nb)
            A.out[set * width + 9 + t] = (float)((double)hist[t] / dn);
    } else if (t < nb) {
        A.out[set * width + 9 + t] = t == 0 ? 1.0f : 0.0f;
    }
    if (t == 0) {
        for (int k = 0; k < 9; k++)
            A.out[set * width + k] = (float)head[k];
        if (A.eig) {
            for (int k = 0; k < 3; k++) {
                A.eig[set * 6 + k] = lam[k];
                A.eig[set * 6 + 3 + k] = v1[k];
            }
        }
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

int launch_small_sets(const PcaArgs &A, int64_t max_points, hipStream_t s)
{
    // keep each block's LDS near 48 KiB so that three blocks share a CU
    if (max_points <= 1024)
        return launch_wave_kernel<4>(A, max_points, s);
    return launch_wave_kernel<2>(A, max_points, s);
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

    if (!d_offsets) {
        FxProfileScope prof(s);
        if (N <= WAVE_SET_MAX_POINTS)
            return launch_small_sets(A, N < 3 ? 3 : N, s);
        hipLaunchKernelGGL(pca_cloud_block_kernel, dim3((unsigned)B), dim3(CLOUD_BLOCK), 0, s, A);
        FX_HIP_CHECK(hipGetLastError());
        return FX_OK;
    }

    // Ragged batch: read the offsets back once, route small sets to the LDS
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
    int64_t n_small = 0, n_large = 0, max_small = 3;
    bool ok = true;
    for (int64_t b = 0; b < B; b++) {
        const int64_t n = h_off[b + 1] - h_off[b];
        if (n < 0)
            ok = false;
        if (n <= WAVE_SET_MAX_POINTS) {
            h_ids[n_small++] = (int32_t)b;
            max_small = n > max_small ? n : max_small;
        }
    }
    for (int64_t b = 0; b < B; b++) {
        const int64_t n = h_off[b + 1] - h_off[b];
        if (n > WAVE_SET_MAX_POINTS)
            h_ids[n_small + n_large++] = (int32_t)b;
    }
    free(h_off);
    if (!ok) {
        free(h_ids);
        fx_set_error("fx_pca_batch_f32: offsets are not non-decreasing");
        return FX_EINVAL;
    }
    int32_t *d_ids = (int32_t *)fx_scratch((size_t)B * sizeof(int32_t), 0);
    if (!d_ids) {
        free(h_ids);
        return FX_ENOMEM;
    }
    e = hipMemcpyAsync(d_ids, h_ids, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess)
        e = hipStreamSynchronize(s);  // h_ids is pageable; it is freed right below
    free(h_ids);
    if (e != hipSuccess) {
        fx_set_error("fx_pca_batch_f32: uploading the work list failed: %s", hipGetErrorString(e));
        return FX_EHIP;
    }

    FxProfileScope prof(s);
    if (n_small > 0) {
        PcaArgs S = A;
        S.set_ids = d_ids;
        S.n_sets = n_small;
        int rc = launch_small_sets(S, max_small, s);
        if (rc != FX_OK)
            return rc;
    }
    if (n_large > 0) {
        PcaArgs L = A;
        L.set_ids = d_ids + n_small;
        L.n_sets = n_large;
        hipLaunchKernelGGL(pca_cloud_block_kernel, dim3((unsigned)n_large), dim3(CLOUD_BLOCK), 0, s, L);
        FX_HIP_CHECK(hipGetLastError());
    }
    return FX_OK;
}

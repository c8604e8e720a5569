// What a kernel that only reads can stream from HBM on this device: every block reads its own contiguous
// segment with 16-byte loads, U loads per lane in flight, and folds the data into one word.  Variants: plain or
// non-temporal loads, a block per segment or a resident grid that walks the segments.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int U, bool NT, bool WALK>
__global__ __launch_bounds__(256) void read_kernel(const u32x4 *buf, int64_t seg_vec, int64_t n_seg, uint32_t *out)
{
    uint32_t acc = 0;
    for (int64_t seg = blockIdx.x; seg < n_seg; seg += WALK ? gridDim.x : n_seg) {
        const u32x4 *p = buf + seg * seg_vec;
        for (int64_t i = threadIdx.x; i < seg_vec; i += 256 * U) {
            u32x4 v[U];
#pragma unroll
            for (int k = 0; k < U; k++) {
                const int64_t j = i + 256 * k;
                const u32x4 *q = p + (j < seg_vec ? j : i);
                v[k] = NT ? __builtin_nontemporal_load(q) : *q;
            }
#pragma unroll
            for (int k = 0; k < U; k++)
                acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    }
    if (acc == 0x12345678u)
        out[blockIdx.x] = acc;
}
// The band kernel's pattern on a 512 x 512 image (a segment of 256 KB): eight bands of 64 rows, a band per
// 32 lanes, four rows of a band in flight -- a block reads at eight places at once, 512 bytes each.
// PAIR: the two halves of a wave take adjacent rows instead (a wave reads 1 KB at one place, four waves at four).
// Column strips: wave w takes the 128-byte quarter w of every row of the 512-byte-wide image, eight rows a load,
// four loads in flight; the four waves walk down side by side (if they stay together, 16 KB at one place).
__global__ __launch_bounds__(256) void strip_kernel(const u32x4 *buf, int64_t seg_vec, int64_t n_seg, uint32_t *out)
{
    uint32_t acc = 0;
    const u32x4 *p = buf + (int64_t)blockIdx.x * seg_vec;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, tx = lane & 7, bl = lane >> 3;
    for (int r = 0; r < 512; r += 32) {
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            v[k] = __builtin_nontemporal_load(p + (int64_t)(r + 8 * k + bl) * 32 + wave * 8 + tx);
#pragma unroll
        for (int k = 0; k < 4; k++)
            acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u)
        out[blockIdx.x] = acc;
}
// Point data: a thread takes four points (48 bytes) with three 16-byte loads, so an instruction touches a third
// of every line it reads (V3 = false); or a point per lane with one 12-byte load, lanes side by side (V3 = true).
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
template <bool V3, bool NT>
__global__ __launch_bounds__(256) void point_kernel(const u32x4 *buf, int64_t seg_vec, int64_t n_seg, uint32_t *out)
{
    uint32_t acc = 0;
    const u32x4 *p = buf + (int64_t)blockIdx.x * seg_vec;
    if (!V3) {
        for (int64_t g = threadIdx.x; 3 * g + 2 < seg_vec; g += 256 * 4) {   // groups of four points, four groups in flight
            u32x4 v[12];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int64_t gg = 3 * (g + 256 * k) + 2 < seg_vec ? g + 256 * k : g;
#pragma unroll
                for (int q = 0; q < 3; q++)
                    v[3 * k + q] = NT ? __builtin_nontemporal_load(p + 3 * gg + q) : p[3 * gg + q];
            }
#pragma unroll
            for (int k = 0; k < 12; k++)
                acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    } else {
        const u32x3 *q3 = reinterpret_cast<const u32x3 *>(p);
        const int64_t n_pts = seg_vec * 16 / 12;
        for (int64_t i = threadIdx.x; i < n_pts; i += 256 * 16) {   // sixteen points in flight
            u32x3 v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int64_t j = i + 256 * k < n_pts ? i + 256 * k : i;
                v[k] = NT ? __builtin_nontemporal_load(q3 + j) : q3[j];
            }
#pragma unroll
            for (int k = 0; k < 16; k++)
                acc ^= v[k].x ^ v[k].y ^ v[k].z;
        }
    }
    if (acc == 0x12345678u)
        out[blockIdx.x] = acc;
}
// A point per lane read as 16 bytes at a 12-byte stride (the fourth word is the next point's first).
template <bool NT>
__global__ __launch_bounds__(256) void point16_kernel(const u32x4 *buf, int64_t seg_vec, int64_t n_seg, uint32_t *out)
{
    typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
    uint32_t acc = 0;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(buf + (int64_t)blockIdx.x * seg_vec);
    const int64_t n_pts = seg_vec * 16 / 12 - 1;
    for (int64_t i = threadIdx.x; i < n_pts; i += 256 * 16) {
        u32x4u v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int64_t j = i + 256 * k < n_pts ? i + 256 * k : i;
            const u32x4u *q = reinterpret_cast<const u32x4u *>(p + 3 * j);
            v[k] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int k = 0; k < 16; k++)
            acc ^= v[k].x ^ v[k].y ^ v[k].z;
    }
    if (acc == 0x12345678u)
        out[blockIdx.x] = acc;
}
// Sets of 12 KB read the way a set kernel with prefetch would: a wave brings its next set (or SETW words of it)
// to LDS with loads that have no register destination (1 KB in one piece per instruction), takes the one that
// has arrived out of LDS a point (12 bytes) per lane, and starts the next transfer into the same buffer.
template <bool NT, int SETW, int OCC>
__global__ __launch_bounds__(256, OCC) void glds_kernel(const u32x4 *buf, int64_t n_sets, uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds[4][SETW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    uint32_t *mine = lds[wave];
    const int64_t stride = (int64_t)gridDim.x * 4;
    auto issue = [&](int64_t set) {
        const u32x4 *g = buf + set * (SETW / 4) + lane;
#pragma unroll
        for (int k = 0; k < SETW / 256; k++)
            __builtin_amdgcn_global_load_lds((glb_void *)(g + 64 * k), (lds_void *)(mine + 256 * k), 16, 0, NT ? 2 : 0);
    };
    uint32_t acc = 0;
    int64_t set = (int64_t)blockIdx.x * 4 + wave;
    if (set < n_sets)
        issue(set);
    for (; set < n_sets; set += stride) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        u32x3 v[SETW / 192];
#pragma unroll
        for (int i = 0; i < SETW / 192; i++)
            v[i] = *reinterpret_cast<const u32x3 *>(mine + 3 * (lane + 64 * i));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (set + stride < n_sets)
            issue(set + stride);
#pragma unroll
        for (int i = 0; i < SETW / 192; i++)
            acc ^= v[i].x ^ v[i].y ^ v[i].z;
    }
    if (acc == 0x12345678u)
        out[blockIdx.x] = acc;
}
template <bool NT, int SETW, int OCC>
void run_glds(const char *name, const u32x4 *buf, int64_t bytes, uint32_t *out, int cus)
{
    const int64_t n_sets = bytes / (SETW * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((glds_kernel<NT, SETW, OCC>), dim3((unsigned)(cus * OCC)), dim3(256), 0, 0, buf, n_sets, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best)
            best = ms;
    }
    printf("%-44s  %2d KB per wave, %d blocks per CU: %7.3f ms  %6.0f GB/s\n", name, SETW * 4 / 1024, OCC, best, n_sets * SETW * 4 / best / 1e6);
}
template <bool NT>
void run_point16(const char *name, const u32x4 *buf, int64_t bytes, uint32_t *out)
{
    const int64_t seg_bytes = 98304, seg_vec = seg_bytes / 16, n_seg = bytes / seg_bytes;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((point16_kernel<NT>), dim3((unsigned)n_seg), dim3(256), 0, 0, buf, seg_vec, n_seg, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best)
            best = ms;
    }
    printf("%-44s segment %5lld KB: %7.3f ms  %6.0f GB/s\n", name, (long long)(seg_bytes >> 10), best, n_seg * seg_bytes / best / 1e6);
}
template <bool V3, bool NT>
void run_points(const char *name, const u32x4 *buf, int64_t bytes, uint32_t *out)
{
    const int64_t seg_bytes = 98304, seg_vec = seg_bytes / 16, n_seg = bytes / seg_bytes;   // 8192 points
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((point_kernel<V3, NT>), dim3((unsigned)n_seg), dim3(256), 0, 0, buf, seg_vec, n_seg, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best)
            best = ms;
    }
    printf("%-44s segment %5lld KB: %7.3f ms  %6.0f GB/s\n", name, (long long)(seg_bytes >> 10), best, n_seg * seg_bytes / best / 1e6);
}
template <bool PAIR>
__global__ __launch_bounds__(256) void band_kernel(const u32x4 *buf, int64_t seg_vec, int64_t n_seg, uint32_t *out)
{
    uint32_t acc = 0;
    const u32x4 *p = buf + (int64_t)blockIdx.x * seg_vec;
    const int t = threadIdx.x, tx = t & 31;
    if (!PAIR) {
        const int band = t >> 5;
        for (int r = 0; r < 64; r += 4) {
            u32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                v[k] = __builtin_nontemporal_load(p + (int64_t)(band * 64 + r + k) * 32 + tx);
#pragma unroll
            for (int k = 0; k < 4; k++)
                acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    } else {
        const int wave = t >> 6, half = (t >> 5) & 1;
        for (int r = 0; r < 128; r += 8) {
            u32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                v[k] = __builtin_nontemporal_load(p + (int64_t)(wave * 128 + r + 2 * k + half) * 32 + tx);
#pragma unroll
            for (int k = 0; k < 4; k++)
                acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
        }
    }
    if (acc == 0x12345678u)
        out[blockIdx.x] = acc;
}
template <bool PAIR>
void run_band(const char *name, const u32x4 *buf, int64_t bytes, uint32_t *out)
{
    const int64_t seg_bytes = 262144, seg_vec = seg_bytes / 16, n_seg = bytes / seg_bytes;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((band_kernel<PAIR>), dim3((unsigned)n_seg), dim3(256), 0, 0, buf, seg_vec, n_seg, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best)
            best = ms;
    }
    printf("%-44s segment %5lld KB: %7.3f ms  %6.0f GB/s\n", name, (long long)(seg_bytes >> 10), best, bytes / best / 1e6);
}
template <int U, bool NT, bool WALK>
void run(const char *name, const u32x4 *buf, int64_t bytes, int64_t seg_bytes, uint32_t *out, int cus)
{
    const int64_t seg_vec = seg_bytes / 16, n_seg = bytes / seg_bytes;
    const int grid = WALK ? cus * 8 : (int)n_seg;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((read_kernel<U, NT, WALK>), dim3(grid), dim3(256), 0, 0, buf, seg_vec, n_seg, out);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best)
            best = ms;
    }
    printf("%-44s segment %5lld KB: %7.3f ms  %6.0f GB/s\n", name, (long long)(seg_bytes >> 10), best, bytes / best / 1e6);
}
int main(int argc, char **argv)
{
    const int64_t bytes = (argc > 1 ? atoll(argv[1]) : 16) << 30;
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    u32x4 *buf;
    uint32_t *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 1 << 22) != hipSuccess)
        return 1;
    (void)hipMemset(buf, 1, bytes);
    const int cus = p.multiProcessorCount;
    {
        const int64_t seg_vec = 262144 / 16, n_seg = bytes / 262144;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(strip_kernel, dim3((unsigned)n_seg), dim3(256), 0, 0, buf, seg_vec, n_seg, out);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best)
                best = ms;
        }
        printf("%-44s segment   256 KB: %7.3f ms  %6.0f GB/s\n", "nt, four column strips of 128 bytes", best, bytes / best / 1e6);
    }
    run_points<false, false>("points, 3 x 16 B per thread, plain", buf, bytes, out);
    run_points<false, true>("points, 3 x 16 B per thread, nt", buf, bytes, out);
    run_points<true, false>("points, 12 B per lane, plain", buf, bytes, out);
    run_points<true, true>("points, 12 B per lane, nt", buf, bytes, out);
    run_point16<false>("points, 16 B per lane at 12 B stride, plain", buf, bytes, out);
    run_point16<true>("points, 16 B per lane at 12 B stride, nt", buf, bytes, out);
    run_glds<false, 3072, 3>("to LDS, plain", buf, bytes, out, cus);
    run_glds<true, 3072, 3>("to LDS, nt", buf, bytes, out, cus);
    run_glds<true, 1536, 4>("to LDS, nt", buf, bytes, out, cus);
    run_glds<false, 1536, 4>("to LDS, plain", buf, bytes, out, cus);
    run_glds<true, 1536, 6>("to LDS, nt", buf, bytes, out, cus);
    run_glds<true, 768, 8>("to LDS, nt", buf, bytes, out, cus);
    run_band<false>("nt, eight bands of 64 rows per block", buf, bytes, out);
    run_band<true>("nt, four bands of 128 rows, halves on adjacent rows", buf, bytes, out);
    for (int64_t seg : {65536ll, 262144ll, 1048576ll}) {
        run<4, false, false>("plain, 4 in flight, block per segment", buf, bytes, seg, out, cus);
        run<4, true, false>("nt,    4 in flight, block per segment", buf, bytes, seg, out, cus);
        run<8, true, false>("nt,    8 in flight, block per segment", buf, bytes, seg, out, cus);
        run<16, true, false>("nt,   16 in flight, block per segment", buf, bytes, seg, out, cus);
        run<4, true, true>("nt,    4 in flight, resident grid", buf, bytes, seg, out, cus);
        run<8, true, true>("nt,    8 in flight, resident grid", buf, bytes, seg, out, cus);
        run<8, false, true>("plain, 8 in flight, resident grid", buf, bytes, seg, out, cus);
    }
    return 0;
}

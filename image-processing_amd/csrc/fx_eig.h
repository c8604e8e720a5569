// Symmetric 3x3 eigen-decomposition and the descriptor head shared by the
// point-set kernel and the contour kernel (device code, float64, no FMA
// contraction: the translation unit is built with -ffp-contract=off and uses
// fma() only where stated).
#pragma once

#include "fx_common.h"

#ifdef __HIPCC__

// One Jacobi rotation that zeroes a_pq; r is the third index.  app/aqq/apq are
// the 2x2 block, arp/arq the couplings to the third row, V the accumulated
// eigenvector matrix (columns p, q are rotated).
__device__ __forceinline__ void fx_jacobi_rotate(double &app, double &aqq, double &apq, double &arp,
                                                 double &arq, double (&V)[3][3], int p, int q)
{
    if (apq == 0.0)
        return;
    // t = tan(rotation angle), the root of t^2 + 2*tau*t - 1 = 0 of smaller
    // magnitude with tau = (aqq - app) / (2 apq), written without forming tau:
    // t = sgn(d) * 2 apq / (|d| + sqrt(d^2 + 4 apq^2)).
    double d = aqq - app;
    double two = apq + apq;
    double h = sqrt(d * d + two * two);
    double t = two / (fabs(d) + h);
    if (d < 0.0)
        t = -t;
    double c = 1.0 / sqrt(t * t + 1.0);
    double s = t * c;
    double tap = t * apq;
    app = app - tap;
    aqq = aqq + tap;
    apq = 0.0;
    double np_ = c * arp - s * arq;
    double nq_ = s * arp + c * arq;
    arp = np_;
    arq = nq_;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double vp = V[k][p], vq = V[k][q];
        V[k][p] = c * vp - s * vq;
        V[k][q] = s * vp + c * vq;
    }
}

// One Newton step on the hardware estimates (about 26 bits): full double
// precision up to an ulp or two.
__device__ __forceinline__ double fx_rcp_n1(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return fma(r, fma(-x, r, 1.0), r);
}
__device__ __forceinline__ double fx_rsq_n1(double x)
{
    const double r = __builtin_amdgcn_rsq(x);
    return fma(r * 0.5, fma(-x * r, r, 1.0), r);
}

__device__ __forceinline__ void fx_jacobi_rotate_n1(double &app, double &aqq, double &apq, double &arp,
                                                    double &arq, double (&V)[3][3], int p, int q)
{
    if (apq == 0.0)
        return;
    const double d = aqq - app;
    const double two = apq + apq;
    const double h2 = fma(d, d, two * two);
    const double h = h2 * fx_rsq_n1(h2);
    double t = two * fx_rcp_n1(fabs(d) + h);
    if (d < 0.0)
        t = -t;
    const double c = fx_rsq_n1(fma(t, t, 1.0));  // c^2 + s^2 = 1 within 1e-15
    const double s = t * c;
    const double tap = t * apq;
    app = app - tap;
    aqq = aqq + tap;
    apq = 0.0;
    const double np_ = c * arp - s * arq;
    const double nq_ = s * arp + c * arq;
    arp = np_;
    arq = nq_;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double vp = V[k][p], vq = V[k][q];
        V[k][p] = c * vp - s * vq;
        V[k][q] = s * vp + c * vq;
    }
}

// Eigenvalues (descending) and the eigenvector of the largest one of the
// symmetric matrix [[a00 a01 a02][a01 a11 a12][a02 a12 a22]] by cyclic Jacobi
// sweeps.  A coordinate that is exactly uncoupled (its off-diagonal entries are
// 0.0, e.g. z of a planar set) keeps its diagonal entry unchanged, so a planar
// set gets an exact 0.0 eigenvalue as LAPACK's eigh returns for it.
// FAST: reciprocals from the hardware estimate plus one Newton step in the
// rotations (a rotation only has to be orthogonal to rounding level and to
// shrink a_pq; the last bits of the angle do not matter to the converged
// result) and the matrix scaled to
// unit magnitude first (so that squares cannot leave the double range); used
// where the solve sits on the critical path of a streaming kernel.
template <bool FAST = false>
__device__ __forceinline__ void fx_eig3(double a00, double a01, double a02, double a11, double a12,
                                        double a22, double (&lam)[3], double (&v1)[3])
{
    double V[3][3] = {{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}};
    int scale_exp = 0;
    if (FAST) {
        const double big = fmax(fmax(fabs(a00), fabs(a11)), fabs(a22));
        if (big > 0.0 && big < 1.7e308) {
            (void)frexp(big, &scale_exp);
            a00 = ldexp(a00, -scale_exp);
            a01 = ldexp(a01, -scale_exp);
            a02 = ldexp(a02, -scale_exp);
            a11 = ldexp(a11, -scale_exp);
            a12 = ldexp(a12, -scale_exp);
            a22 = ldexp(a22, -scale_exp);
        }
    }
    for (int sweep = 0; sweep < 12; sweep++) {
        double off = fabs(a01) + fabs(a02) + fabs(a12);
        double diag = fabs(a00) + fabs(a11) + fabs(a22);
        // FAST stops one sweep earlier: off-diagonal 1e-13 leaves eigenvalue errors of 1e-26 and
        // eigenvector errors of 1e-13 / gap, far inside every tolerance of the descriptor
        if (off == 0.0 || off <= (FAST ? 1e-13 : 1e-19) * diag)
            break;
        if (FAST) {
            fx_jacobi_rotate_n1(a00, a11, a01, a02, a12, V, 0, 1);
            fx_jacobi_rotate_n1(a00, a22, a02, a01, a12, V, 0, 2);
            fx_jacobi_rotate_n1(a11, a22, a12, a01, a02, V, 1, 2);
        } else {
            fx_jacobi_rotate(a00, a11, a01, a02, a12, V, 0, 1);
            fx_jacobi_rotate(a00, a22, a02, a01, a12, V, 0, 2);
            fx_jacobi_rotate(a11, a22, a12, a01, a02, V, 1, 2);
        }
    }
    if (FAST && scale_exp != 0) {
        a00 = ldexp(a00, scale_exp);
        a11 = ldexp(a11, scale_exp);
        a22 = ldexp(a22, scale_exp);
    }
    double w0 = a00, w1 = a11, w2 = a22;
    int i0 = 0, i1 = 1, i2 = 2;
    // descending order of (w, column) pairs; no run-time array indexing
#define FX_CSWAP(wa, wb, ia, ib) \
    if (wa < wb) {               \
        double tw = wa;          \
        wa = wb;                 \
        wb = tw;                 \
        int ti = ia;             \
        ia = ib;                 \
        ib = ti;                 \
    }
    FX_CSWAP(w0, w1, i0, i1)
    FX_CSWAP(w1, w2, i1, i2)
    FX_CSWAP(w0, w1, i0, i1)
#undef FX_CSWAP
    (void)i1;
    (void)i2;
    lam[0] = w0;
    lam[1] = w1;
    lam[2] = w2;
#pragma unroll
    for (int k = 0; k < 3; k++)
        v1[k] = (i0 == 0) ? V[k][0] : ((i0 == 1) ? V[k][1] : V[k][2]);
}

// The nine scalars in front of the histogram (reference
// src/features/extract_3d_pca_features.py:97-113,136-141): negative
// eigenvalues clipped to 0, r_i = lambda_i / max(trace, 1e-12) (the reference
// tests `trace > 1e-12`), ratio12, ratio23 with a 1e-12 guard, trace.
__device__ __forceinline__ void fx_pca_head(double (&lam)[3], double (&head)[9])
{
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (lam[k] < 0.0)
            lam[k] = 0.0;
    double trace = lam[0] + lam[1] + lam[2];
    double safe = trace > 1e-12 ? trace : 1e-12;
    head[0] = lam[0];
    head[1] = lam[1];
    head[2] = lam[2];
    head[3] = lam[0] / safe;
    head[4] = lam[1] / safe;
    head[5] = lam[2] / safe;
    head[6] = lam[0] / (lam[1] + 1e-12);
    head[7] = lam[1] / (lam[2] + 1e-12);
    head[8] = trace;
}

// Bin of value v for np.histogram(proj, bins=np.linspace(0, top, n_bins+1)):
// edges e_i = i*step (i < n_bins), e_n = top; bin i holds e_i <= v < e_{i+1},
// the last bin is closed on the right.
__device__ __forceinline__ int fx_hist_bin(double v, double step, double inv_step, int n_bins)
{
    int k = (int)(v * inv_step);
    if (k > n_bins - 1)
        k = n_bins - 1;
    if (k < 0)
        k = 0;
    while (k > 0 && v < (double)k * step)
        k--;
    while (k < n_bins - 1 && v >= (double)(k + 1) * step)
        k++;
    return k;
}

#endif  // __HIPCC__

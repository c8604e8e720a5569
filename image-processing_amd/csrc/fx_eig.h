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
// Two Newton steps: the reciprocal to within an ulp, as the hardware division sequence forms it.
__device__ __forceinline__ double fx_rcp_n2(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    return fma(r, fma(-x, r, 1.0), r);
}
// n / d from r = fx_rcp_n2(d): quotient estimate, exact residual, one correction -- the steps of the
// compiler's IEEE division without the scaling and the fix-up that only denormal, infinite or zero
// operands need.  Correctly rounded for ordinary operands; in particular exact whenever n / d is
// representable.
__device__ __forceinline__ double fx_div_r(double n, double d, double r)
{
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
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

// First principal axis of the symmetric 2x2 matrix [[a00 a01][a01 a11]]: the one Jacobi rotation that
// diagonalises it (what fx_eig3 does to [[a00 a01 0][a01 a11 0][0 0 0]]: the third coordinate never
// couples), with the reciprocals from the hardware estimates plus a Newton step -- the axis is good to
// a few ulp, as far from LAPACK's as LAPACK's is from the exact one.
__device__ __forceinline__ void fx_eig2_axis(double a00, double a01, double a11, double &vx, double &vy)
{
    double c = 1.0, s = 0.0;
    const double off = fabs(a01), diag = fabs(a00) + fabs(a11);
    if (!(off == 0.0 || off <= 1e-19 * diag)) {
        const double d = a11 - a00;
        const double two = a01 + a01;
        const double h2 = fma(d, d, two * two);
        const double h = h2 * fx_rsq_n1(h2);
        double t = two * fx_rcp_n1(fabs(d) + h);
        if (d < 0.0)
            t = -t;
        c = fx_rsq_n1(fma(t, t, 1.0));
        s = t * c;
        const double tap = t * a01;
        a00 = a00 - tap;
        a11 = a11 + tap;
    }
    // columns of the rotation: (c, -s) belongs to a00, (s, c) to a11; the zero eigenvalue of the third
    // coordinate wins only over two negative ones
    const bool second = a00 < a11;
    const double w = second ? a11 : a00;
    vx = second ? s : c;
    vy = second ? c : -s;
    if (w < 0.0)
        vx = vy = 0.0;
}

// Direct solve for well-conditioned matrices (three clearly separated,
// clearly positive eigenvalues -- what a cloud of points in general position
// gives), a few hundred instructions without a loop over sweeps:
//   * the largest eigenvalue from the characteristic polynomial of the
//     trace-free, scaled matrix B = (A - q I) / p: lambda1 = q + 2 p c with c the
//     root of 4 c^3 - 3 c = det(B) / 2 in [1/2, 1] (c = cos(acos(.) / 3)), by
//     Newton steps from 1/2 + sqrt((1 + r) / 2) / 2;
//   * its eigenvector as the largest cross product of two rows of A - lambda1 I
//     (error about eps * lambda1 / (lambda1 - lambda2));
//   * the other two eigenvalues from the 2 x 2 matrix A induces on the plane
//     orthogonal to that vector (closed form, the smaller one as product over
//     the larger one: absolute error about eps * lambda1, where the roots of the
//     cubic themselves lose half the digits when lambda2 and lambda3 are close),
//     and lambda1 again as the Rayleigh quotient of the vector.
// Returns false, with nothing usable in lam / v1, when the matrix is not of that
// kind (lambda1 - lambda2 < 1e-3 lambda1, lambda3 < 1e-5 lambda1, a zero or
// non-finite matrix): the caller then takes fx_eig3, whose sweeps keep exact
// zeros and small eigenvalues to full relative accuracy.
__device__ __forceinline__ bool fx_eig3_direct(double a00, double a01, double a02, double a11, double a12, double a22,
                                               double (&lam)[3], double (&v1)[3])
{
    const double q = ((a00 + a11) + a22) * (1.0 / 3.0);
    const double b00 = a00 - q, b11 = a11 - q, b22 = a22 - q;
    const double p1 = fma(a01, a01, fma(a02, a02, a12 * a12));
    const double p2 = fma(b00, b00, fma(b11, b11, fma(b22, b22, p1 + p1)));
    if (!(p2 > 1e-290) || !(p2 < 1e290))
        return false;
    const double s6 = p2 * (1.0 / 6.0);
    const double ip = fx_rsq_n1(s6);   // 1 / p
    const double pp = s6 * ip;         // p
    const double c00 = b00 * ip, c11 = b11 * ip, c22 = b22 * ip, c01 = a01 * ip, c02 = a02 * ip, c12 = a12 * ip;
    double r = 0.5 * fma(c00, fma(c11, c22, -c12 * c12), fma(-c01, fma(c01, c22, -c12 * c02), c02 * fma(c01, c12, -c11 * c02)));
    r = fmin(1.0, fmax(-1.0, r));
    const double g = fma(0.5, r, 0.5);
    double c = fma(0.5, g > 1e-300 ? g * fx_rsq_n1(g) : 0.0, 0.5);
#pragma unroll
    for (int it = 0; it < 5; it++) {
        const double f = fma(fma(4.0 * c, c, -3.0), c, -r), fp = fma(12.0 * c, c, -3.0);
        if (fp > 1e-30)
            c = fma(-f, fx_rcp_n1(fp), c);
    }
    const double l1 = fma(pp + pp, c, q);
    // rows of A - l1 I and their cross products
    const double d0 = a00 - l1, d1 = a11 - l1, d2 = a22 - l1;
    const double x0 = fma(a01, a12, -a02 * d1), y0 = fma(a02, a01, -d0 * a12), z0 = fma(d0, d1, -a01 * a01);    // r0 x r1
    const double x1 = fma(a01, d2, -a02 * a12), y1 = fma(a02, a02, -d0 * d2), z1 = fma(d0, a12, -a01 * a02);    // r0 x r2
    const double x2 = fma(d1, d2, -a12 * a12), y2 = fma(a12, a02, -a01 * d2), z2 = fma(a01, a12, -d1 * a02);    // r1 x r2
    const double n0 = fma(x0, x0, fma(y0, y0, z0 * z0)), n1 = fma(x1, x1, fma(y1, y1, z1 * z1)),
                 n2 = fma(x2, x2, fma(y2, y2, z2 * z2));
    double vx = x0, vy = y0, vz = z0, nn = n0;
    if (n1 > nn) {
        vx = x1, vy = y1, vz = z1, nn = n1;
    }
    if (n2 > nn) {
        vx = x2, vy = y2, vz = z2, nn = n2;
    }
    if (!(nn > 1e-290) || !(nn < 1e290))
        return false;
    const double in = fx_rsq_n1(nn);
    vx *= in, vy *= in, vz *= in;
    // an orthonormal pair (u, w) in the plane orthogonal to v: u = v x e for the axis e on which v is smallest
    const double ax = fabs(vx), ay = fabs(vy), az = fabs(vz);
    double ux, uy, uz;
    if (ax <= ay && ax <= az) {
        ux = 0.0, uy = vz, uz = -vy;
    } else if (ay <= az) {
        ux = -vz, uy = 0.0, uz = vx;
    } else {
        ux = vy, uy = -vx, uz = 0.0;
    }
    const double iu = fx_rsq_n1(fma(ux, ux, fma(uy, uy, uz * uz)));   // |u|^2 >= 2/3
    ux *= iu, uy *= iu, uz *= iu;
    const double wx = fma(vy, uz, -vz * uy), wy = fma(vz, ux, -vx * uz), wz = fma(vx, uy, -vy * ux);
    const double Aux = fma(a00, ux, fma(a01, uy, a02 * uz)), Auy = fma(a01, ux, fma(a11, uy, a12 * uz)),
                 Auz = fma(a02, ux, fma(a12, uy, a22 * uz));
    const double Awx = fma(a00, wx, fma(a01, wy, a02 * wz)), Awy = fma(a01, wx, fma(a11, wy, a12 * wz)),
                 Awz = fma(a02, wx, fma(a12, wy, a22 * wz));
    const double Avx = fma(a00, vx, fma(a01, vy, a02 * vz)), Avy = fma(a01, vx, fma(a11, vy, a12 * vz)),
                 Avz = fma(a02, vx, fma(a12, vy, a22 * vz));
    const double m00 = fma(ux, Aux, fma(uy, Auy, uz * Auz)), m01 = fma(ux, Awx, fma(uy, Awy, uz * Awz)),
                 m11 = fma(wx, Awx, fma(wy, Awy, wz * Awz));
    const double l1r = fma(vx, Avx, fma(vy, Avy, vz * Avz));
    const double h = 0.5 * (m00 - m11), mid = 0.5 * (m00 + m11);
    const double rad2 = fma(h, h, m01 * m01);
    const double l2 = mid + (rad2 > 1e-300 ? rad2 * fx_rsq_n1(rad2) : 0.0);
    if (!(l2 > 0.0))
        return false;
    const double l3 = fma(m00, m11, -m01 * m01) * fx_rcp_n1(l2);
    if (!(l1r - l2 >= 1e-3 * l1r) || !(l3 >= 1e-5 * l1r) || !(l1r < 1e300))
        return false;
    lam[0] = l1r, lam[1] = l2, lam[2] = l3;
    v1[0] = vx, v1[1] = vy, v1[2] = vz;
    return true;
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

// The same bin from a first guess that is at most one off (v * inv_step with inv_step within 1e-6 of
// n_bins / top): one step down or up instead of the loops.
__device__ __forceinline__ int fx_hist_bin_near(double v, double step, double inv_step, int n_bins)
{
    int k = (int)(v * inv_step);
    k = k > n_bins - 1 ? n_bins - 1 : k;
    k = k < 0 ? 0 : k;
    const bool down = k > 0 && v < (double)k * step;
    const bool up = k < n_bins - 1 && v >= (double)(k + 1) * step;
    return down ? k - 1 : up ? k + 1 : k;
}

#endif  // __HIPCC__

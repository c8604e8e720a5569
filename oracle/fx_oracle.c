/*
 * fx_oracle.c -- CPU restatement of the reference's Hu-moment path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the HIP kernels are
 * compared against.  Nothing in the product package may call it; only tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py load it.
 *
 * What it restates
 * ----------------
 * reference src/features/extract_2d_hu.py:443-478 (compute_hu_from_image):
 *     moments = cv2.moments(img.astype(np.uint8))      (:449)
 *     hu      = cv2.HuMoments(moments)                 (:450)
 *     val     = -sign(h) * log10(|h| + 1e-300)         (:471-478)
 * reference src/features/extract_2d_hu.py:641-644: per-sample mean / std
 * (ddof=0) over the per-view rows in float32.
 * reference src/features/extract_2d_hu.py:163,258: (mask > 127) * 255
 * binarisation that precedes the moment call on the silhouette path.
 *
 * The arithmetic of cv2.moments / cv2.HuMoments lives in OpenCV
 * (opencv-python, version NOT pinned by the reference: its requirements.txt
 * lists only tabulate).  OpenCV is absent from this image, so its published
 * algorithm (modules/imgproc/src/moments.cpp: momentsInTile<uchar,int,int>
 * over 32x32 tiles accumulated into doubles, completeMomentState, HuMoments)
 * is restated here.  PARITY PIN: the reference holds no test vectors for this
 * function; the restatement is pinned by (a) agreement with the textbook
 * definition of central moments, (b) the 21 silhouette panels and the feature
 * table the reference committed (tests/golden).  Per-view last-digit parity
 * with a particular OpenCV build is therefore "parity unpinned".
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 * FMA contraction must stay off: OpenCV's x86-64 baseline build evaluates the
 * finalisation in plain IEEE double multiply/add order.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FXO_TILE 32

/* Binarisation fused in front of the moment sum: thresh < 0 keeps the pixel
 * value as the weight (cv2.moments binaryImage=False on a gray image);
 * thresh >= 0 maps v > thresh to on_value and everything else to 0, the
 * reference's `(mask > 127).astype(np.uint8) * 255`. */
static inline int fxo_weight(uint8_t v, int thresh, int on_value)
{
    if (thresh < 0)
        return (int)v;
    return ((int)v > thresh) ? on_value : 0;
}

/* completeMomentState: central (mu) and normalised (nu) moments from the ten
 * raw ones.  m[0..9] = m00 m10 m01 m20 m11 m02 m30 m21 m12 m03,
 * m[10..16] = mu20 mu11 mu02 mu30 mu21 mu12 mu03,
 * m[17..23] = nu20 nu11 nu02 nu30 nu21 nu12 nu03. */
void fxo_complete_moments(double m[24])
{
    const double m00 = m[0], m10 = m[1], m01 = m[2], m20 = m[3], m11 = m[4];
    const double m02 = m[5], m30 = m[6], m21 = m[7], m12 = m[8], m03 = m[9];
    double cx = 0.0, cy = 0.0, inv_m00 = 0.0;

    if (fabs(m00) > DBL_EPSILON) {
        inv_m00 = 1.0 / m00;
        cx = m10 * inv_m00;
        cy = m01 * inv_m00;
    }

    double mu20 = m20 - m10 * cx;
    double mu11 = m11 - m10 * cy;
    double mu02 = m02 - m01 * cy;
    m[10] = mu20;
    m[11] = mu11;
    m[12] = mu02;

    m[13] = m30 - cx * (3 * mu20 + cx * m10);
    mu11 += mu11;
    m[14] = m21 - cx * (mu11 + cx * m01) - cy * mu20;
    m[15] = m12 - cy * (mu11 + cy * m10) - cx * mu02;
    m[16] = m03 - cy * (3 * mu02 + cy * m01);

    double inv_sqrt_m00 = sqrt(fabs(inv_m00));
    double s2 = inv_m00 * inv_m00, s3 = s2 * inv_sqrt_m00;

    m[17] = m[10] * s2;
    m[18] = m[11] * s2;
    m[19] = m[12] * s2;
    m[20] = m[13] * s3;
    m[21] = m[14] * s3;
    m[22] = m[15] * s3;
    m[23] = m[16] * s3;
}

/* cv2.moments on an 8-bit single-channel image, OpenCV evaluation order:
 * exact int32 sums inside every 32x32 tile, tile results shifted to image
 * coordinates and accumulated in double, tiles visited row-major. */
void fxo_moments_u8(const uint8_t *img, int H, int W, int64_t pitch,
                    int thresh, int on_value, double out[24])
{
    double M[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    for (int ty = 0; ty < H; ty += FXO_TILE) {
        int th = (H - ty < FXO_TILE) ? (H - ty) : FXO_TILE;
        for (int tx = 0; tx < W; tx += FXO_TILE) {
            int tw = (W - tx < FXO_TILE) ? (W - tx) : FXO_TILE;
            int mom[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

            for (int y = 0; y < th; y++) {
                const uint8_t *row = img + (int64_t)(ty + y) * pitch + tx;
                int x0 = 0, x1 = 0, x2 = 0, x3 = 0;
                for (int x = 0; x < tw; x++) {
                    int p = fxo_weight(row[x], thresh, on_value);
                    int xp = x * p, xxp = xp * x;
                    x0 += p;
                    x1 += xp;
                    x2 += xxp;
                    x3 += xxp * x;
                }
                int py = y * x0, sy = y * y;
                mom[9] += py * sy;
                mom[8] += x1 * sy;
                mom[7] += x2 * y;
                mom[6] += x3;
                mom[5] += x0 * sy;
                mom[4] += x1 * y;
                mom[3] += x2;
                mom[2] += py;
                mom[1] += x1;
                mom[0] += x0;
            }

            double t[10];
            for (int k = 0; k < 10; k++)
                t[k] = (double)mom[k];
            double x = (double)tx, y = (double)ty;
            double xm = x * t[0], ym = y * t[0];

            M[0] += t[0];
            M[1] += t[1] + xm;
            M[2] += t[2] + ym;
            M[3] += t[3] + x * (t[1] * 2 + xm);
            M[4] += t[4] + x * (t[2] + ym) + y * t[1];
            M[5] += t[5] + y * (t[2] * 2 + ym);
            M[6] += t[6] + x * (3. * t[3] + x * (3. * t[1] + xm));
            M[7] += t[7] + x * (2 * (t[4] + y * t[1]) + x * (t[2] + ym)) + y * t[3];
            M[8] += t[8] + y * (2 * (t[4] + x * t[2]) + y * (t[1] + xm)) + x * t[5];
            M[9] += t[9] + y * (3. * t[5] + y * (3. * t[2] + ym));
        }
    }

    memset(out, 0, 24 * sizeof(double));
    for (int k = 0; k < 10; k++)
        out[k] = M[k];
    fxo_complete_moments(out);
}

/* Round an unsigned 128-bit integer to the nearest double (ties to even). */
static double fxo_u128_to_double(unsigned __int128 v)
{
    if (v == 0)
        return 0.0;
    uint64_t hi = (uint64_t)(v >> 64), lo = (uint64_t)v;
    int top = hi ? (127 - __builtin_clzll(hi)) : (63 - __builtin_clzll(lo));
    if (top <= 52)
        return (double)lo;
    int shift = top - 52;
    unsigned __int128 keep = v >> shift;
    unsigned __int128 rem = v & ((((unsigned __int128)1) << shift) - 1);
    unsigned __int128 half = ((unsigned __int128)1) << (shift - 1);
    if (rem > half || (rem == half && (keep & 1)))
        keep += 1;
    return ldexp((double)(uint64_t)keep, shift);
}

/* The same ten raw moments as exact integers (128-bit), each rounded once to
 * double.  Equals fxo_moments_u8 bit for bit while every partial result stays
 * below 2^53 (any image up to 512x512 at weight 255); for larger images
 * OpenCV's double accumulation rounds per tile and this is the value it
 * approximates. */
void fxo_moments_exact_u8(const uint8_t *img, int H, int W, int64_t pitch,
                          int thresh, int on_value, double out[24])
{
    unsigned __int128 M[10];
    for (int k = 0; k < 10; k++)
        M[k] = 0;

    for (int y = 0; y < H; y++) {
        const uint8_t *row = img + (int64_t)y * pitch;
        uint64_t r0 = 0, r1 = 0, r2 = 0;
        unsigned __int128 r3 = 0;
        for (int x = 0; x < W; x++) {
            uint64_t p = (uint64_t)fxo_weight(row[x], thresh, on_value);
            uint64_t xp = (uint64_t)x * p;
            r0 += p;
            r1 += xp;
            r2 += xp * (uint64_t)x;
            r3 += (unsigned __int128)(xp * (uint64_t)x) * (uint64_t)x;
        }
        unsigned __int128 yy = (unsigned __int128)y;
        M[0] += r0;
        M[1] += r1;
        M[2] += yy * r0;
        M[3] += r2;
        M[4] += yy * r1;
        M[5] += yy * yy * r0;
        M[6] += r3;
        M[7] += yy * r2;
        M[8] += yy * yy * r1;
        M[9] += yy * yy * yy * r0;
    }

    memset(out, 0, 24 * sizeof(double));
    for (int k = 0; k < 10; k++)
        out[k] = fxo_u128_to_double(M[k]);
    fxo_complete_moments(out);
}

/* cv2.HuMoments: seven invariants from nu20 nu11 nu02 nu30 nu21 nu12 nu03. */
void fxo_hu(const double m[24], double hu[7])
{
    const double nu20 = m[17], nu11 = m[18], nu02 = m[19];
    const double nu30 = m[20], nu21 = m[21], nu12 = m[22], nu03 = m[23];

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
}

/* reference extract_2d_hu.py:471-478: -sign(h)*log10(|h|+1e-300), inf/nan
 * mapped to 0.  h == 0 gives 0 (sign is 0). */
void fxo_hu_log(const double hu[7], double out[7])
{
    for (int i = 0; i < 7; i++) {
        double h = hu[i];
        double sgn = (h > 0) - (h < 0);
        double v = -sgn * log10(fabs(h) + 1e-300);
        if (isinf(v) || isnan(v))
            v = 0.0;
        if (sgn == 0)
            v = 0.0;
        out[i] = v;
    }
}

/* compute_hu_from_image for a batch of equally sized images.  exact != 0
 * selects the 128-bit integer raw moments instead of OpenCV's tile order.
 * Any output pointer may be NULL. */
void fxo_hu_batch_u8(const uint8_t *img, int64_t B, int H, int W, int64_t pitch,
                     int64_t img_stride, int thresh, int on_value, int exact,
                     double *moments24, double *hu7, double *hulog7_f64,
                     float *hulog7)
{
    for (int64_t b = 0; b < B; b++) {
        double m[24], hu[7], lg[7];
        const uint8_t *p = img + b * img_stride;
        if (exact)
            fxo_moments_exact_u8(p, H, W, pitch, thresh, on_value, m);
        else
            fxo_moments_u8(p, H, W, pitch, thresh, on_value, m);
        fxo_hu(m, hu);
        fxo_hu_log(hu, lg);
        if (moments24)
            memcpy(moments24 + b * 24, m, sizeof(m));
        if (hu7)
            memcpy(hu7 + b * 7, hu, sizeof(hu));
        if (hulog7_f64)
            memcpy(hulog7_f64 + b * 7, lg, sizeof(lg));
        if (hulog7)
            for (int i = 0; i < 7; i++)
                hulog7[b * 7 + i] = (float)lg[i];
    }
}

/* reference extract_2d_hu.py:641-644 in NumPy's float32 evaluation order:
 * mean = (sequential f32 sum over views) / V; std = sqrt(sum((x-mean)^2)/V),
 * every step rounded to float32.  offsets has G+1 entries (rows of group g are
 * offsets[g] .. offsets[g+1]-1); out is G x 14 = mean[7] || std[7].  An empty
 * group yields NaN like np.mean of an empty array. */
void fxo_hu_groupstat(const float *hulog7, const int64_t *offsets, int64_t G,
                      float *out14)
{
    for (int64_t g = 0; g < G; g++) {
        int64_t a = offsets[g], b = offsets[g + 1];
        float cnt = (float)(b - a);
        for (int j = 0; j < 7; j++) {
            float s = 0.0f;
            for (int64_t r = a; r < b; r++)
                s = s + hulog7[r * 7 + j];
            float mean = s / cnt;
            float q = 0.0f;
            for (int64_t r = a; r < b; r++) {
                float d = hulog7[r * 7 + j] - mean;
                float dd = d * d;
                q = q + dd;
            }
            float var = q / cnt;
            out14[g * 14 + j] = mean;
            out14[g * 14 + 7 + j] = sqrtf(var);
        }
    }
}

/* Contour pixels of the binarised image: foreground pixels (v > thresh, or
 * v > 127 when thresh < 0) with at least one 4-neighbour that is background
 * or outside the image.  Row-major order; writes (x, y) pairs.  Returns the
 * number found (may exceed cap; only the first cap are written).  This point
 * set is a build-defined input of the PCA descriptor (the reference has no
 * 2-D contour extractor); the descriptor itself is
 * extract_3d_pca_features.py:59-145 applied to (x, y, 0). */
int64_t fxo_contour_points_u8(const uint8_t *img, int H, int W, int64_t pitch,
                              int thresh, int32_t *xy, int64_t cap)
{
    int t = thresh < 0 ? 127 : thresh;
    int64_t n = 0;
    for (int y = 0; y < H; y++) {
        const uint8_t *row = img + (int64_t)y * pitch;
        for (int x = 0; x < W; x++) {
            if ((int)row[x] <= t)
                continue;
            int up = (y > 0) && ((int)row[x - pitch] > t);
            int dn = (y < H - 1) && ((int)row[x + pitch] > t);
            int lf = (x > 0) && ((int)row[x - 1] > t);
            int rt = (x < W - 1) && ((int)row[x + 1] > t);
            if (up && dn && lf && rt)
                continue;
            if (n < cap) {
                xy[2 * n] = x;
                xy[2 * n + 1] = y;
            }
            n++;
        }
    }
    return n;
}

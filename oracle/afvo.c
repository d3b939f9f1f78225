/*
 * afvo.c — CPU ORACLE (test infrastructure; see afvo.h header: PARITY UNPINNED).
 * Build: gcc -O3 -march=native -ffp-contract=off -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off matters: every float expression below must round exactly once per operator so
 * that the HIP kernels (also built with -ffp-contract=off) reproduce it bit for bit.
 */
#include "afvo.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef uint8_t u8;

/* cvRound(float/double): SSE cvtss2si => round-half-to-even in the default rounding mode */
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------------------
 * Tables / geometry
 * ---------------------------------------------------------------------------------------------- */

/* cv::ORB pyramid geometry (orb.cpp detectAndCompute; called from Feature_orb32.cpp:34,48):
 *   scale_l = (float)pow((double)scaleFactor, l); size_l = cvRound(cols * (1.f/scale_l)).
 * cv::ORB stores scaleFactor as double, created from the float 1.2f default (Feature_orb32.cpp:21). */
void afvo_level_geometry(int w, int h, int nlevels, float scale_factor, int *lw, int *lh, float *lscale) {
    for (int l = 0; l < nlevels; ++l) {
        float scale = (float)pow((double)scale_factor, (double)l);
        float inv = 1.0f / scale;
        lscale[l] = scale;
        lw[l] = cv_round_f((float)w * inv);
        lh[l] = cv_round_f((float)h * inv);
    }
}

/* FeatureExtractor.cpp:97-108 — quadtree quotas mnFeaturesPerLevel */
void afvo_quotas_extractor(int nfeatures, int nlevels, float scale_factor, int *q) {
    float factor = 1.0f / scale_factor;
    float desired = (float)nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) {
        q[l] = cv_round_f(desired);
        sum += q[l];
        desired *= factor;
    }
    q[nlevels - 1] = imax(nfeatures - sum, 0);
}

/* cv::ORB computeKeyPoints quotas, nfeatures = 10 * extractor nfeatures (Feature_orb32.cpp:22) */
void afvo_quotas_cvorb(int nfeatures, int nlevels, float scale_factor, int *q) {
    double sf = (double)scale_factor;
    float factor = (float)(1.0 / sf);
    float desired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) {
        q[l] = cv_round_f(desired);
        sum += q[l];
        desired *= factor;
    }
    q[nlevels - 1] = imax(nfeatures - sum, 0);
}

/* ORBextractor.cc:124-139 (same code as cv::ORB): circular patch row ends, half patch 15 */
void afvo_umax(int *umax) {
    const int hp = 15;
    int v, v0, vmax = (int)floor(hp * sqrt(2.f) / 2 + 1);
    int vmin = (int)ceil(hp * sqrt(2.f) / 2);
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt((double)hp * hp - (double)v * v));
    for (v = hp, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

/* GaussianBlur(7x7, sigma 2) taps as used by the 8U separable filter: float kernel * 256 rounded
 * (createSeparableLinearFilter: _rowKernel.convertTo(CV_32S, 1<<8)) */
void afvo_gauss7_taps(int *taps) {
    double g[7], sum = 0;
    for (int i = 0; i < 7; ++i) {
        double x = i - 3;
        g[i] = exp(-0.5 * x * x / (2.0 * 2.0));
        sum += g[i];
    }
    for (int i = 0; i < 7; ++i) taps[i] = cv_round_d((double)(float)(g[i] / sum) * 256.0);
}

static const int8_t k_brief_pattern[1024] = {
#include "brief_pattern.inc"
};
const int8_t *afvo_brief_pattern(void) { return k_brief_pattern; }

/* cv::fastAtan2 (OpenCV core/mathfuncs_core: atan_f32), degrees in [0,360) */
float afvo_fast_atan2(float y, float x) {
    const float rad2deg = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * rad2deg;
    const float p3 = -0.3258083974640975f * rad2deg;
    const float p5 = 0.1555786518463281f * rad2deg;
    const float p7 = -0.04432655554792128f * rad2deg;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* cos/sin of the keypoint angle as rBRIEF needs them (FeatureExtractor.h:181-183: angle*factorPI in
 * float, then (float)cos / (float)sin).  libm and the ROCm device library are not guaranteed to agree in
 * the last bit, so oracle and kernels share ONE explicit algorithm: Cody-Waite reduction by pi/2 and
 * degree-15/16 Taylor polynomials, all in IEEE double without contraction, rounded once to float.  The
 * result is the correctly rounded float cos/sin except for ~1e-9 of inputs. */
static void sincos_rad_f64(double t, double *c_out, double *s_out) {
    const double two_over_pi = 0.63661977236758138;
    const double pio2_hi = 1.5707963267341256e+00; /* 33 high bits of pi/2 */
    const double pio2_lo = 6.0771005065061922e-11; /* pi/2 - pio2_hi */
    double kd = floor(t * two_over_pi + 0.5);
    int k = (int)kd;
    double r = (t - kd * pio2_hi) - kd * pio2_lo;
    double z = r * r;
    double sp = 1.0 + z * (-1.6666666666666666e-01 + z * (8.3333333333333332e-03 + z * (-1.9841269841269841e-04 +
                z * (2.7557319223985893e-06 + z * (-2.5052108385441720e-08 + z * (1.6059043836821613e-10 +
                z * (-7.6471637318198164e-13)))))));
    double s = r * sp;
    double c = 1.0 + z * (-0.5 + z * (4.1666666666666664e-02 + z * (-1.3888888888888889e-03 + z * (2.4801587301587302e-05 +
               z * (-2.7557319223985888e-07 + z * (2.0876756987868100e-09 + z * (-1.1470745597729725e-11 +
               z * (4.7794773323873853e-14))))))));
    switch (k & 3) {
    case 0: *c_out = c; *s_out = s; break;
    case 1: *c_out = -s; *s_out = c; break;
    case 2: *c_out = -c; *s_out = -s; break;
    default: *c_out = s; *s_out = -c; break;
    }
}

void afvo_sincos_deg(float angle_deg, float *cos_out, float *sin_out) {
    const float factor_pi = (float)(3.1415926535897932384626433832795 / 180.f); /* FeatureExtractor.h:177 */
    float angle = angle_deg * factor_pi;
    double c, s;
    sincos_rad_f64((double)angle, &c, &s);
    *cos_out = (float)c;
    *sin_out = (float)s;
}

/* Feature_orb32.cpp:59-61 */
float afvo_keypoint_size(int octave, float scale_factor) { return powf(scale_factor, (float)octave); }

/* FeatureExtractor.cpp:132-172 with the settings of FeatureExtractor.cpp:52-55:
 * maxKeyPtSize0 = maxKeyPtSize = pow(1.2f, 7.0f), minKeyPtSize = 1.0f */
void afvo_size_sigma(const afvo_keypoint *kps, int n, float scale_factor, float *size, float *sigma2, float *inf) {
    const float scale_factor_orb = 1.2f;
    const float max_size0 = powf(scale_factor_orb, (float)(8 - 1.0));
    const float max_size = max_size0, min_size = 1.0f;
    for (int i = 0; i < n; ++i) {
        float s = afvo_keypoint_size(kps[i].octave, scale_factor);
        float norm = max_size;
        if (max_size > min_size) norm = 1.0f + (s - min_size) * (max_size0 - 1.0f) / (max_size - min_size);
        size[i] = norm;
        float s2 = norm * norm;
        sigma2[i] = s2;
        inf[i] = 1.0f / s2;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Image stages
 * ---------------------------------------------------------------------------------------------- */

/* resize(..., INTER_LINEAR_EXACT) for CV_8UC1 (OpenCV imgproc/resize.cpp resize_bitExact with
 * interpolation_linear<uchar>, ufixedpoint16 8.8 coefficients).  Per destination index d:
 *   f = (src/dst)*(d+0.5)-0.5 (IEEE double), i=floor(f); i<0 -> left edge; i>=src-1 -> right edge;
 *   c1 = cvRound((f-i)*256), c0 = 256-c1.  Horizontal pass keeps 8.8 fixed point, vertical pass
 *   rounds (+2^15)>>16. */
static void resize_coeffs(int src, int dst, int *ofs, int *c1) {
    double inv_scale = (double)dst / (double)src;
    double scale = 1.0 / inv_scale;
    for (int d = 0; d < dst; ++d) {
        double f = scale * ((double)d + 0.5) - 0.5;
        int i = (int)floor(f);
        if (i >= 0 && src > 1) {
            if (i < src - 1) {
                ofs[d] = i;
                c1[d] = cv_round_d((f - (double)i) * 256.0);
            } else {
                ofs[d] = src - 1;
                c1[d] = 0;
            }
        } else {
            ofs[d] = 0;
            c1[d] = 0;
        }
    }
}

void afvo_resize_linear_exact(const u8 *src, int sw, int sh, int sstride, u8 *dst, int dw, int dh, int dstride) {
    int *xo = (int *)malloc(sizeof(int) * (size_t)(2 * dw + 2 * dh));
    int *xc = xo + dw, *yo = xc + dw, *yc = yo + dh;
    resize_coeffs(sw, dw, xo, xc);
    resize_coeffs(sh, dh, yo, yc);
    uint16_t *h0 = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)dw * 2);
    uint16_t *h1 = h0 + dw;
    for (int y = 0; y < dh; ++y) {
        const u8 *r0 = src + (size_t)yo[y] * sstride;
        const u8 *r1 = src + (size_t)imin(yo[y] + 1, sh - 1) * sstride;
        for (int x = 0; x < dw; ++x) {
            int o = xo[x], o1 = imin(o + 1, sw - 1), c = xc[x];
            h0[x] = (uint16_t)((256 - c) * r0[o] + c * r0[o1]);
            h1[x] = (uint16_t)((256 - c) * r1[o] + c * r1[o1]);
        }
        int cy = yc[y];
        u8 *d = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; ++x) {
            uint32_t v = (uint32_t)h0[x] * (uint32_t)(256 - cy) + (uint32_t)h1[x] * (uint32_t)cy;
            d[x] = (u8)((v + 32768u) >> 16);
        }
    }
    free(h0);
    free(xo);
}

static inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * n - 2 - p;
    }
    return p;
}

/* copyMakeBorder(..., BORDER_REFLECT_101) */
void afvo_make_border101(const u8 *src, int w, int h, int sstride, u8 *dst, int border) {
    int bw = w + 2 * border;
    for (int y = -border; y < h + border; ++y) {
        const u8 *s = src + (size_t)reflect101(y, h) * sstride;
        u8 *d = dst + (size_t)(y + border) * bw;
        for (int x = -border; x < w + border; ++x) d[x + border] = s[reflect101(x, w)];
    }
}

/* FAST-9/16 ring, radius 3, clockwise from (0,3) — OpenCV fast_score.cpp makeOffsets */
static const int k_ring_dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int k_ring_dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

/* corner test + cornerScore<16> (OpenCV fast.cpp FAST_t<16>, fast_score.cpp): returns 0 for a non-corner,
 * else the largest threshold for which the pixel is still a 9-arc corner (as uchar). */
static int fast_corner_score(const u8 *p, int stride, int threshold) {
    int v = p[0];
    {   /* FAST_t's cheap rejection (fast.cpp): every antipodal ring pair must hold a darker (1) / brighter (2) pixel.
           Pure speed-up of the CPU baseline: a 9-arc always satisfies it, so results are unchanged. */
        const int lo = v - threshold, hi = v + threshold;
#define CLS(k) ((p[k_ring_dy[k] * stride + k_ring_dx[k]] < lo ? 1 : 0) | (p[k_ring_dy[k] * stride + k_ring_dx[k]] > hi ? 2 : 0))
        int c = CLS(0) | CLS(8);
        if (!c) return 0;
        c &= CLS(2) | CLS(10);
        c &= CLS(4) | CLS(12);
        c &= CLS(6) | CLS(14);
        if (!c) return 0;
        c &= CLS(1) | CLS(9);
        c &= CLS(3) | CLS(11);
        c &= CLS(5) | CLS(13);
        c &= CLS(7) | CLS(15);
        if (!c) return 0;
#undef CLS
    }
    int d[25];
    for (int k = 0; k < 16; ++k) d[k] = v - p[k_ring_dy[k] * stride + k_ring_dx[k]];
    for (int k = 16; k < 25; ++k) d[k] = d[k - 16];
    /* is it a corner?  >=9 contiguous ring px with x < v-t (d > t) or x > v+t (d < -t) */
    int is_corner = 0, cnt_dark = 0, cnt_bright = 0;
    for (int k = 0; k < 25; ++k) {
        if (d[k] > threshold) { if (++cnt_dark > 8) is_corner = 1; } else cnt_dark = 0;
        if (d[k] < -threshold) { if (++cnt_bright > 8) is_corner = 1; } else cnt_bright = 0;
    }
    if (!is_corner) return 0;
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = imin(d[k + 1], d[k + 2]);
        a = imin(a, d[k + 3]);
        if (a <= a0) continue;
        a = imin(a, d[k + 4]); a = imin(a, d[k + 5]); a = imin(a, d[k + 6]);
        a = imin(a, d[k + 7]); a = imin(a, d[k + 8]);
        a0 = imax(a0, imin(a, d[k]));
        a0 = imax(a0, imin(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = imax(d[k + 1], d[k + 2]);
        b = imax(b, d[k + 3]); b = imax(b, d[k + 4]); b = imax(b, d[k + 5]);
        if (b >= b0) continue;
        b = imax(b, d[k + 6]); b = imax(b, d[k + 7]); b = imax(b, d[k + 8]);
        b0 = imin(b0, imax(b, d[k]));
        b0 = imin(b0, imax(b, d[k + 9]));
    }
    return (u8)(-b0 - 1);
}

/* pre-NMS score map: 0 outside rows/cols [3, dim-4] and for non-corners */
void afvo_fast_score_map(const u8 *img, int w, int h, int stride, int threshold, u8 *score) {
    threshold = imin(imax(threshold, 0), 255);
    memset(score, 0, (size_t)w * h);
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) score[(size_t)y * w + x] = (u8)fast_corner_score(img + (size_t)y * stride + x, stride, threshold);
}

/* FastFeatureDetector(threshold, nonmaxSuppression=true, TYPE_9_16)::detect — raster order output */
int afvo_fast9_16(const u8 *img, int w, int h, int stride, int threshold, int32_t *xs, int32_t *ys, int32_t *scores, int cap) {
    if (w < 7 || h < 7) return 0;
    u8 *score = (u8 *)malloc((size_t)w * h);
    afvo_fast_score_map(img, w, h, stride, threshold, score);
    int n = 0;
    for (int y = 3; y < h - 3; ++y) {
        const u8 *pp = score + (size_t)(y - 1) * w, *pc = score + (size_t)y * w, *pn = score + (size_t)(y + 1) * w;
        for (int x = 3; x < w - 3; ++x) {
            int s = pc[x];
            if (!s) continue;
            if (s > pc[x - 1] && s > pc[x + 1] && s > pp[x - 1] && s > pp[x] && s > pp[x + 1] && s > pn[x - 1] && s > pn[x] &&
                s > pn[x + 1]) {
                if (n < cap) { xs[n] = x; ys[n] = y; scores[n] = s; }
                ++n;
            }
        }
    }
    free(score);
    return n;
}

/* HarrisResponses (OpenCV orb.cpp), blockSize 7: integer sums over the 7x7 block whose origin is (x-3,y-3) */
void afvo_harris_sums(const u8 *bordered, int bstride, int x, int y, int *pa, int *pb, int *pc) {
    const u8 *p0 = bordered + (size_t)(y - 3 + AFVO_BORDER) * bstride + (x - 3 + AFVO_BORDER);
    int a = 0, b = 0, c = 0;
    for (int i = 0; i < 7; ++i)
        for (int j = 0; j < 7; ++j) {
            const u8 *p = p0 + i * bstride + j;
            int Ix = (p[1] - p[-1]) * 2 + (p[-bstride + 1] - p[-bstride - 1]) + (p[bstride + 1] - p[bstride - 1]);
            int Iy = (p[bstride] - p[-bstride]) * 2 + (p[bstride - 1] - p[-bstride - 1]) + (p[bstride + 1] - p[-bstride + 1]);
            a += Ix * Ix;
            b += Iy * Iy;
            c += Ix * Iy;
        }
    *pa = a; *pb = b; *pc = c;
}

float afvo_harris_response(int a, int b, int c) {
    const float harris_k = 0.04f;
    float scale = 1.f / ((1 << 2) * 7 * 255.f);
    float scale_sq_sq = scale * scale * scale * scale;
    return ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
}

/* ICAngles (OpenCV orb.cpp) == IC_Angle (ORBextractor.cc:143-170) with half patch 15 */
float afvo_ic_angle(const u8 *bordered, int bstride, int x, int y) {
    int umax[17];
    afvo_umax(umax);
    const u8 *center = bordered + (size_t)(y + AFVO_BORDER) * bstride + (x + AFVO_BORDER);
    int m_01 = 0, m_10 = 0;
    for (int u = -15; u <= 15; ++u) m_10 += u * center[u];
    for (int v = 1; v <= 15; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * bstride], val_minus = center[u - v * bstride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return afvo_fast_atan2((float)m_01, (float)m_10);
}

/* GaussianBlur(7x7, 2, 2, BORDER_REFLECT_101) on a pyramid ROI (orb.cpp).  The ROI is a sub-matrix, so
 * OpenCV takes the generic 8U separable path: integer taps [18,34,49,55,49,34,18] (/256) in both passes,
 * row pass exact int32, column pass = S/65536 rounded.  The vectorised column filter (SymmColumnVec_32s8u)
 * evaluates S/65536 exactly in float and rounds half-to-even (cvtps2dq); that is the rule used here for
 * every pixel (its scalar tail, <8 px per row, rounds half-up instead — not modelled). */
void afvo_gaussian_blur7(const u8 *bordered, int w, int h, int bstride, u8 *dst, int dstride) {
    int taps[7];
    afvo_gauss7_taps(taps);
    int32_t *rows = (int32_t *)malloc(sizeof(int32_t) * (size_t)w * (size_t)(h + 6));
    for (int y = -3; y < h + 3; ++y) {
        const u8 *s = bordered + (size_t)(y + AFVO_BORDER) * bstride + AFVO_BORDER;
        int32_t *r = rows + (size_t)(y + 3) * w;
        for (int x = 0; x < w; ++x) {
            int32_t acc = 0;
            for (int k = 0; k < 7; ++k) acc += taps[k] * s[x + k - 3];
            r[x] = acc;
        }
    }
    for (int y = 0; y < h; ++y) {
        u8 *d = dst + (size_t)y * dstride;
        for (int x = 0; x < w; ++x) {
            int32_t S = 0;
            for (int k = 0; k < 7; ++k) S += taps[k] * rows[(size_t)(y + k) * w + x];
            int32_t q = S >> 16, r = S & 0xFFFF;
            if (r > 32768 || (r == 32768 && (q & 1))) ++q;
            d[x] = (u8)(q > 255 ? 255 : q);
        }
    }
    free(rows);
}

/* computeOrbDescriptor (FeatureExtractor.h:178-217) == OpenCV computeOrbDescriptors, WTA_K = 2 */
void afvo_brief_descriptor(const u8 *bb, int bstride, int cx, int cy, float angle_deg, u8 *desc) {
    float a, b;
    afvo_sincos_deg(angle_deg, &a, &b);
    const u8 *center = bb + (size_t)(cy + AFVO_BORDER) * bstride + (cx + AFVO_BORDER);
    const int8_t *pat = k_brief_pattern;
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int t = 0; t < 8; ++t, pat += 4) {
            float x0 = (float)pat[0] * a - (float)pat[1] * b, y0 = (float)pat[0] * b + (float)pat[1] * a;
            float x1 = (float)pat[2] * a - (float)pat[3] * b, y1 = (float)pat[2] * b + (float)pat[3] * a;
            int t0 = center[cv_round_f(y0) * bstride + cv_round_f(x0)];
            int t1 = center[cv_round_f(y1) * bstride + cv_round_f(x1)];
            val |= (t0 < t1) << t;
        }
        desc[i] = (u8)val;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Selection
 * ---------------------------------------------------------------------------------------------- */
static int cmp_float_desc(const void *a, const void *b) {
    float x = *(const float *)a, y = *(const float *)b;
    return (x < y) - (x > y);
}

/* KeyPointsFilter::retainBest set semantics: if n > k keep everything >= the k-th largest response */
int afvo_retain_best_mask(const float *resp, int n, int k, u8 *keep) {
    if (k < 0 || n <= k) {
        memset(keep, 1, (size_t)n);
        return n;
    }
    if (k == 0) {
        memset(keep, 0, (size_t)n);
        return 0;
    }
    float *tmp = (float *)malloc(sizeof(float) * (size_t)n);
    memcpy(tmp, resp, sizeof(float) * (size_t)n);
    qsort(tmp, (size_t)n, sizeof(float), cmp_float_desc);
    float thr = tmp[k - 1];
    free(tmp);
    int m = 0;
    for (int i = 0; i < n; ++i) {
        keep[i] = resp[i] >= thr;
        m += keep[i];
    }
    return m;
}

/* ---- DistributeOctTree (ORBextractor.cc:239-458) + ExtractorNode::DivideNode (:181-237) ---- */
typedef struct {
    int x0, y0, x1, y1; /* UL.x, UL.y, UR.x, BR.y */
    int *idx;
    int n;
    int no_more;
    int prev, next; /* std::list links */
    int alive;
} qnode;

typedef struct {
    qnode *nodes;
    int count, cap;
    int head, tail, size;
} qlist;

static int ql_new(qlist *L) {
    if (L->count == L->cap) {
        L->cap = L->cap ? L->cap * 2 : 64;
        L->nodes = (qnode *)realloc(L->nodes, sizeof(qnode) * (size_t)L->cap);
    }
    qnode *q = &L->nodes[L->count];
    memset(q, 0, sizeof(*q));
    q->prev = q->next = -1;
    return L->count++;
}
static void ql_push_front(qlist *L, int i) {
    qnode *q = &L->nodes[i];
    q->prev = -1; q->next = L->head; q->alive = 1;
    if (L->head >= 0) L->nodes[L->head].prev = i;
    L->head = i;
    if (L->tail < 0) L->tail = i;
    L->size++;
}
static void ql_push_back(qlist *L, int i) {
    qnode *q = &L->nodes[i];
    q->next = -1; q->prev = L->tail; q->alive = 1;
    if (L->tail >= 0) L->nodes[L->tail].next = i;
    L->tail = i;
    if (L->head < 0) L->head = i;
    L->size++;
}
static int ql_erase(qlist *L, int i) { /* returns next */
    qnode *q = &L->nodes[i];
    int nx = q->next;
    if (q->prev >= 0) L->nodes[q->prev].next = q->next; else L->head = q->next;
    if (q->next >= 0) L->nodes[q->next].prev = q->prev; else L->tail = q->prev;
    q->alive = 0;
    free(q->idx);
    q->idx = NULL;
    L->size--;
    return nx;
}

/* DivideNode: creates 4 children in the pool (not yet in the list); returns their pool indices */
static void divide_node(qlist *L, int parent, const float *px, const float *py, int ch[4]) {
    for (int k = 0; k < 4; ++k) ch[k] = ql_new(L);
    qnode *P = &L->nodes[parent];
    const int halfX = (int)ceilf((float)(P->x1 - P->x0) / 2);
    const int halfY = (int)ceilf((float)(P->y1 - P->y0) / 2);
    qnode *n1 = &L->nodes[ch[0]], *n2 = &L->nodes[ch[1]], *n3 = &L->nodes[ch[2]], *n4 = &L->nodes[ch[3]];
    n1->x0 = P->x0; n1->y0 = P->y0; n1->x1 = P->x0 + halfX; n1->y1 = P->y0 + halfY;
    n2->x0 = P->x0 + halfX; n2->y0 = P->y0; n2->x1 = P->x1; n2->y1 = P->y0 + halfY;
    n3->x0 = P->x0; n3->y0 = P->y0 + halfY; n3->x1 = P->x0 + halfX; n3->y1 = P->y1;
    n4->x0 = P->x0 + halfX; n4->y0 = P->y0 + halfY; n4->x1 = P->x1; n4->y1 = P->y1;
    for (int k = 0; k < 4; ++k) L->nodes[ch[k]].idx = (int *)malloc(sizeof(int) * (size_t)(P->n > 0 ? P->n : 1));
    const float midx = (float)n1->x1, midy = (float)n1->y1;
    for (int i = 0; i < P->n; ++i) {
        int id = P->idx[i];
        qnode *t;
        if (px[id] < midx) t = (py[id] < midy) ? n1 : n3;
        else t = (py[id] < midy) ? n2 : n4;
        t->idx[t->n++] = id;
    }
    for (int k = 0; k < 4; ++k)
        if (L->nodes[ch[k]].n == 1) L->nodes[ch[k]].no_more = 1;
}

typedef struct { int size; int node; } size_node;
static int cmp_size_node(const void *a, const void *b) {
    const size_node *x = (const size_node *)a, *y = (const size_node *)b;
    if (x->size != y->size) return (x->size > y->size) - (x->size < y->size);
    return (x->node > y->node) - (x->node < y->node); /* pointer tie-break -> creation sequence */
}

/* push the non-empty children of a split (n1..n4 order, each to the FRONT) and record expandable ones */
static void add_children(qlist *L, const int ch[4], size_node **vec, int *vn, int *vcap, int *n_to_expand) {
    for (int k = 0; k < 4; ++k) {
        qnode *c = &L->nodes[ch[k]];
        if (c->n > 0) {
            ql_push_front(L, ch[k]);
            if (c->n > 1) {
                if (n_to_expand) (*n_to_expand)++;
                if (*vn == *vcap) {
                    *vcap = *vcap ? *vcap * 2 : 64;
                    *vec = (size_node *)realloc(*vec, sizeof(size_node) * (size_t)*vcap);
                }
                (*vec)[*vn].size = c->n;
                (*vec)[*vn].node = ch[k];
                (*vn)++;
            }
        } else {
            free(c->idx);
            c->idx = NULL;
        }
    }
}

int afvo_quadtree(const float *px, const float *py, const float *resp, const int64_t *tiebreak, int n, int min_x, int max_x,
                  int min_y, int max_y, int N, int32_t *out_idx, int cap) {
    qlist L;
    memset(&L, 0, sizeof(L));
    L.head = L.tail = -1;
    const int nIni = (int)roundf((float)(max_x - min_x) / (float)(max_y - min_y));
    const float hX = (float)(max_x - min_x) / (float)nIni;
    int *ini = (int *)malloc(sizeof(int) * (size_t)(nIni > 0 ? nIni : 1));
    for (int i = 0; i < nIni; ++i) {
        int q = ql_new(&L);
        qnode *ni = &L.nodes[q];
        ni->x0 = (int)(hX * (float)i);
        ni->x1 = (int)(hX * (float)(i + 1));
        ni->y0 = 0;
        ni->y1 = max_y - min_y;
        ni->idx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
        ql_push_back(&L, q);
        ini[i] = q;
    }
    for (int i = 0; i < n; ++i) {
        qnode *r = &L.nodes[ini[(size_t)(px[i] / hX)]];
        r->idx[r->n++] = i;
    }
    free(ini);
    for (int it = L.head; it >= 0;) {
        qnode *q = &L.nodes[it];
        if (q->n == 1) { q->no_more = 1; it = q->next; }
        else if (q->n == 0) it = ql_erase(&L, it);
        else it = q->next;
    }
    int finish = 0;
    size_node *vec = NULL;
    int vn = 0, vcap = 0;
    while (!finish) {
        int prev_size = L.size;
        int n_to_expand = 0;
        vn = 0;
        for (int it = L.head; it >= 0;) {
            if (L.nodes[it].no_more) { it = L.nodes[it].next; continue; }
            int ch[4];
            divide_node(&L, it, px, py, ch);
            add_children(&L, ch, &vec, &vn, &vcap, &n_to_expand);
            it = ql_erase(&L, it);
        }
        if (L.size >= N || L.size == prev_size) {
            finish = 1;
        } else if (L.size + n_to_expand * 3 > N) {
            while (!finish) {
                prev_size = L.size;
                int pn = vn;
                size_node *prev = (size_node *)malloc(sizeof(size_node) * (size_t)(pn > 0 ? pn : 1));
                memcpy(prev, vec, sizeof(size_node) * (size_t)pn);
                vn = 0;
                qsort(prev, (size_t)pn, sizeof(size_node), cmp_size_node);
                for (int j = pn - 1; j >= 0; --j) {
                    int ch[4];
                    divide_node(&L, prev[j].node, px, py, ch);
                    add_children(&L, ch, &vec, &vn, &vcap, NULL);
                    ql_erase(&L, prev[j].node);
                    if (L.size >= N) break;
                }
                free(prev);
                if (L.size >= N || L.size == prev_size) finish = 1;
            }
        }
    }
    free(vec);
    int m = 0;
    for (int it = L.head; it >= 0; it = L.nodes[it].next) {
        qnode *q = &L.nodes[it];
        int best = q->idx[0];
        for (int k = 1; k < q->n; ++k) {
            int id = q->idx[k];
            if (resp[id] > resp[best] || (tiebreak && resp[id] == resp[best] && tiebreak[id] < tiebreak[best])) best = id;
        }
        if (m < cap) out_idx[m] = best;
        ++m;
    }
    for (int i = 0; i < L.count; ++i) free(L.nodes[i].idx);
    free(L.nodes);
    return m;
}

/* ------------------------------------------------------------------------------------------------
 * Full extraction
 * ---------------------------------------------------------------------------------------------- */
static u8 *build_bordered(const u8 *img, int w, int h, int stride) {
    u8 *b = (u8 *)malloc((size_t)(w + 2 * AFVO_BORDER) * (size_t)(h + 2 * AFVO_BORDER));
    afvo_make_border101(img, w, h, stride, b, AFVO_BORDER);
    return b;
}

/* one cv::ORB::compute()-style pass for level L: (re)build levels 0..L, blur each, return blurred level L
 * with the unblurred apron (orb.cpp: GaussianBlur on imagePyramid(layerInfo[level]) in place). */
static u8 *faithful_compute_level(const u8 *gray, int w, int h, int stride, const int *lw, const int *lh, int L) {
    u8 *prev = (u8 *)malloc((size_t)w * h);
    for (int y = 0; y < h; ++y) memcpy(prev + (size_t)y * w, gray + (size_t)y * stride, (size_t)w);
    u8 *result = NULL;
    u8 **bord = (u8 **)calloc((size_t)L + 1, sizeof(u8 *));
    for (int l = 0; l <= L; ++l) {
        if (l > 0) {
            u8 *cur = (u8 *)malloc((size_t)lw[l] * lh[l]);
            afvo_resize_linear_exact(prev, lw[l - 1], lh[l - 1], lw[l - 1], cur, lw[l], lh[l], lw[l]);
            free(prev);
            prev = cur;
        }
        bord[l] = build_bordered(prev, lw[l], lh[l], lw[l]);
    }
    free(prev);
    for (int l = 0; l <= L; ++l) {
        int bs = lw[l] + 2 * AFVO_BORDER;
        u8 *bl = (u8 *)malloc((size_t)lw[l] * lh[l]);
        afvo_gaussian_blur7(bord[l], lw[l], lh[l], bs, bl, lw[l]);
        for (int y = 0; y < lh[l]; ++y) memcpy(bord[l] + (size_t)(y + AFVO_BORDER) * bs + AFVO_BORDER, bl + (size_t)y * lw[l], (size_t)lw[l]);
        free(bl);
        if (l == L) result = bord[l]; else free(bord[l]);
    }
    free(bord);
    return result;
}

int afvo_orb_extract_trace(const afvo_params *p, const u8 *gray, int w, int h, int stride, afvo_trace *tr, afvo_keypoint *kps,
                           u8 *desc32, int cap, int *n_out) {
    const int nl = p->nlevels;
    if (nl < 1 || nl > AFVO_MAX_LEVELS) return -1;
    int variant = tr ? 0 : 0;
    (void)variant;
    int lw[AFVO_MAX_LEVELS], lh[AFVO_MAX_LEVELS], q[AFVO_MAX_LEVELS], cvq[AFVO_MAX_LEVELS];
    float ls[AFVO_MAX_LEVELS];
    afvo_level_geometry(w, h, nl, p->scale_factor, lw, lh, ls);
    afvo_quotas_extractor(p->nfeatures, nl, p->scale_factor, q);
    afvo_quotas_cvorb(p->nfeatures * 10, nl, p->scale_factor, cvq);
    u8 *lev[AFVO_MAX_LEVELS], *bord[AFVO_MAX_LEVELS];
    memset(lev, 0, sizeof(lev));
    memset(bord, 0, sizeof(bord));
    /* E2: pyramid, level l resized from level l-1 */
    for (int l = 0; l < nl; ++l) {
        lev[l] = (u8 *)malloc((size_t)lw[l] * lh[l]);
        if (l == 0) for (int y = 0; y < h; ++y) memcpy(lev[0] + (size_t)y * w, gray + (size_t)y * stride, (size_t)w);
        else afvo_resize_linear_exact(lev[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], lev[l], lw[l], lh[l], lw[l]);
        bord[l] = build_bordered(lev[l], lw[l], lh[l], lw[l]);
    }
    /* E3..E6 per level */
    int total_cap = 0;
    for (int l = 0; l < nl; ++l) total_cap += ((lw[l] + 1) / 2) * ((lh[l] + 1) / 2) + 16;
    afvo_candidate *cand = (afvo_candidate *)malloc(sizeof(afvo_candidate) * (size_t)total_cap);
    u8 *keep1 = (u8 *)malloc((size_t)total_cap), *keep2 = (u8 *)malloc((size_t)total_cap);
    int ncand = 0, n = 0;
    if (tr) memset(tr, 0, sizeof(*tr));
    for (int l = 0; l < nl; ++l) {
        int lcap = total_cap - ncand;
        int32_t *xs = (int32_t *)malloc(sizeof(int32_t) * (size_t)lcap * 3), *ys = xs + lcap, *sc = ys + lcap;
        int m = afvo_fast9_16(lev[l], lw[l], lh[l], lw[l], p->fast_threshold, xs, ys, sc, lcap);
        afvo_candidate *c = cand + ncand;
        u8 *k1 = keep1 + ncand, *k2 = keep2 + ncand;
        float *resp = (float *)malloc(sizeof(float) * (size_t)(m > 0 ? m : 1));
        for (int i = 0; i < m; ++i) {
            c[i].x = xs[i]; c[i].y = ys[i]; c[i].level = l; c[i].fast_score = sc[i];
            c[i].ha = c[i].hb = c[i].hc = 0; c[i].response = 0.f;
            resp[i] = (float)sc[i];
        }
        free(xs);
        /* E4a: retainBest(2*featuresNum) on the FAST score */
        afvo_retain_best_mask(resp, m, 2 * cvq[l], k1);
        /* E5: Harris for the survivors */
        int bs = lw[l] + 2 * AFVO_BORDER;
        int m1 = 0;
        for (int i = 0; i < m; ++i) {
            if (!k1[i]) { resp[i] = -FLT_MAX; continue; }
            afvo_harris_sums(bord[l], bs, c[i].x, c[i].y, &c[i].ha, &c[i].hb, &c[i].hc);
            c[i].response = afvo_harris_response(c[i].ha, c[i].hb, c[i].hc);
            resp[i] = c[i].response;
            ++m1;
        }
        /* E4b: retainBest(featuresNum) on the Harris response, among survivors of E4a */
        {
            float *r1 = (float *)malloc(sizeof(float) * (size_t)(m1 > 0 ? m1 : 1));
            u8 *kk = (u8 *)malloc((size_t)(m1 > 0 ? m1 : 1));
            int j = 0;
            for (int i = 0; i < m; ++i) if (k1[i]) r1[j++] = resp[i];
            afvo_retain_best_mask(r1, m1, cvq[l], kk);
            j = 0;
            for (int i = 0; i < m; ++i) k2[i] = k1[i] ? kk[j++] : 0;
            free(r1);
            free(kk);
        }
        /* E7: quadtree over level-0 coordinates (pt *= scale happens inside cv::ORB::detect) */
        int m2 = 0;
        for (int i = 0; i < m; ++i) m2 += k2[i];
        float *px = (float *)malloc(sizeof(float) * (size_t)(3 * m2 + 3)), *py = px + m2, *pr = py + m2;
        int *src = (int *)malloc(sizeof(int) * (size_t)(m2 + 1));
        int64_t *tb = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m2 + 1));
        int j = 0;
        for (int i = 0; i < m; ++i) {
            if (!k2[i]) continue;
            px[j] = (float)c[i].x * ls[l];
            py[j] = (float)c[i].y * ls[l];
            pr[j] = c[i].response;
            tb[j] = (int64_t)c[i].y * lw[l] + c[i].x;
            src[j++] = i;
        }
        /* saturated: q..q+2 nodes; but the first split round is unconditional (ORBextractor.cc:283-366), so a small quota
           still yields up to 4 * nIni nodes */
        const int sel_cap = imax(q[l], 4 * (int)roundf((float)w / (float)h)) + 8;
        int32_t *sel = (int32_t *)malloc(sizeof(int32_t) * (size_t)sel_cap);
        int ns = 0;
        if (m2 > 0) ns = afvo_quadtree(px, py, pr, tb, m2, 0, w, 0, h, q[l], sel, sel_cap);
        if (tr) tr->t_counts[l] = ns;
        /* E6 + E8..E10 for the selected keypoints of this level */
        u8 *bb = NULL;
        if (ns > 0) {
            bb = (u8 *)malloc((size_t)bs * (size_t)(lh[l] + 2 * AFVO_BORDER));
            memcpy(bb, bord[l], (size_t)bs * (size_t)(lh[l] + 2 * AFVO_BORDER));
            u8 *bl = (u8 *)malloc((size_t)lw[l] * lh[l]);
            afvo_gaussian_blur7(bord[l], lw[l], lh[l], bs, bl, lw[l]);
            for (int y = 0; y < lh[l]; ++y) memcpy(bb + (size_t)(y + AFVO_BORDER) * bs + AFVO_BORDER, bl + (size_t)y * lw[l], (size_t)lw[l]);
            if (tr) tr->blurred[l] = bl; else free(bl);
        } else if (tr) {
            u8 *bl = (u8 *)malloc((size_t)lw[l] * lh[l]);
            afvo_gaussian_blur7(bord[l], lw[l], lh[l], bs, bl, lw[l]);
            tr->blurred[l] = bl;
        }
        for (int s = 0; s < ns; ++s) {
            const afvo_candidate *cc = &c[src[sel[s]]];
            if (n < cap) {
                afvo_keypoint *kp = &kps[n];
                kp->x = (float)cc->x * ls[l];
                kp->y = (float)cc->y * ls[l];
                kp->size = 31 * ls[l];
                kp->angle = afvo_ic_angle(bord[l], bs, cc->x, cc->y);
                kp->response = cc->response;
                kp->octave = l;
                kp->class_id = -1;
                /* BRIEF centre: cvRound(pt * (1/scale)) (orb.cpp computeOrbDescriptors) */
                float inv = 1.f / ls[l];
                int cx = cv_round_f(kp->x * inv), cy = cv_round_f(kp->y * inv);
                afvo_brief_descriptor(bb, bs, cx, cy, kp->angle, desc32 + (size_t)n * 32);
            }
            ++n;
        }
        free(bb);
        free(sel); free(tb); free(src); free(px); free(resp);
        ncand += m;
    }
    if (tr) {
        tr->nlevels = nl;
        for (int l = 0; l < nl; ++l) { tr->lw[l] = lw[l]; tr->lh[l] = lh[l]; tr->lscale[l] = ls[l]; tr->level[l] = lev[l]; }
        tr->cand = cand; tr->ncand = ncand; tr->keep1 = keep1; tr->keep2 = keep2;
    } else {
        for (int l = 0; l < nl; ++l) free(lev[l]);
        free(cand); free(keep1); free(keep2);
    }
    for (int l = 0; l < nl; ++l) free(bord[l]);
    *n_out = n < cap ? n : cap;
    return n > cap ? -2 : 0;
}

void afvo_trace_free(afvo_trace *tr) {
    for (int l = 0; l < tr->nlevels; ++l) { free(tr->level[l]); free(tr->blurred[l]); }
    free(tr->cand); free(tr->keep1); free(tr->keep2);
    memset(tr, 0, sizeof(*tr));
}

int afvo_orb_extract(const afvo_params *p, const u8 *gray, int w, int h, int stride, int variant, afvo_keypoint *kps, u8 *desc32,
                     int cap, int *n_out) {
    int rc = afvo_orb_extract_trace(p, gray, w, h, stride, NULL, kps, desc32, cap, n_out);
    if (rc != 0 || variant == 0) return rc;
    /* variant 1: additionally perform the redundant work of the reference's call pattern
     * (Feature_orb32.cpp:42-53: one cv::ORB::compute per populated level, each rebuilding and blurring
     * levels 0..L) and recompute the descriptors from those buffers.  Outputs are identical. */
    int lw[AFVO_MAX_LEVELS], lh[AFVO_MAX_LEVELS];
    float ls[AFVO_MAX_LEVELS];
    afvo_level_geometry(w, h, p->nlevels, p->scale_factor, lw, lh, ls);
    int i = 0;
    while (i < *n_out) {
        int L = kps[i].octave;
        u8 *bb = faithful_compute_level(gray, w, h, stride, lw, lh, L);
        int bs = lw[L] + 2 * AFVO_BORDER;
        float inv = 1.f / ls[L];
        for (; i < *n_out && kps[i].octave == L; ++i)
            afvo_brief_descriptor(bb, bs, cv_round_f(kps[i].x * inv), cv_round_f(kps[i].y * inv), kps[i].angle, desc32 + (size_t)i * 32);
        free(bb);
    }
    return 0;
}

/* computeDescriptors on its own (Feature_orb32.cpp:42-53: orb32_extractor->compute(img.grayImg, keypoints, descriptors_level[level])):
 * cv::ORB::compute = detectAndCompute(..., useProvidedKeypoints = true) - levels 0 .. (octave of the keypoints) are rebuilt from the image and
 * blurred, every keypoint is described at cvRound(pt * (1 / scale)) of its own octave with the angle it carries (orb.cpp computeOrbDescriptors);
 * KeyPointsFilter::runByImageBorder with edgeThreshold 0 (Feature_orb32.cpp:23) removes nothing.  desc32[n][32] in the order of kps.
 * Returns -1 for a keypoint whose octave is not a level or whose centre lies off its level image (the sampling would leave the apron). */
int afvo_orb_compute(const afvo_params *p, const u8 *gray, int w, int h, int stride, const afvo_keypoint *kps, int n, u8 *desc32) {
    const int nl = p->nlevels;
    if (nl < 1 || nl > AFVO_MAX_LEVELS) return -1;
    int lw[AFVO_MAX_LEVELS], lh[AFVO_MAX_LEVELS];
    float ls[AFVO_MAX_LEVELS];
    afvo_level_geometry(w, h, nl, p->scale_factor, lw, lh, ls);
    u8 *bb[AFVO_MAX_LEVELS];
    memset(bb, 0, sizeof(bb));
    int rc = 0;
    for (int i = 0; i < n && rc == 0; ++i) {
        const int L = kps[i].octave;
        if (L < 0 || L >= nl) { rc = -1; break; }
        const float inv = 1.f / ls[L];
        const int cx = cv_round_f(kps[i].x * inv), cy = cv_round_f(kps[i].y * inv);
        if (cx < 0 || cx > lw[L] || cy < 0 || cy > lh[L]) { rc = -1; break; }
        if (!bb[L]) bb[L] = faithful_compute_level(gray, w, h, stride, lw, lh, L);
        afvo_brief_descriptor(bb[L], lw[L] + 2 * AFVO_BORDER, cx, cy, kps[i].angle, desc32 + (size_t)i * 32);
    }
    for (int l = 0; l < nl; ++l) free(bb[l]);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Matching
 * ---------------------------------------------------------------------------------------------- */

/* Feature_orb32.cpp:67-84 — SWAR popcount over 8 x u32 */
int afvo_hamming256(const u8 *a, const u8 *b) {
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

/* cv::norm(a,b,NORM_HAMMING) over nbytes (Feature_akaze61.cpp:75-77) */
int afvo_hamming_bytes(const u8 *a, const u8 *b, int nbytes) {
    int d = 0;
    for (int i = 0; i < nbytes; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}

/* cv::norm(a,b,NORM_L2SQR) for CV_32F (Feature_sift128.cpp:132-134): normL2Sqr<float,double> — float
 * differences, double squares, 4-way unrolled partial sums added into a double accumulator, returned as
 * double and narrowed to Descriptor_Distance_Type = float */
float afvo_l2sqr(const float *a, const float *b, int n) {
    double s = 0;
    int i = 0;
    for (; i <= n - 4; i += 4) {
        double v0 = (double)(a[i] - b[i]), v1 = (double)(a[i + 1] - b[i + 1]), v2 = (double)(a[i + 2] - b[i + 2]),
               v3 = (double)(a[i + 3] - b[i + 3]);
        s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    }
    for (; i < n; ++i) {
        double v = (double)(a[i] - b[i]);
        s += v * v;
    }
    return (float)s;
}

static float bow_dist(const afvo_bow_job *j, int i1, int i2) {
    if (j->float_dim > 0)
        return afvo_l2sqr((const float *)(const void *)j->desc1 + (size_t)i1 * j->float_dim, (const float *)(const void *)j->desc2 + (size_t)i2 * j->float_dim,
                          j->float_dim);
    const u8 *a = j->desc1 + (size_t)i1 * j->desc_bytes, *b = j->desc2 + (size_t)i2 * j->desc_bytes;
    if (j->desc_bytes == 32) return (float)afvo_hamming256(a, b);
    return (float)afvo_hamming_bytes(a, b, j->desc_bytes);
}

/* FeatureMatcher.cc:1587-1599 (rotFactor = 1/30, :1579-1585) */
int afvo_rotation_bin(float a1, float a2) {
    const float rot_factor = 1.0f / 30.0f;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)roundf(rot * rot_factor);
    if (bin == 30) bin = 0;
    return bin;
}

/* FeatureMatcher.cc:1631-1668 */
void afvo_three_maxima(const int *hs, int L, int *ind1, int *ind2, int *ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    *ind1 = *ind2 = *ind3 = -1;
    for (int i = 0; i < L; ++i) {
        const int s = hs[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; *ind3 = *ind2; *ind2 = *ind1; *ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; *ind3 = *ind2; *ind2 = i; }
        else if (s > max3) { max3 = s; *ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

typedef struct { int *key; int *bin; int n; } ori_log;

/* drop matches whose rotation bin is not among the three maxima; keys index `out` */
static int apply_orientation(ori_log *lg, int32_t *out, int nmatches) {
    int hs[30];
    memset(hs, 0, sizeof(hs));
    for (int i = 0; i < lg->n; ++i) hs[lg->bin[i]]++;
    int i1, i2, i3;
    afvo_three_maxima(hs, 30, &i1, &i2, &i3);
    for (int i = 0; i < lg->n; ++i) {
        int b = lg->bin[i];
        if (b == i1 || b == i2 || b == i3) continue;
        out[lg->key[i]] = -1;
        nmatches--;
    }
    return nmatches;
}

/* merge-join driver shared by M2/M3/M4: calls node(ctx, seg1, n1, seg2, n2) for every node id on both sides */
typedef void (*node_fn)(void *ctx, const int32_t *s1, int n1, const int32_t *s2, int n2);
static void for_shared_nodes(const afvo_bow_job *j, node_fn fn, void *ctx) {
    if (j->nnodes1 == 0 || j->nnodes2 == 0) {
        int32_t *all = (int32_t *)malloc(sizeof(int32_t) * (size_t)(imax(j->n1, j->n2) + 1));
        for (int i = 0; i < imax(j->n1, j->n2); ++i) all[i] = i;
        fn(ctx, all, j->n1, all, j->n2);
        free(all);
        return;
    }
    int a = 0, b = 0;
    while (a < j->nnodes1 && b < j->nnodes2) {
        if (j->node_id1[a] == j->node_id2[b]) {
            fn(ctx, j->seg_idx1 + j->seg_ptr1[a], j->seg_ptr1[a + 1] - j->seg_ptr1[a], j->seg_idx2 + j->seg_ptr2[b],
               j->seg_ptr2[b + 1] - j->seg_ptr2[b]);
            ++a; ++b;
        } else if (j->node_id1[a] < j->node_id2[b]) {
            while (a < j->nnodes1 && j->node_id1[a] < j->node_id2[b]) ++a; /* lower_bound */
        } else {
            while (b < j->nnodes2 && j->node_id2[b] < j->node_id1[a]) ++b;
        }
    }
}

typedef struct {
    const afvo_bow_job *j;
    int32_t *out;
    u8 *matched2;
    int nmatches;
    ori_log lg;
} m2_ctx;

/* FeatureMatcher.cc:587-641 */
static void m2_node(void *vctx, const int32_t *s1, int n1, const int32_t *s2, int n2) {
    m2_ctx *c = (m2_ctx *)vctx;
    const afvo_bow_job *j = c->j;
    for (int a = 0; a < n1; ++a) {
        const int idx1 = s1[a];
        if (j->valid1 && !j->valid1[idx1]) continue;
        float best1 = FLT_MAX, best2 = FLT_MAX;
        int best_idx2 = -1;
        for (int b = 0; b < n2; ++b) {
            const int idx2 = s2[b];
            if (c->matched2[idx2] || (j->valid2 && !j->valid2[idx2])) continue;
            float d = bow_dist(j, idx1, idx2);
            if (d < best1) { best2 = best1; best1 = d; best_idx2 = idx2; }
            else if (d < best2) { best2 = d; }
        }
        if (best1 < j->th_low) {
            if (best1 < j->nnratio * best2) {
                c->out[idx1] = best_idx2;
                c->matched2[best_idx2] = 1;
                c->nmatches++;
                if (j->check_orientation) {
                    c->lg.key[c->lg.n] = idx1;
                    c->lg.bin[c->lg.n] = afvo_rotation_bin(j->angle1[idx1], j->angle2[best_idx2]);
                    c->lg.n++;
                }
            }
        }
    }
}

int afvo_search_by_bow_kf_kf(const afvo_bow_job *j, int32_t *match12) {
    m2_ctx c;
    memset(&c, 0, sizeof(c));
    c.j = j; c.out = match12;
    c.matched2 = (u8 *)calloc((size_t)j->n2 + 1, 1);
    c.lg.key = (int *)malloc(sizeof(int) * (size_t)(2 * j->n1 + 2));
    c.lg.bin = c.lg.key + j->n1 + 1;
    for (int i = 0; i < j->n1; ++i) match12[i] = -1;
    for_shared_nodes(j, m2_node, &c);
    int nm = c.nmatches;
    if (j->check_orientation) nm = apply_orientation(&c.lg, match12, nm);
    free(c.matched2);
    free(c.lg.key);
    return nm;
}

/* FeatureMatcher.cc:212-263: out is indexed by the FRAME feature (side 2) */
static void m3_node(void *vctx, const int32_t *s1, int n1, const int32_t *s2, int n2) {
    m2_ctx *c = (m2_ctx *)vctx;
    const afvo_bow_job *j = c->j;
    for (int a = 0; a < n1; ++a) {
        const int idx_kf = s1[a];
        if (j->valid1 && !j->valid1[idx_kf]) continue;
        float best1 = FLT_MAX, best2 = FLT_MAX;
        int best_f = -1;
        for (int b = 0; b < n2; ++b) {
            const int idx_f = s2[b];
            if (c->out[idx_f] >= 0) continue;
            float d = bow_dist(j, idx_kf, idx_f);
            if (d < best1) { best2 = best1; best1 = d; best_f = idx_f; }
            else if (d < best2) { best2 = d; }
        }
        if (best1 <= j->th_low) {
            if (best1 < j->nnratio * best2) {
                c->out[best_f] = idx_kf;
                c->nmatches++;
                if (j->check_orientation) {
                    c->lg.key[c->lg.n] = best_f;
                    c->lg.bin[c->lg.n] = afvo_rotation_bin(j->angle1[idx_kf], j->angle2[best_f]);
                    c->lg.n++;
                }
            }
        }
    }
}

int afvo_search_by_bow_kf_frame(const afvo_bow_job *j, int32_t *matchF) {
    m2_ctx c;
    memset(&c, 0, sizeof(c));
    c.j = j; c.out = matchF;
    c.lg.key = (int *)malloc(sizeof(int) * (size_t)(2 * j->n1 + 2));
    c.lg.bin = c.lg.key + j->n1 + 1;
    for (int i = 0; i < j->n2; ++i) matchF[i] = -1;
    for_shared_nodes(j, m3_node, &c);
    int nm = c.nmatches;
    if (j->check_orientation) nm = apply_orientation(&c.lg, matchF, nm);
    free(c.lg.key);
    return nm;
}

typedef struct { const afvo_tri_job *t; int32_t *out; int nmatches; } m4_ctx;

/* FeatureMatcher.cc:165-182 */
static int check_dist_epipolar(float x1, float y1, float x2, float y2, const float *F, float sigma2_kp2) {
    const float a = x1 * F[0] + y1 * F[3] + F[6];
    const float b = x1 * F[1] + y1 * F[4] + F[7];
    const float c = x1 * F[2] + y1 * F[5] + F[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84f * sigma2_kp2;
}

/* FeatureMatcher.cc:695-764 */
static void m4_node(void *vctx, const int32_t *s1, int n1, const int32_t *s2, int n2) {
    m4_ctx *c = (m4_ctx *)vctx;
    const afvo_tri_job *t = c->t;
    const afvo_bow_job *j = &t->bow;
    for (int a = 0; a < n1; ++a) {
        const int idx1 = s1[a];
        if (j->valid1 && j->valid1[idx1]) continue; /* already has a MapPoint */
        const int stereo1 = t->u_right1 && t->u_right1[idx1] >= 0.0f; /* :705 */
        if (t->only_stereo && !stereo1) continue;                      /* :707-709 */
        float best_dist = j->th_low;
        int best_idx2 = -1;
        for (int b = 0; b < n2; ++b) {
            const int idx2 = s2[b];
            if (j->valid2 && j->valid2[idx2]) continue;
            const int stereo2 = t->u_right2 && t->u_right2[idx2] >= 0.0f; /* :727 */
            if (t->only_stereo && !stereo2) continue;                      /* :729-731 */
            const float d = bow_dist(j, idx1, idx2);
            if (d > j->th_low || d > best_dist) continue;
            if (!stereo1 && !stereo2) { /* :741-748 */
                const float distex = t->ex - t->x2[idx2], distey = t->ey - t->y2[idx2];
                if (distex * distex + distey * distey < 100.0f * sqrtf(t->sigma2_2[idx2])) continue;
            }
            if (check_dist_epipolar(t->x1[idx1], t->y1[idx1], t->x2[idx2], t->y2[idx2], t->F12, t->sigma2_2[idx2])) {
                best_idx2 = idx2;
                best_dist = d;
            }
        }
        if (best_idx2 >= 0) {
            c->out[idx1] = best_idx2;
            c->nmatches++;
        }
    }
}

int afvo_search_for_triangulation(const afvo_tri_job *t, int32_t *match12) {
    m4_ctx c;
    c.t = t; c.out = match12; c.nmatches = 0;
    for (int i = 0; i < t->bow.n1; ++i) match12[i] = -1;
    for_shared_nodes(&t->bow, m4_node, &c);
    return c.nmatches;
}

/* M8 distance inside the M2 control flow, brute force (single node), no orientation check */
int afvo_match_l2_bruteforce(const afvo_l2_job *j, int32_t *match12) {
    u8 *matched2 = (u8 *)calloc((size_t)j->n2 + 1, 1);
    int nm = 0;
    for (int i = 0; i < j->n1; ++i) {
        match12[i] = -1;
        if (j->valid1 && !j->valid1[i]) continue;
        float best1 = FLT_MAX, best2 = FLT_MAX;
        int bi = -1;
        for (int k = 0; k < j->n2; ++k) {
            if (matched2[k] || (j->valid2 && !j->valid2[k])) continue;
            float d = afvo_l2sqr(j->desc1 + (size_t)i * j->dim, j->desc2 + (size_t)k * j->dim, j->dim);
            if (d < best1) { best2 = best1; best1 = d; bi = k; }
            else if (d < best2) best2 = d;
        }
        if (best1 < j->th_low && best1 < j->nnratio * best2) {
            match12[i] = bi;
            matched2[bi] = 1;
            ++nm;
        }
    }
    free(matched2);
    return nm;
}

/* ------------------------------------------------------------------------------------------------
 * projection-guided matching core
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int *cell_ptr; int *cell_idx; } proj_grid;

/* Frame::AssignFeaturesToGrid + PosInGrid (Frame.cc:225-240, 383-394): cell (ix, iy) lists, feature order ascending */
static void build_grid(const afvo_proj_job *j, proj_grid *g) {
    const int nc = j->grid_cols * j->grid_rows;
    g->cell_ptr = (int *)calloc((size_t)nc + 1, sizeof(int));
    g->cell_idx = (int *)malloc(sizeof(int) * (size_t)(j->n > 0 ? j->n : 1));
    int *cell = (int *)malloc(sizeof(int) * (size_t)(j->n > 0 ? j->n : 1));
    for (int i = 0; i < j->n; ++i) {
        const int px = (int)roundf((j->x[i] - j->min_x) * j->grid_inv_w);
        const int py = (int)roundf((j->y[i] - j->min_y) * j->grid_inv_h);
        cell[i] = (px < 0 || px >= j->grid_cols || py < 0 || py >= j->grid_rows) ? -1 : px * j->grid_rows + py;
        if (cell[i] >= 0) g->cell_ptr[cell[i] + 1]++;
    }
    for (int c = 0; c < nc; ++c) g->cell_ptr[c + 1] += g->cell_ptr[c];
    int *fill = (int *)malloc(sizeof(int) * (size_t)nc);
    memcpy(fill, g->cell_ptr, sizeof(int) * (size_t)nc);
    for (int i = 0; i < j->n; ++i)
        if (cell[i] >= 0) g->cell_idx[fill[cell[i]]++] = i;
    free(fill);
    free(cell);
}

/* FeatureMatcher::DescriptorDistance (FeatureMatcher.cc:1508-1531) between query q and feature idx: Descriptor_Distance_Type = float */
static float proj_dist(const afvo_proj_job *j, int q, int idx) {
    if (j->float_dim > 0)
        return afvo_l2sqr((const float *)(const void *)j->qdesc + (size_t)q * j->float_dim, (const float *)(const void *)j->desc + (size_t)idx * j->float_dim,
                          j->float_dim);
    const u8 *a = j->qdesc + (size_t)q * j->desc_bytes, *b = j->desc + (size_t)idx * j->desc_bytes;
    return (float)(j->desc_bytes == 32 ? afvo_hamming256(a, b) : afvo_hamming_bytes(a, b, j->desc_bytes));
}

int afvo_match_projection(const afvo_proj_job *j, int32_t *assign) {
    proj_grid g;
    build_grid(j, &g);
    u8 *occ = (u8 *)calloc((size_t)j->n + 1, 1);
    if (j->occupied) memcpy(occ, j->occupied, (size_t)j->n);
    for (int i = 0; i < j->n; ++i) assign[i] = -1;
    int nmatches = 0;
    int *okey = (int *)malloc(sizeof(int) * (size_t)(2 * j->nq + 2)), *obin = okey + j->nq + 1, on = 0;
    for (int q = 0; q < j->nq; ++q) {
        if (j->qvalid && !j->qvalid[q]) continue;
        const float x = j->qu[q], y = j->qv[q], r = j->qr[q], min_size = j->qmin_size[q], max_size = j->qmax_size[q];
        /* Frame::GetFeaturesInArea (Frame.cc:333-382) */
        const int min_cx = imax(0, (int)floorf((x - j->min_x - r) * j->grid_inv_w));
        if (min_cx >= j->grid_cols) continue;
        const int max_cx = imin(j->grid_cols - 1, (int)ceilf((x - j->min_x + r) * j->grid_inv_w));
        if (max_cx < 0) continue;
        const int min_cy = imax(0, (int)floorf((y - j->min_y - r) * j->grid_inv_h));
        if (min_cy >= j->grid_rows) continue;
        const int max_cy = imin(j->grid_rows - 1, (int)ceilf((y - j->min_y + r) * j->grid_inv_h));
        if (max_cy < 0) continue;
        float best = FLT_MAX, best2 = FLT_MAX, best_size = -1.0f, best_size2 = -1.0f;
        int best_idx = -1;
        for (int ix = min_cx; ix <= max_cx; ++ix)
            for (int iy = min_cy; iy <= max_cy; ++iy) {
                const int c = ix * j->grid_rows + iy;
                for (int k = g.cell_ptr[c]; k < g.cell_ptr[c + 1]; ++k) {
                    const int idx = g.cell_idx[k];
                    if (j->size[idx] < min_size) continue;
                    if (j->size[idx] > max_size) continue;
                    const float dx = j->x[idx] - x, dy = j->y[idx] - y;
                    if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
                    /* matching loop (FeatureMatcher.cc:108-139 / :1362-1386) */
                    if (occ[idx]) continue;
                    if (j->u_right && j->u_right[idx] > 0.0f) { /* :114-119 / :1367-1372 */
                        const float er = fabsf(j->q_ur[q] - j->u_right[idx]);
                        if (er > j->q_er_max[q]) continue;
                    }
                    const float d = proj_dist(j, q, idx);
                    if (d < best) {
                        best2 = best; best = d; best_idx = idx;
                        best_size2 = best_size; best_size = j->size[idx];
                    } else if (j->mode == 0 && d < best2) {
                        best2 = d; best_size2 = j->size[idx];
                    }
                }
            }
        if (best <= j->th_high) {
            if (j->mode == 0) { /* ratio test only if best and second lie in the same scale band (FeatureMatcher.cc:142-148) */
                if ((best_size / best_size2 < j->size_tol) && (best_size / best_size2 > j->inv_size_tol) && (best_size2 > 0.0f)) {
                    if (best > j->nnratio * best2) continue;
                }
            }
            assign[best_idx] = q;
            if (!j->qoccupies || j->qoccupies[q]) occ[best_idx] = 1;
            nmatches++;
            if (j->mode == 1 && j->check_orientation) { /* :1392-1393 */
                okey[on] = best_idx;
                obin[on] = afvo_rotation_bin(j->qangle[q], j->angle[best_idx]);
                on++;
            }
        }
    }
    if (j->mode == 1 && j->check_orientation) { /* filterMatchesWithOrientation (Pt version, :1601-1613) */
        int hs[30], i1, i2, i3;
        memset(hs, 0, sizeof(hs));
        for (int i = 0; i < on; ++i) hs[obin[i]]++;
        afvo_three_maxima(hs, 30, &i1, &i2, &i3);
        for (int i = 0; i < on; ++i) {
            if (obin[i] == i1 || obin[i] == i2 || obin[i] == i3) continue;
            assign[okey[i]] = -1;
            nmatches--;
        }
    }
    free(okey); free(occ); free(g.cell_ptr); free(g.cell_idx);
    return nmatches;
}

int afvo_match_fuse(const afvo_proj_job *j, int32_t *best_out) {
    proj_grid g;
    build_grid(j, &g);
    int nfound = 0;
    for (int q = 0; q < j->nq; ++q) {
        best_out[q] = -1;
        if (j->qvalid && !j->qvalid[q]) continue;
        const float u = j->qu[q], v = j->qv[q], r = j->qr[q], min_size = j->qmin_size[q], max_size = j->qmax_size[q];
        /* KeyFrame::GetFeaturesInArea (KeyFrame.cc:613-652) */
        const int min_cx = imax(0, (int)floorf((u - j->min_x - r) * j->grid_inv_w));
        if (min_cx >= j->grid_cols) continue;
        const int max_cx = imin(j->grid_cols - 1, (int)ceilf((u - j->min_x + r) * j->grid_inv_w));
        if (max_cx < 0) continue;
        const int min_cy = imax(0, (int)floorf((v - j->min_y - r) * j->grid_inv_h));
        if (min_cy >= j->grid_rows) continue;
        const int max_cy = imin(j->grid_rows - 1, (int)ceilf((v - j->min_y + r) * j->grid_inv_h));
        if (max_cy < 0) continue;
        float best = FLT_MAX;
        int best_idx = -1;
        for (int ix = min_cx; ix <= max_cx; ++ix)
            for (int iy = min_cy; iy <= max_cy; ++iy) {
                const int c = ix * j->grid_rows + iy;
                for (int k = g.cell_ptr[c]; k < g.cell_ptr[c + 1]; ++k) {
                    const int idx = g.cell_idx[k];
                    const float dx = j->x[idx] - u, dy = j->y[idx] - v;
                    if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
                    const float sz = j->size[idx];
                    if ((sz < min_size) || (sz > max_size)) continue;             /* :871-873 */
                    const float ex = u - j->x[idx], ey = v - j->y[idx];
                    if (j->inf && j->u_right && j->u_right[idx] >= 0.0f) {        /* :880-894: reprojection error in stereo */
                        const float er = j->q_ur[q] - j->u_right[idx];
                        const float e2 = ex * ex + ey * ey + er * er;
                        if (e2 * j->inf[idx] > 7.8) continue;
                    } else {
                        const float e2 = ex * ex + ey * ey;
                        if (j->inf && e2 * j->inf[idx] > 5.99) continue;          /* :897-898 (float product vs double); no gate in Fuse(Sim3) / SearchBySim3 */
                    }
                    const float d = proj_dist(j, q, idx);
                    if (d < best) { best = d; best_idx = idx; }
                }
            }
        if (best <= j->th_high) { /* caller passes TH_LOW here (:915) */
            best_out[q] = best_idx;
            nfound++;
        }
    }
    free(g.cell_ptr); free(g.cell_idx);
    return nfound;
}

/* SearchBySim3 (FeatureMatcher.cc:1066-1287): both directed searches are the gate-less fuse core with TH_HIGH, then the
 * agreement check (:1268-1284).  j12: queries = KF1 features (their map points) searched in KF2; j21 the other way. */
int afvo_match_sim3(const afvo_proj_job *j12, const afvo_proj_job *j21, int32_t *match12) {
    int32_t *m1 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(j12->nq + 1)), *m2 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(j21->nq + 1));
    afvo_proj_job a = *j12, b = *j21;
    a.inf = NULL; /* gate-less: SearchBySim3 has no chi-square test on the reprojection (:1130-1180, :1210-1262) */
    b.inf = NULL;
    afvo_match_fuse(&a, m1);
    afvo_match_fuse(&b, m2);
    int nfound = 0;
    for (int i1 = 0; i1 < j12->nq; ++i1) {
        match12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && idx2 < j21->nq && m2[idx2] == i1) { match12[i1] = idx2; nfound++; }
    }
    free(m1); free(m2);
    return nfound;
}

/* SearchForInitialization (FeatureMatcher.cc:399-557, active code :480-556).  queries = F1 features (qvalid = octave 0,
 * qu/qv = vbPrevMatched, qr = windowSize, size band [0, F1.maxKeyPtSize]); features = F2 with its grid; th_high carries
 * TH_LOW.  match12[q] = F2 index or -1. */
int afvo_match_initialization(const afvo_proj_job *j, int32_t *match12) {
    proj_grid g;
    build_grid(j, &g);
    float *mdist = (float *)malloc(sizeof(float) * (size_t)(j->n + 1));
    int *m21 = (int *)malloc(sizeof(int) * (size_t)(j->n + 1));
    for (int i = 0; i < j->n; ++i) { mdist[i] = FLT_MAX; m21[i] = -1; }
    for (int q = 0; q < j->nq; ++q) match12[q] = -1;
    int nmatches = 0;
    int hist[30]; memset(hist, 0, sizeof hist);
    int *oq = (int *)malloc(sizeof(int) * (size_t)(2 * j->nq + 2)), *obin = oq + j->nq + 1, on = 0;
    for (int q = 0; q < j->nq; ++q) {
        if (j->qvalid && !j->qvalid[q]) continue;                                  /* :489-491 */
        const float x = j->qu[q], y = j->qv[q], r = j->qr[q], min_size = j->qmin_size[q], max_size = j->qmax_size[q];
        const int min_cx = imax(0, (int)floorf((x - j->min_x - r) * j->grid_inv_w));
        if (min_cx >= j->grid_cols) continue;
        const int max_cx = imin(j->grid_cols - 1, (int)ceilf((x - j->min_x + r) * j->grid_inv_w));
        if (max_cx < 0) continue;
        const int min_cy = imax(0, (int)floorf((y - j->min_y - r) * j->grid_inv_h));
        if (min_cy >= j->grid_rows) continue;
        const int max_cy = imin(j->grid_rows - 1, (int)ceilf((y - j->min_y + r) * j->grid_inv_h));
        if (max_cy < 0) continue;
        float best = FLT_MAX, best2 = FLT_MAX;
        int best_idx = -1;
        for (int ix = min_cx; ix <= max_cx; ++ix)
            for (int iy = min_cy; iy <= max_cy; ++iy) {
                const int c = ix * j->grid_rows + iy;
                for (int k = g.cell_ptr[c]; k < g.cell_ptr[c + 1]; ++k) {
                    const int idx = g.cell_idx[k];
                    if (j->size[idx] < min_size) continue;
                    if (j->size[idx] > max_size) continue;
                    const float dx = j->x[idx] - x, dy = j->y[idx] - y;
                    if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
                    const float d = proj_dist(j, q, idx);
                    if (mdist[idx] <= d) continue;                                 /* :513-514 */
                    if (d < best) { best2 = best; best = d; best_idx = idx; }
                    else if (d < best2) best2 = d;
                }
            }
        if (best <= j->th_high) {                                                  /* :527 */
            if ((float)best < (float)best2 * j->nnratio) {                         /* :529 */
                if (m21[best_idx] >= 0) { match12[m21[best_idx]] = -1; nmatches--; }
                match12[q] = best_idx;
                m21[best_idx] = q;
                mdist[best_idx] = best;
                nmatches++;
                if (j->check_orientation) {
                    const int bin = afvo_rotation_bin(j->qangle[q], j->angle[best_idx]);
                    oq[on] = q; obin[on] = bin; on++; hist[bin]++;
                }
            }
        }
    }
    if (j->check_orientation) {                                                    /* :1615-1629, int flavour */
        int i1, i2, i3;
        afvo_three_maxima(hist, 30, &i1, &i2, &i3);
        for (int e = 0; e < on; ++e) {
            if (obin[e] == i1 || obin[e] == i2 || obin[e] == i3) continue;
            if (match12[oq[e]] >= 0) { nmatches--; match12[oq[e]] = -1; }
        }
    }
    free(oq); free(mdist); free(m21); free(g.cell_ptr); free(g.cell_idx);
    return nmatches;
}

/* ------------------------------------------------------------------------------------------------
 * BoW quantisation (DBoW2 transform, upstream semantics; parity unpinned)
 * ---------------------------------------------------------------------------------------------- */
/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:279-349): N x N distances, per row the sorted row's entry 0.5 * (N - 1), least median
 * wins (strict <: the first row on ties).  Returns the row, *median_out its median; -1 for n = 0. */
static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }
int afvo_distinctive_descriptor(const uint8_t *desc, int n, int desc_bytes, int *median_out) {
    if (n <= 0) { if (median_out) *median_out = 0; return -1; }
    int *D = (int *)malloc(sizeof(int) * (size_t)n * n), *row = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        D[(size_t)i * n + i] = 0;
        for (int j = i + 1; j < n; ++j) {
            const int d = desc_bytes == 32 ? afvo_hamming256(desc + (size_t)i * 32, desc + (size_t)j * 32)
                                           : afvo_hamming_bytes(desc + (size_t)i * desc_bytes, desc + (size_t)j * desc_bytes, desc_bytes);
            D[(size_t)i * n + j] = D[(size_t)j * n + i] = d;
        }
    }
    int best = 0, best_median = 0x7fffffff;
    for (int i = 0; i < n; ++i) {
        memcpy(row, D + (size_t)i * n, sizeof(int) * (size_t)n);
        qsort(row, (size_t)n, sizeof(int), cmp_int);
        const int median = row[(size_t)(0.5 * (n - 1))];
        if (median < best_median) { best_median = median; best = i; }
    }
    free(D); free(row);
    if (median_out) *median_out = best_median;
    return best;
}

static int cmp_float(const void *a, const void *b) {
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}
/* the same with Descriptor_Distance_Type = float distances of float descriptors (DescriptorDistance -> afvo_l2sqr) */
int afvo_distinctive_descriptor_f32(const float *desc, int n, int dim, float *median_out) {
    if (n <= 0) { if (median_out) *median_out = 0.0f; return -1; }
    float *D = (float *)malloc(sizeof(float) * (size_t)n * n), *row = (float *)malloc(sizeof(float) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        D[(size_t)i * n + i] = 0.0f;
        for (int j = i + 1; j < n; ++j) {
            const float d = afvo_l2sqr(desc + (size_t)i * dim, desc + (size_t)j * dim, dim);
            D[(size_t)i * n + j] = D[(size_t)j * n + i] = d;
        }
    }
    int best = 0;
    float best_median = FLT_MAX;
    for (int i = 0; i < n; ++i) {
        memcpy(row, D + (size_t)i * n, sizeof(float) * (size_t)n);
        qsort(row, (size_t)n, sizeof(float), cmp_float);
        const float median = row[(size_t)(0.5 * (n - 1))];
        if (median < best_median) { best_median = median; best = i; }
    }
    free(D); free(row);
    if (median_out) *median_out = best_median;
    return best;
}

/* float descriptors (Vocabulary.cpp:158-187): DBoW2's float classes take the distance as the squared differences evaluated in float,
 * accumulated in double in index order (upstream FSurf64::distance); first minimum wins.  v->desc = float[nnodes][dim], v->desc_bytes = 4 * dim. */
void afvo_bow_transform_f32(const afvo_vocab *v, const float *desc, int n, int levelsup, int32_t *leaf_node, int32_t *node_at_level) {
    const int nid_level = v->L - levelsup, dim = v->desc_bytes / 4;
    const float *nd = (const float *)v->desc;
    for (int i = 0; i < n; ++i) {
        const float *f = desc + (size_t)i * dim;
        int final_id = 0, level = 0, nid = 0;
        while (v->child_ptr[final_id + 1] > v->child_ptr[final_id]) {
            ++level;
            const int b = v->child_ptr[final_id], e = v->child_ptr[final_id + 1];
            int best = -1;
            double best_d = 0.0;
            for (int c = b; c < e; ++c) {
                const int id = v->child_idx[c];
                const float *g = nd + (size_t)id * dim;
                double sqd = 0.0;
                for (int t = 0; t < dim; ++t) {
                    const float df = f[t] - g[t];
                    const float sq = df * df;
                    sqd += (double)sq;
                }
                if (best < 0 || sqd < best_d) { best_d = sqd; best = id; }
            }
            final_id = best;
            if (level == nid_level) nid = final_id;
        }
        leaf_node[i] = final_id;
        node_at_level[i] = nid_level <= 0 ? 0 : nid;
    }
}

void afvo_bow_transform(const afvo_vocab *v, const uint8_t *desc, int n, int levelsup, int32_t *leaf_node, int32_t *node_at_level) {
    const int nid_level = v->L - levelsup;
    for (int i = 0; i < n; ++i) {
        const u8 *f = desc + (size_t)i * v->desc_bytes;
        int final_id = 0, level = 0, nid = 0;
        while (v->child_ptr[final_id + 1] > v->child_ptr[final_id]) { /* !isLeaf() */
            ++level;
            const int b = v->child_ptr[final_id], e = v->child_ptr[final_id + 1];
            int best = v->child_idx[b];
            int best_d = v->desc_bytes == 32 ? afvo_hamming256(f, v->desc + (size_t)best * 32)
                                             : afvo_hamming_bytes(f, v->desc + (size_t)best * v->desc_bytes, v->desc_bytes);
            for (int c = b + 1; c < e; ++c) {
                const int id = v->child_idx[c];
                const int d = v->desc_bytes == 32 ? afvo_hamming256(f, v->desc + (size_t)id * 32)
                                                  : afvo_hamming_bytes(f, v->desc + (size_t)id * v->desc_bytes, v->desc_bytes);
                if (d < best_d) { best_d = d; best = id; }
            }
            final_id = best;
            if (level == nid_level) nid = final_id;
        }
        leaf_node[i] = final_id;
        node_at_level[i] = nid_level <= 0 ? 0 : nid;
    }
}

/* oracle/akaze.c — see akaze.h.  PARITY UNPINNED (libAKAZE fork absent; restated from upstream libAKAZE 1.5).
 * TEST INFRASTRUCTURE ONLY: nothing under anyfeature-vslam_amd/ may link or call this. */
#include "akaze.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int iclamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}
static inline int f_round(float x) { return (int)(x + 0.5f); } /* libAKAZE fRound */

void akz_default_options(akz_options *o) {
    o->omax = 2; o->nsublevels = 4; o->soffset = 1.6f; o->derivative_factor = 1.5f; o->dthreshold = 0.0005f;
    o->min_dthreshold = 0.00001f; o->kcontrast_percentile = 0.7f; o->kcontrast_nbins = 300;
}

/* ---- FED (libAKAZE fed.cpp) ---- */
static int fed_is_prime(int n) {
    if (n <= 1) return 0;
    if (n == 2 || n == 3 || n == 5 || n == 7) return 1;
    if (n % 2 == 0 || n % 3 == 0 || n % 5 == 0 || n % 7 == 0) return 0;
    const int upper = (int)(sqrt((double)n + 1.0));
    for (int d = 11; d <= upper; d += 2)
        if (n % d == 0) return 0;
    return 1;
}
static int fed_tau_internal(int n, float scale, float tau_max, float *tau) {
    if (n <= 0) return 0;
    float tauh[AKZ_MAX_FED];
    const float c = 1.0f / (4.0f * (float)n + 2.0f);
    const float d = scale * tau_max / 2.0f;
    for (int k = 0; k < n; ++k) {
        const float h = cosf(3.14159265358979323846f * (2.0f * (float)k + 1.0f) * c);
        tauh[k] = d / (h * h);
    }
    /* reordering: kappa-cycle with the next prime >= n + 1 */
    const int kappa = n / 2;
    int prime = n + 1;
    while (!fed_is_prime(prime)) prime++;
    for (int k = 0, l = 0; l < n; ++k, ++l) {
        int index;
        while ((index = ((k + 1) * kappa) % prime - 1) >= n) k++;
        tau[l] = tauh[index];
    }
    return n;
}
static int fed_tau_by_process_time(float T, int M, float tau_max, float *tau) {
    const int n = (int)(ceilf(sqrtf(3.0f * T / ((float)M * tau_max) + 0.25f) - 0.5f - 1.0e-8f) + 0.5f);
    if (n > AKZ_MAX_FED) return -1;
    const float scale = 3.0f * T / (tau_max * (float)(n * (n + 1)));
    return fed_tau_internal(n, scale, tau_max, tau);
}

/* cv::getGaussianKernel(n, sigma, CV_32F) for sigma > 0 */
static void gauss_taps(float sigma, int n, float *k) {
    const double s = (double)sigma, scale2x = -0.5 / (s * s);
    double t[64], sum = 0;
    for (int i = 0; i < n; ++i) {
        const double x = i - (n - 1) * 0.5;
        t[i] = exp(scale2x * x * x);
        sum += t[i];
    }
    for (int i = 0; i < n; ++i) k[i] = (float)(t[i] / sum);
}
static int gauss_ksize(float sigma) { /* libAKAZE gaussian_2D_convolution with ksize 0 */
    int ks = (int)ceilf(2.0f * (1.0f + (sigma - 0.8f) / 0.3f));
    if ((ks % 2) == 0) ks += 1;
    return ks;
}

int akz_make_plan(const akz_options *o, int w, int h, akz_plan *p) {
    memset(p, 0, sizeof *p);
    p->w = w; p->h = h;
    int n = 0;
    for (int i = 0; i < o->omax; ++i) {
        const float rfactor = 1.0f / powf(2.0f, (float)i);
        const int lh = (int)((float)h * rfactor), lw = (int)((float)w * rfactor);
        if ((lw < 80 || lh < 40) && i != 0) break; /* smallest octave libAKAZE keeps */
        for (int j = 0; j < o->nsublevels; ++j) {
            if (n >= AKZ_MAX_LEVELS) return -1;
            akz_level_info *L = &p->lv[n++];
            L->w = lw; L->h = lh; L->octave = i; L->sublevel = j;
            L->esigma = o->soffset * powf(2.0f, (float)j / (float)o->nsublevels + (float)i);
            L->etime = 0.5f * (L->esigma * L->esigma);
            L->sigma_size = f_round(L->esigma * o->derivative_factor / powf(2.0f, (float)i));
        }
    }
    p->nlevels = n;
    for (int i = 1; i < n; ++i) {
        const float ttime = p->lv[i].etime - p->lv[i - 1].etime;
        const int ns = fed_tau_by_process_time(ttime, 1, 0.25f, p->lv[i].tau);
        if (ns < 0) return -1;
        p->lv[i].nsteps = ns;
    }
    p->ksize_soffset = gauss_ksize(o->soffset);
    p->ksize_one = gauss_ksize(1.0f);
    if (p->ksize_soffset > 31 || p->ksize_one > 7) return -1;
    gauss_taps(o->soffset, p->ksize_soffset, p->gauss_soffset);
    gauss_taps(1.0f, p->ksize_one, p->gauss_one);
    return 0;
}

/* img.grayImg.convertTo(img_32, CV_32F, 1.0/255.0, 0) (Feature_akaze61.cpp:28) */
void akz_convert(const uint8_t *gray, int stride, int w, int h, float *dst) {
    const float a = (float)(1.0 / 255.0);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) dst[(size_t)y * w + x] = (float)gray[(size_t)y * stride + x] * a;
}

/* cv::GaussianBlur(src, dst, ksize, sigma, sigma, BORDER_REPLICATE) as a symmetric separable float filter: rows then columns,
 * s = k[r] * c + sum_{j=1..r} k[r+j] * (S[+j] + S[-j]) in that order */
void akz_gauss(const float *src, int w, int h, const float *k, int ksize, float *dst) {
    const int r = ksize / 2;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float *row = src + (size_t)y * w;
            float s = k[r] * row[x];
            for (int j = 1; j <= r; ++j) s += k[r + j] * (row[iclamp(x + j, 0, w - 1)] + row[iclamp(x - j, 0, w - 1)]);
            tmp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float s = k[r] * tmp[(size_t)y * w + x];
            for (int j = 1; j <= r; ++j)
                s += k[r + j] * (tmp[(size_t)iclamp(y + j, 0, h - 1) * w + x] + tmp[(size_t)iclamp(y - j, 0, h - 1) * w + x]);
            dst[(size_t)y * w + x] = s;
        }
    free(tmp);
}

/* cv::Scharr(src, dst, CV_32F, 1, 0) / (0, 1), BORDER_REFLECT_101: derivative [-1 0 1], smoothing [3 10 3] */
static inline float scharr_x(const float *s, int w, int h, int x, int y) {
    const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w), ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
    const float t0 = s[(size_t)ym * w + xp] - s[(size_t)ym * w + xm];
    const float t1 = s[(size_t)y * w + xp] - s[(size_t)y * w + xm];
    const float t2 = s[(size_t)yp * w + xp] - s[(size_t)yp * w + xm];
    return 10.0f * t1 + 3.0f * (t0 + t2);
}
static inline float scharr_y(const float *s, int w, int h, int x, int y) {
    const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w), ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
    const float u0 = 10.0f * s[(size_t)ym * w + x] + 3.0f * (s[(size_t)ym * w + xm] + s[(size_t)ym * w + xp]);
    const float u2 = 10.0f * s[(size_t)yp * w + x] + 3.0f * (s[(size_t)yp * w + xm] + s[(size_t)yp * w + xp]);
    return u2 - u0;
}

/* compute_k_percentile(img, perc, gscale = 1, nbins, 0, 0) */
float akz_kcontrast(const float *img, int w, int h, const akz_plan *p, const akz_options *o) {
    const int nbins = o->kcontrast_nbins;
    float *g = (float *)malloc(sizeof(float) * (size_t)w * h);
    akz_gauss(img, w, h, p->gauss_one, p->ksize_one, g);
    float hmax = 0.0f;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            const float lx = scharr_x(g, w, h, x, y), ly = scharr_y(g, w, h, x, y);
            const float m = sqrtf(lx * lx + ly * ly);
            if (m > hmax) hmax = m;
        }
    int *hist = (int *)calloc((size_t)nbins, sizeof(int));
    int npoints = 0;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            const float lx = scharr_x(g, w, h, x, y), ly = scharr_y(g, w, h, x, y);
            const float m = sqrtf(lx * lx + ly * ly);
            if (m != 0.0f) {
                int nbin = (int)floorf((float)nbins * (m / hmax));
                if (nbin == nbins) nbin--;
                hist[nbin]++;
                npoints++;
            }
        }
    const int nthreshold = (int)((float)npoints * o->kcontrast_percentile);
    int k = 0, nelements = 0;
    for (k = 0; nelements < nthreshold && k < nbins; ++k) nelements += hist[k];
    /* hmax == 0 (constant image): upstream would return 0 and turn the conductivity into NaN; keep its own fallback value */
    float kperc = (nelements < nthreshold || hmax == 0.0f) ? 0.03f : hmax * ((float)k / (float)nbins);
    free(hist); free(g);
    return kperc;
}

/* halfsample_image: cv::resize(..., INTER_AREA) by exactly 2 = mean of the 2x2 block, ((a + b) + (c + d)) * 0.25f */
void akz_halfsample(const float *src, int w, int h, float *dst, int dw, int dh) {
    (void)h;
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            const float *r0 = src + (size_t)(2 * y) * w + 2 * x, *r1 = r0 + w;
            dst[(size_t)y * dw + x] = ((r0[0] + r0[1]) + (r1[0] + r1[1])) * 0.25f;
        }
}

/* image_derivatives_scharr x2 + pm_g2: g = 1 / (1 + (Lx^2 + Ly^2) * (1 / k^2)) */
void akz_flow_g2(const float *Ls, int w, int h, float k, float *flow) {
    const float k2inv = 1.0f / (k * k);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float lx = scharr_x(Ls, w, h, x, y), ly = scharr_y(Ls, w, h, x, y);
            flow[(size_t)y * w + x] = 1.0f / (1.0f + (lx * lx + ly * ly) * k2inv);
        }
}

/* nld_step_scalar: out = Lt + 0.5 * tau * div(c grad Lt); one-sided differences on the image border (zero flux) */
void akz_nld_step(const float *L, const float *c, int w, int h, float tau, float *out) {
    const double hs = 0.5 * (double)tau; /* upstream writes 0.5*stepsize*(...): the product is formed in double */
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const size_t i = (size_t)y * w + x;
            const float xpos = x + 1 < w ? (c[i] + c[i + 1]) * (L[i + 1] - L[i]) : 0.0f;
            const float xneg = x > 0 ? (c[i - 1] + c[i]) * (L[i] - L[i - 1]) : 0.0f;
            const float ypos = y + 1 < h ? (c[i] + c[i + w]) * (L[i + w] - L[i]) : 0.0f;
            const float yneg = y > 0 ? (c[i - w] + c[i]) * (L[i] - L[i - w]) : 0.0f;
            const float sum = ((xpos - xneg) + ypos) - yneg; /* upstream: xpos-xneg + ypos-yneg, left to right */
            out[i] = L[i] + (float)(hs * (double)sum);
        }
}

/* compute_scharr_derivatives(src, dst, xorder, yorder, scale) for scale >= 2: 3 sparse taps at distance `scale`,
 * derivative (-1, 0, 1), smoothing (norm, w * norm, norm), w = 10/3, norm = 1 / (2 * scale * (w + 2)); BORDER_REFLECT_101.
 * sepFilter2D runs the row (x) kernel first, then the column kernel. */
static void scharr_scaled(const float *src, int w, int h, int xorder, int scale, float *dst, float *tmp) {
    const float wgt = 10.0f / 3.0f, norm = 1.0f / (2.0f * (float)scale * (wgt + 2.0f)), mid = wgt * norm;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float a = src[(size_t)y * w + reflect101(x - scale, w)], b = src[(size_t)y * w + reflect101(x + scale, w)];
            tmp[(size_t)y * w + x] = xorder ? (b - a) : (mid * src[(size_t)y * w + x] + norm * (a + b));
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float a = tmp[(size_t)reflect101(y - scale, h) * w + x], b = tmp[(size_t)reflect101(y + scale, h) * w + x];
            dst[(size_t)y * w + x] = xorder ? (mid * tmp[(size_t)y * w + x] + norm * (a + b)) : (b - a);
        }
}

void akz_hessian(const float *Ls, int w, int h, int s, float *Lx, float *Ly, float *Ldet) {
    const size_t n = (size_t)w * h;
    float *tmp = (float *)malloc(sizeof(float) * n), *Lxx = (float *)malloc(sizeof(float) * n), *Lyy = (float *)malloc(sizeof(float) * n),
          *Lxy = (float *)malloc(sizeof(float) * n);
    scharr_scaled(Ls, w, h, 1, s, Lx, tmp);
    scharr_scaled(Ls, w, h, 0, s, Ly, tmp);
    scharr_scaled(Lx, w, h, 1, s, Lxx, tmp);
    scharr_scaled(Ly, w, h, 0, s, Lyy, tmp);
    scharr_scaled(Lx, w, h, 0, s, Lxy, tmp);
    const float fs = (float)s, fs2 = (float)(s * s);
    for (size_t i = 0; i < n; ++i) {
        const float lxx = Lxx[i] * fs2, lyy = Lyy[i] * fs2, lxy = Lxy[i] * fs2;
        Lx[i] = Lx[i] * fs;
        Ly[i] = Ly[i] * fs;
        Ldet[i] = lxx * lyy - lxy * lxy;
    }
    free(tmp); free(Lxx); free(Lyy); free(Lxy);
}

void akz_scale_space(const akz_plan *p, const akz_options *o, const uint8_t *gray, int stride, akz_planes *lv, float *k0) {
    const int w = p->w, h = p->h;
    float *img = (float *)malloc(sizeof(float) * (size_t)w * h);
    akz_convert(gray, stride, w, h, img);
    akz_gauss(img, w, h, p->gauss_soffset, p->ksize_soffset, lv[0].Lt);
    memcpy(lv[0].Lsmooth, lv[0].Lt, sizeof(float) * (size_t)w * h);
    float kc = akz_kcontrast(img, w, h, p, o);
    if (k0) *k0 = kc;
    free(img);
    float *flow = (float *)malloc(sizeof(float) * (size_t)w * h), *pong = (float *)malloc(sizeof(float) * (size_t)w * h);
    for (int i = 1; i < p->nlevels; ++i) {
        const akz_level_info *L = &p->lv[i], *P = &p->lv[i - 1];
        const size_t n = (size_t)L->w * L->h;
        if (L->octave > P->octave) {
            akz_halfsample(lv[i - 1].Lt, P->w, P->h, lv[i].Lt, L->w, L->h);
            kc = kc * 0.75f;
        } else {
            memcpy(lv[i].Lt, lv[i - 1].Lt, sizeof(float) * n);
        }
        akz_gauss(lv[i].Lt, L->w, L->h, p->gauss_one, p->ksize_one, lv[i].Lsmooth);
        akz_flow_g2(lv[i].Lsmooth, L->w, L->h, kc, flow);
        for (int j = 0; j < L->nsteps; ++j) {
            akz_nld_step(lv[i].Lt, flow, L->w, L->h, L->tau[j], pong);
            memcpy(lv[i].Lt, pong, sizeof(float) * n);
        }
    }
    free(flow); free(pong);
}

/* ------------------------------------------------------------------------------------------------
 * Feature_Detection (libAKAZE AKAZE::Find_Scale_Space_Extrema / Do_Subpixel_Refinement)
 * ---------------------------------------------------------------------------------------------- */
static int is_out_of_bounds(float px, float py, int sigma_size, int w, int h) {
    const float smax = 10.0f * sqrtf(2.0f); /* MLDB sampling reach */
    const int left_x = f_round(px - smax * (float)sigma_size) - 1, right_x = f_round(px + smax * (float)sigma_size) + 1;
    const int up_y = f_round(py - smax * (float)sigma_size) - 1, down_y = f_round(py + smax * (float)sigma_size) + 1;
    return left_x < 0 || right_x >= w || up_y < 0 || down_y >= h;
}

int akz_level_candidates(const akz_plan *p, const akz_options *o, int level, const float *Ldet, int32_t *out_idx, int cap) {
    const akz_level_info *L = &p->lv[level];
    const int w = L->w, h = L->h;
    const float psize = L->esigma * o->derivative_factor, ratio = powf(2.0f, (float)L->octave);
    const int sigma_size = f_round(psize / ratio);
    int n = 0;
    for (int iy = 1; iy < h - 1; ++iy) {
        const float *m = Ldet + (size_t)(iy - 1) * w, *c = m + w, *q = c + w;
        for (int jx = 1; jx < w - 1; ++jx) {
            const float v = c[jx];
            if (v > o->dthreshold && v >= o->min_dthreshold && v > c[jx - 1] && v > c[jx + 1] && v > m[jx - 1] && v > m[jx] && v > m[jx + 1] &&
                v > q[jx - 1] && v > q[jx] && v > q[jx + 1]) {
                /* a point that fails the border test never changes kpts_aux, so it can be dropped before the ordered pass */
                if (is_out_of_bounds((float)jx, (float)iy, sigma_size, w, h)) continue;
                if (n < cap) out_idx[n] = iy * w + jx;
                n++;
            }
        }
    }
    return n;
}

typedef struct { int *slot; int n, cap; } akz_cell;
typedef struct { akz_cell *cells; int gw, gh; float cell; } akz_grid;
static void grid_init(akz_grid *g, int w, int h, float cell) {
    g->cell = cell; g->gw = (int)((float)w / cell) + 2; g->gh = (int)((float)h / cell) + 2;
    g->cells = (akz_cell *)calloc((size_t)g->gw * g->gh, sizeof(akz_cell));
}
static void grid_free(akz_grid *g) {
    for (int i = 0; i < g->gw * g->gh; ++i) free(g->cells[i].slot);
    free(g->cells);
}
static akz_cell *grid_cell(akz_grid *g, float x, float y) {
    int cx = (int)(x / g->cell), cy = (int)(y / g->cell);
    cx = iclamp(cx, 0, g->gw - 1); cy = iclamp(cy, 0, g->gh - 1);
    return &g->cells[cy * g->gw + cx];
}
static void grid_add(akz_grid *g, float x, float y, int slot) {
    akz_cell *c = grid_cell(g, x, y);
    if (c->n == c->cap) { c->cap = c->cap ? 2 * c->cap : 8; c->slot = (int *)realloc(c->slot, sizeof(int) * (size_t)c->cap); }
    c->slot[c->n++] = slot;
}
static void grid_remove(akz_grid *g, float x, float y, int slot) {
    akz_cell *c = grid_cell(g, x, y);
    for (int i = 0; i < c->n; ++i)
        if (c->slot[i] == slot) { c->slot[i] = c->slot[--c->n]; return; }
}

int akz_find_extrema(const akz_plan *p, const akz_options *o, const akz_planes *lv, akz_keypoint *out, int cap) {
    const int W = p->w, H = p->h;
    int aux_cap = 1 << 14, naux = 0;
    akz_keypoint *aux = (akz_keypoint *)malloc(sizeof(akz_keypoint) * (size_t)aux_cap);
    /* the linear "first entry in kpts_aux order that is close enough" scan of upstream, served from a uniform grid over
     * level-0 coordinates: the first match is the smallest slot among the matches */
    float max_size = 0;
    for (int i = 0; i < p->nlevels; ++i) max_size = fmaxf(max_size, p->lv[i].esigma * o->derivative_factor);
    akz_grid g;
    grid_init(&g, W, H, max_size + 1.0f);
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)W * H / 4 + 64);
    for (int i = 0; i < p->nlevels; ++i) {
        const akz_level_info *L = &p->lv[i];
        const int nc = akz_level_candidates(p, o, i, lv[i].Ldet, cand, W * H / 4);
        const float psize = L->esigma * o->derivative_factor, ratio = powf(2.0f, (float)L->octave);
        for (int k = 0; k < nc; ++k) {
            const int iy = cand[k] / L->w, jx = cand[k] - iy * L->w;
            akz_keypoint pt;
            pt.response = fabsf(lv[i].Ldet[cand[k]]);
            pt.size = psize; pt.octave = L->octave; pt.class_id = i; pt.angle = 0;
            pt.x = (float)jx; pt.y = (float)iy;
            const float sx = pt.x * ratio, sy = pt.y * ratio;
            int first = -1;
            const int cx = (int)(sx / g.cell), cy = (int)(sy / g.cell);
            for (int gy = cy - 1; gy <= cy + 1; ++gy)
                for (int gx = cx - 1; gx <= cx + 1; ++gx) {
                    if (gx < 0 || gy < 0 || gx >= g.gw || gy >= g.gh) continue;
                    const akz_cell *c = &g.cells[gy * g.gw + gx];
                    for (int e = 0; e < c->n; ++e) {
                        const akz_keypoint *a = &aux[c->slot[e]];
                        if ((pt.class_id - 1) == a->class_id || pt.class_id == a->class_id) {
                            const float dist = (sx - a->x) * (sx - a->x) + (sy - a->y) * (sy - a->y);
                            if (dist <= pt.size * pt.size && (first < 0 || c->slot[e] < first)) first = c->slot[e];
                        }
                    }
                }
            int is_extremum = 1, is_repeated = 0;
            if (first >= 0) {
                if (pt.response > aux[first].response) is_repeated = 1;
                else is_extremum = 0;
            }
            if (!is_extremum) continue;
            pt.x = sx; pt.y = sy; /* point.pt.x *= ratio */
            if (!is_repeated) {
                if (naux == aux_cap) { aux_cap *= 2; aux = (akz_keypoint *)realloc(aux, sizeof(akz_keypoint) * (size_t)aux_cap); }
                aux[naux] = pt;
                grid_add(&g, pt.x, pt.y, naux);
                naux++;
            } else {
                grid_remove(&g, aux[first].x, aux[first].y, first);
                aux[first] = pt;
                grid_add(&g, pt.x, pt.y, first);
            }
        }
    }
    free(cand);
    /* "Now filter points with the upper scale level" */
    int n = 0;
    for (int i = 0; i < naux; ++i) {
        const akz_keypoint *a = &aux[i];
        int is_repeated = 0;
        const int cx = (int)(a->x / g.cell), cy = (int)(a->y / g.cell);
        for (int gy = cy - 1; gy <= cy + 1 && !is_repeated; ++gy)
            for (int gx = cx - 1; gx <= cx + 1 && !is_repeated; ++gx) {
                if (gx < 0 || gy < 0 || gx >= g.gw || gy >= g.gh) continue;
                const akz_cell *c = &g.cells[gy * g.gw + gx];
                for (int e = 0; e < c->n; ++e) {
                    const int j = c->slot[e];
                    if (j <= i) continue;
                    const akz_keypoint *b = &aux[j];
                    if ((a->class_id + 1) == b->class_id) {
                        const float dist = (a->x - b->x) * (a->x - b->x) + (a->y - b->y) * (a->y - b->y);
                        if (dist <= a->size * a->size && a->response < b->response) { is_repeated = 1; break; }
                    }
                }
            }
        if (!is_repeated) {
            if (n < cap) out[n] = *a;
            n++;
        }
    }
    grid_free(&g);
    free(aux);
    return n;
}

int akz_subpixel(const akz_plan *p, const akz_planes *lv, akz_keypoint *kpts, int n) {
    int m = 0;
    for (int i = 0; i < n; ++i) {
        akz_keypoint k = kpts[i];
        const akz_level_info *L = &p->lv[k.class_id];
        const float ratio = powf(2.0f, (float)k.octave);
        const int x = f_round(k.x / ratio), y = f_round(k.y / ratio), w = L->w;
        const float *D = lv[k.class_id].Ldet;
#define LD(yy, xx) D[(size_t)(yy) * w + (xx)]
        /* derivatives: upstream mixes double literals into float expressions; the roundings are kept */
        const float Dx = (float)(0.5 * (double)(LD(y, x + 1) - LD(y, x - 1)));
        const float Dy = (float)(0.5 * (double)(LD(y + 1, x) - LD(y - 1, x)));
        const float Dxx = (float)((double)(LD(y, x + 1) + LD(y, x - 1)) - 2.0 * (double)LD(y, x));
        const float Dyy = (float)((double)(LD(y + 1, x) + LD(y - 1, x)) - 2.0 * (double)LD(y, x));
        const float Dxy = (float)(0.25 * (double)(LD(y + 1, x + 1) + LD(y - 1, x - 1)) - 0.25 * (double)(LD(y - 1, x + 1) + LD(y + 1, x - 1)));
#undef LD
        /* cv::solve(A, b, dst, DECOMP_LU) for a 2x2 float system: Cramer's rule in double */
        const double det = (double)Dxx * (double)Dyy - (double)Dxy * (double)Dxy;
        if (det == 0.0) continue; /* dst stays at its previous content upstream; a singular system has no refinement: dropped */
        const double b0 = -(double)Dx, b1 = -(double)Dy, inv = 1.0 / det;
        const float d0 = (float)((b0 * (double)Dyy - b1 * (double)Dxy) * inv);
        const float d1 = (float)((b1 * (double)Dxx - b0 * (double)Dxy) * inv);
        if (fabsf(d0) <= 1.0f && fabsf(d1) <= 1.0f) {
            const int power = (int)powf(2.0f, (float)L->octave);
            k.x = ((float)x + d0) * (float)power;
            k.y = ((float)y + d1) * (float)power;
            k.angle = 0.0f;
            k.size = k.size * 2.0f;
            kpts[m++] = k;
        }
    }
    return m;
}

/* ------------------------------------------------------------------------------------------------
 * Compute_Descriptors
 * ---------------------------------------------------------------------------------------------- */
#define AKZ_PI 3.14159265358979323846 /* CV_PI */

/* atan(z), z >= 0 (or +inf / NaN), in double: argument reduction at tan(pi/8), tan(3pi/8), Taylor to t^23 */
static double atan_pos_f64(double z) {
    const double pio2 = 1.5707963267948966, pio4 = 0.78539816339744831;
    double base, t;
    if (z > 2.414213562373095) { base = pio2; t = -1.0 / z; }
    else if (z > 0.4142135623730950) { base = pio4; t = (z - 1.0) / (z + 1.0); }
    else { base = 0.0; t = z; }
    const double w = t * t;
    double s = 1.0 / 23.0;
    s = 1.0 / 21.0 - w * s;
    s = 1.0 / 19.0 - w * s;
    s = 1.0 / 17.0 - w * s;
    s = 1.0 / 15.0 - w * s;
    s = 1.0 / 13.0 - w * s;
    s = 1.0 / 11.0 - w * s;
    s = 1.0 / 9.0 - w * s;
    s = 1.0 / 7.0 - w * s;
    s = 1.0 / 5.0 - w * s;
    s = 1.0 / 3.0 - w * s;
    s = 1.0 - w * s;
    return base + t * s;
}
static float atanf_det(float z) { return (float)atan_pos_f64((double)z); }

/* libAKAZE get_angle: quadrant-wise atanf, result in [0, 2 pi) */
float akz_get_angle(float x, float y) {
    if (x >= 0 && y >= 0) return atanf_det(y / x);
    if (x < 0 && y >= 0) return (float)(AKZ_PI - (double)atanf_det(-y / x));
    if (x < 0 && y < 0) return (float)(AKZ_PI + (double)atanf_det(y / x));
    if (x >= 0 && y < 0) return (float)(2.0 * AKZ_PI - (double)atanf_det(-y / x));
    return 0.0f;
}

/* same Cody-Waite + Taylor cos/sin as oracle/afvo.c (shared with the kernels) */
static void sincos_f64(double t, double *c_out, double *s_out) {
    const double two_over_pi = 0.63661977236758138;
    const double pio2_hi = 1.5707963267341256e+00, pio2_lo = 6.0771005065061922e-11;
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

/* SURF's 7 x 7 Gaussian (sigma 2.5) weight table used by Compute_Main_Orientation */
static const float k_gauss25[7][7] = {
    {0.02546481f, 0.02350698f, 0.01849125f, 0.01239505f, 0.00708017f, 0.00344629f, 0.00142946f},
    {0.02350698f, 0.02169968f, 0.01706957f, 0.01144208f, 0.00653582f, 0.00318132f, 0.00131956f},
    {0.01849125f, 0.01706957f, 0.01342740f, 0.00900066f, 0.00514126f, 0.00250252f, 0.00103800f},
    {0.01239505f, 0.01144208f, 0.00900066f, 0.00603332f, 0.00344629f, 0.00167749f, 0.00069579f},
    {0.00708017f, 0.00653582f, 0.00514126f, 0.00344629f, 0.00196855f, 0.00095820f, 0.00039744f},
    {0.00344629f, 0.00318132f, 0.00250252f, 0.00167749f, 0.00095820f, 0.00046640f, 0.00019346f},
    {0.00142946f, 0.00131956f, 0.00103800f, 0.00069579f, 0.00039744f, 0.00019346f, 0.00008024f}};

void akz_main_orientation(const akz_plan *p, const akz_planes *lv, akz_keypoint *kp) {
    static const int id[13] = {6, 5, 4, 3, 2, 1, 0, 1, 2, 3, 4, 5, 6};
    const int level = kp->class_id;
    const akz_level_info *L = &p->lv[level];
    const float ratio = (float)(1 << L->octave);
    const int s = f_round((float)(0.5 * (double)kp->size / (double)ratio));
    const float xf = kp->x / ratio, yf = kp->y / ratio;
    float resX[109], resY[109], Ang[109];
    int idx = 0;
    for (int i = -6; i <= 6; ++i)
        for (int j = -6; j <= 6; ++j)
            if (i * i + j * j < 36) {
                const int iy = iclamp(f_round(yf + (float)(j * s)), 0, L->h - 1), ix = iclamp(f_round(xf + (float)(i * s)), 0, L->w - 1);
                const float gw = k_gauss25[id[i + 6]][id[j + 6]];
                resX[idx] = gw * lv[level].Lx[(size_t)iy * L->w + ix];
                resY[idx] = gw * lv[level].Ly[(size_t)iy * L->w + ix];
                Ang[idx] = akz_get_angle(resX[idx], resY[idx]);
                ++idx;
            }
    float max = 0.0f, angle = kp->angle;
    for (float ang1 = 0; (double)ang1 < 2.0 * AKZ_PI; ang1 += 0.15f) {
        const float ang2 = (float)((double)ang1 + AKZ_PI / 3.0 > 2.0 * AKZ_PI ? (double)ang1 - 5.0 * AKZ_PI / 3.0 : (double)ang1 + AKZ_PI / 3.0);
        float sumX = 0.f, sumY = 0.f;
        for (int k = 0; k < 109; ++k) {
            const float ang = Ang[k];
            if (ang1 < ang2 && ang1 < ang && ang < ang2) { sumX += resX[k]; sumY += resY[k]; }
            else if (ang2 < ang1 && ((ang > 0 && ang < ang2) || (ang > ang1 && (double)ang < 2.0 * AKZ_PI))) { sumX += resX[k]; sumY += resY[k]; }
        }
        if (sumX * sumX + sumY * sumY > max) {
            max = sumX * sumX + sumY * sumY;
            angle = akz_get_angle(sumX, sumY);
        }
    }
    kp->angle = angle;
}

static inline int32_t toggle_flt(int32_t x) { return x ^ (x < 0 ? 0x7fffffff : 0); } /* CV_TOGGLE_FLT */

void akz_mldb(const akz_plan *p, const akz_planes *lv, const akz_keypoint *kp, uint8_t *desc) {
    const int level = kp->class_id;
    const akz_level_info *L = &p->lv[level];
    const int w = L->w, h = L->h;
    const float ratio = (float)(1 << kp->octave);
    const float scale = (float)f_round(0.5f * kp->size / ratio);
    const float xf = kp->x / ratio, yf = kp->y / ratio;
    double cd, sd;
    sincos_f64((double)kp->angle, &cd, &sd);
    const float co = (float)cd, si = (float)sd;
    const int pattern_size = 10;
    static const int sample_step[3] = {10, 7, 5}; /* ceil(10 * {1, 2/3, 1/2}) */
    memset(desc, 0, 61);
    int dpos = 0;
    for (int lvl = 0; lvl < 3; ++lvl) {
        const int step = sample_step[lvl], val_count = (lvl + 2) * (lvl + 2);
        float values[16 * 3];
        int valpos = 0;
        for (int i = -pattern_size; i < pattern_size; i += step)
            for (int j = -pattern_size; j < pattern_size; j += step) {
                float di = 0, dx = 0, dy = 0;
                int nsamples = 0;
                for (int k = i; k < i + step; ++k)
                    for (int l = j; l < j + step; ++l) {
                        const float sample_y = yf + ((float)l * co * scale + (float)k * si * scale);
                        const float sample_x = xf + (-(float)l * si * scale + (float)k * co * scale);
                        const int y1 = iclamp(f_round(sample_y), 0, h - 1), x1 = iclamp(f_round(sample_x), 0, w - 1);
                        const size_t o = (size_t)y1 * w + x1;
                        const float ri = lv[level].Lt[o], rx = lv[level].Lx[o], ry = lv[level].Ly[o];
                        di += ri;
                        const float rry = rx * co + ry * si, rrx = -rx * si + ry * co;
                        dx += rrx;
                        dy += rry;
                        nsamples++;
                    }
                di /= (float)nsamples; dx /= (float)nsamples; dy /= (float)nsamples;
                values[valpos] = di; values[valpos + 1] = dx; values[valpos + 2] = dy;
                valpos += 3;
            }
        /* MLDB_Binary_Comparisons: per channel, all ordered pairs (i < j): bit = value_i > value_j (integer compare of toggled floats) */
        int32_t iv[16 * 3];
        memcpy(iv, values, sizeof(float) * (size_t)val_count * 3);
        for (int i = 0; i < val_count * 3; ++i) iv[i] = toggle_flt(iv[i]);
        for (int pos = 0; pos < 3; ++pos)
            for (int i = 0; i < val_count; ++i) {
                const int32_t ival = iv[3 * i + pos];
                for (int j = i + 1; j < val_count; ++j) {
                    if (ival > iv[3 * j + pos]) desc[dpos >> 3] |= (uint8_t)(1 << (dpos & 7));
                    dpos++;
                }
            }
    }
}

void akz_compute_descriptors(const akz_plan *p, const akz_planes *lv, akz_keypoint *kpts, int n, uint8_t *desc) {
    for (int i = 0; i < n; ++i) {
        akz_main_orientation(p, lv, &kpts[i]);
        akz_mldb(p, lv, &kpts[i], desc + (size_t)i * 61);
    }
}

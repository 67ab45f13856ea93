/*
 * lk_oracle.c -- CPU restatement of cv::calcOpticalFlowPyrLK as Tracking::GetSceneFlowObj calls it
 * (src/Tracking.cc:896: winSize 22 x 22, maxLevel 5, TermCriteria(COUNT | EPS, 20, 0.01), flags 0, minEigThreshold 1e-4).
 * TEST INFRASTRUCTURE ONLY (checker of amos-slam_amd/csrc/amos_flow.hip's k_lk_*).
 *
 * PARITY UNPINNED, and more than the other OpenCV stages: this follows OpenCV 4.5's lkpyramid.cpp / pyramids.cpp as published
 * (buildOpticalFlowPyramid: pyrDown 5 x 5 [1 4 6 4 1] with (sum + 128) >> 8, REFLECT_101 image borders of winSize pixels,
 * zero borders for the derivatives; calcScharrDeriv; the 14-bit fixed-point bilinear window with CV_DESCALE; float 2 x 2 system,
 * minimum-eigenvalue test, iteration and early-exit rules), but OpenCV accumulates A11, A12, A22, b1, b2 in float through the SIMD
 * lanes of its build (four partial sums + reduce on SSE / AVX, multiply-add fused or not), so the last bits of every iterate are
 * a property of that binary.  Here the accumulation order is DEFINED as the scalar path's: row by row, left to right.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LK_W_BITS 14
#define LK_DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

static int lk_refl101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * n - 2 - i;
    }
    return i;
}

static int lk_round(float v) { return (int)lrintf(v); } /* cvRound: nearest even */

typedef struct {
    int w, h, pw, ph;  /* level size, padded size (w + 2 win, h + 2 win) */
    uint8_t *img;      /* padded, REFLECT_101 */
    int16_t *deriv;    /* padded, zeros; interleaved (dx, dy) */
} lk_level;

static void lk_pad_image(const uint8_t *src, size_t stride, int w, int h, int win, uint8_t *dst)
{
    const int pw = w + 2 * win;
    for (int y = -win; y < h + win; y++)
        for (int x = -win; x < w + win; x++) dst[(size_t)(y + win) * pw + x + win] = src[(size_t)lk_refl101(y, h) * stride + lk_refl101(x, w)];
}

/* cv::pyrDown, 8-bit: separable [1 4 6 4 1], BORDER_REFLECT_101 on the source index, (sum + 128) >> 8 */
static void lk_pyr_down(const uint8_t *src, size_t stride, int w, int h, uint8_t *dst, int dw, int dh)
{
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int sum = 0;
            for (int j = 0; j < 5; j++) {
                const uint8_t *row = src + (size_t)lk_refl101(2 * y + j - 2, h) * stride;
                int hs = 0;
                for (int i = 0; i < 5; i++) hs += k[i] * row[lk_refl101(2 * x + i - 2, w)];
                sum += k[j] * hs;
            }
            dst[(size_t)y * dw + x] = (uint8_t)((sum + 128) >> 8);
        }
}

/* calcScharrDeriv (lkpyramid.cpp): dx = [-1 0 1] x [3 10 3]^T, dy = [3 10 3] x [-1 0 1]^T, REFLECT_101, shorts */
static void lk_scharr(const uint8_t *img, int w, int h, int pw, int win, int16_t *deriv)
{
    memset(deriv, 0, sizeof(int16_t) * 2 * (size_t)pw * (h + 2 * win));
    int *t0 = (int *)malloc(sizeof(int) * (w + 2)), *t1 = (int *)malloc(sizeof(int) * (w + 2));
    for (int y = 0; y < h; y++) {
        const uint8_t *r0 = img + (size_t)(lk_refl101(y - 1, h) + win) * pw + win, *r1 = img + (size_t)(y + win) * pw + win,
                      *r2 = img + (size_t)(lk_refl101(y + 1, h) + win) * pw + win;
        for (int x = 0; x < w; x++) {
            t0[x + 1] = (int16_t)((r0[x] + r2[x]) * 3 + r1[x] * 10);
            t1[x + 1] = (int16_t)(r2[x] - r0[x]);
        }
        t0[0] = t0[w > 1 ? 2 : 1]; t0[w + 1] = t0[w > 1 ? w - 1 : 1];
        t1[0] = t1[w > 1 ? 2 : 1]; t1[w + 1] = t1[w > 1 ? w - 1 : 1];
        int16_t *d = deriv + 2 * ((size_t)(y + win) * pw + win);
        for (int x = 0; x < w; x++) {
            d[2 * x] = (int16_t)(t0[x + 2] - t0[x]);
            d[2 * x + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
    free(t0); free(t1);
}

/* levels 0 .. return value of buildOpticalFlowPyramid(img, winSize, maxLevel) */
static int lk_build(const uint8_t *gray, size_t stride, int w, int h, int win, int max_level, lk_level *lv, int with_deriv)
{
    int level = 0;
    for (;; level++) {
        lk_level *L = &lv[level];
        L->w = w; L->h = h; L->pw = w + 2 * win; L->ph = h + 2 * win;
        L->img = (uint8_t *)malloc((size_t)L->pw * L->ph);
        if (level == 0) lk_pad_image(gray, stride, w, h, win, L->img);
        else {
            uint8_t *tmp = (uint8_t *)malloc((size_t)w * h);
            const lk_level *P = &lv[level - 1];
            lk_pyr_down(P->img + (size_t)win * P->pw + win, (size_t)P->pw, P->w, P->h, tmp, w, h);
            lk_pad_image(tmp, (size_t)w, w, h, win, L->img);
            free(tmp);
        }
        L->deriv = NULL;
        if (with_deriv) {
            L->deriv = (int16_t *)malloc(sizeof(int16_t) * 2 * (size_t)L->pw * L->ph);
            lk_scharr(L->img, w, h, L->pw, win, L->deriv);
        }
        if (level == max_level) break;
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= win || h <= win) break;
    }
    return level;
}

static void lk_free(lk_level *lv, int n)
{
    for (int i = 0; i <= n; i++) { free(lv[i].img); free(lv[i].deriv); }
}

/* Returns the number of pyramid levels used - 1 (buildOpticalFlowPyramid's return), or a negative error. */
int orc_lk_track(const uint8_t *prev, size_t prev_stride, const uint8_t *next, size_t next_stride, int w, int h, const float *prev_pts, int n,
                 int win, int max_level, int max_count, double epsilon, float min_eig_threshold, float *next_pts, uint8_t *status, float *err)
{
    if (win < 3 || win > 31 || max_level < 0 || max_level > 7 || w <= win || h <= win) return -1;
    lk_level P[8], N[8];
    const int lp = lk_build(prev, prev_stride, w, h, win, max_level, P, 1), ln = lk_build(next, next_stride, w, h, win, max_level, N, 0);
    const int top = lp < ln ? lp : ln;
    if (max_count < 0) max_count = 0;
    if (max_count > 100) max_count = 100;
    if (epsilon < 0) epsilon = 0;
    if (epsilon > 10) epsilon = 10;
    epsilon *= epsilon;
    const float half = (win - 1) * 0.5f, scale20 = 1.f / (1 << 20);
    short *Iw = (short *)malloc(sizeof(short) * win * win), *dIw = (short *)malloc(sizeof(short) * 2 * win * win);
    for (int i = 0; i < n; i++) { status[i] = 1; if (err) err[i] = 0; }
    for (int level = top; level >= 0; level--) {
        const lk_level *I = &P[level], *J = &N[level];
        const int pw = I->pw;
        for (int pt = 0; pt < n; pt++) {
            float px = prev_pts[2 * pt] * (float)(1. / (1 << level)), py = prev_pts[2 * pt + 1] * (float)(1. / (1 << level));
            float nx, ny;
            if (level == top) { nx = px; ny = py; }
            else { nx = next_pts[2 * pt] * 2.f; ny = next_pts[2 * pt + 1] * 2.f; }
            next_pts[2 * pt] = nx; next_pts[2 * pt + 1] = ny;
            px -= half; py -= half;
            const int ipx = (int)floorf(px), ipy = (int)floorf(py);
            if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
                if (level == 0) { status[pt] = 0; if (err) err[pt] = 0; }
                continue;
            }
            float a = px - ipx, b = py - ipy;
            int iw00 = lk_round((1.f - a) * (1.f - b) * (1 << LK_W_BITS)), iw01 = lk_round(a * (1.f - b) * (1 << LK_W_BITS)),
                iw10 = lk_round((1.f - a) * b * (1 << LK_W_BITS)), iw11 = (1 << LK_W_BITS) - iw00 - iw01 - iw10;
            float iA11 = 0, iA12 = 0, iA22 = 0;
            for (int y = 0; y < win; y++) {
                const uint8_t *src = I->img + (size_t)(y + ipy + win) * pw + ipx + win;
                const int16_t *dsrc = I->deriv + 2 * ((size_t)(y + ipy + win) * pw + ipx + win);
                for (int x = 0; x < win; x++, dsrc += 2) {
                    const int ival = LK_DESCALE(src[x] * iw00 + src[x + 1] * iw01 + src[x + pw] * iw10 + src[x + pw + 1] * iw11, LK_W_BITS - 5);
                    const int ixval = LK_DESCALE(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[2 * pw] * iw10 + dsrc[2 * pw + 2] * iw11, LK_W_BITS);
                    const int iyval = LK_DESCALE(dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[2 * pw + 1] * iw10 + dsrc[2 * pw + 3] * iw11, LK_W_BITS);
                    Iw[y * win + x] = (short)ival; dIw[2 * (y * win + x)] = (short)ixval; dIw[2 * (y * win + x) + 1] = (short)iyval;
                    iA11 += (float)(ixval * ixval); iA12 += (float)(ixval * iyval); iA22 += (float)(iyval * iyval);
                }
            }
            const float A11 = iA11 * scale20, A12 = iA12 * scale20, A22 = iA22 * scale20;
            float D = A11 * A22 - A12 * A12;
            const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
            if (minEig < min_eig_threshold || D < 1.1920929e-7f) {
                if (level == 0) status[pt] = 0;
                continue;
            }
            D = 1.f / D;
            nx -= half; ny -= half;
            float pdx = 0, pdy = 0;
            for (int j = 0; j < max_count; j++) {
                const int inx = (int)floorf(nx), iny = (int)floorf(ny);
                if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
                    if (level == 0) status[pt] = 0;
                    break;
                }
                a = nx - inx; b = ny - iny;
                iw00 = lk_round((1.f - a) * (1.f - b) * (1 << LK_W_BITS)); iw01 = lk_round(a * (1.f - b) * (1 << LK_W_BITS));
                iw10 = lk_round((1.f - a) * b * (1 << LK_W_BITS)); iw11 = (1 << LK_W_BITS) - iw00 - iw01 - iw10;
                float ib1 = 0, ib2 = 0;
                for (int y = 0; y < win; y++) {
                    const uint8_t *Jp = J->img + (size_t)(y + iny + win) * pw + inx + win;
                    for (int x = 0; x < win; x++) {
                        const int diff = LK_DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + pw] * iw10 + Jp[x + pw + 1] * iw11, LK_W_BITS - 5) - Iw[y * win + x];
                        ib1 += (float)(diff * dIw[2 * (y * win + x)]); ib2 += (float)(diff * dIw[2 * (y * win + x) + 1]);
                    }
                }
                const float b1 = ib1 * scale20, b2 = ib2 * scale20;
                const float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
                nx += dx; ny += dy;
                next_pts[2 * pt] = nx + half; next_pts[2 * pt + 1] = ny + half;
                if ((double)dx * dx + (double)dy * dy <= epsilon) break;
                if (j > 0 && fabsf(dx + pdx) < 0.01 && fabsf(dy + pdy) < 0.01) {
                    next_pts[2 * pt] -= dx * 0.5f; next_pts[2 * pt + 1] -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
            if (status[pt] && err && level == 0) {
                const float ex = next_pts[2 * pt] - half, ey = next_pts[2 * pt + 1] - half;
                const int iex = (int)floorf(ex), iey = (int)floorf(ey);
                if (iex < -win || iex >= J->w || iey < -win || iey >= J->h) { status[pt] = 0; continue; }
                const float aa = ex - iex, bb = ey - iey;
                iw00 = lk_round((1.f - aa) * (1.f - bb) * (1 << LK_W_BITS)); iw01 = lk_round(aa * (1.f - bb) * (1 << LK_W_BITS));
                iw10 = lk_round((1.f - aa) * bb * (1 << LK_W_BITS)); iw11 = (1 << LK_W_BITS) - iw00 - iw01 - iw10;
                float errval = 0.f;
                for (int y = 0; y < win; y++) {
                    const uint8_t *Jp = J->img + (size_t)(y + iey + win) * pw + iex + win;
                    for (int x = 0; x < win; x++) {
                        const int diff = LK_DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + pw] * iw10 + Jp[x + pw + 1] * iw11, LK_W_BITS - 5) - Iw[y * win + x];
                        errval += fabsf((float)diff);
                    }
                }
                err[pt] = errval * 1.f / (32 * win * win);
            }
        }
    }
    free(Iw); free(dIw);
    lk_free(P, lp); lk_free(N, ln);
    return top;
}

/* one pyramid level image (unpadded) and its derivative, for the unit tests */
int orc_lk_pyramid_level(const uint8_t *gray, size_t stride, int w, int h, int win, int max_level, int level, uint8_t *img, int16_t *deriv, int *lw, int *lh)
{
    lk_level P[8];
    const int top = lk_build(gray, stride, w, h, win, max_level, P, 1);
    if (level > top) { lk_free(P, top); return -1; }
    const lk_level *L = &P[level];
    *lw = L->w; *lh = L->h;
    for (int y = 0; y < L->h; y++) {
        if (img) memcpy(img + (size_t)y * L->w, L->img + (size_t)(y + win) * L->pw + win, (size_t)L->w);
        if (deriv) memcpy(deriv + 2 * (size_t)y * L->w, L->deriv + 2 * ((size_t)(y + win) * L->pw + win), sizeof(int16_t) * 2 * (size_t)L->w);
    }
    lk_free(P, top);
    return top;
}

/* ---------------------------------------------------------------- cv::cvtColor(COLOR_BGR2Lab), 8-bit (src/cluster.cc:310) ----
 * OpenCV 4.5's RGB2Lab_b: gamma table (sRGB, 3 extra bits), 12-bit XYZ coefficients divided by the D65 white point, cube-root table
 * of 3072 entries with 15 fractional bits, L = (296 fY - 1336934 + 2^14) >> 15, a = (500 (fX - fY) + 128 * 2^15 + 2^14) >> 15,
 * b = (200 (fY - fZ) + 128 * 2^15 + 2^14) >> 15, saturated to 8 bits.  PARITY UNPINNED: OpenCV builds the two tables with its softfloat
 * pow / cbrt; here they come from libm in double (an entry can differ by one where a value falls within an ulp of a rounding
 * boundary).  blue_idx = 0 for BGR input, 2 for RGB. */
static int lab_tables_ready = 0;
static uint16_t lab_gamma_tab[256], lab_cbrt_tab[3072];
static int lab_coeffs[9];

static void lab_init(void)
{
    if (lab_tables_ready) return;
    for (int i = 0; i < 256; i++) {
        const float x = (float)i / 255.f;
        const double g = x <= 0.04045f ? (double)x / 12.92 : pow(((double)x + 0.055) / 1.055, 2.4);
        lab_gamma_tab[i] = (uint16_t)lrint(255.0 * 8.0 * g);
    }
    for (int i = 0; i < 3072; i++) {
        const float x = (float)i / (255.f * 8.f);
        const double f = x < 0.008856f ? (double)x * 7.787 + 0.13793103448275862 : cbrt((double)x);
        lab_cbrt_tab[i] = (uint16_t)lrint(32768.0 * f);
    }
    static const double xyz[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
    static const double white[3] = {0.950456, 1., 1.088754};
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) lab_coeffs[3 * i + k] = (int)lrint(4096.0 * xyz[3 * i + k] / white[i]);  /* by (R, G, B) */
    lab_tables_ready = 1;
}

void orc_lab_tables(uint16_t *gamma, uint16_t *cbrt_tab, int32_t *coeffs)
{
    lab_init();
    if (gamma) memcpy(gamma, lab_gamma_tab, sizeof(lab_gamma_tab));
    if (cbrt_tab) memcpy(cbrt_tab, lab_cbrt_tab, sizeof(lab_cbrt_tab));
    if (coeffs) for (int i = 0; i < 9; i++) coeffs[i] = lab_coeffs[i];
}

static uint8_t lab_sat(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

void orc_bgr_to_lab(const uint8_t *src, size_t n_px, int blue_idx, uint8_t *dst)
{
    lab_init();
    const int Lscale = (116 * 255 + 50) / 100, Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    for (size_t i = 0; i < n_px; i++) {
        const int R = lab_gamma_tab[src[3 * i + (blue_idx ^ 2)]], G = lab_gamma_tab[src[3 * i + 1]], B = lab_gamma_tab[src[3 * i + blue_idx]];
        const int fX = lab_cbrt_tab[LK_DESCALE(R * lab_coeffs[0] + G * lab_coeffs[1] + B * lab_coeffs[2], 12)];
        const int fY = lab_cbrt_tab[LK_DESCALE(R * lab_coeffs[3] + G * lab_coeffs[4] + B * lab_coeffs[5], 12)];
        const int fZ = lab_cbrt_tab[LK_DESCALE(R * lab_coeffs[6] + G * lab_coeffs[7] + B * lab_coeffs[8], 12)];
        dst[3 * i] = lab_sat(LK_DESCALE(Lscale * fY + Lshift, 15));
        dst[3 * i + 1] = lab_sat(LK_DESCALE(500 * (fX - fY) + 128 * (1 << 15), 15));
        dst[3 * i + 2] = lab_sat(LK_DESCALE(200 * (fY - fZ) + 128 * (1 << 15), 15));
    }
}

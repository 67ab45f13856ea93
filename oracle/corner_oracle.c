/* corner_oracle.c -- CPU restatement of the corner source of Tracking::GetSceneFlowObj (src/Tracking.cc:894-895):
 *
 *     cv::goodFeaturesToTrack(imlast, prepoint, 1000, 0.01, 8, cv::Mat(), 3, true, 0.04);
 *     cv::cornerSubPix(imlast, prepoint, cv::Size(10, 10), cv::Size(-1, -1), cv::TermCriteria(ITER | EPS, 20, 0.03));
 *
 * TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline may use it; the product never does).
 *
 * PARITY UNPINNED: both functions live in OpenCV 4.5 (absent from this image; the reference ships no vectors for them).  The
 * restatement follows the published implementation (modules/imgproc/src/featureselect.cpp, corner.cpp, cornersubpix.cpp,
 * samplers.cpp) with every float operation rounded on its own, in the order the scalar code writes them; OpenCV's SIMD paths may
 * fuse multiply-adds or reorder sums depending on the build, so the last bit of a Harris response belongs to that binary.
 *
 * goodFeaturesToTrack(useHarrisDetector = true, blockSize 3, gradientSize 3):
 *   1. cornerHarris: Dx, Dy = Sobel 3 x 3 of the 8-bit image as float, scaled by 1 / (4 * blockSize * 255) (the scale multiplies the
 *      smoothing kernel [1 2 1]), BORDER_REFLECT_101; cov = (Dx Dx, Dx Dy, Dy Dy); unnormalised 3 x 3 box sums (double accumulator,
 *      one rounding to float), BORDER_REFLECT_101; R = a c - b b - k (a + c) (a + c) in float.
 *   2. threshold at (float)(max R * qualityLevel) to zero; a candidate is a pixel of rows 1 .. h - 2, columns 1 .. w - 2 whose value is
 *      non-zero and equals the maximum of its 3 x 3 neighbourhood (dilate with the default border: outside pixels do not take part).
 *   3. candidates sorted by value, descending; equal values: the LATER pixel first (the library compares pointers into the image).
 *   4. greedy selection in that order: a candidate is kept unless a kept one lies within minDistance (squared distance < minDistance^2),
 *      looked up through a grid of cvRound(minDistance)-pixel cells; stops at maxCorners.
 * cornerSubPix: per corner at most max_iter iterations of  c += G^-1 b  over a (2 win + 1)^2 window with weights
 *   exp(-(i / win)^2) exp(-(j / win)^2), gradients from a bilinear (2 win + 3)^2 patch around the current position (getRectSubPix with
 *   replicated borders), sums in double in row-major order; stops when the step is below eps, the determinant vanishes or the point
 *   leaves the image; a point that moved further than the window is put back.
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int refl101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

/* cornerHarris(src, dst, 3, 3, k, BORDER_DEFAULT) for an 8-bit image; dst: w * h floats */
void orc_corner_harris(const uint8_t *img, size_t stride, int w, int h, double k, float *dst)
{
    const double scale = 1.0 / (4.0 * 3.0 * 255.0);
    const float k1 = (float)(1.0 * scale), k2 = (float)(2.0 * scale);  /* [1 2 1] * scale as a float kernel (Mat *= double) */
    const float kf = (float)k;
    float *dx = (float *)malloc(sizeof(float) * (size_t)w * h), *dy = (float *)malloc(sizeof(float) * (size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t *r0 = img + (size_t)refl101(y - 1, h) * stride, *r1 = img + (size_t)y * stride, *r2 = img + (size_t)refl101(y + 1, h) * stride;
        for (int x = 0; x < w; x++) {
            const int xm = refl101(x - 1, w), xp = refl101(x + 1, w);
            /* Dx: rows filtered by [-1 0 1] (exact), columns by [s 2s s]: (S0 + S2) * f1 + S1 * f0 */
            const float a0 = (float)(r0[xp] - r0[xm]), a1 = (float)(r1[xp] - r1[xm]), a2 = (float)(r2[xp] - r2[xm]);
            const float p = (a0 + a2) * k1, q = a1 * k2;
            dx[(size_t)y * w + x] = p + q;
            /* Dy: rows filtered by [s 2s s]: S[0] * k0 + (S[-1] + S[1]) * k1, columns by [-1 0 1] */
            const float b0 = (float)r0[x] * k2 + (float)(r0[xm] + r0[xp]) * k1;
            const float b2 = (float)r2[x] * k2 + (float)(r2[xm] + r2[xp]) * k1;
            dy[(size_t)y * w + x] = b2 - b0;
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double sa = 0, sb = 0, sc = 0;
            for (int j = -1; j <= 1; j++) {
                const int yy = refl101(y + j, h);
                double ra = 0, rb = 0, rc = 0;
                for (int i = -1; i <= 1; i++) {
                    const int xx = refl101(x + i, w);
                    const float gx = dx[(size_t)yy * w + xx], gy = dy[(size_t)yy * w + xx];
                    const float xx2 = gx * gx, xy = gx * gy, yy2 = gy * gy;
                    ra += (double)xx2;
                    rb += (double)xy;
                    rc += (double)yy2;
                }
                sa += ra;
                sb += rb;
                sc += rc;
            }
            const float a = (float)sa, b = (float)sb, c = (float)sc;
            const float ac = a * c, bb = b * b, tr = a + c;
            const float kt = kf * tr;
            dst[(size_t)y * w + x] = (ac - bb) - kt * tr;
        }
    free(dx);
    free(dy);
}

typedef struct {
    float v;
    int idx;
} corner_cand;

static int cand_cmp(const void *pa, const void *pb)
{
    const corner_cand *a = (const corner_cand *)pa, *b = (const corner_cand *)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return a->idx > b->idx ? -1 : (a->idx < b->idx ? 1 : 0);  /* greaterThanPtr: the higher address first */
}

/* Returns the number of corners written to xy (x0, y0, x1, y1, ...), at most max_corners (<= 0: no limit, capacity cap). */
int orc_good_features_to_track(const uint8_t *img, size_t stride, int w, int h, int max_corners, double quality, double min_distance, double k,
                               float *xy, int cap, float *response_out /* w * h floats or NULL */)
{
    float *eig = (float *)malloc(sizeof(float) * (size_t)w * h);
    orc_corner_harris(img, stride, w, h, k, eig);
    if (response_out) memcpy(response_out, eig, sizeof(float) * (size_t)w * h);
    double maxVal = 0;
    {
        float m = eig[0];
        for (size_t i = 1; i < (size_t)w * h; i++) m = eig[i] > m ? eig[i] : m;
        maxVal = m;
    }
    const float thr = (float)(maxVal * quality);
    for (size_t i = 0; i < (size_t)w * h; i++) eig[i] = eig[i] > thr ? eig[i] : 0.f;
    corner_cand *cand = (corner_cand *)malloc(sizeof(corner_cand) * (size_t)w * h);
    int n = 0;
    for (int y = 1; y < h - 1; y++)
        for (int x = 1; x < w - 1; x++) {
            const float v = eig[(size_t)y * w + x];
            if (v == 0.f) continue;
            float m = v;
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++) {
                    const float t = eig[(size_t)(y + j) * w + x + i];
                    m = t > m ? t : m;
                }
            if (v == m) { cand[n].v = v; cand[n].idx = y * w + x; n++; }
        }
    qsort(cand, (size_t)n, sizeof(corner_cand), cand_cmp);
    int out = 0;
    if (min_distance >= 1) {
        const int cell = (int)lrint(min_distance);  /* cvRound */
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        int *head = (int *)malloc(sizeof(int) * (size_t)gw * gh), *next = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
        for (int i = 0; i < gw * gh; i++) head[i] = -1;
        const double md2 = min_distance * min_distance;
        for (int i = 0; i < n && out < cap; i++) {
            const int y = cand[i].idx / w, x = cand[i].idx - y * w;
            const int xc = x / cell, yc = y / cell;
            int x1 = xc - 1, y1 = yc - 1, x2 = xc + 1, y2 = yc + 1;
            x1 = x1 < 0 ? 0 : x1; y1 = y1 < 0 ? 0 : y1; x2 = x2 > gw - 1 ? gw - 1 : x2; y2 = y2 > gh - 1 ? gh - 1 : y2;
            int good = 1;
            for (int yy = y1; yy <= y2 && good; yy++)
                for (int xx = x1; xx <= x2 && good; xx++)
                    for (int j = head[yy * gw + xx]; j >= 0; j = next[j]) {
                        const float dx = (float)x - xy[2 * j], dy = (float)y - xy[2 * j + 1];
                        if ((double)(dx * dx + dy * dy) < md2) { good = 0; break; }
                    }
            if (!good) continue;
            xy[2 * out] = (float)x;
            xy[2 * out + 1] = (float)y;
            next[out] = head[yc * gw + xc];
            head[yc * gw + xc] = out;
            out++;
            if (max_corners > 0 && out == max_corners) break;
        }
        free(head);
        free(next);
    } else {
        for (int i = 0; i < n && out < cap; i++) {
            xy[2 * out] = (float)(cand[i].idx % w);
            xy[2 * out + 1] = (float)(cand[i].idx / w);
            out++;
            if (max_corners > 0 && out == max_corners) break;
        }
    }
    free(cand);
    free(eig);
    return out;
}

/* getRectSubPix(src 8-bit, Size(pw, ph), center, dst CV_32F): bilinear, replicated borders (samplers.cpp getRectSubPix_Cn_ + adjustRect) */
static void rect_subpix(const uint8_t *src, size_t step, int sw, int sh, int pw, int ph, float cx, float cy, float *dst)
{
    cx -= (pw - 1) * 0.5f;
    cy -= (ph - 1) * 0.5f;
    const int ipx = (int)floorf(cx), ipy = (int)floorf(cy);
    const float a = cx - ipx, b = cy - ipy;
    const float a11 = (1.f - a) * (1.f - b), a12 = a * (1.f - b), a21 = (1.f - a) * b, a22 = a * b, b1 = 1.f - b, b2 = b;
    if (0 <= ipx && ipx < sw - pw && 0 <= ipy && ipy < sh - ph) {
        const uint8_t *s = src + (size_t)ipy * step + ipx;
        for (int i = 0; i < ph; i++, s += step, dst += pw)
            for (int j = 0; j < pw; j++) dst[j] = s[j] * a11 + s[j + 1] * a12 + s[j + step] * a21 + s[j + step + 1] * a22;
        return;
    }
    /* adjustRect */
    int rx, ry, rw, rh;
    const uint8_t *s = src;
    if (ipx >= 0) { s += ipx; rx = 0; } else { rx = -ipx; if (rx > pw) rx = pw; }
    if (ipx < sw - pw) rw = pw; else { rw = sw - ipx - 1; if (rw < 0) { s += rw; rw = 0; } }
    if (ipy >= 0) { s += (size_t)ipy * step; ry = 0; } else ry = -ipy;
    if (ipy < sh - ph) rh = ph; else { rh = sh - ipy - 1; if (rh < 0) { s += (ptrdiff_t)rh * (ptrdiff_t)step; rh = 0; } }
    s -= rx;
    for (int i = 0; i < ph; i++, dst += pw) {
        const uint8_t *s2 = s + step;
        if (i < ry || i >= rh) s2 -= step;
        int j = 0;
        for (; j < rx; j++) dst[j] = s[rx] * b1 + s2[rx] * b2;
        for (; j < rw; j++) dst[j] = s[j] * a11 + s[j + 1] * a12 + s2[j] * a21 + s2[j + 1] * a22;
        for (; j < pw; j++) dst[j] = s[rw] * b1 + s2[rw] * b2;
        if (i < rh) s = s2;
    }
}

/* the window weights of cornerSubPix: (2 win + 1)^2 floats */
void orc_corner_subpix_mask(int win, float *mask)
{
    const int ww = 2 * win + 1;
    for (int i = 0; i < ww; i++) {
        const float y = (float)(i - win) / win;
        const float vy = expf(-y * y);
        for (int j = 0; j < ww; j++) {
            const float x = (float)(j - win) / win;
            mask[i * ww + j] = (float)(vy * expf(-x * x));
        }
    }
}

/* cornerSubPix(src, corners, Size(win, win), Size(-1, -1), TermCriteria(COUNT | EPS, max_count, epsilon)); xy in / out */
int orc_corner_subpix(const uint8_t *img, size_t stride, int w, int h, float *xy, int n, int win, int max_count, double epsilon)
{
    if (win < 1 || win > 15 || w < 2 * win + 5 || h < 2 * win + 5) return -1;
    const int ww = 2 * win + 1, pw = ww + 2;
    double eps = epsilon > 0 ? epsilon : 0;
    eps *= eps;
    int max_iters = max_count < 1 ? 1 : (max_count > 100 ? 100 : max_count);
    float *mask = (float *)malloc(sizeof(float) * ww * ww), *buf = (float *)malloc(sizeof(float) * pw * pw);
    orc_corner_subpix_mask(win, mask);
    for (int p = 0; p < n; p++) {
        const float ctx = xy[2 * p], cty = xy[2 * p + 1];
        float cix = ctx, ciy = cty;
        int iter = 0;
        double err = 0;
        do {
            double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
            rect_subpix(img, stride, w, h, pw, pw, cix, ciy, buf);
            const float *sp = buf + pw + 1;
            for (int i = 0, k = 0; i < ww; i++, sp += pw)
                for (int j = 0; j < ww; j++, k++) {
                    const double m = mask[k];
                    const double tgx = sp[j + 1] - sp[j - 1];
                    const double tgy = sp[j + pw] - sp[j - pw];
                    const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m;
                    const double px = j - win, py = i - win;
                    a += gxx;
                    b += gxy;
                    c += gyy;
                    bb1 += gxx * px + gxy * py;
                    bb2 += gxy * px + gyy * py;
                }
            const double det = a * c - b * b;
            if (fabs(det) <= DBL_EPSILON * DBL_EPSILON) break;
            const double scale = 1.0 / det;
            const float c2x = (float)(cix + c * scale * bb1 - b * scale * bb2);
            const float c2y = (float)(ciy - b * scale * bb1 + a * scale * bb2);
            err = (double)((c2x - cix) * (c2x - cix) + (c2y - ciy) * (c2y - ciy));
            cix = c2x;
            ciy = c2y;
            if (cix < 0 || cix >= w || ciy < 0 || ciy >= h) break;
        } while (++iter < max_iters && err > eps);
        if (fabsf(cix - ctx) > win || fabsf(ciy - cty) > win) { cix = ctx; ciy = cty; }
        xy[2 * p] = cix;
        xy[2 * p + 1] = ciy;
    }
    free(mask);
    free(buf);
    return 0;
}

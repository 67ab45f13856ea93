/*
 * orb_oracle.c -- CPU restatement (plain C, single thread) of the Amos-SLAM front-end hot path.
 * TEST INFRASTRUCTURE ONLY -- see orb_oracle.h for who may use it and for the parity status
 * ("PARITY UNPINNED" for the stages that restate OpenCV 4.5 primitives).
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * Build with -ffp-contract=off: every place where the shipped reference binary fuses a
 * multiply-add is written as an explicit fmaf() below.
 */
#include "orb_oracle.h"
#include "../include/amos_orb_pattern.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define EDGE_THRESHOLD 19   /* ORBextractor.cc:93 */
#define PATCH_SIZE 31       /* ORBextractor.cc:91 */
#define HALF_PATCH_SIZE 15  /* ORBextractor.cc:92 */

/* cvRound: round half to even (cvtss2si / lrint), SURVEY Appendix A.0 */
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) { return (int)floorf(v); }

struct orc_extractor {
    amos_orb_params p;
    int nlevels;
    float scale[AMOS_MAX_LEVELS], inv_scale[AMOS_MAX_LEVELS];
    float sigma2[AMOS_MAX_LEVELS], inv_sigma2[AMOS_MAX_LEVELS];
    int quota[AMOS_MAX_LEVELS];
    int umax[HALF_PATCH_SIZE + 1];
    int width, height;
    int lw[AMOS_MAX_LEVELS], lh[AMOS_MAX_LEVELS];
    uint8_t *padded[AMOS_MAX_LEVELS];  /* (lw+38) x (lh+38), stride lw+38 */
    uint8_t *blurred[AMOS_MAX_LEVELS]; /* lw x lh, stride lw */
    amos_keypoint *cand[AMOS_MAX_LEVELS];
    int ncand[AMOS_MAX_LEVELS], capcand[AMOS_MAX_LEVELS];
    amos_keypoint *kps[AMOS_MAX_LEVELS];
    int nkps[AMOS_MAX_LEVELS], capkps[AMOS_MAX_LEVELS];
    uint8_t *closed;
    int detected;
};

/* ------------------------------------------------------------------------------------------ */
/* ORBextractor::ORBextractor, ORBextractor.cc:492-609                                          */
orc_extractor *orc_create(const amos_orb_params *p)
{
    if (!p || p->n_levels < 1 || p->n_levels > AMOS_MAX_LEVELS || p->n_features < 1) return NULL;
    orc_extractor *e = (orc_extractor *)calloc(1, sizeof(*e));
    e->p = *p;
    e->nlevels = p->n_levels;
    e->scale[0] = 1.0f; /* :503-510 */
    e->sigma2[0] = 1.0f;
    for (int i = 1; i < e->nlevels; i++) {
        e->scale[i] = e->scale[i - 1] * p->scale_factor;
        e->sigma2[i] = e->scale[i] * e->scale[i];
    }
    for (int i = 0; i < e->nlevels; i++) { /* :514-518 */
        e->inv_scale[i] = 1.0f / e->scale[i];
        e->inv_sigma2[i] = 1.0f / e->sigma2[i];
    }
    /* :524-537 quota per level */
    float factor = 1.0f / p->scale_factor;
    float nDesired = (float)p->n_features * (1.0f - factor) /
                     (1.0f - (float)pow((double)factor, (double)e->nlevels));
    int sum = 0;
    for (int level = 0; level < e->nlevels - 1; level++) {
        e->quota[level] = cv_round_f(nDesired);
        sum += e->quota[level];
        nDesired *= factor;
    }
    e->quota[e->nlevels - 1] = p->n_features - sum > 0 ? p->n_features - sum : 0;
    /* :579-608 umax */
    int v, v0;
    int vmax = cv_floor_f(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = (int)ceilf(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) e->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    return e;
}

static void free_frame(orc_extractor *e)
{
    for (int l = 0; l < AMOS_MAX_LEVELS; l++) {
        free(e->padded[l]); e->padded[l] = NULL;
        free(e->blurred[l]); e->blurred[l] = NULL;
        free(e->cand[l]); e->cand[l] = NULL;
        free(e->kps[l]); e->kps[l] = NULL;
        e->ncand[l] = e->nkps[l] = e->capcand[l] = e->capkps[l] = 0;
    }
    free(e->closed); e->closed = NULL;
    e->detected = 0;
}

void orc_destroy(orc_extractor *e)
{
    if (!e) return;
    free_frame(e);
    free(e);
}

void orc_tables(const orc_extractor *e, float *scale, float *inv_scale, float *sigma2,
                float *inv_sigma2, int32_t *features_per_level, int32_t *umax)
{
    for (int i = 0; i < e->nlevels; i++) {
        if (scale) scale[i] = e->scale[i];
        if (inv_scale) inv_scale[i] = e->inv_scale[i];
        if (sigma2) sigma2[i] = e->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = e->inv_sigma2[i];
        if (features_per_level) features_per_level[i] = e->quota[i];
    }
    if (umax) for (int i = 0; i <= HALF_PATCH_SIZE; i++) umax[i] = e->umax[i];
}

/* ORBextractor.cc:1832-1834 */
int orc_level_sizes(const orc_extractor *e, int width, int height, int32_t *lw, int32_t *lh)
{
    for (int l = 0; l < e->nlevels; l++) {
        float s = e->inv_scale[l];
        if (lw) lw[l] = cv_round_f((float)width * s);
        if (lh) lh[l] = cv_round_f((float)height * s);
    }
    return e->nlevels;
}

/* ------------------------------------------------------------------------------------------ */
/* cv::resize 8UC1 INTER_LINEAR (ORBextractor.cc:1848), SURVEY Appendix A.1.                    */
static inline short sat_short_round(float v)
{
    int i = cv_round_f(v);
    return (short)(i < SHRT_MIN ? SHRT_MIN : i > SHRT_MAX ? SHRT_MAX : i);
}

void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, size_t sstride, uint8_t *dst, int dw,
                          int dh, size_t dstride)
{
    /* cv::resize computes inv_scale = dsize/ssize, hal::resize then scale = 1/inv_scale */
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * dw);
    int *yofs = (int *)malloc(sizeof(int) * dh);
    short *ibeta = (short *)malloc(sizeof(short) * 2 * dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short_round((1.f - fx) * 2048);
        ialpha[2 * dx + 1] = sat_short_round(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy);
        fy -= sy;
        yofs[dy] = sy; /* rows are clipped when fetched, fy is NOT reset (resizeGeneric_Invoker) */
        ibeta[2 * dy] = sat_short_round((1.f - fy) * 2048);
        ibeta[2 * dy + 1] = sat_short_round(fy * 2048);
    }
    int *row0 = (int *)malloc(sizeof(int) * dw), *row1 = (int *)malloc(sizeof(int) * dw);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1;
        sy0 = sy0 < 0 ? 0 : sy0 > sh - 1 ? sh - 1 : sy0;
        sy1 = sy1 < 0 ? 0 : sy1 > sh - 1 ? sh - 1 : sy1;
        const uint8_t *S0 = src + (size_t)sy0 * sstride, *S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; dx++) { /* HResizeLinear: second tap index clamped, weight 0 */
            int sx = xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sx;
            row0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx1] * ialpha[2 * dx + 1];
            row1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx1] * ialpha[2 * dx + 1];
        }
        short b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++) /* VResizeLinear 8u fixed point */
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(row0); free(row1); free(xofs); free(ialpha); free(yofs); free(ibeta);
}

/* BORDER_REFLECT_101 index map, SURVEY Appendix A.5 (single reflection suffices: border < dim) */
static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * n - 2 - i;
    }
    return i;
}

/* copyMakeBorder(..., 19,19,19,19, BORDER_REFLECT_101), ORBextractor.cc:1859,1880 */
static void make_border(uint8_t *padded, int w, int h)
{
    const int ps = w + 2 * EDGE_THRESHOLD;
    for (int y = 0; y < h + 2 * EDGE_THRESHOLD; y++) {
        int sy = reflect101(y - EDGE_THRESHOLD, h);
        uint8_t *drow = padded + (size_t)y * ps;
        const uint8_t *srow = padded + (size_t)(sy + EDGE_THRESHOLD) * ps + EDGE_THRESHOLD;
        for (int x = 0; x < ps; x++) {
            int sx = reflect101(x - EDGE_THRESHOLD, w);
            if (y >= EDGE_THRESHOLD && y < h + EDGE_THRESHOLD && x >= EDGE_THRESHOLD && x < w + EDGE_THRESHOLD)
                continue;
            drow[x] = srow[sx];
        }
    }
}

/* ORBextractor::ComputePyramid, ORBextractor.cc:1826-1886 */
static void compute_pyramid(orc_extractor *e, const uint8_t *gray, size_t stride)
{
    for (int l = 0; l < e->nlevels; l++) {
        const int w = e->lw[l], h = e->lh[l], ps = w + 2 * EDGE_THRESHOLD;
        e->padded[l] = (uint8_t *)malloc((size_t)ps * (h + 2 * EDGE_THRESHOLD));
        uint8_t *roi = e->padded[l] + (size_t)EDGE_THRESHOLD * ps + EDGE_THRESHOLD;
        if (l == 0) {
            for (int y = 0; y < h; y++) memcpy(roi + (size_t)y * ps, gray + (size_t)y * stride, w);
        } else {
            const int pps = e->lw[l - 1] + 2 * EDGE_THRESHOLD;
            const uint8_t *prev = e->padded[l - 1] + (size_t)EDGE_THRESHOLD * pps + EDGE_THRESHOLD;
            orc_resize_linear_u8(prev, e->lw[l - 1], e->lh[l - 1], pps, roi, w, h, ps);
        }
        make_border(e->padded[l], w, h);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* cv::FAST TYPE_9_16 with non-max suppression, SURVEY Appendix A.3 (FAST_t<16>, cornerScore<16>) */
static int fast_corner_score16(const uint8_t *ptr, const int pixel[25], int threshold)
{
    const int K = 8, N = K * 3 + 1;
    int k, v = ptr[0];
    short d[25];
    for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        a = a < d[k + 3] ? a : d[k + 3];
        if (a <= a0) continue;
        a = a < d[k + 4] ? a : d[k + 4];
        a = a < d[k + 5] ? a : d[k + 5];
        a = a < d[k + 6] ? a : d[k + 6];
        a = a < d[k + 7] ? a : d[k + 7];
        a = a < d[k + 8] ? a : d[k + 8];
        int t0 = a < d[k] ? a : d[k];
        a0 = a0 > t0 ? a0 : t0;
        int t1 = a < d[k + 9] ? a : d[k + 9];
        a0 = a0 > t1 ? a0 : t1;
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        b = b > d[k + 3] ? b : d[k + 3];
        b = b > d[k + 4] ? b : d[k + 4];
        b = b > d[k + 5] ? b : d[k + 5];
        if (b >= b0) continue;
        b = b > d[k + 6] ? b : d[k + 6];
        b = b > d[k + 7] ? b : d[k + 7];
        b = b > d[k + 8] ? b : d[k + 8];
        int t0 = b > d[k] ? b : d[k];
        b0 = b0 < t0 ? b0 : t0;
        int t1 = b > d[k + 9] ? b : d[k + 9];
        b0 = b0 < t1 ? b0 : t1;
    }
    return -b0 - 1;
}

int orc_fast9_16(const uint8_t *img, size_t stride, int w, int h, int threshold,
                 amos_keypoint *out, int cap)
{
    static const int offs[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = offs[k][0] + offs[k][1] * (int)stride;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    threshold = threshold < 0 ? 0 : threshold > 255 ? 255 : threshold;
    int nout = 0;
    if (w < 7 || h < 7) return 0;
    uint8_t *buf[3];
    int *cpbuf[3];
    for (int i = 0; i < 3; i++) {
        buf[i] = (uint8_t *)calloc(w, 1);
        cpbuf[i] = (int *)calloc(w + 1, sizeof(int));
    }
    for (int i = 3; i < h - 2; i++) {
        const uint8_t *ptr = img + (size_t)i * stride + 3;
        uint8_t *curr = buf[(i - 3) % 3];
        int *cornerpos = cpbuf[(i - 3) % 3] + 1;
        memset(curr, 0, w);
        int ncorners = 0;
        if (i < h - 3) {
            for (int j = 3; j < w - 3; j++, ptr++) {
                int v = ptr[0], is_corner = 0;
                int vt = v - threshold, count = 0;
                for (int k = 0; k < N; k++) { /* darker arc */
                    if (ptr[pixel[k]] < vt) { if (++count > K) { is_corner = 1; break; } }
                    else count = 0;
                }
                if (!is_corner) {
                    vt = v + threshold; count = 0;
                    for (int k = 0; k < N; k++) { /* brighter arc */
                        if (ptr[pixel[k]] > vt) { if (++count > K) { is_corner = 1; break; } }
                        else count = 0;
                    }
                }
                if (is_corner) {
                    cornerpos[ncorners++] = j;
                    curr[j] = (uint8_t)fast_corner_score16(ptr, pixel, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t *prev = buf[(i - 4 + 3) % 3];
        const uint8_t *pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3] + 1;
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; k++) {
            int j = cornerpos[k];
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
                if (nout < cap) {
                    amos_keypoint kp = {(float)j, (float)(i - 1), 7.f, -1.f, (float)score, 0, -1};
                    out[nout] = kp;
                }
                nout++;
            }
        }
    }
    for (int i = 0; i < 3; i++) { free(buf[i]); free(cpbuf[i]); }
    return nout;
}

/* ------------------------------------------------------------------------------------------ */
/* ExtractorNode + DistributeOctTree, ORBextractor.cc:635-703, 706-1049                         */
typedef struct onode {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    int *keys; /* indices into the input array, order preserved */
    int nkeys;
    int noMore;
    long seq; /* creation order: stands in for the heap address in the (size, pointer) sort */
    struct onode *prev, *next;
} onode;

typedef struct { onode *head, *tail; int size; long seq; } olist;

static onode *node_new(olist *L, int cap)
{
    onode *n = (onode *)calloc(1, sizeof(onode));
    n->keys = (int *)malloc(sizeof(int) * (cap > 0 ? cap : 1));
    n->seq = L->seq++;
    return n;
}
static void list_push_front(olist *L, onode *n)
{
    n->prev = NULL; n->next = L->head;
    if (L->head) L->head->prev = n; else L->tail = n;
    L->head = n; L->size++;
}
static void list_push_back(olist *L, onode *n)
{
    n->next = NULL; n->prev = L->tail;
    if (L->tail) L->tail->next = n; else L->head = n;
    L->tail = n; L->size++;
}
static onode *list_erase(olist *L, onode *n) /* returns next */
{
    onode *nx = n->next;
    if (n->prev) n->prev->next = n->next; else L->head = n->next;
    if (n->next) n->next->prev = n->prev; else L->tail = n->prev;
    L->size--;
    free(n->keys); free(n);
    return nx;
}

/* ExtractorNode::DivideNode, ORBextractor.cc:635-703 */
static void divide_node(olist *L, const onode *p, const amos_keypoint *pts, onode *c[4])
{
    const int halfX = (int)ceilf((float)(p->URx - p->ULx) / 2);
    const int halfY = (int)ceilf((float)(p->BRy - p->ULy) / 2);
    for (int i = 0; i < 4; i++) c[i] = node_new(L, p->nkeys);
    onode *n1 = c[0], *n2 = c[1], *n3 = c[2], *n4 = c[3];
    n1->ULx = p->ULx; n1->ULy = p->ULy;
    n1->URx = p->ULx + halfX; n1->URy = p->ULy;
    n1->BLx = p->ULx; n1->BLy = p->ULy + halfY;
    n1->BRx = p->ULx + halfX; n1->BRy = p->ULy + halfY;
    n2->ULx = n1->URx; n2->ULy = n1->URy;
    n2->URx = p->URx; n2->URy = p->URy;
    n2->BLx = n1->BRx; n2->BLy = n1->BRy;
    n2->BRx = p->URx; n2->BRy = p->ULy + halfY;
    n3->ULx = n1->BLx; n3->ULy = n1->BLy;
    n3->URx = n1->BRx; n3->URy = n1->BRy;
    n3->BLx = p->BLx; n3->BLy = p->BLy;
    n3->BRx = n1->BRx; n3->BRy = p->BLy;
    n4->ULx = n3->URx; n4->ULy = n3->URy;
    n4->URx = n2->BRx; n4->URy = n2->BRy;
    n4->BLx = n3->BRx; n4->BLy = n3->BRy;
    n4->BRx = p->BRx; n4->BRy = p->BRy;
    for (int i = 0; i < p->nkeys; i++) {
        const amos_keypoint *kp = &pts[p->keys[i]];
        onode *dst;
        if (kp->x < (float)n1->URx) dst = (kp->y < (float)n1->BRy) ? n1 : n3;
        else dst = (kp->y < (float)n1->BRy) ? n2 : n4;
        dst->keys[dst->nkeys++] = p->keys[i];
    }
    for (int i = 0; i < 4; i++) if (c[i]->nkeys == 1) c[i]->noMore = 1;
}

typedef struct { int size; onode *node; } size_node;
static int cmp_size_node(const void *a, const void *b)
{
    const size_node *x = (const size_node *)a, *y = (const size_node *)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    /* the reference compares heap addresses here (ORBextractor.cc:948); this restatement fixes the
     * rule to creation order, i.e. addresses that grow with allocation order */
    return x->node->seq < y->node->seq ? -1 : x->node->seq > y->node->seq ? 1 : 0;
}

/* pushes the non-empty children of a division to the list front (ORBextractor.cc:842-899,962-1002) */
static void push_children(olist *L, onode *c[4], size_node *vec, int *nvec, int *nToExpand)
{
    for (int i = 0; i < 4; i++) {
        if (c[i]->nkeys > 0) {
            list_push_front(L, c[i]);
            if (c[i]->nkeys > 1) {
                if (nToExpand) (*nToExpand)++;
                vec[*nvec].size = c[i]->nkeys;
                vec[*nvec].node = c[i];
                (*nvec)++;
            }
        } else {
            free(c[i]->keys); free(c[i]);
        }
    }
}

int orc_distribute_octree(const amos_keypoint *pts, int n, int minX, int maxX, int minY, int maxY,
                          int N, amos_keypoint *out, int cap)
{
    olist L = {0, 0, 0, 0};
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY)); /* :718 */
    if (nIni < 1) return AMOS_ERR_INVALID; /* the reference divides by zero here */
    const float hX = (float)(maxX - minX) / nIni;
    onode **ini = (onode **)malloc(sizeof(onode *) * nIni);
    for (int i = 0; i < nIni; i++) { /* :731-755 */
        onode *ni = node_new(&L, n);
        ni->ULx = (int)(hX * (float)i); ni->ULy = 0;
        ni->URx = (int)(hX * (float)(i + 1)); ni->URy = 0;
        ni->BLx = ni->ULx; ni->BLy = maxY - minY;
        ni->BRx = ni->URx; ni->BRy = maxY - minY;
        list_push_back(&L, ni);
        ini[i] = ni;
    }
    for (int i = 0; i < n; i++) { /* :758-763 */
        int idx = (int)(pts[i].x / hX);
        if (idx < 0) idx = 0;
        if (idx >= nIni) idx = nIni - 1; /* out of range is UB in the reference */
        ini[idx]->keys[ini[idx]->nkeys++] = i;
    }
    free(ini);
    for (onode *lit = L.head; lit;) { /* :769-786 */
        if (lit->nkeys == 1) { lit->noMore = 1; lit = lit->next; }
        else if (lit->nkeys == 0) lit = list_erase(&L, lit);
        else lit = lit->next;
    }
    int bFinish = 0;
    size_node *vec = (size_node *)malloc(sizeof(size_node) * (4 * (size_t)(n > 0 ? n : 1) + 16));
    size_node *prevvec = (size_node *)malloc(sizeof(size_node) * (4 * (size_t)(n > 0 ? n : 1) + 16));
    int nvec = 0;
    while (!bFinish) { /* :800-1021 */
        int prevSize = L.size;
        int nToExpand = 0;
        nvec = 0;
        onode *lit = L.head;
        while (lit) {
            if (lit->noMore) { lit = lit->next; continue; }
            onode *c[4];
            divide_node(&L, lit, pts, c);
            push_children(&L, c, vec, &nvec, &nToExpand);
            lit = list_erase(&L, lit);
        }
        if (L.size >= N || L.size == prevSize) {
            bFinish = 1;
        } else if (L.size + nToExpand * 3 > N) { /* :936 */
            while (!bFinish) {
                prevSize = L.size;
                int nprev = nvec;
                memcpy(prevvec, vec, sizeof(size_node) * nvec);
                nvec = 0;
                qsort(prevvec, nprev, sizeof(size_node), cmp_size_node);
                for (int j = nprev - 1; j >= 0; j--) {
                    onode *c[4];
                    divide_node(&L, prevvec[j].node, pts, c);
                    push_children(&L, c, vec, &nvec, NULL);
                    list_erase(&L, prevvec[j].node);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prevSize) bFinish = 1;
            }
        }
    }
    free(vec); free(prevvec);
    int nout = 0;
    for (onode *lit = L.head; lit; lit = lit->next) { /* :1024-1046 */
        int best = lit->keys[0];
        float maxResponse = pts[best].response;
        for (int k = 1; k < lit->nkeys; k++)
            if (pts[lit->keys[k]].response > maxResponse) {
                best = lit->keys[k];
                maxResponse = pts[best].response;
            }
        if (nout < cap) out[nout] = pts[best];
        nout++;
    }
    while (L.head) list_erase(&L, L.head);
    return nout;
}

/* ------------------------------------------------------------------------------------------ */
/* cv::fastAtan2, SURVEY Appendix A.4 (atan_f32 in mathfuncs.cpp; evaluation order as written)  */
float orc_fast_atan2(float y, float x)
{
    static const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
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

/* IC_Angle, ORBextractor.cc:108-161 */
static float ic_angle(const uint8_t *image, size_t step, float ptx, float pty, const int *u_max)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = image + (ptrdiff_t)cv_round_f(pty) * (ptrdiff_t)step + cv_round_f(ptx);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * (ptrdiff_t)step], val_minus = center[u - v * (ptrdiff_t)step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------------------------------------------------------ */
/* ORBextractor::ComputeKeyPointsOctTree, ORBextractor.cc:1052-1199                            */
static int compute_keypoints_octree(orc_extractor *e)
{
    const float W = 30;
    for (int level = 0; level < e->nlevels; ++level) {
        const int lw = e->lw[level], lh = e->lh[level], ps = lw + 2 * EDGE_THRESHOLD;
        const uint8_t *img = e->padded[level] + (size_t)EDGE_THRESHOLD * ps + EDGE_THRESHOLD;
        const int minBorderX = EDGE_THRESHOLD - 3;
        const int minBorderY = minBorderX;
        const int maxBorderX = lw - EDGE_THRESHOLD + 3;
        const int maxBorderY = lh - EDGE_THRESHOLD + 3;
        const float width = (float)(maxBorderX - minBorderX);
        const float height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W);
        const int nRows = (int)(height / W);
        if (nCols < 1 || nRows < 1) return AMOS_ERR_INVALID; /* division by zero in the reference */
        const int wCell = (int)ceilf(width / nCols);
        const int hCell = (int)ceilf(height / nRows);

        int cap = 1024, ncand = 0;
        amos_keypoint *cand = (amos_keypoint *)malloc(sizeof(amos_keypoint) * cap);
        const int cellcap = ((wCell + 7) / 2) * ((hCell + 7) / 2) + 4;
        amos_keypoint *cell = (amos_keypoint *)malloc(sizeof(amos_keypoint) * cellcap);
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                const uint8_t *sub = img + (size_t)(int)iniY * ps + (int)iniX;
                const int sw = (int)maxX - (int)iniX, sh = (int)maxY - (int)iniY;
                int nc = orc_fast9_16(sub, ps, sw, sh, e->p.ini_th_fast, cell, cellcap);
                if (nc == 0) nc = orc_fast9_16(sub, ps, sw, sh, e->p.min_th_fast, cell, cellcap);
                if (nc > cellcap) return AMOS_ERR_CAPACITY;
                for (int k = 0; k < nc; k++) {
                    cell[k].x += j * wCell;
                    cell[k].y += i * hCell;
                    if (ncand == cap) { cap *= 2; cand = (amos_keypoint *)realloc(cand, sizeof(amos_keypoint) * cap); }
                    cand[ncand++] = cell[k];
                }
            }
        }
        free(cell);
        e->cand[level] = cand; e->ncand[level] = ncand; e->capcand[level] = cap;

        const int N = e->quota[level];
        const int kcap = N + 8 + 4 * ncand;
        e->kps[level] = (amos_keypoint *)malloc(sizeof(amos_keypoint) * kcap);
        e->capkps[level] = kcap;
        int nk = orc_distribute_octree(cand, ncand, minBorderX, maxBorderX, minBorderY, maxBorderY, N,
                                       e->kps[level], kcap);
        if (nk < 0) return nk;
        e->nkps[level] = nk;
        const int scaledPatchSize = (int)(PATCH_SIZE * e->scale[level]); /* :1177 */
        for (int i = 0; i < nk; i++) {
            e->kps[level][i].x += minBorderX;
            e->kps[level][i].y += minBorderY;
            e->kps[level][i].octave = level;
            e->kps[level][i].size = (float)scaledPatchSize;
        }
    }
    for (int level = 0; level < e->nlevels; ++level) { /* :1194-1198 computeOrientation */
        const int ps = e->lw[level] + 2 * EDGE_THRESHOLD;
        const uint8_t *img = e->padded[level] + (size_t)EDGE_THRESHOLD * ps + EDGE_THRESHOLD;
        for (int i = 0; i < e->nkps[level]; i++)
            e->kps[level][i].angle = ic_angle(img, ps, e->kps[level][i].x, e->kps[level][i].y, e->umax);
    }
    return AMOS_OK;
}

int orc_detect(orc_extractor *e, const uint8_t *gray, size_t stride, int width, int height)
{
    if (!e || !gray || width < 1 || height < 1) return AMOS_ERR_INVALID;
    free_frame(e);
    e->width = width; e->height = height;
    orc_level_sizes(e, width, height, e->lw, e->lh);
    compute_pyramid(e, gray, stride);
    int rc = compute_keypoints_octree(e);
    if (rc == AMOS_OK) e->detected = 1;
    return rc;
}

int orc_level_count(const orc_extractor *e, int level)
{
    if (!e || level < 0 || level >= e->nlevels) return AMOS_ERR_INVALID;
    return e->nkps[level];
}
int orc_level_keypoints(const orc_extractor *e, int level, amos_keypoint *out, int cap)
{
    if (!e || level < 0 || level >= e->nlevels) return AMOS_ERR_INVALID;
    if (cap < e->nkps[level]) return AMOS_ERR_CAPACITY;
    memcpy(out, e->kps[level], sizeof(amos_keypoint) * e->nkps[level]);
    return e->nkps[level];
}
int orc_set_level_keypoints(orc_extractor *e, int level, const amos_keypoint *kps, int n)
{
    if (!e || level < 0 || level >= e->nlevels || n < 0) return AMOS_ERR_INVALID;
    if (n > e->capkps[level]) {
        e->kps[level] = (amos_keypoint *)realloc(e->kps[level], sizeof(amos_keypoint) * n);
        e->capkps[level] = n;
    }
    memcpy(e->kps[level], kps, sizeof(amos_keypoint) * n);
    e->nkps[level] = n;
    return AMOS_OK;
}
int orc_level_candidates(const orc_extractor *e, int level, amos_keypoint *out, int cap)
{
    if (!e || level < 0 || level >= e->nlevels) return AMOS_ERR_INVALID;
    if (cap < e->ncand[level]) return AMOS_ERR_CAPACITY;
    memcpy(out, e->cand[level], sizeof(amos_keypoint) * e->ncand[level]);
    return e->ncand[level];
}
int orc_level_image(const orc_extractor *e, int level, uint8_t *dst, size_t dst_stride, int padded)
{
    if (!e || !e->detected || level < 0 || level >= e->nlevels) return AMOS_ERR_INVALID;
    const int ps = e->lw[level] + 2 * EDGE_THRESHOLD;
    if (padded) {
        for (int y = 0; y < e->lh[level] + 2 * EDGE_THRESHOLD; y++)
            memcpy(dst + (size_t)y * dst_stride, e->padded[level] + (size_t)y * ps, ps);
    } else {
        for (int y = 0; y < e->lh[level]; y++)
            memcpy(dst + (size_t)y * dst_stride,
                   e->padded[level] + (size_t)(y + EDGE_THRESHOLD) * ps + EDGE_THRESHOLD, e->lw[level]);
    }
    return AMOS_OK;
}
int orc_blurred_image(const orc_extractor *e, int level, uint8_t *dst, size_t dst_stride)
{
    if (!e || level < 0 || level >= e->nlevels || !e->blurred[level]) return AMOS_ERR_INVALID;
    for (int y = 0; y < e->lh[level]; y++)
        memcpy(dst + (size_t)y * dst_stride, e->blurred[level] + (size_t)y * e->lw[level], e->lw[level]);
    return AMOS_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* cv::GaussianBlur(7x7, sigma 2) 8-bit fixed-point path, SURVEY Appendix A.2.  Taps are the
 * 8.8 fixed-point kernel after OpenCV's error-diffusion normalisation (sum 256). */
static const int k_gauss7[7] = {18, 34, 48, 56, 48, 34, 18};

void orc_gaussian_blur7(const uint8_t *src, size_t sstride, int w, int h, uint8_t *dst, size_t dstride)
{
    uint16_t *tmp = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h);
    for (int y = 0; y < h; y++) { /* horizontal: u8 x 8.8 -> exact in 16 bits */
        const uint8_t *s = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            unsigned acc = 0;
            for (int k = -3; k <= 3; k++) acc += (unsigned)k_gauss7[k + 3] * s[reflect101(x + k, w)];
            tmp[(size_t)y * w + x] = (uint16_t)acc;
        }
    }
    for (int y = 0; y < h; y++) { /* vertical: 16.16 accumulate, round, shift */
        for (int x = 0; x < w; x++) {
            uint32_t acc = 0;
            for (int k = -3; k <= 3; k++)
                acc += (uint32_t)k_gauss7[k + 3] * tmp[(size_t)reflect101(y + k, h) * w + x];
            dst[(size_t)y * dstride + x] = (uint8_t)((acc + 32768u) >> 16);
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------------------------------ */
/* sincosf as glibc >= 2.28 computes it (ARM optimized-routines, sysdeps/ieee754/flt-32/
 * s_sincosf.c + sincosf_poly.h), restated for the only range the path uses, |x| < 120.
 * Double arithmetic, no FMA.  tests/test_oracle_math.py compares it with the host libm. */
typedef struct { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; } sincos_tab;
static const sincos_tab k_sincos_tab[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2,
     0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3,
     0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2,
     -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3,
     0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};

static inline uint32_t abstop12(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    return (u >> 20) & 0x7ff;
}

static void sincosf_poly(double x, double x2, const sincos_tab *p, int n, float *sinp, float *cosp)
{
    double x3, x4, x5, x6, s, c, c1, c2, s1;
    x4 = x2 * x2;
    x3 = x2 * x;
    c2 = p->c3 + x2 * p->c4;
    s1 = p->s2 + x2 * p->s3;
    float *tmp = (n & 1 ? cosp : sinp);
    cosp = (n & 1 ? sinp : cosp);
    sinp = tmp;
    c1 = p->c0 + x2 * p->c1;
    x5 = x3 * x2;
    x6 = x4 * x2;
    s = x + x3 * p->s1;
    c = c1 + x4 * p->c2;
    *sinp = (float)(s + x5 * s1);
    *cosp = (float)(c + x6 * c2);
}

void orc_sincosf(float y, float *sinp, float *cosp)
{
    double x = y;
    const sincos_tab *p = &k_sincos_tab[0];
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) { /* |y| < pi/4 */
        double x2 = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) {
            *sinp = y;
            *cosp = 1.0f;
            return;
        }
        sincosf_poly(x, x2, p, 0, sinp, cosp);
    } else if (abstop12(y) < abstop12(120.0f)) {
        double r = x * p->hpi_inv; /* reduce_fast, !TOINT_INTRINSICS form */
        int n = ((int32_t)r + 0x800000) >> 24;
        x = x - n * p->hpi;
        double s = p->sign[n & 3];
        if (n & 2) p = &k_sincos_tab[1];
        sincosf_poly(x * s, x * x, p, n, sinp, cosp);
    } else { /* outside the range an angle in [0,360) degrees can reach */
        *sinp = sinf(y);
        *cosp = cosf(y);
    }
}

/* computeOrbDescriptor, ORBextractor.cc:173-227.  The shipped binary fuses the second product of
 * each coordinate into an FMA: row = rn(fma(px, b, py*a)), col = rn(fma(px, a, -(py*b))). */
static void compute_orb_descriptor(const amos_keypoint *kpt, const uint8_t *img, size_t step, uint8_t *desc)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = kpt->angle * factorPI;
    float a, b;
    orc_sincosf(angle, &b, &a);
    const uint8_t *center = img + (ptrdiff_t)cv_round_f(kpt->y) * (ptrdiff_t)step + cv_round_f(kpt->x);
    const signed char *pattern = amos_orb_pattern;
    for (int i = 0; i < 32; ++i, pattern += 32) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            float x0 = (float)pattern[4 * k + 0], y0 = (float)pattern[4 * k + 1];
            float x1 = (float)pattern[4 * k + 2], y1 = (float)pattern[4 * k + 3];
            int t0 = center[(ptrdiff_t)cv_round_f(fmaf(x0, b, y0 * a)) * (ptrdiff_t)step +
                            cv_round_f(fmaf(x0, a, -(y0 * b)))];
            int t1 = center[(ptrdiff_t)cv_round_f(fmaf(x1, b, y1 * a)) * (ptrdiff_t)step +
                            cv_round_f(fmaf(x1, a, -(y1 * b)))];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ORBextractor::ProcessDesp, ORBextractor.cc:1747-1820 (== tail of the 4-arg operator()) */
int orc_describe(orc_extractor *e, amos_keypoint *kps, uint8_t *desc, int cap, int *n)
{
    if (!e || !e->detected || !n) return AMOS_ERR_STATE;
    int nkeypoints = 0;
    for (int l = 0; l < e->nlevels; l++) nkeypoints += e->nkps[l];
    *n = nkeypoints;
    if (nkeypoints > cap) return AMOS_ERR_CAPACITY;
    int offset = 0;
    for (int level = 0; level < e->nlevels; ++level) {
        int nl = e->nkps[level];
        if (nl == 0) continue;
        const int w = e->lw[level], h = e->lh[level], ps = w + 2 * EDGE_THRESHOLD;
        free(e->blurred[level]);
        e->blurred[level] = (uint8_t *)malloc((size_t)w * h);
        /* workingMat = clone of the ROI: the blur sees only the level image and reflects itself */
        orc_gaussian_blur7(e->padded[level] + (size_t)EDGE_THRESHOLD * ps + EDGE_THRESHOLD, ps, w, h,
                           e->blurred[level], w);
        for (int i = 0; i < nl; i++) {
            compute_orb_descriptor(&e->kps[level][i], e->blurred[level], w, desc + (size_t)(offset + i) * 32);
            kps[offset + i] = e->kps[level][i];
            if (level != 0) { /* :1804-1813 */
                kps[offset + i].x *= e->scale[level];
                kps[offset + i].y *= e->scale[level];
            }
        }
        offset += nl;
    }
    return AMOS_OK;
}

int orc_extract(orc_extractor *e, const uint8_t *gray, size_t stride, int width, int height,
                amos_keypoint *kps, uint8_t *desc, int cap, int *n)
{
    int rc = orc_detect(e, gray, stride, width, height);
    if (rc != AMOS_OK) return rc;
    return orc_describe(e, kps, desc, cap, n);
}

/* ------------------------------------------------------------------------------------------ */
/* getStructuringElement(MORPH_ELLIPSE, 31x31) + dilate + erode, SURVEY Appendix A.6            */
static void ellipse31_rows(int j1[31], int j2[31])
{
    const int r = 15, c = 15;
    const double inv_r2 = 1. / ((double)r * r);
    for (int i = 0; i < 31; i++) {
        int dy = i - r;
        int dx = cv_round_d(c * sqrt((r * r - dy * dy) * inv_r2));
        j1[i] = c - dx > 0 ? c - dx : 0;
        j2[i] = c + dx + 1 < 31 ? c + dx + 1 : 31;
    }
}

static void morph31(const uint8_t *src, size_t sstride, int w, int h, uint8_t *dst, size_t dstride, int dilate)
{
    int j1[31], j2[31];
    ellipse31_rows(j1, j2);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int acc = dilate ? 0 : 255; /* outside the image never wins (morphologyDefaultBorderValue) */
            for (int i = 0; i < 31; i++) {
                int yy = y + i - 15;
                if (yy < 0 || yy >= h) continue;
                for (int j = j1[i]; j < j2[i]; j++) {
                    int xx = x + j - 15;
                    if (xx < 0 || xx >= w) continue;
                    int v = src[(size_t)yy * sstride + xx];
                    if (dilate ? v > acc : v < acc) acc = v;
                }
            }
            dst[(size_t)y * dstride + x] = (uint8_t)acc;
        }
}

void orc_close_ellipse31(const uint8_t *src, size_t sstride, int w, int h, uint8_t *dst, size_t dstride)
{
    uint8_t *tmp = (uint8_t *)malloc((size_t)w * h);
    morph31(src, sstride, w, h, tmp, w, 1);
    morph31(tmp, w, w, h, dst, dstride, 0);
    free(tmp);
}

/* ORBextractor::MovingKeyPoints, ORBextractor.cc:1688-1745 */
int orc_gate(orc_extractor *e, const uint8_t *mask, size_t mask_stride, const double *labels,
             size_t lstride, const int32_t *center_ids, int n_centers, const int32_t *rm_vector,
             int n_rm, amos_keypoint *removed, int cap, int *n_removed)
{
    if (!e || !e->detected || !mask || !n_removed) return AMOS_ERR_STATE;
    const int w = e->width, h = e->height;
    free(e->closed);
    e->closed = (uint8_t *)malloc((size_t)w * h);
    orc_close_ellipse31(mask, mask_stride, w, h, e->closed, w);
    int nrem = 0;
    for (int level = 0; level < e->nlevels; ++level) {
        int nl = e->nkps[level];
        if (nl == 0) continue;
        float scale = level != 0 ? e->scale[level] : 1.f;
        int keep = 0;
        for (int i = 0; i < nl; i++) {
            const amos_keypoint *kp = &e->kps[level][i];
            float sx = kp->x * scale, sy = kp->y * scale;
            int ix = (int)sx, iy = (int)sy;
            if (ix < 0 || iy < 0 || ix >= w || iy >= h) return AMOS_ERR_INVALID; /* UB in the reference */
            int dyna_flag = 0;
            if (labels) {
                double super_pixel = labels[(size_t)iy * lstride + ix];
                long ci = (long)(super_pixel - 1);
                if (ci < 0 || ci >= n_centers) return AMOS_ERR_INVALID;
                int id = center_ids[ci];
                if (id < 0 || id >= n_rm) return AMOS_ERR_INVALID;
                if (rm_vector[id] == 1) dyna_flag = 1;
            }
            int label_coord = e->closed[(size_t)iy * w + ix];
            if (label_coord != 0 || dyna_flag == 1) {
                if (nrem < cap && removed) removed[nrem] = *kp;
                nrem++;
            } else {
                e->kps[level][keep++] = *kp;
            }
        }
        e->nkps[level] = keep;
    }
    *n_removed = nrem;
    return nrem > cap && removed ? AMOS_ERR_CAPACITY : AMOS_OK;
}

int orc_closed_mask(const orc_extractor *e, uint8_t *dst, size_t dst_stride)
{
    if (!e || !e->closed) return AMOS_ERR_STATE;
    for (int y = 0; y < e->height; y++)
        memcpy(dst + (size_t)y * dst_stride, e->closed + (size_t)y * e->width, e->width);
    return AMOS_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* ORBmatcher::DescriptorDistance, ORBmatcher.cc:1913-1933                                      */
int orc_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

void orc_distances(const uint8_t *q, int nq, const uint8_t *t, int nt, uint16_t *out)
{
    for (int i = 0; i < nq; i++)
        for (int j = 0; j < nt; j++)
            out[(size_t)i * nt + j] = (uint16_t)orc_descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j);
}

void orc_list_distances(const uint8_t *q, int nq, const uint8_t *t, const int32_t *cand_off,
                        const int32_t *cand_idx, uint16_t *out)
{
    for (int i = 0; i < nq; i++)
        for (int k = cand_off[i]; k < cand_off[i + 1]; k++)
            out[k] = (uint16_t)orc_descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)cand_idx[k]);
}

/* best / second-best update exactly as ORBmatcher.cc:135-147 */
static inline void best2_update(amos_best2 *r, int dist, int idx)
{
    if (dist < r->best_dist) {
        r->second_dist = r->best_dist;
        r->second_idx = r->best_idx;
        r->best_dist = dist;
        r->best_idx = idx;
    } else if (dist < r->second_dist) {
        r->second_dist = dist;
        r->second_idx = idx;
    }
}

void orc_list_best2(const uint8_t *q, int nq, const uint8_t *t, const int32_t *cand_off,
                    const int32_t *cand_idx, int init_dist, amos_best2 *out)
{
    for (int i = 0; i < nq; i++) {
        amos_best2 r = {-1, init_dist, -1, init_dist};
        for (int k = cand_off[i]; k < cand_off[i + 1]; k++)
            best2_update(&r, orc_descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)cand_idx[k]), cand_idx[k]);
        out[i] = r;
    }
}

void orc_bruteforce_best2(const uint8_t *q, int nq, const uint8_t *t, int nt, int init_dist, amos_best2 *out)
{
    for (int i = 0; i < nq; i++) {
        amos_best2 r = {-1, init_dist, -1, init_dist};
        for (int j = 0; j < nt; j++)
            best2_update(&r, orc_descriptor_distance(q + 32 * (size_t)i, t + 32 * (size_t)j), j);
        out[i] = r;
    }
}

/* ORBmatcher::ComputeThreeMaxima, ORBmatcher.cc:1866-1908 */
void orc_three_maxima(const int32_t *histo, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            *ind3 = *ind2; *ind2 = *ind1; *ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            *ind3 = *ind2; *ind2 = i;
        } else if (s > max3) {
            max3 = s; *ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { *ind3 = -1; }
}

/* ------------------------------------------------------------------------------------------ */
/* Frame grid (Frame.cc:431-461, 1007-1030) and Frame::GetFeaturesInArea (Frame.cc:894-1003).   */
typedef struct {
    int *cell_start; /* [COLS*ROWS + 1] */
    int *items;      /* feature indices, insertion order inside a cell */
    float winv, hinv;
} ogrid;

static ogrid grid_build(const amos_frame_view *f)
{
    ogrid g;
    const int nc = AMOS_FRAME_GRID_COLS * AMOS_FRAME_GRID_ROWS;
    g.winv = (float)AMOS_FRAME_GRID_COLS / (float)(f->max_x - f->min_x);
    g.hinv = (float)AMOS_FRAME_GRID_ROWS / (float)(f->max_y - f->min_y);
    int *cell = (int *)malloc(sizeof(int) * (f->n > 0 ? f->n : 1));
    g.cell_start = (int *)calloc(nc + 1, sizeof(int));
    for (int i = 0; i < f->n; i++) {
        int px = (int)roundf((f->keys_un[i].x - f->min_x) * g.winv);
        int py = (int)roundf((f->keys_un[i].y - f->min_y) * g.hinv);
        if (px < 0 || px >= AMOS_FRAME_GRID_COLS || py < 0 || py >= AMOS_FRAME_GRID_ROWS) { cell[i] = -1; continue; }
        cell[i] = px * AMOS_FRAME_GRID_ROWS + py;
        g.cell_start[cell[i] + 1]++;
    }
    for (int c = 0; c < nc; c++) g.cell_start[c + 1] += g.cell_start[c];
    g.items = (int *)malloc(sizeof(int) * (f->n > 0 ? f->n : 1));
    int *fill = (int *)calloc(nc, sizeof(int));
    for (int i = 0; i < f->n; i++)
        if (cell[i] >= 0) g.items[g.cell_start[cell[i]] + fill[cell[i]]++] = i;
    free(fill); free(cell);
    return g;
}
static void grid_free(ogrid *g) { free(g->cell_start); free(g->items); }

static int grid_area(const ogrid *g, const amos_frame_view *f, float x, float y, float r, int minLevel, int maxLevel,
                     int32_t *out, int cap)
{
    int n = 0;
    int nMinCellX = (int)floorf((x - f->min_x - r) * g->winv); if (nMinCellX < 0) nMinCellX = 0;
    if (nMinCellX >= AMOS_FRAME_GRID_COLS) return 0;
    int nMaxCellX = (int)ceilf((x - f->min_x + r) * g->winv); if (nMaxCellX > AMOS_FRAME_GRID_COLS - 1) nMaxCellX = AMOS_FRAME_GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    int nMinCellY = (int)floorf((y - f->min_y - r) * g->hinv); if (nMinCellY < 0) nMinCellY = 0;
    if (nMinCellY >= AMOS_FRAME_GRID_ROWS) return 0;
    int nMaxCellY = (int)ceilf((y - f->min_y + r) * g->hinv); if (nMaxCellY > AMOS_FRAME_GRID_ROWS - 1) nMaxCellY = AMOS_FRAME_GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * AMOS_FRAME_GRID_ROWS + iy;
            for (int j = g->cell_start[c]; j < g->cell_start[c + 1]; j++) {
                const amos_keypoint *kp = &f->keys_un[g->items[j]];
                if (bCheckLevels) {
                    if (kp->octave < minLevel) continue;
                    if (maxLevel >= 0 && kp->octave > maxLevel) continue;
                }
                const float distx = kp->x - x, disty = kp->y - y;
                if (fabsf(distx) < r && fabsf(disty) < r) {
                    if (n < cap) out[n] = g->items[j];
                    n++;
                }
            }
        }
    return n;
}

int orc_features_in_area(const amos_frame_view *f, float x, float y, float r, int min_level, int max_level,
                         int32_t *out, int cap)
{
    ogrid g = grid_build(f);
    int n = grid_area(&g, f, x, y, r, min_level, max_level, out, cap);
    grid_free(&g);
    return n;
}

static void prune_histogram(int **hist, int *hn, int32_t *match, int *nmatches, int by_value)
{
    int32_t sizes[AMOS_HISTO_LENGTH];
    for (int i = 0; i < AMOS_HISTO_LENGTH; i++) sizes[i] = hn[i];
    int ind1 = -1, ind2 = -1, ind3 = -1;
    orc_three_maxima(sizes, AMOS_HISTO_LENGTH, &ind1, &ind2, &ind3);
    for (int i = 0; i < AMOS_HISTO_LENGTH; i++) {
        if (i == ind1 || i == ind2 || i == ind3) continue;
        for (int j = 0; j < hn[i]; j++) {
            if (by_value) { /* SearchForInitialization: only still-matched entries count (:626-630) */
                if (match[hist[i][j]] >= 0) { match[hist[i][j]] = -1; (*nmatches)--; }
            } else {        /* SearchByProjection(F,F): unconditional (:1716-1717) */
                match[hist[i][j]] = -1; (*nmatches)--;
            }
        }
    }
}

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono), :1569-1728 */
int orc_search_by_projection_frame(const amos_frame_view *cur, const amos_proj_query *q, int nq, int32_t *cur_match,
                                   const float *scale_factors, float mbf, float th, int forward, int backward,
                                   int check_orientation)
{
    ogrid g = grid_build(cur);
    int nmatches = 0;
    int *hist[AMOS_HISTO_LENGTH], hn[AMOS_HISTO_LENGTH];
    for (int i = 0; i < AMOS_HISTO_LENGTH; i++) { hist[i] = (int *)malloc(sizeof(int) * (nq + 1)); hn[i] = 0; }
    const float factor = AMOS_HISTO_LENGTH / 360.0f;
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (cur->n + 1));
    for (int i = 0; i < nq; i++) {
        const amos_proj_query *p = &q[i];
        const int nLastOctave = p->octave;
        const float radius = th * scale_factors[nLastOctave];
        int nc;
        if (forward) nc = grid_area(&g, cur, p->u, p->v, radius, nLastOctave, -1, cand, cur->n);
        else if (backward) nc = grid_area(&g, cur, p->u, p->v, radius, 0, nLastOctave, cand, cur->n);
        else nc = grid_area(&g, cur, p->u, p->v, radius, nLastOctave - 1, nLastOctave + 1, cand, cur->n);
        if (nc == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int k = 0; k < nc; k++) {
            const int i2 = cand[k];
            if (cur_match[i2] >= 0 && q[cur_match[i2]].has_obs) continue;
            if (cur->u_right && cur->u_right[i2] > 0) {
                const float ur = p->u - mbf * p->invz;
                const float er = fabsf(ur - cur->u_right[i2]);
                if (er > radius) continue;
            }
            const int dist = orc_descriptor_distance(p->desc, cur->descriptors + 32 * (size_t)i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= AMOS_TH_HIGH) {
            cur_match[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                float rot = p->angle - cur->keys_un[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == AMOS_HISTO_LENGTH) bin = 0;
                hist[bin][hn[bin]++] = bestIdx2;
            }
        }
    }
    if (check_orientation) prune_histogram(hist, hn, cur_match, &nmatches, 0);
    for (int i = 0; i < AMOS_HISTO_LENGTH; i++) free(hist[i]);
    free(cand);
    grid_free(&g);
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th), :70-175 */
int orc_search_by_projection_points(const amos_frame_view *f, const amos_map_query *q, int nq, int32_t *cur_match,
                                    uint8_t *cur_has_obs, const float *scale_factors, float th, float nn_ratio)
{
    ogrid g = grid_build(f);
    int nmatches = 0;
    const int bFactor = th != 1.0;
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (f->n + 1));
    for (int iMP = 0; iMP < nq; iMP++) {
        const amos_map_query *mp = &q[iMP];
        const int nPredictedLevel = mp->level;
        float r = mp->view_cos > 0.998 ? 2.5f : 4.0f; /* RadiusByViewingCos, :176-182 */
        if (bFactor) r *= th;
        const int nc = grid_area(&g, f, mp->proj_x, mp->proj_y, r * scale_factors[nPredictedLevel], nPredictedLevel - 1,
                                 nPredictedLevel, cand, f->n);
        if (nc == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < nc; k++) {
            const int idx = cand[k];
            if (cur_has_obs[idx]) continue;
            if (f->u_right && f->u_right[idx] > 0) {
                const float er = fabsf(mp->proj_xr - f->u_right[idx]);
                if (er > r * scale_factors[nPredictedLevel]) continue;
            }
            const int dist = orc_descriptor_distance(mp->desc, f->descriptors + 32 * (size_t)idx);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = f->keys_un[idx].octave;
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = f->keys_un[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= AMOS_TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nn_ratio * bestDist2) continue;
            cur_match[bestIdx] = iMP;
            cur_has_obs[bestIdx] = mp->has_obs != 0;
            nmatches++;
        }
    }
    free(cand);
    grid_free(&g);
    return nmatches;
}

/* ORBmatcher::SearchForInitialization, :515-643 */
int orc_search_for_initialization(const amos_frame_view *f1, const amos_frame_view *f2, float *prev_matched,
                                  int32_t *matches12, int window_size, float nn_ratio, int check_orientation)
{
    ogrid g = grid_build(f2);
    int nmatches = 0;
    for (int i = 0; i < f1->n; i++) matches12[i] = -1;
    int *hist[AMOS_HISTO_LENGTH], hn[AMOS_HISTO_LENGTH];
    for (int i = 0; i < AMOS_HISTO_LENGTH; i++) { hist[i] = (int *)malloc(sizeof(int) * (f1->n + 1)); hn[i] = 0; }
    const float factor = AMOS_HISTO_LENGTH / 360.0f;
    int *vMatchedDistance = (int *)malloc(sizeof(int) * (f2->n + 1));
    int *vnMatches21 = (int *)malloc(sizeof(int) * (f2->n + 1));
    for (int i = 0; i < f2->n; i++) { vMatchedDistance[i] = INT_MAX; vnMatches21[i] = -1; }
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (f2->n + 1));
    for (int i1 = 0; i1 < f1->n; i1++) {
        const amos_keypoint *kp1 = &f1->keys_un[i1];
        const int level1 = kp1->octave;
        if (level1 > 0) continue;
        const int nc = grid_area(&g, f2, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window_size, level1, level1, cand, f2->n);
        if (nc == 0) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int k = 0; k < nc; k++) {
            const int i2 = cand[k];
            const int dist = orc_descriptor_distance(f1->descriptors + 32 * (size_t)i1, f2->descriptors + 32 * (size_t)i2);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= AMOS_TH_LOW) {
            if (bestDist < (float)bestDist2 * nn_ratio) {
                if (vnMatches21[bestIdx2] >= 0) { matches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                matches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) {
                    float rot = f1->keys_un[i1].angle - f2->keys_un[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == AMOS_HISTO_LENGTH) bin = 0;
                    hist[bin][hn[bin]++] = i1;
                }
            }
        }
    }
    if (check_orientation) prune_histogram(hist, hn, matches12, &nmatches, 1);
    for (int i1 = 0; i1 < f1->n; i1++)
        if (matches12[i1] >= 0) {
            prev_matched[2 * i1] = f2->keys_un[matches12[i1]].x;
            prev_matched[2 * i1 + 1] = f2->keys_un[matches12[i1]].y;
        }
    for (int i = 0; i < AMOS_HISTO_LENGTH; i++) free(hist[i]);
    free(vMatchedDistance); free(vnMatches21); free(cand);
    grid_free(&g);
    return nmatches;
}

/* ------------------------------------------------------------------------------------------ */
/* Callers either side of the path.                                                  */
void orc_color_to_gray(const uint8_t *src, size_t sstride, int w, int h, int channels, int rgb_order, uint8_t *dst,
                       size_t dstride)
{
    const int RY15 = 9798, GY15 = 19235, BY15 = 3735; /* color.hpp: gray_shift = 15 */
    const int c0 = rgb_order ? RY15 : BY15, c2 = rgb_order ? BY15 : RY15;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const uint8_t *p = src + (size_t)y * sstride + (size_t)x * channels;
            dst[(size_t)y * dstride + x] = (uint8_t)((p[0] * c0 + p[1] * GY15 + p[2] * c2 + (1 << 14)) >> 15);
        }
}

float orc_depth_convert(uint16_t raw, float factor) { return (float)raw * factor; }

/* cv::undistortPoints(pts, pts, K, distCoef, Mat(), K) as Frame::UndistortKeyPoints / ComputeImageBounds call
 * it (Frame.cc:1052-1118, 1121-1170).  OpenCV 4.5 cvUndistortPointsInternal restated (PARITY UNPINNED, like the
 * other OpenCV stages): default criteria = exactly 5 iterations, double arithmetic, coefficients beyond k3 zero,
 * R = I, P = K.  dist = (k1, k2, p1, p2[, k3]); n_dist == 0 or k1 == 0 copies the input (Frame.cc:1057). */
void orc_undistort_points(const float *xy, int n, float fx_, float fy_, float cx_, float cy_, const float *dist, int n_dist,
                          float *out)
{
    double k[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < n_dist && i < 5; i++) k[i] = dist[i];
    if (n_dist == 0 || dist[0] == 0.0f) {
        memcpy(out, xy, sizeof(float) * 2 * (size_t)n);
        return;
    }
    const double fx = fx_, fy = fy_, cx = cx_, cy = cy_, ifx = 1. / fx, ify = 1. / fy;
    for (int i = 0; i < n; i++) {
        const double u = xy[2 * i], v = xy[2 * i + 1];
        double x = (u - cx) * ifx, y = (v - cy) * ify;
        const double x0 = x, y0 = y;
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + 0 * r2 + 0 * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
        out[2 * i] = (float)(xx * ww);
        out[2 * i + 1] = (float)(yy * ww);
    }
}

/* Frame::ComputeImageBounds, Frame.cc:1121-1170: bounds = {mnMinX, mnMaxX, mnMinY, mnMaxY}. */
void orc_image_bounds(int width, int height, float fx, float fy, float cx, float cy, const float *dist, int n_dist, float *bounds)
{
    if (n_dist == 0 || dist[0] == 0.0f) { bounds[0] = 0.f; bounds[1] = (float)width; bounds[2] = 0.f; bounds[3] = (float)height; return; }
    const float c[8] = {0.f, 0.f, (float)width, 0.f, 0.f, (float)height, (float)width, (float)height};
    float u[8];
    orc_undistort_points(c, 4, fx, fy, cx, cy, dist, n_dist, u);
    bounds[0] = u[0] < u[4] ? u[0] : u[4];
    bounds[1] = u[2] > u[6] ? u[2] : u[6];
    bounds[2] = u[1] < u[3] ? u[1] : u[3];
    bounds[3] = u[5] > u[7] ? u[5] : u[7];
}

void orc_rgbd_glue(const amos_keypoint *kps, const amos_keypoint *kps_un, int n, const float *depth, size_t depth_stride_elems, int w, int h,
                   float mbf, float min_x, float max_x, float min_y, float max_y, float *u_right, float *depth_out, int32_t *grid_cell)
{
    if (!kps_un) kps_un = kps; /* zero-distortion camera: mvKeysUn == mvKeys */
    const float winv = (float)AMOS_FRAME_GRID_COLS / (float)(max_x - min_x);
    const float hinv = (float)AMOS_FRAME_GRID_ROWS / (float)(max_y - min_y);
    for (int i = 0; i < n; i++) {
        const float v = kps[i].y, u = kps[i].x;
        u_right[i] = -1.f;
        depth_out[i] = -1.f;
        if ((int)u >= 0 && (int)v >= 0 && (int)u < w && (int)v < h) { /* outside is UB in the reference */
            const float d = depth[(size_t)(int)v * depth_stride_elems + (int)u];
            if (d > 0) {
                depth_out[i] = d;
                u_right[i] = kps_un[i].x - mbf / d; /* kpU.pt.x - mbf/d, Frame.cc:1607 */
            }
        }
        const int px = (int)roundf((kps_un[i].x - min_x) * winv), py = (int)roundf((kps_un[i].y - min_y) * hinv);
        grid_cell[i] = (px < 0 || px >= AMOS_FRAME_GRID_COLS || py < 0 || py >= AMOS_FRAME_GRID_ROWS) ? -1 : px * AMOS_FRAME_GRID_ROWS + py;
    }
}

/* Checker of amos_match_window_best2_batch_device: Frame::GetFeaturesInArea (Frame.cc:894-1003) feeding
 * the best / second-best loop of ORBmatcher::SearchByProjection(F, LastF) (ORBmatcher.cc:1629-1690)
 * WITHOUT the greedy already-matched skip (which the caller resolves).  query_uv / query_invz may be
 * NULL (own position; no stereo gate). */
void orc_window_best2(const amos_frame_view *train, const amos_keypoint *qk, const uint8_t *qdesc, int nq, const float *query_uv,
                      const float *query_invz, const float *scale_factors, float th, float mbf, int mode, int init_dist,
                      amos_best2 *out)
{
    ogrid g = grid_build(train);
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (train->n + 1));
    for (int i = 0; i < nq; i++) {
        const float u = query_uv ? query_uv[2 * i] : qk[i].x, v = query_uv ? query_uv[2 * i + 1] : qk[i].y;
        const int oct = qk[i].octave;
        const float radius = th * scale_factors[oct];
        int nc;
        if (mode == 1) nc = grid_area(&g, train, u, v, radius, oct, -1, cand, train->n);
        else if (mode == 2) nc = grid_area(&g, train, u, v, radius, 0, oct, cand, train->n);
        else nc = grid_area(&g, train, u, v, radius, oct - 1, oct + 1, cand, train->n);
        amos_best2 r = {-1, init_dist, -1, init_dist};
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            if (query_invz && train->u_right && train->u_right[i2] > 0) {
                const float ur = u - mbf * query_invz[i];
                const float er = fabsf(ur - train->u_right[i2]);
                if (er > radius) continue;
            }
            const int d = orc_descriptor_distance(qdesc + (size_t)i * 32, train->descriptors + (size_t)i2 * 32);
            if (d < r.best_dist) { r.second_dist = r.best_dist; r.second_idx = r.best_idx; r.best_dist = d; r.best_idx = i2; }
            else if (d < r.second_dist) { r.second_dist = d; r.second_idx = i2; }
        }
        out[i] = r;
    }
    free(cand);
    grid_free(&g);
}

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist), ORBmatcher.cc:1731-1863,
 * from the candidate search on (the projection / filtering of :1745-1795 is the caller's, see amos_kf_query). */
int orc_search_by_projection_kf(const amos_frame_view *cur, const amos_kf_query *q, int nq, int32_t *cur_match,
                                const float *scale_factors, float th, int orb_dist, int check_orientation)
{
    ogrid g = grid_build(cur);
    int nmatches = 0;
    int *hist[AMOS_HISTO_LENGTH], hn[AMOS_HISTO_LENGTH];
    for (int i = 0; i < AMOS_HISTO_LENGTH; i++) { hist[i] = (int *)malloc(sizeof(int) * (nq + 1)); hn[i] = 0; }
    const float factor = AMOS_HISTO_LENGTH / 360.0f;
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (cur->n + 1));
    for (int i = 0; i < nq; i++) {
        const int lvl = q[i].level;
        const float radius = th * scale_factors[lvl];
        const int nc = grid_area(&g, cur, q[i].u, q[i].v, radius, lvl - 1, lvl + 1, cand, cur->n);
        if (nc == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            if (cur_match[i2] != AMOS_MATCH_FREE) continue; /* :1816-1817 */
            const int dist = orc_descriptor_distance(q[i].desc, cur->descriptors + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= orb_dist) {
            cur_match[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                float rot = q[i].angle - cur->keys_un[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)round(rot * factor);
                if (bin == AMOS_HISTO_LENGTH) bin = 0;
                hist[bin][hn[bin]++] = bestIdx2;
            }
        }
    }
    if (check_orientation) {
        int32_t sizes[AMOS_HISTO_LENGTH];
        for (int i = 0; i < AMOS_HISTO_LENGTH; i++) sizes[i] = hn[i];
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(sizes, AMOS_HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < AMOS_HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < hn[i]; j++) { cur_match[hist[i][j]] = AMOS_MATCH_FREE; nmatches--; }
        }
    }
    for (int i = 0; i < AMOS_HISTO_LENGTH; i++) free(hist[i]);
    free(cand);
    grid_free(&g);
    return nmatches;
}

/* ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vpMapPointMatches), ORBmatcher.cc:230-382. */
int orc_search_by_bow(const amos_bow_view *kf, const amos_bow_view *f, int32_t *matches_f, float nn_ratio, int check_orientation)
{
    for (int i = 0; i < f->n; i++) matches_f[i] = -1;
    int nmatches = 0;
    int *hist[AMOS_HISTO_LENGTH], hn[AMOS_HISTO_LENGTH];
    for (int i = 0; i < AMOS_HISTO_LENGTH; i++) { hist[i] = (int *)malloc(sizeof(int) * (f->n + 1)); hn[i] = 0; }
    const float factor = AMOS_HISTO_LENGTH / 360.0f;
    int a = 0, b = 0;
    while (a < kf->n_nodes && b < f->n_nodes) {
        if (kf->node_ids[a] == f->node_ids[b]) {
            for (int ik = kf->node_off[a]; ik < kf->node_off[a + 1]; ik++) {
                const int realIdxKF = kf->node_idx[ik];
                if (kf->has_point && !kf->has_point[realIdxKF]) continue;
                const uint8_t *dKF = kf->descriptors + (size_t)realIdxKF * 32;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int jf = f->node_off[b]; jf < f->node_off[b + 1]; jf++) {
                    const int realIdxF = f->node_idx[jf];
                    if (matches_f[realIdxF] >= 0) continue;
                    const int dist = orc_descriptor_distance(dKF, f->descriptors + (size_t)realIdxF * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= AMOS_TH_LOW) {
                    if ((float)bestDist1 < nn_ratio * (float)bestDist2) {
                        matches_f[bestIdxF] = realIdxKF;
                        if (check_orientation) {
                            float rot = kf->keys[realIdxKF].angle - f->keys[bestIdxF].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)round(rot * factor);
                            if (bin == AMOS_HISTO_LENGTH) bin = 0;
                            hist[bin][hn[bin]++] = bestIdxF;
                        }
                        nmatches++;
                    }
                }
            }
            a++; b++;
        } else if (kf->node_ids[a] < f->node_ids[b]) {
            while (a < kf->n_nodes && kf->node_ids[a] < f->node_ids[b]) a++;
        } else {
            while (b < f->n_nodes && f->node_ids[b] < kf->node_ids[a]) b++;
        }
    }
    if (check_orientation) {
        int32_t sizes[AMOS_HISTO_LENGTH];
        for (int i = 0; i < AMOS_HISTO_LENGTH; i++) sizes[i] = hn[i];
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(sizes, AMOS_HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < AMOS_HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < hn[i]; j++) { matches_f[hist[i][j]] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < AMOS_HISTO_LENGTH; i++) free(hist[i]);
    return nmatches;
}

/* ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vpMatches12), ORBmatcher.cc:656-808. */
int orc_search_by_bow_kf(const amos_bow_view *k1, const amos_bow_view *k2, int32_t *matches12, float nn_ratio, int check_orientation)
{
    for (int i = 0; i < k1->n; i++) matches12[i] = -1;
    uint8_t *matched2 = (uint8_t *)calloc(k2->n + 1, 1);
    int nmatches = 0;
    int *hist[AMOS_HISTO_LENGTH], hn[AMOS_HISTO_LENGTH];
    for (int i = 0; i < AMOS_HISTO_LENGTH; i++) { hist[i] = (int *)malloc(sizeof(int) * (k1->n + 1)); hn[i] = 0; }
    const float factor = AMOS_HISTO_LENGTH / 360.0f;
    int a = 0, b = 0;
    while (a < k1->n_nodes && b < k2->n_nodes) {
        if (k1->node_ids[a] == k2->node_ids[b]) {
            for (int i1 = k1->node_off[a]; i1 < k1->node_off[a + 1]; i1++) {
                const int idx1 = k1->node_idx[i1];
                if (k1->has_point && !k1->has_point[idx1]) continue;
                const uint8_t *d1 = k1->descriptors + (size_t)idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int i2 = k2->node_off[b]; i2 < k2->node_off[b + 1]; i2++) {
                    const int idx2 = k2->node_idx[i2];
                    if (matched2[idx2] || (k2->has_point && !k2->has_point[idx2])) continue;
                    const int dist = orc_descriptor_distance(d1, k2->descriptors + (size_t)idx2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < AMOS_TH_LOW) {
                    if ((float)bestDist1 < nn_ratio * (float)bestDist2) {
                        matches12[idx1] = bestIdx2;
                        matched2[bestIdx2] = 1;
                        if (check_orientation) {
                            float rot = k1->keys[idx1].angle - k2->keys[bestIdx2].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)round(rot * factor);
                            if (bin == AMOS_HISTO_LENGTH) bin = 0;
                            hist[bin][hn[bin]++] = idx1;
                        }
                        nmatches++;
                    }
                }
            }
            a++; b++;
        } else if (k1->node_ids[a] < k2->node_ids[b]) {
            while (a < k1->n_nodes && k1->node_ids[a] < k2->node_ids[b]) a++;
        } else {
            while (b < k2->n_nodes && k2->node_ids[b] < k1->node_ids[a]) b++;
        }
    }
    if (check_orientation) {
        int32_t sizes[AMOS_HISTO_LENGTH];
        for (int i = 0; i < AMOS_HISTO_LENGTH; i++) sizes[i] = hn[i];
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(sizes, AMOS_HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < AMOS_HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < hn[i]; j++) { matches12[hist[i][j]] = -1; nmatches--; }
        }
    }
    for (int i = 0; i < AMOS_HISTO_LENGTH; i++) free(hist[i]);
    free(matched2);
    return nmatches;
}

/* ORBmatcher::CheckDistEpipolarLine, ORBmatcher.cc:188-215 (plain float arithmetic, no fused multiply-add). */
static int check_dist_epipolar_line(const amos_keypoint *kp1, const amos_keypoint *kp2, const float *F12, float sigma2)
{
    const float a = kp1->x * F12[0] + kp1->y * F12[3] + F12[6];
    const float b = kp1->x * F12[1] + kp1->y * F12[4] + F12[7];
    const float c = kp1->x * F12[2] + kp1->y * F12[5] + F12[8];
    const float num = a * kp2->x + b * kp2->y + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2;
}

/* ORBmatcher::SearchForTriangulation, ORBmatcher.cc:810-1018.  pairs: (idx1, idx2) in ascending idx1. */
int orc_search_for_triangulation(const amos_bow_view *k1, const amos_bow_view *k2, const float *F12, float ex, float ey,
                                 const float *scale_factors2, const float *level_sigma2_2, int only_stereo, int check_orientation,
                                 int32_t *pairs, int cap)
{
    int nmatches = 0;
    uint8_t *matched2 = (uint8_t *)calloc(k2->n + 1, 1);
    int32_t *m12 = (int32_t *)malloc(sizeof(int32_t) * (k1->n + 1));
    for (int i = 0; i < k1->n; i++) m12[i] = -1;
    int *hist[AMOS_HISTO_LENGTH], hn[AMOS_HISTO_LENGTH];
    for (int i = 0; i < AMOS_HISTO_LENGTH; i++) { hist[i] = (int *)malloc(sizeof(int) * (k1->n + 1)); hn[i] = 0; }
    const float factor = AMOS_HISTO_LENGTH / 360.0f;
    int a = 0, b = 0;
    while (a < k1->n_nodes && b < k2->n_nodes) {
        if (k1->node_ids[a] == k2->node_ids[b]) {
            for (int i1 = k1->node_off[a]; i1 < k1->node_off[a + 1]; i1++) {
                const int idx1 = k1->node_idx[i1];
                if (k1->has_point && k1->has_point[idx1]) continue;
                const int bStereo1 = k1->u_right ? k1->u_right[idx1] >= 0 : 0;
                if (only_stereo && !bStereo1) continue;
                const amos_keypoint *kp1 = &k1->keys[idx1];
                const uint8_t *d1 = k1->descriptors + (size_t)idx1 * 32;
                int bestDist = AMOS_TH_LOW, bestIdx2 = -1;
                for (int i2 = k2->node_off[b]; i2 < k2->node_off[b + 1]; i2++) {
                    const int idx2 = k2->node_idx[i2];
                    if (matched2[idx2] || (k2->has_point && k2->has_point[idx2])) continue;
                    const int bStereo2 = k2->u_right ? k2->u_right[idx2] >= 0 : 0;
                    if (only_stereo && !bStereo2) continue;
                    const int dist = orc_descriptor_distance(d1, k2->descriptors + (size_t)idx2 * 32);
                    if (dist > AMOS_TH_LOW || dist > bestDist) continue;
                    const amos_keypoint *kp2 = &k2->keys[idx2];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - kp2->x, distey = ey - kp2->y;
                        if (distex * distex + distey * distey < 100 * scale_factors2[kp2->octave]) continue;
                    }
                    if (check_dist_epipolar_line(kp1, kp2, F12, level_sigma2_2[kp2->octave])) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    m12[idx1] = bestIdx2;
                    matched2[bestIdx2] = 1;
                    nmatches++;
                    if (check_orientation) {
                        float rot = kp1->angle - k2->keys[bestIdx2].angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)round(rot * factor);
                        if (bin == AMOS_HISTO_LENGTH) bin = 0;
                        hist[bin][hn[bin]++] = idx1;
                    }
                }
            }
            a++; b++;
        } else if (k1->node_ids[a] < k2->node_ids[b]) {
            while (a < k1->n_nodes && k1->node_ids[a] < k2->node_ids[b]) a++;
        } else {
            while (b < k2->n_nodes && k2->node_ids[b] < k1->node_ids[a]) b++;
        }
    }
    if (check_orientation) {
        int32_t sizes[AMOS_HISTO_LENGTH];
        for (int i = 0; i < AMOS_HISTO_LENGTH; i++) sizes[i] = hn[i];
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(sizes, AMOS_HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < AMOS_HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j = 0; j < hn[i]; j++) { m12[hist[i][j]] = -1; nmatches--; }
        }
    }
    int np = 0;
    for (int i = 0; i < k1->n; i++)
        if (m12[i] >= 0 && np < cap) { pairs[2 * np] = i; pairs[2 * np + 1] = m12[i]; np++; }
    for (int i = 0; i < AMOS_HISTO_LENGTH; i++) free(hist[i]);
    free(matched2); free(m12);
    return nmatches;
}

/* The per-map-point search of Fuse (ORBmatcher.cc:1085-1132 with the chi2 gate, :1251-1273 without),
 * SearchByProjection(pKF, Scw, ...) (:455-500, greedy on `occupied`) and both passes of SearchBySim3
 * (:1394-1425, :1474-1505): window th * scale[level] with KeyFrame::GetFeaturesInArea (no level filter), level gate
 * nPredictedLevel-1 .. nPredictedLevel, best by strict <, accepted at bestDist <= max_dist.
 * best_idx[q] = accepted feature or -1; occupied (may be NULL): AMOS_MATCH_FREE entries only are eligible and an
 * accepted feature becomes the query's index.  Returns the number of accepted queries. */
int orc_window_search(const amos_frame_view *kf, const amos_window_query *q, int nq, const float *scale_factors,
                      const float *inv_level_sigma2, float th, int max_dist, int32_t *occupied, int32_t *best_idx)
{
    ogrid g = grid_build(kf);
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (kf->n + 1));
    int naccepted = 0;
    for (int i = 0; i < nq; i++) {
        const int lvl = q[i].level;
        const float radius = th * scale_factors[lvl];
        const float u = q[i].u, v = q[i].v, ur = q[i].ur;
        const int nc = grid_area(&g, kf, u, v, radius, -1, -1, cand, kf->n);
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            if (occupied && occupied[idx] != AMOS_MATCH_FREE) continue;
            const amos_keypoint *kp = &kf->keys_un[idx];
            const int kpLevel = kp->octave;
            if (kpLevel < lvl - 1 || kpLevel > lvl) continue;
            if (inv_level_sigma2) {
                if (kf->u_right && kf->u_right[idx] >= 0) {
                    const float ex = u - kp->x, ey = v - kp->y, er = ur - kf->u_right[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
                } else {
                    const float ex = u - kp->x, ey = v - kp->y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = orc_descriptor_distance(q[i].desc, kf->descriptors + (size_t)idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        best_idx[i] = -1;
        if (bestDist <= max_dist) {
            best_idx[i] = bestIdx;
            if (occupied) occupied[bestIdx] = i;
            naccepted++;
        }
    }
    free(cand);
    grid_free(&g);
    return naccepted;
}

/* ORBmatcher::SearchBySim3, ORBmatcher.cc:1314-1565, from the two projected point sets on. */
int orc_search_by_sim3(const amos_frame_view *kf1, const amos_frame_view *kf2, const amos_window_query *q12, int n12,
                       const amos_window_query *q21, int n21, const float *scale_factors1, const float *scale_factors2, float th,
                       int32_t *matches12)
{
    int32_t *m1 = (int32_t *)malloc(sizeof(int32_t) * (kf1->n + 1)), *m2 = (int32_t *)malloc(sizeof(int32_t) * (kf2->n + 1));
    int32_t *b12 = (int32_t *)malloc(sizeof(int32_t) * (n12 + 1)), *b21 = (int32_t *)malloc(sizeof(int32_t) * (n21 + 1));
    for (int i = 0; i < kf1->n; i++) m1[i] = -1;
    for (int i = 0; i < kf2->n; i++) m2[i] = -1;
    orc_window_search(kf2, q12, n12, scale_factors2, NULL, th, AMOS_TH_HIGH, NULL, b12);
    for (int i = 0; i < n12; i++) if (b12[i] >= 0) m1[q12[i].src] = b12[i];
    orc_window_search(kf1, q21, n21, scale_factors1, NULL, th, AMOS_TH_HIGH, NULL, b21);
    for (int i = 0; i < n21; i++) if (b21[i] >= 0) m2[q21[i].src] = b21[i];
    int nFound = 0;
    for (int i1 = 0; i1 < kf1->n; i1++) {
        matches12[i1] = -1;
        const int idx2 = m1[i1];
        if (idx2 >= 0 && m2[idx2] == i1) { matches12[i1] = idx2; nFound++; }
    }
    free(m1); free(m2); free(b12); free(b21);
    return nFound;
}

/* ---------------------------------------------------------------- cluster::SLIC (src/cluster.cc) ----------
 * From the Lab image on (cluster.cc:310's cvtColor stays with the caller).  Sequential, loop for loop as the reference:
 * Sobel + addWeighted (:314-316), initilizeCenters (:212-244), fituneCenter (:246-298), `iterations` x { clustering
 * (:88-158), updateCenter (:160-210) }.  Doubles throughout, no fused multiply-add.  Returns the centre count. */
static int orc_refl101(int i, int n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; return i; }

int orc_slic(const uint8_t *lab, const uint16_t *depth, int w, int h, int len, int m, int iterations, double *labels,
             amos_slic_center *centers, int cap)
{
    const size_t px = (size_t)w * h;
    /* cv::Sobel(imageLAB, CV_64F, 0, 1, 3), cv::Sobel(.., 1, 0, 3), addWeighted(0.5, 0.5): 3 channels, REFLECT_101 */
    double *grad = (double *)malloc(sizeof(double) * px * 3);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            for (int c = 0; c < 3; c++) {
                const int ym = orc_refl101(y - 1, h), yp = orc_refl101(y + 1, h), xm = orc_refl101(x - 1, w), xp = orc_refl101(x + 1, w);
#define LABP(yy, xx) ((int)lab[((size_t)(yy) * w + (xx)) * 3 + c])
                const double sx = (double)((LABP(yp, xm) + 2 * LABP(yp, x) + LABP(yp, xp)) - (LABP(ym, xm) + 2 * LABP(ym, x) + LABP(ym, xp)));
                const double sy = (double)((LABP(ym, xp) + 2 * LABP(y, xp) + LABP(yp, xp)) - (LABP(ym, xm) + 2 * LABP(y, xm) + LABP(yp, xm)));
#undef LABP
                grad[((size_t)y * w + x) * 3 + c] = sx * 0.5 + sy * 0.5;
            }
    /* initilizeCenters */
    int n = 0;
    for (int i = 0; i < h; i += len) {
        const int cy = i + len / 2;
        if (cy >= h) continue;
        for (int j = 0; j < w; j += len) {
            const int cx = j + len / 2;
            if (cx >= w) continue;
            if (n < cap) {
                amos_slic_center *ce = &centers[n];
                ce->x = cx; ce->y = cy;
                ce->L = lab[((size_t)cy * w + cx) * 3]; ce->A = lab[((size_t)cy * w + cx) * 3 + 1]; ce->B = lab[((size_t)cy * w + cx) * 3 + 2];
                ce->label = n + 1;
                ce->D = depth[(size_t)cy * w + cx];
                ce->id = 0;
            }
            n++;
        }
    }
    if (n > cap) { free(grad); return n; }
    /* fituneCenter */
    for (int ck = 0; ck < n; ck++) {
        amos_slic_center cent = centers[ck];
        if (cent.x - 1 < 0 || cent.x + 1 >= w || cent.y - 1 < 0 || cent.y + 1 >= h) continue;
        double minGradient = 9999999;
        int tempx = 0, tempy = 0;
        for (int mm = -1; mm < 2; mm++)
            for (int nn = -1; nn < 2; nn++) {
                const double *g = &grad[((size_t)(cent.y + mm) * w + (cent.x + nn)) * 3];
                const double gradient = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
                if (gradient < minGradient) { minGradient = gradient; tempy = mm; tempx = nn; }
            }
        cent.x += tempx;
        cent.y += tempy;
        centers[ck].x = cent.x;
        centers[ck].y = cent.y;
        centers[ck].L = lab[((size_t)cent.y * w + cent.x) * 3];
        centers[ck].A = lab[((size_t)cent.y * w + cent.x) * 3 + 1];
        centers[ck].B = lab[((size_t)cent.y * w + cent.x) * 3 + 2];
    }
    free(grad);
    for (size_t p = 0; p < px; p++) labels[p] = 0;
    double *dis_mask = (double *)malloc(sizeof(double) * px);
    for (int time = 0; time < iterations; time++) {
        for (size_t p = 0; p < px; p++) dis_mask[p] = 999999;
        for (int ck = 0; ck < n; ck++) { /* clustering */
            const int cx = centers[ck].x, cy = centers[ck].y, cL = centers[ck].L, cA = centers[ck].A, cB = centers[ck].B;
            for (int i = cy - len; i < cy + len; i++) {
                if (i < 0 || i >= h) continue;
                for (int j = cx - len; j < cx + len; j++) {
                    if (j < 0 || j >= w) continue;
                    const int L = lab[((size_t)i * w + j) * 3], A = lab[((size_t)i * w + j) * 3 + 1], B = lab[((size_t)i * w + j) * 3 + 2];
                    const double disc = sqrt((double)((L - cL) * (L - cL) + (A - cA) * (A - cA) + (B - cB) * (B - cB)));
                    const double diss = sqrt((double)((j - cx) * (j - cx) + (i - cy) * (i - cy)));
                    const double dis = sqrt(disc * disc + m * (diss * diss));
                    if (dis < dis_mask[(size_t)i * w + j]) {
                        dis_mask[(size_t)i * w + j] = dis;
                        labels[(size_t)i * w + j] = centers[ck].label;
                    }
                }
            }
        }
        for (int ck = 0; ck < n; ck++) { /* updateCenter */
            double sumx = 0, sumy = 0, sumL = 0, sumA = 0, sumB = 0, sumNum = 0, sumD = 0;
            const int cx = centers[ck].x, cy = centers[ck].y;
            for (int i = cy - len; i < cy + len; i++) {
                if (i < 0 || i >= h) continue;
                for (int j = cx - len; j < cx + len; j++) {
                    if (j < 0 || j >= w) continue;
                    if (labels[(size_t)i * w + j] == centers[ck].label) {
                        sumL += lab[((size_t)i * w + j) * 3];
                        sumA += lab[((size_t)i * w + j) * 3 + 1];
                        sumB += lab[((size_t)i * w + j) * 3 + 2];
                        sumx += j;
                        sumy += i;
                        sumNum += 1;
                        sumD += (int)depth[(size_t)i * w + j];
                    }
                }
            }
            if (sumNum == 0) sumNum = 0.000000001;
            centers[ck].x = (int)(sumx / sumNum);
            centers[ck].y = (int)(sumy / sumNum);
            centers[ck].L = (int)(sumL / sumNum);
            centers[ck].A = (int)(sumA / sumNum);
            centers[ck].B = (int)(sumB / sumNum);
            centers[ck].D = (int)(sumD / sumNum);
        }
    }
    free(dis_mask);
    return n;
}

/* ---------------------------------------------------------------- cluster::randCent + kmeans (src/cluster.cc:353-460) ----
 * The k-means over the SLIC centres that gives every superpixel its cluster id (centers[label - 1].id, cluster.cc:18-24),
 * loop for loop as the reference, with its three undefined behaviours given a definition:
 *   - rand() (libc state shared with the viewer thread, FrameDrawer.cc:198) -> glibc's TYPE_0 generator on an explicit seed:
 *     state = state * 1103515245 + 12345, value = state & 0x7fffffff;
 *   - dataSet[rand() % rowLen + 1] reads one past the end for the last value (:358) -> index rowLen wraps to 0;
 *     the `while (temp.D <= 0)` redraw stops after 4 * rowLen draws (the reference spins forever on an all-zero depth map);
 *   - the accumulator `center vec;` of the update step is uninitialised (:416) -> zero.
 * distEclud (:374-387): |dD| / 20000 + sqrt(dx^2 + dy^2) / sqrt(640^2 + 480^2), doubles, strict < keeps the first centroid.
 * The reference loops until no assignment changes; max_iter bounds that (returns the passes made, -1 if the bound was hit). */
static uint32_t orc_lcg(uint32_t *state)
{
    *state = *state * 1103515245u + 12345u;
    return *state & 0x7fffffffu;
}

int orc_kmeans(amos_slic_center *centers, int n, int k, uint32_t seed, int max_iter)
{
    if (n < 1 || k < 1) return 0;
    int *cx = (int *)malloc(sizeof(int) * k), *cy = (int *)malloc(sizeof(int) * k), *cd = (int *)malloc(sizeof(int) * k);
    int *assign = (int *)malloc(sizeof(int) * n);
    uint32_t state = seed;
    for (int i = 0; i < k; i++) { /* randCent */
        int idx = (int)(orc_lcg(&state) % (uint32_t)n) + 1;
        if (idx >= n) idx = 0;
        for (int tries = 0; centers[idx].D <= 0 && tries < 4 * n; tries++) {
            idx = (int)(orc_lcg(&state) % (uint32_t)n) + 1;
            if (idx >= n) idx = 0;
        }
        cx[i] = centers[idx].x; cy[i] = centers[idx].y; cd[i] = centers[idx].D;
    }
    for (int i = 0; i < n; i++) assign[i] = -1;
    const double max_D = 20000, max_E = sqrt((double)(640 * 640 + 480 * 480));
    int passes = 0, changed = 1;
    while (changed) {
        if (passes >= max_iter) { passes = -1; break; }
        changed = 0;
        passes++;
        for (int i = 0; i < n; i++) {
            int minIndex = -1;
            double minDist = 2147483647.0;
            for (int j = 0; j < k; j++) {
                const double sum_D = abs(centers[i].D - cd[j]) / max_D;
                const int dx = cx[j] - centers[i].x, dy = cy[j] - centers[i].y;
                const double sum_E = sqrt((double)(dx * dx + dy * dy)) / max_E;
                const double dist = 1 * sum_E + 1 * sum_D;
                if (dist < minDist) { minDist = dist; minIndex = j; }
            }
            if (assign[i] != minIndex) { changed = 1; assign[i] = minIndex; }
        }
        for (int c = 0; c < k; c++) {
            int sx = 0, sy = 0, sd = 0, cnt = 0;
            for (int i = 0; i < n; i++)
                if (assign[i] == c) { cnt++; sx += centers[i].x; sy += centers[i].y; sd += centers[i].D; }
            if (cnt != 0) { sx /= cnt; sy /= cnt; sd /= cnt; }
            cx[c] = sx; cy[c] = sy; cd[c] = sd;
        }
    }
    for (int i = 0; i < n; i++) /* :448-452 + cluster.cc:18-24 */
        if (centers[i].label >= 1 && centers[i].label <= n) centers[centers[i].label - 1].id = assign[i];
    free(cx); free(cy); free(cd); free(assign);
    return passes;
}

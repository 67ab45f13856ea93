// amos_orb.hip -- host side of the ORB extractor: geometry, device buffers, launches and the
// C ABI declared in include/amos_frontend.h.  Kernels are in orb_kernels.h.
//
// Data layout in HBM (per handle, B = max_batch frames):
//   pyramid   [B][frameBytes]   per frame: L padded planes, plane l = (h_l+38) rows of stride_l
//                               bytes (stride multiple of 128, ROI origin at byte 32 of row 19)
//   blurred   [B][frameBytes]   same geometry, only the ROI is written
//   slots     [B][slotTotal]    u32 FAST candidates per cell (worst-case capacity per cell)
//   pts       [B][ptsTotal]     u32 candidates per level, compacted in the reference's order
//   lvKps     [B][kpLevelTotal] amos_keypoint lists per level (level coordinates)
//   outKps / outDesc / outCount [B][kpCap] concatenated result (level-0 coordinates, 32 B rows)
#include "orb_kernels.h"

#include <cmath>
#include <cstdlib>
#include <cstdarg>
#include <cstring>
#include <vector>

#ifndef AMOS_FAST_LDS_PAD
#define AMOS_FAST_LDS_PAD 0  /* experiments (tools/orb_variants.sh): extra LDS bytes per FAST wave, to hold the occupancy down */
#endif
namespace amos {

static thread_local std::string g_error;

void set_error(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
}

static inline int align_up(int v, int a) { return (v + a - 1) / a * a; }
static inline size_t align_up_sz(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cv_round(float v) { return (int)lrintf(v); }  // round half to even

// events of one timed pass: 0 start, 1 after the level-0 import, 2 after the last resize level, 3 before FAST,
// 4 after FAST, 5 after the quad-tree, 6 after the orientation, 7 before rBRIEF, 8 after rBRIEF
constexpr int kTimingEvents = 9;

}  // namespace amos

using namespace amos;

struct amos_orb {
    amos_orb_params p{};
    int maxW = 0, maxH = 0, maxB = 0, device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    // side stream: the blur only needs the pyramid, so it runs beside the latency-bound quad-tree
    hipStream_t streamB = nullptr;
    hipEvent_t evFork = nullptr, evJoin = nullptr, evBlur0 = nullptr, evBlur1 = nullptr;
    bool blurDone = false;  // the blurred planes of the current frame(s) exist
    size_t octLdsAttr = 0;  // dynamic LDS limit last set on k_octree
    // a1 tables
    float scale[AMOS_MAX_LEVELS]{}, invScale[AMOS_MAX_LEVELS]{}, sigma2[AMOS_MAX_LEVELS]{}, invSigma2[AMOS_MAX_LEVELS]{};
    int quota[AMOS_MAX_LEVELS]{};
    int umax[16]{};
    // geometry of the current frame size and of the allocation
    Geom geom{};
    std::vector<Cell> cells;
    std::vector<ResizeTap> taps;
    int curW = 0, curH = 0;
    Geom capGeom{};  // geometry at (maxW, maxH): sizes every buffer
    size_t capCells = 0, capTaps = 0;
    int octNC = 0, octSC = 0;
    // device buffers
    Geom *dGeom = nullptr;
    Cell *dCells = nullptr;
    ResizeTap *dTaps = nullptr;
    uint8_t *dPyr = nullptr, *dBlur = nullptr, *dInput = nullptr;
    int *dSlotCount = nullptr;
    uint32_t *dSlots = nullptr, *dPts = nullptr;
    uint16_t *dNodeOf = nullptr;
    uint8_t *dQuadOf = nullptr;
    int *dCandCount = nullptr, *dLvCount = nullptr, *dOutCount = nullptr;
    amos_keypoint *dLvKps = nullptr, *dOutKps = nullptr, *dRemoved = nullptr, *dScratchKps = nullptr;
    uint8_t *dOutDesc = nullptr;
    uint8_t *dMask = nullptr, *dMaskTmp = nullptr, *dMaskClosed = nullptr;
    uint8_t *hStage = nullptr;  // pinned host staging for fetch_frame: count, one frame's keypoints and descriptors
    uint8_t *hPyrStage = nullptr;  // pinned host staging for one frame's pyramid (level images to host Mats), made on first use
    double *dLabels = nullptr;
    int *dCenterIds = nullptr, *dRm = nullptr, *dNRemoved = nullptr, *dErr = nullptr;
    int capCenters = 0, capRm = 0;
    int inputPitch = 0, maskPitch = 0;
    int nFrames = 0;      // frames of the last detect / batch
    std::vector<hipEvent_t> events;  // [maxRecords][kTimingEvents]
    int maxRecords = 0, nRecords = 0;
    bool detected = false, described = false, gated = false;
};

// ---------------------------------------------------------------------------------------------
// a1: ORBextractor::ORBextractor tables, ORBextractor.cc:492-609
static int build_tables(amos_orb *h)
{
    const int L = h->p.n_levels;
    h->scale[0] = 1.0f;
    h->sigma2[0] = 1.0f;
    for (int i = 1; i < L; i++) {
        h->scale[i] = h->scale[i - 1] * h->p.scale_factor;
        h->sigma2[i] = h->scale[i] * h->scale[i];
    }
    for (int i = 0; i < L; i++) {
        h->invScale[i] = 1.0f / h->scale[i];
        h->invSigma2[i] = 1.0f / h->sigma2[i];
    }
    const float factor = 1.0f / h->p.scale_factor;
    float nDesired = (float)h->p.n_features * (1.0f - factor) / (1.0f - (float)std::pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; l++) {
        h->quota[l] = cv_round(nDesired);
        sum += h->quota[l];
        nDesired *= factor;
    }
    h->quota[L - 1] = std::max(h->p.n_features - sum, 0);
    // umax of the circular patch
    const int vmax = (int)std::floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
    const int vmin = (int)std::ceil(kHalfPatch * std::sqrt(2.f) / 2);
    const double hp2 = kHalfPatch * kHalfPatch;
    for (int v = 0; v <= vmax; ++v) h->umax[v] = (int)lrint(std::sqrt(hp2 - v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (h->umax[v0] == h->umax[v0 + 1]) ++v0;
        h->umax[v] = v0;
        ++v0;
    }
    return AMOS_OK;
}

// cv::resize coefficient tables for one axis: horizontal taps clamp the fraction,
// vertical taps only clip the row index.
static void build_taps(int srcN, int dstN, bool horizontal, ResizeTap *out)
{
    const double inv_scale = (double)dstN / srcN;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dstN; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)std::floor(f);
        f -= s;
        int s0 = s, s1 = s + 1;
        if (horizontal) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= srcN - 1) { f = 0; s = srcN - 1; }
            s0 = s;
            s1 = std::min(s + 1, srcN - 1);
        } else {
            s0 = std::min(std::max(s0, 0), srcN - 1);
            s1 = std::min(std::max(s1, 0), srcN - 1);
        }
        auto sat = [](float v) { int i = cv_round(v); return (short)std::min(std::max(i, -32768), 32767); };
        out[d].ofs = (short)s0;
        out[d].ofs1 = (short)s1;
        out[d].a0 = sat((1.f - f) * 2048);
        out[d].a1 = sat(f * 2048);
    }
}

// Geometry for a w x h frame.  Fails for frames the reference itself cannot process (a level
// with no FAST cell divides by zero at ORBextractor.cc:1083-1086).
static int build_geometry(const amos_orb *h, int W, int Hh, Geom &g, std::vector<Cell> *cells,
                          std::vector<ResizeTap> *taps)
{
    const int L = h->p.n_levels;
    std::memset(&g, 0, sizeof(g));
    g.nLevels = L;
    g.W = W;
    g.H = Hh;
    g.iniTh = h->p.ini_th_fast;
    g.minTh = h->p.min_th_fast;
    size_t off = 0;
    int cellIdx = 0, slotOff = 0, ptsOff = 0, kpOff = 0, tabOff = 0, blurOff = 0, maxTw = 1, maxTh = 1;
    if (cells) cells->clear();
    if (taps) taps->clear();
    for (int l = 0; l < L; l++) {
        LevelGeom &lg = g.lv[l];
        lg.w = cv_round((float)W * h->invScale[l]);
        lg.h = cv_round((float)Hh * h->invScale[l]);
        if (lg.w > 4000 || lg.h > 4000) { set_error("level %d too large (%dx%d)", l, lg.w, lg.h); return AMOS_ERR_INVALID; }
        lg.stride = align_up(kPadLeft + lg.w + kEdge, 128);
        lg.planeOff = (int)off;
        off += align_up_sz((size_t)lg.stride * (lg.h + 2 * kEdge), 256);
        if (off > 0x7fffffffu) { set_error("frame pyramid exceeds 2 GiB"); return AMOS_ERR_INVALID; }
        lg.maxBX = lg.w - kEdge + 3;
        lg.maxBY = lg.h - kEdge + 3;
        const float width = (float)(lg.maxBX - kMinBorder), height = (float)(lg.maxBY - kMinBorder);
        lg.nCols = (int)(width / 30.f);
        lg.nRows = (int)(height / 30.f);
        if (lg.nCols < 1 || lg.nRows < 1 || lg.w <= 2 * kEdge || lg.h <= 2 * kEdge) {
            set_error("level %d (%dx%d) has no FAST cell: frame too small for %d levels", l, lg.w, lg.h, L);
            return AMOS_ERR_INVALID;
        }
        lg.wCell = (int)std::ceil(width / lg.nCols);
        lg.hCell = (int)std::ceil(height / lg.nRows);
        if (lg.wCell > kFastMaxCell || lg.hCell > kFastMaxCell) {
            set_error("level %d cell %dx%d exceeds %d", l, lg.wCell, lg.hCell, kFastMaxCell);
            return AMOS_ERR_INVALID;
        }
        lg.cellStart = cellIdx;
        lg.ptsOff = ptsOff;
        int levelPts = 0;
        for (int i = 0; i < lg.nRows; i++) {  // ORBextractor.cc:1089-1120
            const float iniY = (float)(kMinBorder + i * lg.hCell);
            float maxY = iniY + lg.hCell + 6;
            if (iniY >= lg.maxBY - 3) continue;
            if (maxY > lg.maxBY) maxY = (float)lg.maxBY;
            for (int j = 0; j < lg.nCols; j++) {
                const float iniX = (float)(kMinBorder + j * lg.wCell);
                float maxX = iniX + lg.wCell + 6;
                if (iniX >= lg.maxBX - 6) continue;
                if (maxX > lg.maxBX) maxX = (float)lg.maxBX;
                const int tw = (int)maxX - (int)iniX - 6, th = (int)maxY - (int)iniY - 6;
                if (tw <= 0 || th <= 0) continue;  // FAST tests no pixel of such a sub-image
                Cell c{};
                c.level = (short)l;
                c.x0 = (short)((int)iniX + 3);
                c.y0 = (short)((int)iniY + 3);
                c.tw = (short)tw;
                c.th = (short)th;
                c.slotOff = slotOff;
                c.ndw = (short)((tw + 7 + 15) >> 4);  // bytes 0 .. tw + 6 of the staged row, in 16-byte pieces
                c.groups = (short)((tw + 7) >> 3);  // 8-pixel groups per row (phase 1 of k_fast_cells)
                c.magicDw = (1u << 20) / (unsigned)c.ndw + 1u;
                c.magicG = (1u << 20) / (unsigned)c.groups + 1u;
                maxTw = std::max(maxTw, tw);
                maxTh = std::max(maxTh, th);
                const int cap = ((tw + 1) / 2) * ((th + 1) / 2);  // 3x3 strict NMS keeps <= 1 per 2x2
                slotOff += cap;
                levelPts += cap;
                if (cells) cells->push_back(c);
                cellIdx++;
            }
        }
        lg.nCells = cellIdx - lg.cellStart;
        lg.ptsCap = levelPts;
        ptsOff += levelPts;
        if (levelPts > 0xfffff) { set_error("level %d: more than 2^20 candidate slots", l); return AMOS_ERR_INVALID; }
        lg.quota = h->quota[l];
        lg.nIni = (int)std::round((float)(lg.maxBX - kMinBorder) / (lg.maxBY - kMinBorder));  // :718
        if (lg.nIni < 1) { set_error("level %d aspect ratio %dx%d unsupported (nIni = 0)", l, lg.w, lg.h); return AMOS_ERR_INVALID; }
        lg.nodeCap = std::max(lg.quota, 4 * lg.nIni) + 4;
        lg.kpOff = kpOff;
        kpOff += lg.nodeCap;
        lg.scale = h->scale[l];
        lg.patchSize = (float)(int)(31 * h->scale[l]);
        lg.tabX = lg.tabY = 0;
        if (l > 0) {
            // tables by PADDED destination coordinate (column -32 + i, row -19 + i), reflection folded in
            const int nX = align_up(((kPadLeft + lg.w + kEdge + 3) / 4) * 4, 4), nY = lg.h + 2 * kEdge;
            lg.tabX = tabOff;  // multiple of 4 records = 32 B: the kernel loads 4 records as 2 x 16 B
            lg.tabY = tabOff + nX;
            tabOff += align_up(nX + nY, 4);
            if (taps) {
                taps->resize(tabOff);
                std::vector<ResizeTap> tx(lg.w), tyv(lg.h);
                build_taps(g.lv[l - 1].w, lg.w, true, tx.data());
                build_taps(g.lv[l - 1].h, lg.h, false, tyv.data());
                auto refl = [](int i, int n) { if (i < 0) i = -i; if (i >= n) i = 2 * n - 2 - i; return i; };
                for (int i = 0; i < nX; i++) {
                    const int xo = std::min(std::max(i - kPadLeft, -kEdge), lg.w + kEdge - 1);
                    (*taps)[lg.tabX + i] = tx[refl(xo, lg.w)];
                }
                for (int i = 0; i < nY; i++) (*taps)[lg.tabY + i] = tyv[refl(i - kEdge, lg.h)];
            }
        }
        lg.blurGroups = (lg.w + 7) / 8;
        lg.blurItemStart = blurOff;
        blurOff += lg.blurGroups * ((lg.h + kBlurStrip - 1) / kBlurStrip);
    }
    g.frameBytes = align_up_sz(off, 256);
    g.totalCells = cellIdx;
    g.slotTotal = slotOff;
    g.ptsTotal = ptsOff;
    g.kpLevelTotal = kpOff;
    g.kpCap = kpOff;
    g.blurItems = blurOff;
    // whole 16-byte pieces; + 8: the last 8-pixel group reads one dword past tw + 6.  One of the two strides k_fast_cells is compiled for.
    const int fastStrideNeeded = ((maxTw + 7 + 8 + 15) >> 4) * 4;
    if (fastStrideNeeded > 20) {  // a cell wider than the LDS tile rows k_fast_cells exists for: fail here rather than overrun the tile
        set_error("build_geometry: FAST cell of %d pixels needs %d dwords per tile row, k_fast_cells is compiled for 16 and 20", maxTw, fastStrideNeeded);
        return AMOS_ERR_INVALID;
    }
    g.fastTileStrideDw = fastStrideNeeded <= 16 ? 16 : 20;
    g.fastTileRows = maxTh + 6;
    g.fastMapRows = maxTh + 2;
    g.fastKeptCap = ((maxTw + 1) / 2) * ((maxTh + 1) / 2);
    g.fastWaveBytes = (int)align_up_sz(align_up_sz((size_t)g.fastTileRows * g.fastTileStrideDw * 4, 16) + (size_t)g.fastMapRows * kFastMapStride +
                                       fast_list_bytes(g.fastKeptCap) + AMOS_FAST_LDS_PAD, 16);
    return AMOS_OK;
}

// Capacities of a handle for frames up to maxW x maxH (host only: no device call).
static int compute_capacity(amos_orb *h)
{
    const int max_width = h->maxW, max_height = h->maxH;
    std::vector<Cell> cells;
    std::vector<ResizeTap> taps;
    int rc = build_geometry(h, max_width, max_height, h->capGeom, &cells, &taps);
    if (rc != AMOS_OK) return rc;
    // The allocation must hold EVERY frame up to max_width x max_height, and the cell layout of a smaller frame is
    // not always smaller (fewer, larger cells; another aspect ratio and so more quad-tree roots): inflate the
    // layout-dependent capacities of the largest frame to bounds that hold for all of them.
    {
        Geom &cg = h->capGeom;
        long long slots = 0, ncell = 0, kpl = 0;
        const int nIniMax = std::max(1, (int)std::lround((double)max_width / 48.0));  // width / height of any frame with a FAST cell
        for (int l = 0; l < cg.nLevels; l++) {
            const long long w = cg.lv[l].w, hh = cg.lv[l].h;
            slots += (w * hh * 27 + 99) / 100 + 64;           // sum of ceil(tw/2) * ceil(th/2) over cells of >= 30 px: <= (31/30)^2 / 4 of the area
            ncell += (w / 30 + 2) * (hh / 30 + 2);
            kpl += std::max(cg.lv[l].quota, 4 * nIniMax) + 4;
        }
        cg.slotTotal = (int)std::max<long long>(cg.slotTotal, slots);
        cg.ptsTotal = (int)std::max<long long>(cg.ptsTotal, slots);
        cg.kpLevelTotal = (int)std::max<long long>(cg.kpLevelTotal, kpl);
        cg.kpCap = cg.kpLevelTotal;
        h->capCells = (size_t)std::max<long long>((long long)cells.size(), ncell) + 64;
        cg.totalCells = (int)h->capCells;  // dSlotCount is indexed frame * totalCells + cell with the CURRENT frame's totalCells <= capCells
    }
    h->capTaps = taps.size() + 64 + 8 * (size_t)h->capGeom.nLevels;
    return AMOS_OK;
}

// does a W x H frame fit the handle's capacities?  (the ONE comparison set_geometry and the probe use)
static bool geometry_fits(const amos_orb *h, const Geom &g, size_t nTaps)
{
    const Geom &c = h->capGeom;
    return g.frameBytes <= c.frameBytes && g.totalCells <= c.totalCells && (size_t)g.totalCells <= h->capCells && g.slotTotal <= c.slotTotal &&
           g.ptsTotal <= c.ptsTotal && g.kpLevelTotal <= c.kpLevelTotal && nTaps <= h->capTaps;
}

static int set_geometry(amos_orb *h, int W, int Hh)
{
    if (W == h->curW && Hh == h->curH) return AMOS_OK;
    if (W > h->maxW || Hh > h->maxH) { set_error("frame %dx%d exceeds the handle's %dx%d", W, Hh, h->maxW, h->maxH); return AMOS_ERR_CAPACITY; }
    Geom g;
    std::vector<Cell> cells;
    std::vector<ResizeTap> taps;
    int rc = build_geometry(h, W, Hh, g, &cells, &taps);
    if (rc != AMOS_OK) return rc;
    // the side stream may still be reading dGeom / dPyr (k_blur of a detect-only call): let it finish
    // before the geometry records below are overwritten
    AMOS_HIP_CHECK(hipStreamSynchronize(h->streamB));
    const Geom &c = h->capGeom;
    if (!geometry_fits(h, g, taps.size())) {
        set_error("frame %dx%d needs more scratch than the handle's %dx%d allocation", W, Hh, h->maxW, h->maxH);
        return AMOS_ERR_CAPACITY;
    }
    int nc = 0, sc = 0;
    for (int l = 0; l < g.nLevels; l++) { nc = std::max(nc, g.lv[l].nodeCap); sc = std::max(sc, g.lv[l].nCells); }
    // keep the kpCap (row pitch of the result arrays) of the allocation so batch pointers stay valid
    g.kpCap = c.kpCap;
    h->octNC = align_up(nc, 4);
    h->octSC = align_up(std::max(sc, h->octNC), 4);
    {
        const size_t lds = oct_lds_bytes(h->octNC, h->octSC);
        if (lds > 160 * 1024) { set_error("quad-tree needs %zu B of LDS for a %dx%d frame", lds, W, Hh); return AMOS_ERR_INVALID; }
        if (lds > 48 * 1024 && lds > h->octLdsAttr) {
            AMOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            h->octLdsAttr = lds;
        }
    }
    h->geom = g;
    h->cells.swap(cells);
    h->taps.swap(taps);
    AMOS_HIP_CHECK(hipMemcpyAsync(h->dGeom, &h->geom, sizeof(Geom), hipMemcpyHostToDevice, h->stream));
    AMOS_HIP_CHECK(hipMemcpyAsync(h->dCells, h->cells.data(), sizeof(Cell) * h->cells.size(), hipMemcpyHostToDevice, h->stream));
    if (!h->taps.empty())
        AMOS_HIP_CHECK(hipMemcpyAsync(h->dTaps, h->taps.data(), sizeof(ResizeTap) * h->taps.size(), hipMemcpyHostToDevice, h->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));  // host vectors above are pageable and reused
    h->curW = W;
    h->curH = Hh;
    h->detected = h->described = h->gated = false;
    return AMOS_OK;
}

// ---------------------------------------------------------------------------------------------
static int launch_detect(amos_orb *h, const uint8_t *dSrc, size_t frameStride, size_t rowStride, int nFrames, int channels = 1, int rgbOrder = 0,
                         const MaskPreStageA *fuse = nullptr)
{
    const Geom &g = h->geom;
    hipEvent_t *ev = (h->maxRecords > 0 && h->nRecords < h->maxRecords) ? &h->events[(size_t)h->nRecords * (kTimingEvents)] : nullptr;
    // a previous detect-only call may still have its blur in flight on the side stream, reading the planes
    // that the pyramid kernels below overwrite
    if (h->blurDone) AMOS_HIP_CHECK(hipStreamWaitEvent(h->stream, h->evJoin, 0));
    if (ev) (void)hipEventRecord(ev[0], h->stream);
    for (int l = 0; l < g.nLevels; l++) {
        const LevelGeom &lg = g.lv[l];
        const int groups = (kPadLeft + lg.w + kEdge + 3) / 4;
        dim3 grid((groups + 63) / 64, (lg.h + 2 * kEdge + 4 * kPyrRows - 1) / (4 * kPyrRows), nFrames), block(64, 4);
        if (l == 0 && fuse)
            hipLaunchKernelGGL(k_import_color_mask, dim3((lg.w + kFuseTileW - 1) / kFuseTileW, (lg.h + kFuseTileH - 1) / kFuseTileH, nFrames), dim3(256), 0,
                               h->stream, dSrc, frameStride, rowStride, h->dPyr, h->dGeom, channels, rgbOrder, *fuse);
        else if (l == 0 && channels > 1)
            hipLaunchKernelGGL(k_pyramid_level0_color, grid, block, 0, h->stream, dSrc, frameStride, rowStride, h->dPyr, h->dGeom, channels, rgbOrder);
        else if (l == 0) {
            hipLaunchKernelGGL(k_pyramid_level0_wide, dim3((import_threads(lg.w, lg.h) + 255) / 256, 1, nFrames), dim3(256), 0, h->stream, dSrc,
                               frameStride, rowStride, h->dPyr, h->dGeom);
        }
        else if (h->p.scale_factor < 1.99f)  // the four taps of a thread fit one 8-byte window
            hipLaunchKernelGGL(k_pyramid_level<true>, dim3(xcd_grid((int)(grid.x * grid.y), nFrames)), block, 0, h->stream, h->dPyr, h->dGeom, h->dTaps, l, nFrames);
        else
            hipLaunchKernelGGL(k_pyramid_level<false>, dim3(xcd_grid((int)(grid.x * grid.y), nFrames)), block, 0, h->stream, h->dPyr, h->dGeom, h->dTaps, l, nFrames);
        if (ev && l == 0) (void)hipEventRecord(ev[1], h->stream);
    }
    if (ev) (void)hipEventRecord(ev[2], h->stream);
    // fork: the blur only needs the pyramid; it streams memory on the side stream while FAST (VALU-bound)
    // and the quad-tree (latency-bound) run on the main one
    AMOS_HIP_CHECK(hipEventRecord(h->evFork, h->stream));
    AMOS_HIP_CHECK(hipStreamWaitEvent(h->streamB, h->evFork, 0));
    if (ev) (void)hipEventRecord(h->evBlur0, h->streamB);
    hipLaunchKernelGGL(k_blur, dim3(xcd_grid((g.blurItems + 255) / 256, nFrames)), dim3(256), 0, h->streamB, h->dPyr, h->dBlur, h->dGeom, nFrames);
    if (ev) (void)hipEventRecord(h->evBlur1, h->streamB);
    AMOS_HIP_CHECK(hipEventRecord(h->evJoin, h->streamB));
    h->blurDone = true;
    if (ev) (void)hipEventRecord(ev[3], h->stream);
    {
        const dim3 fgrid(xcd_grid((g.totalCells + kFastCellsPerGroup - 1) / kFastCellsPerGroup, nFrames)), fblock(64 * kFastCellsPerGroup);
        const size_t flds = kFastCellsPerGroup * (size_t)g.fastWaveBytes;
        if (g.fastTileStrideDw == 16)
            hipLaunchKernelGGL(k_fast_cells<16>, fgrid, fblock, flds, h->stream, h->dPyr, h->dGeom, h->dCells, h->dSlotCount, h->dSlots, nFrames);
        else
            hipLaunchKernelGGL(k_fast_cells<20>, fgrid, fblock, flds, h->stream, h->dPyr, h->dGeom, h->dCells, h->dSlotCount, h->dSlots, nFrames);
    }
    if (ev) (void)hipEventRecord(ev[4], h->stream);
    const size_t lds = oct_lds_bytes(h->octNC, h->octSC);
    hipLaunchKernelGGL(k_octree, dim3(nFrames * g.nLevels), dim3(256), lds, h->stream, h->dGeom, h->dCells, h->dSlotCount,
                       h->dSlots, h->dPts, h->dNodeOf, h->dQuadOf, h->dCandCount, h->dLvKps, h->dLvCount, h->octNC, h->octSC);
    if (ev) (void)hipEventRecord(ev[5], h->stream);
    hipLaunchKernelGGL(k_orient, dim3(xcd_grid((g.kpLevelTotal + 15) / 16, nFrames)), dim3(256), 0, h->stream, h->dPyr, h->dGeom,
                       h->dLvKps, h->dLvCount, nFrames);
    if (ev) (void)hipEventRecord(ev[6], h->stream);
    AMOS_HIP_CHECK(hipGetLastError());
    h->nFrames = nFrames;
    h->detected = true;
    h->described = h->gated = false;
    return AMOS_OK;
}

static int launch_describe(amos_orb *h, int nFrames)
{
    const Geom &g = h->geom;
    hipEvent_t *ev = (h->maxRecords > 0 && h->nRecords < h->maxRecords) ? &h->events[(size_t)h->nRecords * (kTimingEvents)] : nullptr;
    AMOS_HIP_CHECK(hipStreamWaitEvent(h->stream, h->evJoin, 0));  // join: blurred planes ready
    if (ev) (void)hipEventRecord(ev[7], h->stream);
    hipLaunchKernelGGL(k_describe, dim3(xcd_grid((g.kpLevelTotal + kDescKps - 1) / kDescKps, nFrames)), dim3(kDescKps * 16), 0, h->stream, h->dBlur, h->dGeom,
                       h->dLvKps, h->dLvCount, h->dOutKps, h->dOutDesc, h->dOutCount, nFrames);
    if (ev) { (void)hipEventRecord(ev[8], h->stream); h->nRecords++; }
    AMOS_HIP_CHECK(hipGetLastError());
    h->described = true;
    return AMOS_OK;
}

// One frame's result to host buffers: count, keypoints and descriptors travel as three asynchronous copies into a pinned
// staging buffer behind ONE synchronisation (the count is not needed to size the copies: whole capacity rows are 78 KB at most
// for 1000 features); three dependent round trips were ~40 us of the single-frame API's 0.22 ms.
static int fetch_frame(amos_orb *h, int frame, amos_keypoint *kps, uint8_t *desc, int cap, int *n)
{
    const size_t kc = (size_t)h->geom.kpCap, base = (size_t)frame * kc;
    uint8_t *st = h->hStage;
    int *stCount = reinterpret_cast<int *>(st);
    amos_keypoint *stKps = reinterpret_cast<amos_keypoint *>(st + 16);
    uint8_t *stDesc = st + 16 + kc * sizeof(amos_keypoint);
    AMOS_HIP_CHECK(hipMemcpyAsync(stCount, h->dOutCount + frame, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (kps) AMOS_HIP_CHECK(hipMemcpyAsync(stKps, h->dOutKps + base, sizeof(amos_keypoint) * kc, hipMemcpyDeviceToHost, h->stream));
    if (desc) AMOS_HIP_CHECK(hipMemcpyAsync(stDesc, h->dOutDesc + base * 32, (size_t)32 * kc, hipMemcpyDeviceToHost, h->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    const int count = *stCount;
    if (n) *n = count;
    if (count > cap) { set_error("result holds %d keypoints, caller capacity %d", count, cap); return AMOS_ERR_CAPACITY; }
    if (count > 0) {
        if (kps) std::memcpy(kps, stKps, sizeof(amos_keypoint) * count);
        if (desc) std::memcpy(desc, stDesc, (size_t)32 * count);
    }
    return AMOS_OK;
}

template <typename T>
static int dev_alloc(T **p, size_t count)
{
    AMOS_HIP_CHECK(hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)));
    return AMOS_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" {

const char *amos_last_error(void) { return g_error.c_str(); }

int amos_device_count(void)
{
    int n = 0;
    AMOS_HIP_CHECK(hipGetDeviceCount(&n));
    return n;
}

int amos_current_device(void)
{
    int d = 0;
    AMOS_HIP_CHECK(hipGetDevice(&d));
    return d;
}

const char *amos_build_variant(void)
{
    // every switch that changes results is named here (and in amos_winograd24.hip through amos_w24_variant_tag)
    static std::string tag;
    if (tag.empty()) {
        std::string t;
        if (AMOS_FAST_EXP != 0) t += " AMOS_FAST_EXP=" + std::to_string(AMOS_FAST_EXP);
        if (AMOS_FAST_LDS_PAD != 0) t += " AMOS_FAST_LDS_PAD=" + std::to_string(AMOS_FAST_LDS_PAD);
        t += amos::w24_variant_tag();
        tag = t.empty() ? "default" : t.substr(1);
    }
    return tag.c_str();
}

int amos_orb_tables_host(const amos_orb_params *params, float *scale_factor, float *inv_scale_factor, float *level_sigma2,
                         float *inv_level_sigma2, int32_t *features_per_level, int32_t *umax)
{
    if (!params || params->n_levels < 1 || params->n_levels > AMOS_MAX_LEVELS || params->n_features < 1 || !(params->scale_factor > 1.0f)) {
        set_error("amos_orb_tables_host: invalid argument");
        return AMOS_ERR_INVALID;
    }
    amos_orb h;  // host fields only
    h.p = *params;
    build_tables(&h);
    return amos_orb_tables(&h, scale_factor, inv_scale_factor, level_sigma2, inv_level_sigma2, features_per_level, umax);
}

int amos_orb_geometry_probe(const amos_orb_params *params, int max_width, int max_height, int width, int height, int32_t need[6], int32_t cap[6])
{
    if (!params || max_width < 1 || max_height < 1 || width < 1 || height < 1 || params->n_levels < 1 || params->n_levels > AMOS_MAX_LEVELS ||
        params->n_features < 1 || !(params->scale_factor > 1.0f) || !(params->scale_factor <= 2.0f)) {
        set_error("amos_orb_geometry_probe: invalid argument");
        return AMOS_ERR_INVALID;
    }
    amos_orb h;  // host fields only: nothing is allocated on a device
    h.p = *params;
    h.maxW = max_width;
    h.maxH = max_height;
    build_tables(&h);
    int rc = compute_capacity(&h);
    if (rc != AMOS_OK) return rc;
    if (width > max_width || height > max_height) { set_error("frame %dx%d exceeds %dx%d", width, height, max_width, max_height); return AMOS_ERR_CAPACITY; }
    Geom g;
    std::vector<Cell> cells;
    std::vector<ResizeTap> taps;
    rc = build_geometry(&h, width, height, g, &cells, &taps);
    if (rc != AMOS_OK) return rc;
    const Geom &c = h.capGeom;
    const long long n[6] = {(long long)g.totalCells, g.slotTotal, g.ptsTotal, g.kpLevelTotal, (long long)taps.size(), (long long)(g.frameBytes >> 8)};
    const long long k[6] = {(long long)c.totalCells, c.slotTotal, c.ptsTotal, c.kpLevelTotal, (long long)h.capTaps, (long long)(c.frameBytes >> 8)};
    for (int i = 0; i < 6; i++) {
        if (need) need[i] = (int32_t)n[i];
        if (cap) cap[i] = (int32_t)k[i];
    }
    return geometry_fits(&h, g, taps.size()) ? AMOS_OK : AMOS_ERR_CAPACITY;
}

int amos_orb_create(const amos_orb_params *params, int max_width, int max_height, int max_batch, int device,
                    void *stream, amos_orb **out)
{
    if (!params || !out || max_width < 1 || max_height < 1 || max_batch < 1 || params->n_levels < 1 ||
        params->n_levels > AMOS_MAX_LEVELS || params->n_features < 1 || !(params->scale_factor > 1.0f) ||
        !(params->scale_factor <= 2.0f)) {  // k_pyramid_level: scaleFactor < 2 uses the 8-byte tap window, 2.0 the four-load variant
        set_error("amos_orb_create: invalid argument");
        return AMOS_ERR_INVALID;
    }
    AMOS_HIP_CHECK(hipSetDevice(device));
    amos_orb *h = new amos_orb();
    h->p = *params;
    h->maxW = max_width;
    h->maxH = max_height;
    h->maxB = max_batch;
    h->device = device;
    build_tables(h);
    static const int kUmax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    if (std::memcmp(kUmax, h->umax, sizeof(kUmax)) != 0) { set_error("umax table mismatch"); delete h; return AMOS_ERR_INVALID; }
    int rc = compute_capacity(h);
    if (rc != AMOS_OK) { delete h; return rc; }
    const Geom &c = h->capGeom;
    if (stream) h->stream = (hipStream_t)stream;
    else {
        hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { set_error("hipStreamCreate: %s", hipGetErrorString(e)); delete h; return AMOS_ERR_DEVICE; }
        h->ownStream = true;
    }
    if (hipStreamCreateWithFlags(&h->streamB, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->evFork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->evJoin, hipEventDisableTiming) != hipSuccess || hipEventCreate(&h->evBlur0) != hipSuccess ||
        hipEventCreate(&h->evBlur1) != hipSuccess) {
        set_error("side stream / event creation failed");
        amos_orb_destroy(h);
        return AMOS_ERR_DEVICE;
    }
    const size_t B = (size_t)max_batch;
    const size_t slack = 4096;  // tile loads may run a few bytes past the last row of the last plane
    h->inputPitch = align_up(max_width, 128);
    h->maskPitch = align_up(max_width, 128);
#define ALLOC(ptr, count) do { rc = dev_alloc(&(ptr), (count)); if (rc != AMOS_OK) { amos_orb_destroy(h); return rc; } } while (0)
    ALLOC(h->dGeom, 1);
    ALLOC(h->dCells, h->capCells);
    ALLOC(h->dTaps, h->capTaps);
    ALLOC(h->dPyr, B * c.frameBytes + slack);
    ALLOC(h->dBlur, B * c.frameBytes + slack);
    ALLOC(h->dInput, (size_t)h->inputPitch * max_height);
    ALLOC(h->dSlotCount, B * c.totalCells);
    ALLOC(h->dSlots, B * c.slotTotal);
    ALLOC(h->dPts, B * c.ptsTotal);
    ALLOC(h->dNodeOf, B * c.ptsTotal);
    ALLOC(h->dQuadOf, B * c.ptsTotal);
    ALLOC(h->dCandCount, B * c.nLevels);
    ALLOC(h->dLvCount, B * c.nLevels);
    ALLOC(h->dOutCount, B);
    ALLOC(h->dLvKps, B * c.kpLevelTotal);
    ALLOC(h->dOutKps, B * c.kpCap);
    ALLOC(h->dOutDesc, B * c.kpCap * 32);
    ALLOC(h->dRemoved, B * c.kpLevelTotal);
    ALLOC(h->dScratchKps, (size_t)c.ptsTotal);
    ALLOC(h->dMask, (size_t)h->maskPitch * max_height);
    ALLOC(h->dMaskTmp, B * (size_t)h->maskPitch * max_height);
    ALLOC(h->dMaskClosed, B * (size_t)h->maskPitch * max_height);
    ALLOC(h->dLabels, (size_t)max_width * max_height);
    ALLOC(h->dNRemoved, B);
    ALLOC(h->dErr, 1);
#undef ALLOC
    if (hipHostMalloc((void **)&h->hStage, 16 + (size_t)c.kpCap * (sizeof(amos_keypoint) + 32), hipHostMallocDefault) != hipSuccess) {
        set_error("hipHostMalloc (result staging) failed");
        amos_orb_destroy(h);
        return AMOS_ERR_DEVICE;
    }
    (void)hipMemsetAsync(h->dPyr, 0, B * c.frameBytes + slack, h->stream);
    (void)hipMemsetAsync(h->dBlur, 0, B * c.frameBytes + slack, h->stream);
    (void)hipMemsetAsync(h->dLvCount, 0, sizeof(int) * B * c.nLevels, h->stream);
    (void)hipMemsetAsync(h->dOutCount, 0, sizeof(int) * B, h->stream);
    // constants
    signed char pat[1024];
    std::memcpy(pat, amos_orb_pattern, 1024);
    int dx[31];
    {  // getStructuringElement(MORPH_ELLIPSE, 31x31), SURVEY A.6
        const int r = 15, cc = 15;
        const double inv_r2 = 1. / ((double)r * r);
        for (int i = 0; i < 31; i++) {
            const int dy = i - r;
            dx[i] = (int)lrint(cc * std::sqrt((r * r - dy * dy) * inv_r2));
        }
    }
    for (int i = 0; i < 31; i++)  // k_morph31 is written for these row widths (getStructuringElement(MORPH_ELLIPSE, 31 x 31))
        if (dx[i] != kMorphDx[std::abs(i - 15)]) { set_error("ellipse row %d: half-width %d, kernel expects %d", i, dx[i], kMorphDx[std::abs(i - 15)]); amos_orb_destroy(h); return AMOS_ERR_INVALID; }
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), pat, sizeof(pat)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(c_umax), h->umax, sizeof(int) * 16) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(c_ellipse_dx), dx, sizeof(dx)) != hipSuccess) {
        set_error("hipMemcpyToSymbol failed");
        amos_orb_destroy(h);
        return AMOS_ERR_DEVICE;
    }
    int maxNC = 0, maxSC = 0;
    for (int l = 0; l < c.nLevels; l++) { maxNC = std::max(maxNC, c.lv[l].nodeCap); maxSC = std::max(maxSC, c.lv[l].nCells); }
    const size_t maxLds = oct_lds_bytes(align_up(maxNC, 4), align_up(std::max(maxSC, maxNC), 4));
    if (maxLds > 160 * 1024) { set_error("quad-tree needs %zu B of LDS (n_features too large)", maxLds); amos_orb_destroy(h); return AMOS_ERR_INVALID; }
    if (maxLds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree), hipFuncAttributeMaxDynamicSharedMemorySize, (int)maxLds);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); amos_orb_destroy(h); return AMOS_ERR_DEVICE; }
        h->octLdsAttr = maxLds;
    }
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { set_error("create sync: %s", hipGetErrorString(e)); amos_orb_destroy(h); return AMOS_ERR_DEVICE; }
    *out = h;
    return AMOS_OK;
}

void amos_orb_destroy(amos_orb *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *ptrs[] = {h->dGeom, h->dCells, h->dTaps, h->dPyr, h->dBlur, h->dInput, h->dSlotCount, h->dSlots, h->dPts,
                    h->dNodeOf, h->dQuadOf, h->dCandCount, h->dLvCount, h->dOutCount, h->dLvKps, h->dOutKps, h->dOutDesc,
                    h->dRemoved, h->dScratchKps, h->dMask, h->dMaskTmp, h->dMaskClosed, h->dLabels, h->dCenterIds, h->dRm,
                    h->dNRemoved, h->dErr};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (h->hStage) (void)hipHostFree(h->hStage);
    if (h->hPyrStage) (void)hipHostFree(h->hPyrStage);
    for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    if (h->streamB) { (void)hipStreamSynchronize(h->streamB); (void)hipStreamDestroy(h->streamB); }
    for (hipEvent_t e : {h->evFork, h->evJoin, h->evBlur0, h->evBlur1}) if (e) (void)hipEventDestroy(e);
    if (h->ownStream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int amos_orb_tables(const amos_orb *h, float *scale_factor, float *inv_scale_factor, float *level_sigma2,
                    float *inv_level_sigma2, int32_t *features_per_level, int32_t *umax)
{
    if (!h) return AMOS_ERR_INVALID;
    for (int i = 0; i < h->p.n_levels; i++) {
        if (scale_factor) scale_factor[i] = h->scale[i];
        if (inv_scale_factor) inv_scale_factor[i] = h->invScale[i];
        if (level_sigma2) level_sigma2[i] = h->sigma2[i];
        if (inv_level_sigma2) inv_level_sigma2[i] = h->invSigma2[i];
        if (features_per_level) features_per_level[i] = h->quota[i];
    }
    if (umax) for (int i = 0; i < 16; i++) umax[i] = h->umax[i];
    return AMOS_OK;
}

int amos_orb_level_sizes(const amos_orb *h, int width, int height, int32_t *level_w, int32_t *level_h)
{
    if (!h) return AMOS_ERR_INVALID;
    for (int l = 0; l < h->p.n_levels; l++) {
        if (level_w) level_w[l] = cv_round((float)width * h->invScale[l]);
        if (level_h) level_h[l] = cv_round((float)height * h->invScale[l]);
    }
    return h->p.n_levels;
}

int amos_orb_detect(amos_orb *h, const uint8_t *gray, size_t stride, int width, int height)
{
    if (!h || !gray || width < 1 || height < 1 || stride < (size_t)width) { set_error("amos_orb_detect: invalid argument"); return AMOS_ERR_INVALID; }
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    int rc = set_geometry(h, width, height);
    if (rc != AMOS_OK) return rc;
    AMOS_HIP_CHECK(hipMemcpy2DAsync(h->dInput, h->inputPitch, gray, stride, width, height, hipMemcpyHostToDevice, h->stream));
    rc = launch_detect(h, h->dInput, 0, h->inputPitch, 1);
    if (rc != AMOS_OK) return rc;
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(h->streamB));  // the blur on the side stream reads dInput's pyramid: nothing of this call is in flight on return
    return AMOS_OK;
}

int amos_orb_level_count(amos_orb *h, int frame, int level)
{
    if (!h || !h->detected || frame < 0 || frame >= h->nFrames || level < 0 || level >= h->geom.nLevels) { set_error("amos_orb_level_count: invalid argument or state"); return AMOS_ERR_INVALID; }
    int n = 0;
    AMOS_HIP_CHECK(hipMemcpyAsync(&n, h->dLvCount + frame * h->geom.nLevels + level, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    return n;
}

int amos_orb_level_keypoints(amos_orb *h, int frame, int level, amos_keypoint *out, int cap)
{
    const int n = amos_orb_level_count(h, frame, level);
    if (n < 0) return n;
    if (n > cap || (!out && n > 0)) { set_error("level holds %d keypoints, caller capacity %d", n, cap); return AMOS_ERR_CAPACITY; }
    if (n > 0) {
        AMOS_HIP_CHECK(hipMemcpyAsync(out, h->dLvKps + (size_t)frame * h->geom.kpLevelTotal + h->geom.lv[level].kpOff,
                                      sizeof(amos_keypoint) * n, hipMemcpyDeviceToHost, h->stream));
        AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    }
    return n;
}

int amos_orb_set_level_keypoints(amos_orb *h, int frame, int level, const amos_keypoint *kps, int n)
{
    if (!h || !h->detected || frame < 0 || frame >= h->nFrames || level < 0 || level >= h->geom.nLevels || n < 0 || (n > 0 && !kps)) {
        set_error("amos_orb_set_level_keypoints: invalid argument or state");
        return AMOS_ERR_INVALID;
    }
    if (n > h->geom.lv[level].nodeCap) { set_error("level %d list capacity %d < %d", level, h->geom.lv[level].nodeCap, n); return AMOS_ERR_CAPACITY; }
    if (n > 0)
        AMOS_HIP_CHECK(hipMemcpyAsync(h->dLvKps + (size_t)frame * h->geom.kpLevelTotal + h->geom.lv[level].kpOff, kps,
                                      sizeof(amos_keypoint) * n, hipMemcpyHostToDevice, h->stream));
    AMOS_HIP_CHECK(hipMemcpyAsync(h->dLvCount + frame * h->geom.nLevels + level, &n, sizeof(int), hipMemcpyHostToDevice, h->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    return AMOS_OK;
}

int amos_orb_level_layout(amos_orb *h, int32_t *offsets, int32_t *caps, int32_t *total)
{
    if (!h || h->curW == 0) { set_error("amos_orb_level_layout: no frame geometry yet"); return AMOS_ERR_STATE; }
    for (int l = 0; l < h->geom.nLevels; l++) {
        if (offsets) offsets[l] = h->geom.lv[l].kpOff;
        if (caps) caps[l] = h->geom.lv[l].nodeCap;
    }
    if (total) *total = h->geom.kpLevelTotal;
    return AMOS_OK;
}

int amos_orb_fetch_levels(amos_orb *h, int frame, int32_t *counts, amos_keypoint *buf, int buf_len)
{
    if (!h || !h->detected || !counts || !buf || frame < 0 || frame >= h->nFrames) { set_error("amos_orb_fetch_levels: invalid argument or state"); return AMOS_ERR_INVALID; }
    if (buf_len < h->geom.kpLevelTotal) { set_error("amos_orb_fetch_levels: buffer holds %d keypoints, need %d", buf_len, h->geom.kpLevelTotal); return AMOS_ERR_CAPACITY; }
    AMOS_HIP_CHECK(hipMemcpyAsync(counts, h->dLvCount + frame * h->geom.nLevels, sizeof(int) * h->geom.nLevels, hipMemcpyDeviceToHost, h->stream));
    AMOS_HIP_CHECK(hipMemcpyAsync(buf, h->dLvKps + (size_t)frame * h->geom.kpLevelTotal, sizeof(amos_keypoint) * h->geom.kpLevelTotal,
                                  hipMemcpyDeviceToHost, h->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    return AMOS_OK;
}

int amos_orb_store_levels(amos_orb *h, int frame, const int32_t *counts, const amos_keypoint *buf, int buf_len)
{
    if (!h || !h->detected || !counts || !buf || frame < 0 || frame >= h->nFrames) { set_error("amos_orb_store_levels: invalid argument or state"); return AMOS_ERR_INVALID; }
    if (buf_len < h->geom.kpLevelTotal) { set_error("amos_orb_store_levels: buffer holds %d keypoints, need %d", buf_len, h->geom.kpLevelTotal); return AMOS_ERR_CAPACITY; }
    for (int l = 0; l < h->geom.nLevels; l++)
        if (counts[l] < 0 || counts[l] > h->geom.lv[l].nodeCap) { set_error("level %d: %d keypoints exceed capacity %d", l, counts[l], h->geom.lv[l].nodeCap); return AMOS_ERR_CAPACITY; }
    AMOS_HIP_CHECK(hipMemcpyAsync(h->dLvCount + frame * h->geom.nLevels, counts, sizeof(int) * h->geom.nLevels, hipMemcpyHostToDevice, h->stream));
    AMOS_HIP_CHECK(hipMemcpyAsync(h->dLvKps + (size_t)frame * h->geom.kpLevelTotal, buf, sizeof(amos_keypoint) * h->geom.kpLevelTotal,
                                  hipMemcpyHostToDevice, h->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    return AMOS_OK;
}

int amos_orb_level_candidates(amos_orb *h, int frame, int level, amos_keypoint *out, int cap)
{
    if (!h || !h->detected || frame < 0 || frame >= h->nFrames || level < 0 || level >= h->geom.nLevels) { set_error("amos_orb_level_candidates: invalid argument or state"); return AMOS_ERR_INVALID; }
    int n = 0;
    AMOS_HIP_CHECK(hipMemcpyAsync(&n, h->dCandCount + frame * h->geom.nLevels + level, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    if (n > cap) { set_error("level holds %d candidates, caller capacity %d", n, cap); return AMOS_ERR_CAPACITY; }
    if (n > 0) {
        hipLaunchKernelGGL(k_unpack_candidates, dim3((n + 255) / 256), dim3(256), 0, h->stream,
                           h->dPts + (size_t)frame * h->geom.ptsTotal + h->geom.lv[level].ptsOff, n, h->dScratchKps);
        AMOS_HIP_CHECK(hipMemcpyAsync(out, h->dScratchKps, sizeof(amos_keypoint) * n, hipMemcpyDeviceToHost, h->stream));
        AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    }
    return n;
}

int amos_orb_gate(amos_orb *h, const uint8_t *mask, size_t mask_stride, const double *labels, size_t lstride,
                  const int32_t *center_ids, int n_centers, const int32_t *rm_vector, int n_rm, amos_keypoint *removed,
                  int cap, int *n_removed)
{
    if (!h || !mask || !n_removed) { set_error("amos_orb_gate: invalid argument"); return AMOS_ERR_INVALID; }
    if (!h->detected) { set_error("amos_orb_gate before amos_orb_detect"); return AMOS_ERR_STATE; }
    if (labels && (!center_ids || !rm_vector || n_centers < 1 || n_rm < 1)) { set_error("amos_orb_gate: label gate needs center ids and rm_vector"); return AMOS_ERR_INVALID; }
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    const Geom &g = h->geom;
    AMOS_HIP_CHECK(hipMemcpy2DAsync(h->dMask, h->maskPitch, mask, mask_stride, g.W, g.H, hipMemcpyHostToDevice, h->stream));
    if (labels) {
        AMOS_HIP_CHECK(hipMemcpy2DAsync(h->dLabels, sizeof(double) * g.W, labels, sizeof(double) * lstride, sizeof(double) * g.W, g.H,
                                        hipMemcpyHostToDevice, h->stream));
        if (n_centers > h->capCenters) {
            if (h->dCenterIds) (void)hipFree(h->dCenterIds);
            h->dCenterIds = nullptr;
            int rc = dev_alloc(&h->dCenterIds, n_centers);
            if (rc != AMOS_OK) return rc;
            h->capCenters = n_centers;
        }
        if (n_rm > h->capRm) {
            if (h->dRm) (void)hipFree(h->dRm);
            h->dRm = nullptr;
            int rc = dev_alloc(&h->dRm, n_rm);
            if (rc != AMOS_OK) return rc;
            h->capRm = n_rm;
        }
        AMOS_HIP_CHECK(hipMemcpyAsync(h->dCenterIds, center_ids, sizeof(int) * n_centers, hipMemcpyHostToDevice, h->stream));
        AMOS_HIP_CHECK(hipMemcpyAsync(h->dRm, rm_vector, sizeof(int) * n_rm, hipMemcpyHostToDevice, h->stream));
    }
    AMOS_HIP_CHECK(hipMemsetAsync(h->dErr, 0, sizeof(int), h->stream));
    dim3 grid((g.W + kMorphTileW - 1) / kMorphTileW, (g.H + kMorphTileH - 1) / kMorphTileH);
    hipLaunchKernelGGL(k_morph31<true>, grid, dim3(256), 0, h->stream, h->dMask, (size_t)0, h->maskPitch, h->dMaskTmp, (size_t)0, h->maskPitch, g.W, g.H);
    hipLaunchKernelGGL(k_morph31<false>, grid, dim3(256), 0, h->stream, h->dMaskTmp, (size_t)0, h->maskPitch, h->dMaskClosed, (size_t)0, h->maskPitch, g.W, g.H);
    hipLaunchKernelGGL(k_gate, dim3(1), dim3(256), 0, h->stream, h->dGeom, h->dLvKps, h->dLvCount, h->dMaskClosed, (size_t)0, h->maskPitch,
                       labels ? h->dLabels : nullptr, g.W, h->dCenterIds, n_centers, h->dRm, n_rm, h->dRemoved, h->dNRemoved, h->dErr);
    AMOS_HIP_CHECK(hipGetLastError());
    int nrem = 0, err = 0;
    AMOS_HIP_CHECK(hipMemcpyAsync(&nrem, h->dNRemoved, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    AMOS_HIP_CHECK(hipMemcpyAsync(&err, h->dErr, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    h->gated = true;
    if (err) { set_error("amos_orb_gate: %s", (err & 1) ? "keypoint outside the mask" : "label / cluster id out of range"); return AMOS_ERR_INVALID; }
    *n_removed = nrem;
    if (removed) {
        if (nrem > cap) { set_error("%d keypoints removed, caller capacity %d", nrem, cap); return AMOS_ERR_CAPACITY; }
        if (nrem > 0) {
            AMOS_HIP_CHECK(hipMemcpyAsync(removed, h->dRemoved, sizeof(amos_keypoint) * nrem, hipMemcpyDeviceToHost, h->stream));
            AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
        }
    }
    return AMOS_OK;
}

int amos_orb_closed_mask(amos_orb *h, uint8_t *dst, size_t dst_stride)
{
    if (!h || !dst || !h->gated) { set_error("amos_orb_closed_mask: no gate has run"); return AMOS_ERR_STATE; }
    AMOS_HIP_CHECK(hipMemcpy2DAsync(dst, dst_stride, h->dMaskClosed, h->maskPitch, h->geom.W, h->geom.H, hipMemcpyDeviceToHost, h->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    return AMOS_OK;
}

int amos_orb_describe(amos_orb *h, amos_keypoint *kps, uint8_t *desc, int cap, int *n)
{
    if (!h || !n) { set_error("amos_orb_describe: invalid argument"); return AMOS_ERR_INVALID; }
    if (!h->detected) { set_error("amos_orb_describe before amos_orb_detect"); return AMOS_ERR_STATE; }
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    int rc = launch_describe(h, 1);
    if (rc != AMOS_OK) return rc;
    return fetch_frame(h, 0, kps, desc, cap, n);
}

int amos_orb_extract(amos_orb *h, const uint8_t *gray, size_t stride, int width, int height, amos_keypoint *kps,
                     uint8_t *desc, int cap, int *n)
{
    if (!h || !gray || !n || width < 1 || height < 1 || stride < (size_t)width) { set_error("amos_orb_extract: invalid argument"); return AMOS_ERR_INVALID; }
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    int rc = set_geometry(h, width, height);
    if (rc != AMOS_OK) return rc;
    AMOS_HIP_CHECK(hipMemcpy2DAsync(h->dInput, h->inputPitch, gray, stride, width, height, hipMemcpyHostToDevice, h->stream));
    rc = launch_detect(h, h->dInput, 0, h->inputPitch, 1);
    if (rc != AMOS_OK) return rc;
    rc = launch_describe(h, 1);
    if (rc != AMOS_OK) return rc;
    return fetch_frame(h, 0, kps, desc, cap, n);
}

// Level planes to host buffers.  The device plane travels as ONE contiguous copy into a pinned staging buffer (DMA at the link rate) and is
// cut into the caller's rows on the host: a strided hipMemcpy2D into pageable memory moves row by row -- 13.6 ms for the eight planes of a
// 640 x 480 frame (measured, tools/r5_dropin.py), where this takes ~0.15 ms.
static int pyr_stage(amos_orb *h)
{
    if (h->hPyrStage) return AMOS_OK;
    if (hipHostMalloc((void **)&h->hPyrStage, (size_t)h->capGeom.frameBytes + 4096, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipHostMalloc (pyramid staging) failed");
        return AMOS_ERR_DEVICE;
    }
    return AMOS_OK;
}

static void cut_plane(const amos_orb *h, int level, const uint8_t *stagedPlane, uint8_t *dst, size_t dst_stride, int padded)
{
    const LevelGeom &lg = h->geom.lv[level];
    const uint8_t *origin = stagedPlane + (size_t)kEdge * lg.stride + kPadLeft;
    const int rows = padded ? lg.h + 2 * kEdge : lg.h, cols = padded ? lg.w + 2 * kEdge : lg.w;
    const uint8_t *src = padded ? origin - (size_t)kEdge * lg.stride - kEdge : origin;
    for (int y = 0; y < rows; y++) std::memcpy(dst + (size_t)y * dst_stride, src + (size_t)y * lg.stride, (size_t)cols);
}

static int copy_plane(amos_orb *h, const uint8_t *dBase, int frame, int level, uint8_t *dst, size_t dst_stride, int padded)
{
    const Geom &g = h->geom;
    const LevelGeom &lg = g.lv[level];
    int rc = pyr_stage(h);
    if (rc != AMOS_OK) return rc;
    const size_t bytes = (size_t)lg.stride * (lg.h + 2 * kEdge);
    AMOS_HIP_CHECK(hipMemcpyAsync(h->hPyrStage, dBase + (size_t)frame * g.frameBytes + lg.planeOff, bytes, hipMemcpyDeviceToHost, h->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    cut_plane(h, level, h->hPyrStage, dst, dst_stride, padded);
    return AMOS_OK;
}

int amos_orb_pyramid_images(amos_orb *h, int frame, uint8_t *const *dst, const size_t *dst_strides, int padded)
{
    if (!h || !dst || !dst_strides || !h->detected || frame < 0 || frame >= h->nFrames) { set_error("amos_orb_pyramid_images: invalid argument or state"); return AMOS_ERR_INVALID; }
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    const Geom &g = h->geom;
    int rc = pyr_stage(h);
    if (rc != AMOS_OK) return rc;
    AMOS_HIP_CHECK(hipMemcpyAsync(h->hPyrStage, h->dPyr + (size_t)frame * g.frameBytes, (size_t)g.frameBytes, hipMemcpyDeviceToHost, h->stream));
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    for (int l = 0; l < g.nLevels; l++)
        if (dst[l]) cut_plane(h, l, h->hPyrStage + g.lv[l].planeOff, dst[l], dst_strides[l], padded);
    return AMOS_OK;
}

int amos_orb_level_image(amos_orb *h, int frame, int level, uint8_t *dst, size_t dst_stride, int padded)
{
    if (!h || !dst || !h->detected || frame < 0 || frame >= h->nFrames || level < 0 || level >= h->geom.nLevels) { set_error("amos_orb_level_image: invalid argument or state"); return AMOS_ERR_INVALID; }
    return copy_plane(h, h->dPyr, frame, level, dst, dst_stride, padded);
}

int amos_orb_blurred_image(amos_orb *h, int frame, int level, uint8_t *dst, size_t dst_stride)
{
    if (!h || !dst || !h->described || frame < 0 || frame >= h->nFrames || level < 0 || level >= h->geom.nLevels) { set_error("amos_orb_blurred_image: invalid argument or state"); return AMOS_ERR_INVALID; }
    return copy_plane(h, h->dBlur, frame, level, dst, dst_stride, 0);
}

int amos_orb_extract_batch_device(amos_orb *h, const uint8_t *d_gray, size_t frame_stride, size_t row_stride, int width,
                                  int height, int n_frames)
{
    if (!h || !d_gray || n_frames < 1 || width < 1 || height < 1 || row_stride < (size_t)width) { set_error("amos_orb_extract_batch_device: invalid argument"); return AMOS_ERR_INVALID; }
    if (n_frames > h->maxB) { set_error("batch of %d frames exceeds the handle's max_batch %d", n_frames, h->maxB); return AMOS_ERR_CAPACITY; }
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    int rc = set_geometry(h, width, height);
    if (rc != AMOS_OK) return rc;
    rc = launch_detect(h, d_gray, frame_stride, row_stride, n_frames);
    if (rc != AMOS_OK) return rc;
    return launch_describe(h, n_frames);
}

int amos_orb_detect_batch_device(amos_orb *h, const uint8_t *d_gray, size_t frame_stride, size_t row_stride, int width,
                                 int height, int n_frames)
{
    if (!h || !d_gray || n_frames < 1 || width < 1 || height < 1 || row_stride < (size_t)width) { set_error("amos_orb_detect_batch_device: invalid argument"); return AMOS_ERR_INVALID; }
    if (n_frames > h->maxB) { set_error("batch of %d frames exceeds the handle's max_batch %d", n_frames, h->maxB); return AMOS_ERR_CAPACITY; }
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    int rc = set_geometry(h, width, height);
    if (rc != AMOS_OK) return rc;
    return launch_detect(h, d_gray, frame_stride, row_stride, n_frames);
}

int amos_orb_gate_batch_device(amos_orb *h, const uint8_t *d_masks, size_t mask_frame_stride, size_t mask_row_stride)
{
    if (!h || !d_masks || mask_row_stride < 1) { set_error("amos_orb_gate_batch_device: invalid argument"); return AMOS_ERR_INVALID; }
    if (!h->detected) { set_error("amos_orb_gate_batch_device before detect"); return AMOS_ERR_STATE; }
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    const Geom &g = h->geom;
    const size_t planeStride = (size_t)h->maskPitch * h->maxH;
    dim3 grid((g.W + kMorphTileW - 1) / kMorphTileW, (g.H + kMorphTileH - 1) / kMorphTileH, h->nFrames);
    AMOS_HIP_CHECK(hipMemsetAsync(h->dErr, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_morph31<true>, grid, dim3(256), 0, h->stream, d_masks, mask_frame_stride, (int)mask_row_stride, h->dMaskTmp, planeStride,
                       h->maskPitch, g.W, g.H);
    hipLaunchKernelGGL(k_morph31<false>, grid, dim3(256), 0, h->stream, h->dMaskTmp, planeStride, h->maskPitch, h->dMaskClosed, planeStride,
                       h->maskPitch, g.W, g.H);
    hipLaunchKernelGGL(k_gate, dim3(h->nFrames), dim3(256), 0, h->stream, h->dGeom, h->dLvKps, h->dLvCount, h->dMaskClosed, planeStride, h->maskPitch,
                       (const double *)nullptr, 0, (const int *)nullptr, 0, (const int *)nullptr, 0, h->dRemoved, h->dNRemoved, h->dErr);
    AMOS_HIP_CHECK(hipGetLastError());
    h->gated = true;
    return AMOS_OK;
}

int amos_orb_describe_batch_device(amos_orb *h)
{
    if (!h || !h->detected) { set_error("amos_orb_describe_batch_device before detect"); return AMOS_ERR_STATE; }
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    return launch_describe(h, h->nFrames);
}

int amos_orb_extract_batch_device_color(amos_orb *h, const uint8_t *d_color, size_t frame_stride, size_t row_stride, int width,
                                        int height, int n_frames, int channels, int rgb_order)
{
    if (!h || !d_color || n_frames < 1 || width < 1 || height < 1 || (channels != 3 && channels != 4) || row_stride < (size_t)width * channels) {
        set_error("amos_orb_extract_batch_device_color: invalid argument");
        return AMOS_ERR_INVALID;
    }
    if (n_frames > h->maxB) { set_error("batch of %d frames exceeds the handle's max_batch %d", n_frames, h->maxB); return AMOS_ERR_CAPACITY; }
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    int rc = set_geometry(h, width, height);
    if (rc != AMOS_OK) return rc;
    rc = launch_detect(h, d_color, frame_stride, row_stride, n_frames, channels, rgb_order != 0);
    if (rc != AMOS_OK) return rc;
    return launch_describe(h, n_frames);
}

int amos_orb_detect_color_with_mask_pre_batch_device(amos_orb *h, amos_mask_pre *pre, const uint8_t *d_color, size_t frame_stride, size_t row_stride,
                                                     int width, int height, int n_frames, int channels, int rgb_order, float *d_net_input)
{
    if (!h || !pre || !d_color || !d_net_input || n_frames < 1 || width < 1 || height < 1 || (channels != 3 && channels != 4) ||
        row_stride < (size_t)width * channels) {
        set_error("amos_orb_detect_color_with_mask_pre_batch_device: invalid argument");
        return AMOS_ERR_INVALID;
    }
    if (n_frames > h->maxB) { set_error("batch of %d frames exceeds the handle's max_batch %d", n_frames, h->maxB); return AMOS_ERR_CAPACITY; }
    MaskPreStageA a;
    int rc = mask_pre_stage_a(pre, &a);
    if (rc != AMOS_OK) return rc;
    if (a.width != width || a.height != height) { set_error("mask pre-processing handle made for %dx%d frames, got %dx%d", a.width, a.height, width, height); return AMOS_ERR_INVALID; }
    if (n_frames > a.maxBatch) { set_error("mask pre-processing handle made for %d frames, got %d", a.maxBatch, n_frames); return AMOS_ERR_CAPACITY; }
    if (width <= 2 * kEdge + 1 || height <= 2 * kEdge + 1) { set_error("frame too small for the fused import"); return AMOS_ERR_INVALID; }
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    rc = set_geometry(h, width, height);
    if (rc != AMOS_OK) return rc;
    rc = launch_detect(h, d_color, frame_stride, row_stride, n_frames, channels, rgb_order != 0, &a);
    if (rc != AMOS_OK) return rc;
    return mask_pre_finish(pre, h->stream, n_frames, d_net_input);  // stages B and C follow on the same stream
}

static int make_undistort_args(float fx, float fy, float cx, float cy, const float *dist_coef, int n_dist, UndistortArgs &a, const char *who)
{
    if (!(fx != 0.f) || !(fy != 0.f) || n_dist < 0 || n_dist > 5 || (n_dist > 0 && !dist_coef)) { set_error("%s: invalid camera", who); return AMOS_ERR_INVALID; }
    a.fx = fx; a.fy = fy; a.cx = cx; a.cy = cy;
    for (int i = 0; i < 5; i++) a.k[i] = i < n_dist ? (double)dist_coef[i] : 0.;
    a.identity = n_dist == 0 || dist_coef[0] == 0.0f;
    return AMOS_OK;
}

int amos_frame_undistort_batch_device(amos_orb *h, float fx, float fy, float cx, float cy, const float *dist_coef, int n_dist,
                                      amos_keypoint *d_kps_un)
{
    if (!h || !d_kps_un) { set_error("amos_frame_undistort_batch_device: invalid argument"); return AMOS_ERR_INVALID; }
    if (!h->described) { set_error("amos_frame_undistort_batch_device before an extraction"); return AMOS_ERR_STATE; }
    UndistortArgs a;
    const int rc = make_undistort_args(fx, fy, cx, cy, dist_coef, n_dist, a, "amos_frame_undistort_batch_device");
    if (rc != AMOS_OK) return rc;
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    hipLaunchKernelGGL(k_undistort, dim3((h->geom.kpCap + 255) / 256, h->nFrames), dim3(256), 0, h->stream, h->dGeom, h->dOutKps, h->dOutCount, a,
                       d_kps_un);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_frame_image_bounds(int width, int height, float fx, float fy, float cx, float cy, const float *dist_coef, int n_dist, float bounds[4])
{
    if (!bounds || width < 1 || height < 1) { set_error("amos_frame_image_bounds: invalid argument"); return AMOS_ERR_INVALID; }
    UndistortArgs a;
    const int rc = make_undistort_args(fx, fy, cx, cy, dist_coef, n_dist, a, "amos_frame_image_bounds");
    if (rc != AMOS_OK) return rc;
    if (a.identity) { bounds[0] = 0.f; bounds[1] = (float)width; bounds[2] = 0.f; bounds[3] = (float)height; return AMOS_OK; }
    const float px[4] = {0.f, (float)width, 0.f, (float)width}, py[4] = {0.f, 0.f, (float)height, (float)height};
    float ux[4], uy[4];
    for (int i = 0; i < 4; i++) undistort_point(px[i], py[i], a.fx, a.fy, a.cx, a.cy, a.k, ux[i], uy[i]);
    bounds[0] = std::min(ux[0], ux[2]);  // Frame.cc:1152-1155
    bounds[1] = std::max(ux[1], ux[3]);
    bounds[2] = std::min(uy[0], uy[1]);
    bounds[3] = std::max(uy[2], uy[3]);
    return AMOS_OK;
}

int amos_frame_rgbd_glue_batch_device(amos_orb *h, const void *d_depth, int depth_is_u16, float depth_map_factor,
                                      size_t depth_frame_stride_bytes, size_t depth_row_stride_bytes, float mbf, float min_x, float max_x,
                                      float min_y, float max_y, const amos_keypoint *d_kps_un, float *d_u_right, float *d_depth_out,
                                      int32_t *d_grid_cell)
{
    if (!h || !d_grid_cell || (d_depth && (!d_u_right || !d_depth_out)) || !(max_x > min_x) || !(max_y > min_y)) { set_error("amos_frame_rgbd_glue_batch_device: invalid argument"); return AMOS_ERR_INVALID; }
    if (!h->described) { set_error("amos_frame_rgbd_glue_batch_device before an extraction"); return AMOS_ERR_STATE; }
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    const float wInv = static_cast<float>(AMOS_FRAME_GRID_COLS) / static_cast<float>(max_x - min_x);  // Frame.cc:302-303
    const float hInv = static_cast<float>(AMOS_FRAME_GRID_ROWS) / static_cast<float>(max_y - min_y);
    hipLaunchKernelGGL(k_rgbd_glue, dim3((h->geom.kpCap + 255) / 256, h->nFrames), dim3(256), 0, h->stream, h->dGeom, h->dOutKps, d_kps_un, h->dOutCount,
                       (const uint8_t *)d_depth, depth_is_u16, depth_map_factor, depth_frame_stride_bytes, depth_row_stride_bytes, mbf, min_x, min_y,
                       wInv, hInv, d_u_right, d_depth_out, d_grid_cell);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_orb_batch_results_device(amos_orb *h, const amos_keypoint **d_kps, const uint8_t **d_desc, const int32_t **d_counts,
                                  int *capacity)
{
    if (!h) return AMOS_ERR_INVALID;
    if (d_kps) *d_kps = h->dOutKps;
    if (d_desc) *d_desc = h->dOutDesc;
    if (d_counts) *d_counts = h->dOutCount;
    if (capacity) *capacity = h->capGeom.kpCap;
    return AMOS_OK;
}

int amos_orb_batch_fetch(amos_orb *h, int frame, amos_keypoint *kps, uint8_t *desc, int cap, int *n)
{
    if (!h || !h->described || frame < 0 || frame >= h->nFrames) { set_error("amos_orb_batch_fetch: invalid argument or state"); return AMOS_ERR_INVALID; }
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    return fetch_frame(h, frame, kps, desc, cap, n);
}

int amos_orb_sync(amos_orb *h)
{
    if (!h) return AMOS_ERR_INVALID;
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    return AMOS_OK;
}

void *amos_orb_stream(amos_orb *h) { return h ? (void *)h->stream : nullptr; }

int amos_orb_timing_enable(amos_orb *h, int max_records)
{
    if (!h || max_records < 0) return AMOS_ERR_INVALID;
    AMOS_HIP_CHECK(hipSetDevice(h->device));
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
    h->events.clear();
    h->maxRecords = h->nRecords = 0;
    for (int i = 0; i < max_records * (kTimingEvents); i++) {
        hipEvent_t e;
        AMOS_HIP_CHECK(hipEventCreate(&e));
        h->events.push_back(e);
    }
    h->maxRecords = max_records;
    return AMOS_OK;
}

int amos_orb_timing_collect(amos_orb *h, float *avg_ms, int *n_records)
{
    if (!h || !avg_ms) return AMOS_ERR_INVALID;
    AMOS_HIP_CHECK(hipStreamSynchronize(h->stream));
    for (int s = 0; s < AMOS_ORB_STAGES; s++) avg_ms[s] = 0.f;
    static const int first[AMOS_ORB_STAGES] = {0, 1, 3, 4, 5, -1, 7}, last[AMOS_ORB_STAGES] = {1, 2, 4, 5, 6, -1, 8};
    for (int r = 0; r < h->nRecords; r++)
        for (int s = 0; s < AMOS_ORB_STAGES; s++) {
            if (first[s] < 0) continue;  // the blur runs on the side stream: timed below from its own events
            float ms = 0.f;
            const hipEvent_t *ev = &h->events[(size_t)r * kTimingEvents];
            AMOS_HIP_CHECK(hipEventElapsedTime(&ms, ev[first[s]], ev[last[s]]));
            avg_ms[s] += ms;
        }
    if (h->nRecords > 0) {
        for (int s = 0; s < AMOS_ORB_STAGES; s++) avg_ms[s] /= (float)h->nRecords;
        AMOS_HIP_CHECK(hipStreamSynchronize(h->streamB));
        float ms = 0.f;  // last pass only (one event pair): the blur overlaps FAST
        AMOS_HIP_CHECK(hipEventElapsedTime(&ms, h->evBlur0, h->evBlur1));
        avg_ms[5] = ms;
    }
    if (n_records) *n_records = h->nRecords;
    h->nRecords = 0;
    return AMOS_OK;
}

}  // extern "C"

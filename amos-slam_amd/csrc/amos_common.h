// amos_common.h -- shared host/device declarations of the MI355X front-end library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>

#include "../../include/amos_frontend.h"

namespace amos {

void set_error(const char *fmt, ...);

#define AMOS_HIP_CHECK(expr)                                                                   \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            amos::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                            __LINE__);                                                         \
            (void)hipGetLastError(); /* reported here: it must not surface again at the next launch's check */ \
            return AMOS_ERR_DEVICE;                                                            \
        }                                                                                      \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting of a kernel: a "set once" flag must be kept per device (a
// host that opens two GPUs in one process would otherwise launch with the default 64 KB limit on the second one) and be safe to reach
// from several threads (setting the same value twice is harmless; the flag is published only after the call succeeded).  The attribute is
// set on the CURRENT device, so the stream the launch goes to must belong to it: the bare-stream entry points (amos_mask_*_device) pass
// their stream and get hipErrorInvalidDevice when it belongs to another device than the caller's current one (a launch there would
// otherwise fail, or run with the 64 KB default, far from the cause).
struct DeviceOnce {
    std::atomic<unsigned long long> done[4] = {};  // bit per device ordinal, 256 ordinals; larger ordinals are simply set every time
};
inline hipError_t set_max_dynamic_lds(DeviceOnce &once, const void *kernel, int bytes, hipStream_t stream = nullptr)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (stream) {
        hipDevice_t sdev = 0;
        e = hipStreamGetDevice(stream, &sdev);
        if (e != hipSuccess) return e;
        if ((int)sdev != dev) return hipErrorInvalidDevice;
    }
    const bool tracked = dev >= 0 && dev < 256;
    const unsigned long long bit = 1ull << (dev & 63);
    if (tracked && (once.done[dev >> 6].load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && tracked) once.done[dev >> 6].fetch_or(bit, std::memory_order_release);
    return e;
}

constexpr int kEdge = AMOS_EDGE_THRESHOLD;  // 19, ORBextractor.cc:93
constexpr int kPadLeft = 32;                // device planes keep the ROI origin 32-byte aligned
constexpr int kHalfPatch = 15;              // ORBextractor.cc:92
constexpr int kMinBorder = kEdge - 3;       // 16, ORBextractor.cc:1067
constexpr int kWave = 64;

// Geometry of one pyramid level for the current frame size (host-built, read by every kernel).
struct LevelGeom {
    int w, h;          // level image size (ORBextractor.cc:1832-1834)
    int stride;        // bytes per row of the padded device plane
    int planeOff;      // byte offset of the plane inside one frame's pyramid
    int maxBX, maxBY;  // maxBorderX/Y = dim - 16 (ORBextractor.cc:1069-1070)
    int nCols, nRows, wCell, hCell;  // FAST cell grid (ORBextractor.cc:1078-1086)
    int cellStart, nCells;           // rows of the cell table that belong to this level
    int ptsOff, ptsCap;              // compacted candidate array of this level inside one frame
    int quota;                       // mnFeaturesPerLevel[level]
    int nIni;                        // root nodes of the quad-tree (ORBextractor.cc:718)
    int nodeCap;                     // capacity of the node list / of the level's keypoint list
    int kpOff;                       // offset of the level's keypoint list inside one frame
    int tabX, tabY;                  // offsets into the resize coefficient tables
    int blurGroups, blurItemStart;   // blur work items: 8-px column groups x kBlurStrip-row strips
    float scale;                     // mvScaleFactor[level]
    float patchSize;                 // (float)(int)(31 * scale), ORBextractor.cc:1177
};

struct Geom {
    int nLevels;
    int W, H;
    int totalCells;      // cells per frame, all levels
    int slotTotal;       // candidate slots per frame, all cells
    int ptsTotal;        // compacted candidate capacity per frame, all levels
    int kpLevelTotal;    // per-level keypoint list capacity per frame (sum of nodeCap)
    int kpCap;           // capacity of the concatenated result per frame
    int iniTh, minTh;
    int blurItems;       // blur work items per frame, all levels
    // FAST LDS carve (per wave = per cell), sized by the largest cell of the geometry
    int fastTileStrideDw, fastTileRows, fastMapRows, fastKeptCap, fastWaveBytes;
    unsigned long long frameBytes;  // bytes of one frame's pyramid
    LevelGeom lv[AMOS_MAX_LEVELS];
};

// One FAST cell: the pixels it tests and where its candidates go.
struct Cell {
    short level;
    short x0, y0;  // first tested pixel (level coordinates) = (iniX + 3, iniY + 3)
    short tw, th;  // tested region size
    short ndw;     // 16-byte pieces per staged tile row = (tw + 7 + 15) / 16
    int slotOff;   // first candidate slot of the cell inside one frame
    unsigned magicDw, magicG;  // (1 << 20) / d + 1 for d = pieces per row, groups: division by multiplication
    short groups;  // 8-pixel groups per row = (tw + 7) / 8
    short pad[3];
};

// Resize coefficients of one destination column / row (cv::resize fixed point, 11 bits).  The tables
// are indexed by PADDED destination coordinate (column -32 .., row -19 ..) with the reflect-101 map
// of the border already applied, so the kernel treats border and interior alike.
struct ResizeTap {
    short ofs;     // source index of the first tap
    short ofs1;    // source index of the second tap (clamped)
    short a0, a1;  // weights, sum 2048
};

// ---- mask pre-processing handle, as the fused import kernel of amos_orb.hip sees it (amos_mask_pre.hip)
constexpr int kMaskMidW = 480, kMaskMidH = 640;  // yolact.cc:220 cv::Size(480, 640): the intermediate image of the mask pre-processing
struct FixTap { int s0, s1, a0, a1; };  // 8-bit cv::resize: source indices and 11-bit weights of one destination index
struct MaskPreStageA {
    const FixTap *tx, *ty;        // destination column / row -> taps (kMidW / kMidH entries)
    const int *firstX, *firstY;   // source column X / row Y -> first destination index whose first tap is >= X (width + 1 / height + 1 entries)
    const float *lut;             // u8 -> float(double(v) / 255.0) * 255.0f
    float *mid;                   // [frames][kMidH][kMidW][3]
    int width, height, maxBatch;
};

}  // namespace amos

struct amos_mask_pre;
namespace amos {
int mask_pre_stage_a(amos_mask_pre *p, MaskPreStageA *out);                              // tables and buffers of stage A
int mask_pre_finish(amos_mask_pre *p, hipStream_t stream, int n_frames, float *d_out);  // stages B and C on `stream`
const char *w24_variant_tag();  // amos_winograd24.hip: " NAME" per timing-experiment switch compiled in, "" for the product build
}  // namespace amos

// orb_kernels.h -- CDNA4 (gfx950) device code of the ORB extractor hot path.
//
// Integer / bitwise work, HBM- and LDS-bound: no MFMA anywhere.  wave = 64 lanes throughout.
// Every kernel takes a batch of frames (blockIdx.y or .z = frame) so a launch fills the chip's
// 256 CUs; a single frame is the batch-of-one case of the same code.
//
// Compiled with -ffp-contract=off; the only fused operations are the explicit __fmaf_rn calls in
// k_describe, which restate the FMA the reference binary executes.
#pragma once
#include <type_traits>

#include "amos_common.h"
#include "../../include/amos_orb_pattern.h"
#include "../../include/amos_host_types.h"

// Wave priority of the latency-bound kernels (quad-tree, orientation, rBRIEF, matcher): with several lanes on the chip their waves share
// SIMDs with the VALU-dense FAST waves of another lane; a higher s_setprio lets the few instructions of a latency-bound wave issue ahead
// of those (the arbiter picks by priority, then age: MI355X_MICROARCH.md), which shortens that lane's critical path at almost no cost
// to FAST.  0 = off (experiments: tools/orb_variants.sh).
#ifndef AMOS_LATENCY_PRIO
#define AMOS_LATENCY_PRIO 0
#endif
#define AMOS_SET_LATENCY_PRIO() do { if (AMOS_LATENCY_PRIO > 0) __builtin_amdgcn_s_setprio(AMOS_LATENCY_PRIO); } while (0)

namespace amos {

__constant__ signed char c_pattern[1024];
// umax of the 31-px circular patch, ORBextractor.cc:579-608 (recomputed and checked on the host)
__constant__ int c_umax[16];

__device__ __forceinline__ int reflect101(int i, int n)
{
    // single reflection: callers keep |overshoot| <= 19 < n
    if (i < 0) i = -i;
    if (i >= n) i = 2 * n - 2 - i;
    return i;
}

// XCD-aware work mapping.  Work-groups are dealt round-robin over the 8 XCDs (b and b + 8 share one,
// each XCD has its own 4 MiB L2), so a 1-D grid is decoded as: XCD = b % 8 owns frames f = 8k + XCD
// and walks them one after the other, chunk by chunk.  All work-groups that touch one frame's
// planes (1.4 MB at 640x480) then run on one XCD close together in time and share its L2; with the
// plain (chunk, frame) grid every XCD fetched every frame.  Placement is a speed matter only.
__device__ __forceinline__ bool xcd_frame_chunk(int b, int chunksPerFrame, int nFrames, int &frame, int &chunk)
{
    const int xcd = b & 7, q = b >> 3;
    const int fk = q / chunksPerFrame;
    frame = fk * 8 + xcd;
    chunk = q - fk * chunksPerFrame;
    return frame < nFrames;
}
__host__ __device__ inline int xcd_grid(int chunksPerFrame, int nFrames) { return ((nFrames + 7) / 8) * 8 * chunksPerFrame; }

__device__ __forceinline__ const uint8_t *level_origin(const uint8_t *pyr, const Geom *g, int frame, int level)
{
    const LevelGeom &lg = g->lv[level];
    return pyr + (size_t)frame * g->frameBytes + lg.planeOff + (size_t)kEdge * lg.stride + kPadLeft;
}
__device__ __forceinline__ uint8_t *level_origin(uint8_t *pyr, const Geom *g, int frame, int level)
{
    const LevelGeom &lg = g->lv[level];
    return pyr + (size_t)frame * g->frameBytes + lg.planeOff + (size_t)kEdge * lg.stride + kPadLeft;
}

// ---------------------------------------------------------------------------------------------
// a2  ORBextractor::ComputePyramid, ORBextractor.cc:1826-1886.
// One launch per level (level l needs level l-1 complete).  A thread produces 4 horizontally
// adjacent bytes of the PADDED plane and stores them as one aligned dword: interior pixels by
// cv::resize's 11-bit fixed-point bilinear, border pixels by the same formula at the
// reflect-101 source coordinate (the host-built tap tables are indexed by padded
// coordinate with the reflection folded in).
// Level 0 is imported by k_pyramid_level0_wide (16 bytes per lane); levels >= 1 by k_pyramid_level with
// grid = xcd_grid(ceil(groups/64) * ceil((h+38)/(4*kPyrRows)), frames), block = (64, 4).
constexpr int kPyrRows = 8;  // padded rows per thread (x taps are loaded once; consecutive rows share source rows)

// Level 0, 16 bytes per lane.  Interior pieces (16 consecutive bytes of the padded plane whose source is
// 16 consecutive image bytes) are one dwordx4 load (any source alignment) and one aligned dwordx4 store,
// kImportRows rows per thread.  The pieces that touch the reflect-101 border gather byte by byte; they
// get their OWN threads (one piece of one row each, at the end of the grid) so that no interior wave
// waits for a lane's sixteen dependent byte loads.
// grid = (ceil((interiorPieces * rowChunks + borderPieces * rows) / 256), 1, frames).
constexpr int kImportRows = 4;

__host__ __device__ inline int import_threads(int w, int h)
{
    const int pieces = (kPadLeft + w + kEdge + 15) >> 4, nInt = w >> 4, nrows = h + 2 * kEdge;
    return nInt * ((nrows + kImportRows - 1) / kImportRows) + (pieces - nInt) * nrows;
}

__global__ __launch_bounds__(256) void k_pyramid_level0_wide(const uint8_t *__restrict__ src, size_t srcFrameStride,
                                                            size_t srcRowStride, uint8_t *__restrict__ pyr,
                                                            const Geom *__restrict__ g)
{
    const LevelGeom &lg = g->lv[0];
    const int frame = blockIdx.z;
    const int pieces = (kPadLeft + lg.w + kEdge + 15) >> 4;  // stride is a multiple of 128: whole pieces fit
    const int nInt = lg.w >> 4, nrows = lg.h + 2 * kEdge;
    const int chunks = (nrows + kImportRows - 1) / kImportRows;
    const int t = blockIdx.x * 256 + threadIdx.x;
    uint8_t *plane = level_origin(pyr, g, frame, 0);
    const uint8_t *s0 = src + (size_t)frame * srcFrameStride;
    if (t < nInt * chunks) {
        const int chunk = t / nInt, piece = t - chunk * nInt;
        const int row0 = chunk * kImportRows, x0 = piece * 16;
        uint4 v[kImportRows];
#pragma unroll
        for (int r = 0; r < kImportRows; r++) {
            const int yo = min(row0 + r, nrows - 1) - kEdge;
            __builtin_memcpy(&v[r], s0 + (size_t)reflect101(yo, lg.h) * srcRowStride + x0, 16);
        }
#pragma unroll
        for (int r = 0; r < kImportRows; r++) {
            const int yo = row0 + r - kEdge;
            if (yo < lg.h + kEdge) *reinterpret_cast<uint4 *>(plane + __mul24(yo, lg.stride) + x0) = v[r];
        }
        return;
    }
    const int nb = pieces - nInt, b = t - nInt * chunks;
    const int row = b / nb, bp = b - row * nb;
    if (row >= nrows) return;
    const int piece = bp < (kPadLeft >> 4) ? bp : bp + nInt;  // left pad pieces, then everything right of the interior
    const int x0 = piece * 16 - kPadLeft, yo = row - kEdge;
    const uint8_t *sr = s0 + (size_t)reflect101(yo, lg.h) * srcRowStride;
    unsigned w[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        w[d] = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int xo = x0 + d * 4 + k;
            xo = xo < -kEdge ? -kEdge : (xo > lg.w + kEdge - 1 ? lg.w + kEdge - 1 : xo);
            w[d] |= (unsigned)sr[reflect101(xo, lg.w)] << (8 * k);
        }
    }
    *reinterpret_cast<uint4 *>(plane + __mul24(yo, lg.stride) + x0) = make_uint4(w[0], w[1], w[2], w[3]);
}

// Level 0 from interleaved colour frames: cv::cvtColor(..., CV_{BGR,RGB}[A]2GRAY) (Tracking.cc:308-321)
// fused into the import.  OpenCV 4.x 8-bit path: gray = (B*3735 + G*19235 + R*9798 + 16384) >> 15.
__global__ __launch_bounds__(256) void k_pyramid_level0_color(const uint8_t *__restrict__ src, size_t srcFrameStride,
                                                             size_t srcRowStride, uint8_t *__restrict__ pyr,
                                                             const Geom *__restrict__ g, int channels, int rgbOrder)
{
    const LevelGeom &lg = g->lv[0];
    const int frame = blockIdx.z;
    const int gx = blockIdx.x * 64 + threadIdx.x;
    const int row0 = (blockIdx.y * 4 + threadIdx.y) * kPyrRows;
    const int groups = (kPadLeft + lg.w + kEdge + 3) >> 2;
    if (gx >= groups) return;
    const int x0 = gx * 4 - kPadLeft;
    int xi[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int xo = x0 + k;
        xo = xo < -kEdge ? -kEdge : (xo > lg.w + kEdge - 1 ? lg.w + kEdge - 1 : xo);
        xi[k] = reflect101(xo, lg.w) * channels;
    }
    const unsigned c0 = rgbOrder ? 9798u : 3735u, c2 = rgbOrder ? 3735u : 9798u;  // weight of byte 0 / byte 2
    uint8_t *plane = level_origin(pyr, g, frame, 0);
    const uint8_t *s0 = src + (size_t)frame * srcFrameStride;
#pragma unroll
    for (int r = 0; r < kPyrRows; r++) {
        const int yo = row0 + r - kEdge;
        if (yo >= lg.h + kEdge) break;
        const uint8_t *s = s0 + (size_t)reflect101(yo, lg.h) * srcRowStride;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint8_t *px = s + xi[k];
            const unsigned y = (px[0] * c0 + px[1] * 19235u + px[2] * c2 + 16384u) >> 15;
            packed |= y << (8 * k);
        }
        *reinterpret_cast<uint32_t *>(plane + __mul24(yo, lg.stride) + x0) = packed;
    }
}

// ONE pass over the colour frame for both of its consumers -- Tracking::GrabImageRGBD's cvtColor into the
// padded level 0 (Tracking.cc:308-321, as k_pyramid_level0_color) and the first stage of the mask pre-processing,
// yolact::evalImage's cv::resize of the BGR frame to W480 x H640 with the u8 -> float step (yolact.cc:220, 385-451,
// as k_mask_pre_a of amos_mask_pre.hip).  A work-group stages a 64 x 16 pixel tile (+ one halo column and row, clamped
// like the resize taps) in LDS, writes the tile's gray pixels (a dword of four per thread; pixels within 19 of an image
// edge also go to their reflect-101 positions in the border) and every pixel of the intermediate image whose FIRST tap
// lies in the tile (the inverse tap tables firstX / firstY give the index ranges; the second tap is the halo at most).
// grid = (ceil(W / 64), ceil(H / 16), frames), block = 256.
constexpr int kFuseTileW = 64, kFuseTileH = 16, kFusePitch = (kFuseTileW + 1) * 3 + 1;

__global__ __launch_bounds__(256) void k_import_color_mask(const uint8_t *__restrict__ src, size_t srcFrameStride, size_t srcRowStride,
                                                          uint8_t *__restrict__ pyr, const Geom *__restrict__ g, int channels, int rgbOrder,
                                                          const MaskPreStageA a)
{
    __shared__ uint8_t tile[kFuseTileH + 1][kFusePitch];
    const LevelGeom &lg = g->lv[0];
    const int W = lg.w, H = lg.h, frame = blockIdx.z, tid = threadIdx.x;
    const int X0 = blockIdx.x * kFuseTileW, Y0 = blockIdx.y * kFuseTileH;
    const uint8_t *s0 = src + (size_t)frame * srcFrameStride;
    for (int idx = tid; idx < (kFuseTileH + 1) * (kFuseTileW + 1); idx += 256) {
        const int r = idx / (kFuseTileW + 1), c = idx - r * (kFuseTileW + 1);
        const uint8_t *px = s0 + (size_t)min(Y0 + r, H - 1) * srcRowStride + (size_t)min(X0 + c, W - 1) * channels;
        tile[r][3 * c] = px[0];
        tile[r][3 * c + 1] = px[1];
        tile[r][3 * c + 2] = px[2];
    }
    __syncthreads();
    {   // gray: thread = (row, four columns)
        const int ly = tid >> 4, lx = (tid & 15) * 4, y = Y0 + ly, x = X0 + lx;
        if (y < H && x < W) {
            const unsigned c0 = rgbOrder ? 9798u : 3735u, c2 = rgbOrder ? 3735u : 9798u;  // weight of byte 0 / byte 2
            uint8_t *plane = level_origin(pyr, g, frame, 0);
            unsigned v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint8_t *t = &tile[ly][3 * (lx + k)];
                v[k] = (t[0] * c0 + t[1] * 19235u + t[2] * c2 + 16384u) >> 15;
            }
            const int yr = (y >= 1 && y <= kEdge) ? -y : (y >= H - 1 - kEdge && y <= H - 2) ? 2 * (H - 1) - y : y;  // reflect-101 image of the row (or itself)
            if (x + 3 < W) *reinterpret_cast<uint32_t *>(plane + (ptrdiff_t)y * lg.stride + x) = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
            else
                for (int k = 0; k < 4 && x + k < W; k++) plane[(ptrdiff_t)y * lg.stride + x + k] = (uint8_t)v[k];
            for (int k = 0; k < 4 && x + k < W; k++) {
                const int xx = x + k;
                const int xr = (xx >= 1 && xx <= kEdge) ? -xx : (xx >= W - 1 - kEdge && xx <= W - 2) ? 2 * (W - 1) - xx : xx;
                if (yr != y) plane[(ptrdiff_t)yr * lg.stride + xx] = (uint8_t)v[k];
                if (xr != xx) {
                    plane[(ptrdiff_t)y * lg.stride + xr] = (uint8_t)v[k];
                    if (yr != y) plane[(ptrdiff_t)yr * lg.stride + xr] = (uint8_t)v[k];
                }
            }
        }
    }
    // the intermediate image of the mask pass: destination indices whose first taps fall into this tile
    const int dx0 = a.firstX[X0], dx1 = a.firstX[min(X0 + kFuseTileW, W)], dy0 = a.firstY[Y0], dy1 = a.firstY[min(Y0 + kFuseTileH, H)];
    const int nx = dx1 - dx0, n = nx * (dy1 - dy0);
    float *mid = a.mid + (size_t)frame * kMaskMidH * kMaskMidW * 3;
    for (int idx = tid; idx < n; idx += 256) {
        const int ry = idx / nx, dy = dy0 + ry, dx = dx0 + (idx - ry * nx);
        const FixTap ax = a.tx[dx], ay = a.ty[dy];
        const uint8_t *r0 = tile[ay.s0 - Y0], *r1 = tile[ay.s1 - Y0];
        const int xa = 3 * (ax.s0 - X0), xb = 3 * (ax.s1 - X0);
        float *o = mid + ((size_t)dy * kMaskMidW + dx) * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int h0 = r0[xa + c] * ax.a0 + r0[xb + c] * ax.a1;
            const int h1 = r1[xa + c] * ax.a0 + r1[xb + c] * ax.a1;
            const int v = (((ay.a0 * (h0 >> 4)) >> 16) + ((ay.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
            o[c] = a.lut[v];  // float(double(v) / 255.0) * 255.0f
        }
    }
}

// Levels >= 1.  Measured on MI355X (profiles/, tools/kt_var.sh): a CU issues about one vector-memory
// instruction per 16 cycles whatever its width and one VALU instruction per cycle, and this kernel was
// bound by the former.  So:
//   * ONE unaligned 8-byte buffer load per source row fetches the taps of all four outputs of a
//     thread (kWide; valid while 3 * scaleFactor + 2 < 8, i.e. scaleFactor < 2 -- otherwise four
//     dword loads); buffer addressing = SGPR descriptor + per-lane column offset + wave-uniform row
//     offset, so there is no vector address arithmetic;
//   * per output and source row one v_perm (two tap bytes -> u16 pair, per-lane selector) and one
//     v_dot2_u32_u16:  H = S[sx]*a0 + S[sx+1]*a1  (a1 == 0 wherever cv::resize clamps sx+1, so the
//     byte after the tap may be read: the planes are padded);
//   * a wave walks kPyrRows consecutive padded rows; rows are wave-uniform, and "the second source
//     row of output row r is the first of row r+1" (most rows at scale 1.2) is a scalar branch that
//     reuses the four H values;
//   * vertical pass ((b*(H>>4))>>16 per tap) = v_and + v_mul_hi_u32_u24 with b << 12.
// grid = xcd_grid(ceil(groups/64) * ceil((h+38)/(4*kPyrRows)), frames), block = (64, 4).
__device__ __forceinline__ unsigned mul_hi_u24(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// the same with the first factor wave-uniform (an SGPR operand: no copy into a vector register)
__device__ __forceinline__ unsigned mul_hi_u24_s(unsigned uniform, unsigned b)
{
    unsigned r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "s"(uniform), "v"(b));
    return r;
}

// Raw buffer descriptor over [p, p + bytes).  Word 3 = 0x00020000: raw 32-bit data format of gfx9.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_buffer(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

template <bool kWide>
__global__ __launch_bounds__(256) void k_pyramid_level(uint8_t *__restrict__ pyr, const Geom *__restrict__ g,
                                                      const ResizeTap *__restrict__ taps, int level, int nFrames)
{
    typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
    typedef unsigned uint2v __attribute__((ext_vector_type(2)));
    const LevelGeom &lg = g->lv[level];
    const int groups = (kPadLeft + lg.w + kEdge + 3) >> 2;
    const int nrows = lg.h + 2 * kEdge;
    // 1-D grid decoded XCD-aware (xcd_frame_chunk): all work-groups of a frame's level run on ONE XCD close together in time, so a source
    // row that several row groups (and the 8-byte windows of neighbouring lanes) read comes from HBM once -- with the (x, y, frame) grid
    // the work-groups of a frame were dealt over all eight XCDs and each L2 fetched the rows again (2.6 x the level's bytes)
    const int nbx = (groups + 63) >> 6, nby = (nrows + 4 * kPyrRows - 1) / (4 * kPyrRows);
    int frame, chunk;
    if (!xcd_frame_chunk(blockIdx.x, nbx * nby, nFrames, frame, chunk)) return;
    const int by = chunk / nbx, bx = chunk - by * nbx;
    const int gx = bx * 64 + threadIdx.x;
    const int row0 = __builtin_amdgcn_readfirstlane((by * 4 + threadIdx.y) * kPyrRows);  // padded row, 0 = yo -19
    if (row0 >= nrows) return;
    const bool active = gx < groups;
    const LevelGeom &pg = g->lv[level - 1];
    const uint8_t *prev = level_origin((const uint8_t *)pyr, g, frame, level - 1);
    // rows 0 .. h-1 of the previous level (+ the right border read after the last column's taps)
    const __amdgpu_buffer_rsrc_t src = make_buffer(prev, (unsigned)(pg.h * pg.stride));
    // destination: the padded plane from its first byte (row -19, column -kPadLeft); lane offset gx * 4
    const __amdgpu_buffer_rsrc_t dst = make_buffer(level_origin(pyr, g, frame, level) - kEdge * lg.stride - kPadLeft, (unsigned)(nrows * lg.stride));
    const uint4 *txp = reinterpret_cast<const uint4 *>(taps + lg.tabX + (active ? gx : 0) * 4);
    const uint4 t01 = txp[0], t23 = txp[1];  // four ResizeTap records
    const unsigned ofs[4] = {t01.x & 0xffffu, t01.z & 0xffffu, t23.x & 0xffffu, t23.z & 0xffffu};  // source columns, >= 0
    const ushort2v wgt[4] = {__builtin_bit_cast(ushort2v, t01.y), __builtin_bit_cast(ushort2v, t01.w),
                             __builtin_bit_cast(ushort2v, t23.y), __builtin_bit_cast(ushort2v, t23.w)};  // a0 | a1 << 16
    const unsigned base = min(min(ofs[0], ofs[1]), min(ofs[2], ofs[3]));  // reflected borders are not monotonic
    unsigned sel[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned o = ofs[k] - base;            // 0 .. 6: byte of the first tap inside the 8-byte window
        sel[k] = 0x0c000c00u | o | ((o + 1) << 16);  // result = (byte o) | (byte o+1) << 16
    }
    const ResizeTap *tyTab = taps + lg.tabY;
    const int pstride = pg.stride, dstride = lg.stride;
    // Row taps of all kPyrRows rows first (wave-uniform and ahead of every store: scalar loads), then
    // every source-row load the rows need (duplicates skipped by scalar branches) so that all of them
    // are in flight together, then the arithmetic.
    // (the table is read through the constant address space: the compiler cannot see that the buffer
    // stores below never touch it, and would otherwise spend a vector-memory slot per record)
    typedef const __attribute__((address_space(4))) uint2v *ConstTapPtr;  // 8-byte aligned records
    const ConstTapPtr tyConst = (ConstTapPtr)(uintptr_t)tyTab;
    ResizeTap ty[kPyrRows];
#pragma unroll
    for (int r = 0; r < kPyrRows; r++) {
        const uint2v t = tyConst[min(row0 + r, nrows - 1)];
        ty[r].ofs = (short)(t.x & 0xffffu); ty[r].ofs1 = (short)(t.x >> 16);
        ty[r].a0 = (short)(t.y & 0xffffu); ty[r].a1 = (short)(t.y >> 16);
    }
    bool need0[kPyrRows], need1[kPyrRows];
    uint2v W0[kPyrRows], W1[kPyrRows];   // kWide: 8-byte windows of the first / second source row
    unsigned Q0[kPyrRows][4], Q1[kPyrRows][4];  // !kWide: one dword per output column
#pragma unroll
    for (int r = 0; r < kPyrRows; r++) {
        need0[r] = r == 0 || ty[r].ofs != ty[r - 1].ofs1;
        need1[r] = true;  // ofs1 == ofs only where cv::resize clamps the second row (a1 == 0): loading it again costs less than a copy in every row
        if (need0[r]) {
            const int so = __builtin_amdgcn_readfirstlane(__mul24((int)ty[r].ofs, pstride));
            if (kWide) W0[r] = __builtin_bit_cast(uint2v, __builtin_amdgcn_raw_buffer_load_b64(src, (int)base, so, 0));
            else
#pragma unroll
                for (int k = 0; k < 4; k++) Q0[r][k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(src, (int)ofs[k], so, 0);
        }
        if (need1[r]) {
            const int so = __builtin_amdgcn_readfirstlane(__mul24((int)ty[r].ofs1, pstride));
            if (kWide) W1[r] = __builtin_bit_cast(uint2v, __builtin_amdgcn_raw_buffer_load_b64(src, (int)base, so, 0));
            else
#pragma unroll
                for (int k = 0; k < 4; k++) Q1[r][k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(src, (int)ofs[k], so, 0);
        }
    }
    // H comes out with its low four bits cleared: that is the form both vertical products want ((b*(H>>4))>>16 as
    // v_mul_hi_u32_u24 of b << 12 and H & ~15), and a row of H serves two output rows
    auto hsum = [&](const uint2v &w, const unsigned (&q)[4], unsigned (&H)[4]) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned pair = kWide ? __builtin_amdgcn_perm(w.y, w.x, sel[k]) : __builtin_amdgcn_perm(0u, q[k], 0x0c010c00u);
            H[k] = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, pair), wgt[k], 0u, false) & ~15u;
        }
    };
    // Two register sets that swap roles every row: the first source row of an even output row lives in HP and its second
    // in HQ, an odd row has them the other way round, so "the second source row of row r is the first of row r + 1"
    // (the usual case) needs no copy at all.
    unsigned HP[4] = {0, 0, 0, 0}, HQ[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < kPyrRows; r++) {
        const int row = row0 + r;
        if (row >= nrows) break;  // wave-uniform
        unsigned(&H0)[4] = (r & 1) ? HQ : HP;
        unsigned(&H1)[4] = (r & 1) ? HP : HQ;
        if (need0[r]) hsum(W0[r], Q0[r], H0);  // otherwise H0 is the H1 of the row before: same registers
        hsum(W1[r], Q1[r], H1);
        const unsigned b0 = (unsigned)ty[r].a0 << 12, b1 = (unsigned)ty[r].a1 << 12;  // <= 2^23, wave-uniform
        unsigned v[4];
#pragma unroll
        for (int k = 0; k < 4; k++)  // ((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2, then >> 2 while packing; H < 2^20
            v[k] = mul_hi_u24_s(b0, H0[k]) + mul_hi_u24_s(b1, H1[k]) + 2u;
        unsigned packed = v[0] >> 2;  // each sum >> 2 is <= 255: the shifted byte is written into its place
        asm("v_lshrrev_b32_sdwa %0, 2, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(packed) : "v"(v[1]));
        asm("v_lshrrev_b32_sdwa %0, 2, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(packed) : "v"(v[2]));
        asm("v_lshrrev_b32_sdwa %0, 2, %1 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(packed) : "v"(v[3]));
        if (active) __builtin_amdgcn_raw_buffer_store_b32(packed, dst, gx * 4, __builtin_amdgcn_readfirstlane(__mul24(row, dstride)), 0);
    }
}

// ---------------------------------------------------------------------------------------------
// a3  per-cell cv::FAST(…, TYPE_9_16, nonmax) with the iniThFAST -> minThFAST fallback,
// ORBextractor.cc:1089-1157, SURVEY A.3.
//
// For a corner the OpenCV score (cornerScore<16>) does not depend on the threshold:
//   score = A - 1,  A = max( max_arcs min_k(v - p_k), max_arcs min_k(p_k - v) ),  corner(t) <=> A > t.
// So one byte map of A (clamped at 0) per cell serves both thresholds; a pixel survives the 3x3
// non-max suppression at threshold t iff A > t and every in-cell neighbour has A_n <= t or A_n < A
// (scores outside the cell's tested region are 0 in OpenCV's row buffers).
//
// Per cell (ONE WAVE; tile + 3-px halo staged in LDS so that the cell's first tested pixel sits at byte 4
// of its row, i.e. 8-pixel groups are 8-byte aligned):
//   1. a cheap necessary test on every pixel, eight pixels per lane: any 9-arc of the 16-pixel circle
//      contains two compass points 90 degrees apart, so a corner at threshold t needs
//      max(min(n,s), min(e,w)) < c - t  (dark) or  min(max(n,s), max(e,w)) > c + t  (bright).  Even and odd
//      pixels of a dword are isolated by one v_and each (odd ones stay shifted left by 8) and everything is
//      unsigned packed 16-bit with saturating c -+ t -- no unpacking; survivors go onto an LDS list;
//   2. the survivors (dense lanes again) get the exact arc value, two per lane (128 per pass) with ONE unsigned
//      packed max-ladder on the raw circle pixels: a candidate is scored in the polarity its compass test
//      passed in, bright ones on complemented bytes (fast_score_chunk2);
//   3. 3x3 strict non-max suppression over the survivors only; the kept ones are ranked by pixel
//      index (row-major = FAST's output order) and written to the cell's slots.
// The cell is processed at iniThFAST first and, only if that leaves it EMPTY, again at minThFAST
// (ORBextractor.cc:1126-1139) -- a block-uniform retry.
#ifndef AMOS_FAST_EXP
#define AMOS_FAST_EXP 0  /* timing experiments, results are wrong (tools/orb_variants.sh): 1 no arc scoring, 2 no phase 3, 3 no candidate stores (1 - 3 also without the second sweep), 4 staging only, 5 no second sweep */
#endif
typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ short2v pk_min(short2v a, short2v b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ short2v pk_max(short2v a, short2v b) { return __builtin_elementwise_max(a, b); }

constexpr int kFastMapStride = 64;   // cell + 2 halo <= 61 columns
constexpr int kFastMaxCell = 59;

// p / d for p < 2^20 / d (d <= 65, p < 4096 here): (p * ((1 << 20) / d + 1)) >> 20
__device__ __forceinline__ int div_small(int p, unsigned magic) { return (int)(((unsigned)p * magic) >> 20); }

__device__ __forceinline__ short2v unpack_lo(unsigned x) { return __builtin_bit_cast(short2v, __builtin_amdgcn_perm(0u, x, 0x0c010c00u)); }
__device__ __forceinline__ short2v unpack_hi(unsigned x) { return __builtin_bit_cast(short2v, __builtin_amdgcn_perm(0u, x, 0x0c030c02u)); }

// sign bits (15 and 31) set where the compass test passes at threshold t, for two packed pixels
__device__ __forceinline__ unsigned fast_compass(short2v c, short2v n, short2v s, short2v e, short2v w, short2v t)
{
    const short2v dN = c - n, dS = c - s, dE = c - e, dW = c - w;
    const short2v dark = pk_min(pk_max(dS, dN), pk_max(dE, dW));
    const short2v bright = short2v{0, 0} - pk_max(pk_min(dS, dN), pk_min(dE, dW));
    const short2v q = pk_max(dark, bright);
    return __builtin_bit_cast(unsigned, t - q) & 0x80008000u;  // t - q < 0  <=>  q > t
}

// grid = (ceil(totalCells / 4), frames), block = 256: ONE WAVE PER CELL, four independent cells per
// work-group and no work-group barrier anywhere (a 30-px cell is ~1000 pixels: one wave's worth of
// work; four waves per cell spent more scalar and barrier time than vector time).  Counters live
// in SGPRs (ballot + popcount), LDS is carved per wave and sized by the geometry's largest cell.
// Candidates of a cell are written in FAST's order (row-major) to the cell's own slot range:
// packed x | y << 12 | score << 24, coordinates relative to (minBorderX, minBorderY) as the
// reference hands them to DistributeOctTree.
// Inclusive prefix sum over the 64 lanes of a wave in six DPP adds (row_shr 1/2/4/8, row_bcast 15/31).
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xe, false);  // row_shr:4, banks 1-3
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xc, false);  // row_shr:8, banks 2-3
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

#ifndef AMOS_FAST_CAND_CAP
#define AMOS_FAST_CAND_CAP 1024
#endif
// candidate list (uint16 entries); one phase-1 iteration appends up to 512, so the list is flushed (scored, forgotten: phase 3 then scans
// the arc map instead) once it holds more than kFastCandCap - 512 with iterations to go.  A 30 x 30-pixel cell has 75 - 150 candidates.
// The list shares its LDS with the overflow path's `kept` list (written only after the last candidate has its arc value): LDS per wave
// decides how many cells a CU works on at a time (5 work-groups of four at 640 x 480, was 4 with separate 2 KB + 1.3 KB lists).  1 024 entries:
// with 768 a 640 x 480 frame runs the same (six work-groups) but the 59-pixel cells of 1920 x 1080 / 12 levels overflow the list more often
// and fall back to scanning the arc map: 34.7 k against 37.7 k frames/s on BASELINE configs[4] (tools/r4_c5.sh).
constexpr int kFastCandCap = AMOS_FAST_CAND_CAP;
__host__ __device__ constexpr size_t fast_list_bytes(int keptCap) { return (size_t)kFastCandCap * 2 > (size_t)keptCap * 4 ? (size_t)kFastCandCap * 2 : (size_t)keptCap * 4; }

// Exact arc value of up to 128 survivors, TWO PER LANE (A in the low halves of the packed registers, B in the high halves; the `count`
// survivors are split in two halves of (count + 1) / 2 so that a short list keeps few lanes busy: the 17 scattered byte reads of a
// candidate are what the LDS spends its conflict cycles on), on the raw pixel values and in ONE polarity per candidate:
//   dark:    A = v - min_arcs max_{k in arc} p_k          bright:  A = max_arcs min_{k in arc} p_k - v  =  v' - min_arcs max p'_k  with x' = 255 - x,
// so a bright candidate's 17 bytes are complemented (one xor with 0x00ff per half) and both kinds run the same unsigned packed max-ladder
// (80 packed operations for two candidates instead of 160 for both polarities).  A 9-arc of one polarity excludes one of the other (9 + 9 >
// 16), so the polarity that can be a corner is the one whose compass test passed (phase 1's test, recomputed here from the four compass
// pixels already in registers): bright iff the bright compass score is the larger one.  A candidate whose BOTH compass scores exceed t
// (a saddle: 0.2 % of the survivors at threshold 20, 1.3 % at 7) may be a corner in either polarity: when a chunk holds one (wave-uniform
// ballot) the ladder runs a second time with every lane's polarity flipped and the maximum of the two passes is kept -- exact for every
// candidate: the polarity that failed the compass test has A <= t, so it changes neither "A > t" nor the value of an A > t.
// (A one-candidate-per-lane ladder on 16-bit scalars -- v_max_u16 / v_xor_b32 issue at twice the packed instructions' rate,
// profiles/r04_valu_issue.md -- was measured too: 13 % more instructions, the same kernel time.)
template <int kStride>
__device__ __forceinline__ void fast_score_chunk2(const uint8_t *tile, uint8_t *amap, const uint16_t *cand, int first, int count, int lane, int t)
{
#if AMOS_FAST_EXP == 1
    return;
#endif
    typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
    const int half = (count + 1) >> 1;  // wave-uniform, <= 64
    const bool hasA = lane < half, hasB = lane + half < count;
    const int pA = cand[first + (hasA ? lane : 0)], pB = hasB ? cand[first + half + lane] : pA;
    constexpr int s = kStride;
    const uint8_t *cA = tile + (pA >> 6) * s + (pA & 63), *cB = tile + (pB >> 6) * s + (pB & 63);
    ushort2v x[16];
#define AMOS_FAST_P(k, off) x[k] = __builtin_bit_cast(ushort2v, (unsigned)cA[off] | ((unsigned)cB[off] << 16));
    AMOS_FAST_P(0, 3 * s) AMOS_FAST_P(1, 3 * s + 1) AMOS_FAST_P(2, 2 * s + 2) AMOS_FAST_P(3, s + 3)
    AMOS_FAST_P(4, 3) AMOS_FAST_P(5, -s + 3) AMOS_FAST_P(6, -2 * s + 2) AMOS_FAST_P(7, -3 * s + 1)
    AMOS_FAST_P(8, -3 * s) AMOS_FAST_P(9, -3 * s - 1) AMOS_FAST_P(10, -2 * s - 2) AMOS_FAST_P(11, -s - 3)
    AMOS_FAST_P(12, -3) AMOS_FAST_P(13, s - 3) AMOS_FAST_P(14, 2 * s - 2) AMOS_FAST_P(15, 3 * s - 1)
#undef AMOS_FAST_P
    const ushort2v vv = __builtin_bit_cast(ushort2v, (unsigned)cA[0] | ((unsigned)cB[0] << 16));
    // compass scores of both polarities (circle pixels 0 / 8 and 4 / 12), as in phase 1
    const ushort2v lo = __builtin_elementwise_max(__builtin_elementwise_min(x[0], x[8]), __builtin_elementwise_min(x[4], x[12]));
    const ushort2v hi = __builtin_elementwise_min(__builtin_elementwise_max(x[0], x[8]), __builtin_elementwise_max(x[4], x[12]));
    const ushort2v qd = __builtin_elementwise_sub_sat(vv, lo), qb = __builtin_elementwise_sub_sat(hi, vv);
    const ushort2v t2 = __builtin_bit_cast(ushort2v, (unsigned)t * 0x00010001u);
    const bool saddle = __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_elementwise_min(qd, qb), t2)) != 0u;
    // 0x00ff in the halves whose candidate is scored as bright: qb > qd  <=>  the 16-bit difference qd - qb is negative
    typedef short short2s __attribute__((ext_vector_type(2)));
    unsigned flip = __builtin_bit_cast(unsigned, (__builtin_bit_cast(short2s, qd) - __builtin_bit_cast(short2s, qb)) >> short2s{15, 15}) & 0x00ff00ffu;
    ushort2v best = ushort2v{0, 0};
    for (int pass = 0;; pass++) {   // one trip; two when the chunk holds a saddle candidate (wave-uniform)
        const ushort2v fv = __builtin_bit_cast(ushort2v, flip);
        ushort2v y[16], hi2[16], hi4[16], arc[16];
#pragma unroll
        for (int k = 0; k < 16; k++) y[k] = x[k] ^ fv;
#pragma unroll
        for (int k = 0; k < 16; k++) hi2[k] = __builtin_elementwise_max(y[k], y[(k + 1) & 15]);
#pragma unroll
        for (int k = 0; k < 16; k++) hi4[k] = __builtin_elementwise_max(hi2[k], hi2[(k + 2) & 15]);
#pragma unroll
        for (int k = 0; k < 16; k++)  // 9 contiguous = two groups of four + the ninth
            arc[k] = __builtin_elementwise_max(__builtin_elementwise_max(hi4[k], hi4[(k + 4) & 15]), y[(k + 8) & 15]);
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1)  // minimum over the 16 arcs as a tree (no serial chain of dependent packed operations)
#pragma unroll
            for (int k = 0; k < w; k++) arc[k] = __builtin_elementwise_min(arc[k], arc[k + w]);
        best = __builtin_elementwise_max(best, __builtin_elementwise_sub_sat(vv ^ fv, arc[0]));
        if (pass == 1 || __ballot(saddle) == 0ull) break;
        flip ^= 0x00ff00ffu;
    }
    const int aA = best.x, aB = best.y;
    if (hasA) amap[((pA >> 6) + 1) * kFastMapStride + (pA & 63) + 1] = (uint8_t)(aA > t ? aA : 0);
    if (hasB) amap[((pB >> 6) + 1) * kFastMapStride + (pB & 63) + 1] = (uint8_t)(aB > t ? aB : 0);
}

#ifndef AMOS_FAST_CELLS_PER_GROUP
#define AMOS_FAST_CELLS_PER_GROUP 4
#endif
constexpr int kFastCellsPerGroup = AMOS_FAST_CELLS_PER_GROUP;  // waves (= cells) per work-group

// Compass flags of the two pixels a dword-set holds in the HIGH bytes of its 16-bit halves (the low bytes may
// hold anything -- "noise"): with lo = max(min(n,s), min(e,w)), hi = min(max(n,s), max(e,w)) as 16-bit values,
//   cd = centre << 8 (low byte 0x00):  sat(cd - lo) = 256 (c - lo_hi) - lo_noise  > 256 t  <=>  c - lo_hi > t
//   cb = centre << 8 | 0xff:           sat(hi - cb) = 256 (hi_hi - c) - (255 - hi_noise) > 256 t  <=>  hi_hi - c > t
// (min / max of 16-bit values order by the high byte first, so lo / hi carry the right high bytes), i.e. the
// comparisons are EXACT although the neighbours are never masked.  q = max of the two is the pixel's compass
// score (scaled by 256): the test at ANY threshold is q > 256 t, which is what lets one sweep answer
// "is there a candidate at iniThFAST" and "could there be one at minThFAST" together.
// Returns 1 in bit 0 / bit 16 where q > t2 (t2 = threshold << 8 in both halves).
template <bool kTrack>
__device__ __forceinline__ unsigned fast_flags(unsigned cd, unsigned cb, unsigned n, unsigned s, unsigned e, unsigned w, unsigned t2, unsigned &qmax)
{
    typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
    const ushort2v n2 = __builtin_bit_cast(ushort2v, n), s2 = __builtin_bit_cast(ushort2v, s), e2 = __builtin_bit_cast(ushort2v, e),
                   w2 = __builtin_bit_cast(ushort2v, w);
    const ushort2v lo = __builtin_elementwise_max(__builtin_elementwise_min(n2, s2), __builtin_elementwise_min(e2, w2));
    const ushort2v hi = __builtin_elementwise_min(__builtin_elementwise_max(n2, s2), __builtin_elementwise_max(e2, w2));
    const ushort2v q = __builtin_elementwise_max(__builtin_elementwise_sub_sat(__builtin_bit_cast(ushort2v, cd), lo),
                                                 __builtin_elementwise_sub_sat(hi, __builtin_bit_cast(ushort2v, cb)));
    if (kTrack) qmax = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(ushort2v, qmax), q));
    const ushort2v f = __builtin_elementwise_sub_sat(q, __builtin_bit_cast(ushort2v, t2));
    unsigned r;  // 1 in bit 0 / bit 16 where the half is non-zero (the compiler turns min(f, 1) into compares)
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(__builtin_bit_cast(unsigned, f)), "v"(0x00010001u));
    return r;
}

// kRowDw: dwords per LDS tile row, a compile-time constant (16 for cells up to 41 px wide, 20 up to 59) so that
// every circle / neighbour offset of phases 1 and 2 is an immediate of the LDS instruction.
template <int kRowDw>
__global__ __launch_bounds__(64 * kFastCellsPerGroup) void k_fast_cells(const uint8_t *__restrict__ pyr, const Geom *__restrict__ g,
                                                   const Cell *__restrict__ cells, int *__restrict__ slotCount,
                                                   uint32_t *__restrict__ slots, int nFrames)
{
    extern __shared__ __align__(16) unsigned char fast_smem[];
    // the wave index is wave-uniform: tell the compiler, so that the cell record, loop bounds and LDS
    // bases live in SGPRs and the per-cell loops are scalar loops
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int frame, chunk;
    if (!xcd_frame_chunk(blockIdx.x, (g->totalCells + kFastCellsPerGroup - 1) / kFastCellsPerGroup, nFrames, frame, chunk)) return;
    const int cellIdx = chunk * kFastCellsPerGroup + wave;
    if (cellIdx >= g->totalCells) return;  // wave-uniform; no work-group barrier below
    const Cell c = cells[cellIdx];
    const LevelGeom &lg = g->lv[c.level];
    constexpr int rowDw = kRowDw, tileStride = kRowDw * 4;
    unsigned char *base = fast_smem + (size_t)wave * g->fastWaveBytes;
    uint32_t *tile32 = reinterpret_cast<uint32_t *>(base);
    uint8_t *amap = base + (((size_t)g->fastTileRows * tileStride + 15) & ~(size_t)15);
    uint16_t *cand = reinterpret_cast<uint16_t *>(amap + (size_t)g->fastMapRows * kFastMapStride);
    uint32_t *kept = reinterpret_cast<uint32_t *>(cand);  // overflow path only, after the candidate list's last use (fast_list_bytes)
    const uint8_t *img = level_origin(pyr, g, frame, c.level);
    const int tw = c.tw, th = c.th;
    // LDS row r holds image row y0 - 3 + r; LDS byte b of a row holds image column x0 - 4 + b.  Rows are
    // fetched as unaligned 16-byte pieces (global memory takes any byte alignment) and land 16-byte aligned.
    const int npc = c.ndw, lh = th + 6;  // pieces per row
    const uint8_t *corner = img + (ptrdiff_t)(c.y0 - 3) * lg.stride + (c.x0 - 4);
    for (int idx = lane; idx < lh * npc; idx += 64) {
        const int r = div_small(idx, c.magicDw), cc = idx - r * npc;
        uint4 v;
        __builtin_memcpy(&v, corner + (ptrdiff_t)r * lg.stride + 16 * cc, 16);
        reinterpret_cast<uint4 *>(tile32 + r * rowDw)[cc] = v;
    }
    for (int idx = lane; idx < (th + 2) * (kFastMapStride / 16); idx += 64) reinterpret_cast<uint4 *>(amap)[idx] = uint4{0, 0, 0, 0};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (AMOS_FAST_EXP == 4) {  // (the compare keeps the staged tile alive)
        if (lane == 0 || tile32[lane] == 0xdeadbeefu) slotCount[(size_t)frame * g->totalCells + cellIdx] = 0;
        return;
    }
    const uint8_t *tile = reinterpret_cast<const uint8_t *>(tile32) + 3 * tileStride + 4;  // pixel (x0, y0)
    const int iniTh = min(max(g->iniTh, 0), 255), minTh = min(max(g->minTh, 0), 255);
    const int groups = c.groups, nitems = th * groups;
    const unsigned long long ltmask = (1ull << lane) - 1ull;
    uint32_t *out = slots + (size_t)frame * g->slotTotal + c.slotOff;
    const int xBase = c.x0 - kMinBorder, yBase = c.y0 - kMinBorder;
    // pixel k of a group <-> bit {0, 1, 16, 17, 2, 3, 18, 19}[k]; pixels beyond tw in the row's last group are masked
    const int nLast = tw - 8 * (groups - 1);  // 1 .. 8, wave-uniform
    const unsigned lastMask = ((1u << (min(nLast, 2) + min(max(nLast - 4, 0), 2))) - 1u) |
                              (((1u << (min(max(nLast - 2, 0), 2) + min(max(nLast - 6, 0), 2))) - 1u) << 16);
    int nkept = 0;
    bool anyLower = true;  // wave-uniform: some pixel's compass score exceeds minThFAST (known after the first sweep)

    // One cv::FAST(cell, t, nonmax) -- ORBextractor.cc:1126 / :1135.  kFirst: the sweep at iniThFAST, which also
    // records whether ANY pixel could pass the compass test at minThFAST: a cell that comes back empty and has
    // no such pixel needs no second sweep (the test is monotone in t) -- flat regions finish after one pass.
    auto sweep = [&](const int t, auto first) {
        constexpr bool kFirst = decltype(first)::value;
        // ---- phase 1 (necessary test, 8 pixels per lane) feeding phase 2 (exact arc value) in dense chunks.
        // Odd pixels of a dword sit in the high bytes of its 16-bit halves as loaded; even pixels are brought
        // there by taking the dword one byte to the left (for the west / east neighbours that is just another
        // alignment of the same row data), and fast_flags compares 16-bit halves whose low bytes are noise.
        const unsigned t2 = ((unsigned)t * 0x00010001u) << 8;
        unsigned qmax = 0;
        int ncand = 0, done = 0;   // list entries [0, done) have their arc value
        bool overflow = false;     // wave-uniform: the list wrapped, phase 3 must scan the arc map instead
        for (int item0 = 0; item0 < nitems; item0 += 64) {
            const int item = item0 + lane;
            unsigned bits = 0;
            int p0 = 0;
            if (item < nitems) {
                const int y = div_small(item, c.magicG), gx = item - y * groups;
                const uint32_t *row = tile32 + (y + 3) * rowDw + 2 * gx;  // dword holding pixels x0 + 8gx - 4 ..
                const unsigned L = row[0], C0 = row[1], C1 = row[2], R = row[3];
                const unsigned N0 = row[1 - 3 * rowDw], N1 = row[2 - 3 * rowDw], S0 = row[1 + 3 * rowDw], S1 = row[2 + 3 * rowDw];
                // odd pixels (x + 1, x + 3 of each dword): west = x - 3 .., east = x + 3 ..
                const unsigned oW0 = __builtin_amdgcn_alignbyte(C0, L, 1u), oW1 = __builtin_amdgcn_alignbyte(C1, C0, 1u);
                const unsigned oE0 = __builtin_amdgcn_alignbyte(C1, C0, 3u), oE1 = __builtin_amdgcn_alignbyte(R, C1, 3u);
                // even pixels, everything one byte to the left: west = the previous dword as it is, east = alignment 2
                const unsigned eE0 = __builtin_amdgcn_alignbyte(C1, C0, 2u), eE1 = __builtin_amdgcn_alignbyte(R, C1, 2u);
                const unsigned bO0 = fast_flags<kFirst>(C0 & 0xff00ff00u, C0 | 0x00ff00ffu, N0, S0, oE0, oW0, t2, qmax);
                const unsigned bO1 = fast_flags<kFirst>(C1 & 0xff00ff00u, C1 | 0x00ff00ffu, N1, S1, oE1, oW1, t2, qmax);
                const unsigned bE0 = fast_flags<kFirst>(__builtin_amdgcn_perm(0u, C0, 0x020c000cu), __builtin_amdgcn_perm(0u, C0, 0x020d000du),
                                                        N0 << 8, S0 << 8, eE0, L, t2, qmax);
                const unsigned bE1 = fast_flags<kFirst>(__builtin_amdgcn_perm(0u, C1, 0x020c000cu), __builtin_amdgcn_perm(0u, C1, 0x020d000du),
                                                        N1 << 8, S1 << 8, eE1, C0, t2, qmax);
                bits = bE0 | (bO0 << 1) | (bE1 << 2) | (bO1 << 3);
                if (gx == groups - 1) bits &= lastMask;
                p0 = (y << 6) | (8 * gx);
            }
            if (ncand + 512 > kFastCandCap) {  // wave-uniform: make room (one iteration adds <= 64 x 8), remember that the list is no longer complete
                if (ncand > done) fast_score_chunk2<tileStride>(tile, amap, cand, done, ncand - done, lane, t);  // < 128 left over
                ncand = done = 0;
                overflow = true;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            {   // append the survivors: one wave-wide prefix sum of the per-lane counts, then up to 8 stores
                if (AMOS_FAST_EXP == 3) bits = 0;
                const int cnt = __popc(bits);
                const int incl = wave_inclusive_scan(cnt);
                int o = ncand + incl - cnt;
                if (bits & 0x00001u) cand[o++] = (uint16_t)p0;
                if (bits & 0x00002u) cand[o++] = (uint16_t)(p0 + 1);
                if (bits & 0x10000u) cand[o++] = (uint16_t)(p0 + 2);
                if (bits & 0x20000u) cand[o++] = (uint16_t)(p0 + 3);
                if (bits & 0x00004u) cand[o++] = (uint16_t)(p0 + 4);
                if (bits & 0x00008u) cand[o++] = (uint16_t)(p0 + 5);
                if (bits & 0x40000u) cand[o++] = (uint16_t)(p0 + 6);
                if (bits & 0x80000u) cand[o] = (uint16_t)(p0 + 7);
                ncand += __builtin_amdgcn_readlane(incl, 63);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            while (ncand - done >= 128) {  // wave-uniform: dense chunks of 128 survivors, two per lane
                fast_score_chunk2<tileStride>(tile, amap, cand, done, 128, lane, t);
                done += 128;
            }
        }
        if (kFirst) {  // compass scores are scaled by 256; pixels outside the cell may be counted (costs a sweep, never a corner)
            const unsigned m2 = ((unsigned)minTh * 0x00010001u) << 8;
            anyLower = __ballot(((qmax & 0xffffu) > (m2 & 0xffffu)) || ((qmax >> 16) > (m2 >> 16))) != 0ull;
        }
        if (ncand > done) fast_score_chunk2<tileStride>(tile, amap, cand, done, ncand - done, lane, t);  // wave-uniform
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- phase 3: strict 3x3 NMS.  For a corner at threshold t (a > t) every non-corner neighbour is
        // smaller than a anyway: keep <=> a > all 8 neighbours.  Over the survivor list, or (list overflowed)
        // over the non-zero entries of the arc map.
        if (AMOS_FAST_EXP == 2) return;
        if (!overflow) {
            // The list is in item order = row-major pixel order and the ordered ballots below keep that order, which
            // is FAST's output order: the kept corners go straight to their final slots.
            for (int i0 = 0; i0 < ncand; i0 += 64) {
                const int i = i0 + lane;
                bool keep = false;
                int p = 0, a = 0;
                if (i < ncand) {
                    p = cand[i];
                    const uint8_t *m = amap + ((p >> 6) + 1) * kFastMapStride + (p & 63) + 1;
                    a = m[0];
                    if (a) {
                        constexpr int s = kFastMapStride;
                        const int nmax = max(max(max((int)m[-s - 1], (int)m[-s]), max((int)m[-s + 1], (int)m[-1])),
                                             max(max((int)m[1], (int)m[s - 1]), max((int)m[s], (int)m[s + 1])));
                        keep = nmax < a;
                    }
                }
                const unsigned long long bal = __ballot(keep);
                if (keep) out[nkept + __popcll(bal & ltmask)] = (uint32_t)(xBase + (p & 63)) | ((uint32_t)(yBase + (p >> 6)) << 12) | ((uint32_t)(a - 1) << 24);
                nkept += __popcll(bal);
            }
        } else {
            const uint32_t *amap32 = reinterpret_cast<const uint32_t *>(amap);
            for (int idx0 = 0; idx0 < th * 16; idx0 += 64) {
                const int idx = idx0 + lane;
                const int y = idx >> 4, dw = idx & 15;
                const unsigned v = idx < th * 16 ? amap32[(y + 1) * 16 + dw] : 0u;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int a = (v >> (8 * k)) & 0xff;
                    bool keep = false;
                    if (a) {
                        const uint8_t *m = amap + (y + 1) * kFastMapStride + 4 * dw + k;
                        constexpr int s = kFastMapStride;
                        const int nmax = max(max(max((int)m[-s - 1], (int)m[-s]), max((int)m[-s + 1], (int)m[-1])),
                                             max(max((int)m[1], (int)m[s - 1]), max((int)m[s], (int)m[s + 1])));
                        keep = nmax < a;
                    }
                    const unsigned long long bal = __ballot(keep);
                    if (keep) kept[nkept + __popcll(bal & ltmask)] = ((uint32_t)((y << 6) | (4 * dw + k - 1)) << 8) | (uint32_t)a;
                    nkept += __popcll(bal);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // the map scan visits (64-entry block, byte k, lane): rank by pixel index = row-major order
            for (int i = lane; i < nkept; i += 64) {
                const uint32_t me = kept[i];
                int rank = 0;
                for (int j = 0; j < nkept; j++) rank += kept[j] < me;
                const int p = (int)(me >> 8), a = (int)(me & 0xff);
                out[rank] = (uint32_t)(xBase + (p & 63)) | ((uint32_t)(yBase + (p >> 6)) << 12) | ((uint32_t)(a - 1) << 24);
            }
        }
    };
    sweep(iniTh, std::true_type{});
    // a second cv::FAST at a higher (or equal) threshold finds nothing new; neither does one on a cell without
    // a single pixel whose compass score exceeds minThFAST
    if (AMOS_FAST_EXP == 0 && nkept == 0 && minTh < iniTh && anyLower) sweep(minTh, std::false_type{});
    if (lane == 0) slotCount[(size_t)frame * g->totalCells + cellIdx] = nkept;
}

// ---------------------------------------------------------------------------------------------
// a4 + a5  ORBextractor::DistributeOctTree (ORBextractor.cc:706-1049) and the fix-up at :1175-1190.
//
// One work-group per (frame, level).  The reference walks a std::list and moves keypoints between
// per-node vectors; here every candidate keeps an index into the CURRENT node list and a pass over
// the candidates counts children per quadrant with LDS atomics.  The list order the reference
// produces is reproduced analytically:
//   * a full pass replaces every multi-point node by its non-empty children, push_front'ed in the
//     order n1..n4 while walking front to back  =>  new list = reverse(creation order) ++ the old
//     single-point nodes in their old order;
//   * the final passes (ORBextractor.cc:936-1020) divide nodes in descending (count, address) order
//     until the list holds N nodes; heap addresses are replaced by creation order.
// The result keeps, per node, the candidate of maximal response, first in input order on ties.
struct OctNode {
    short x0, y0, x1, y1;
};

// Exclusive scan of a[0..n) in LDS by the whole 256-thread work-group; returns the total.
// Per-thread chunk sums, one wave-level shuffle scan, four wave totals through LDS: two barriers.
// The caller has a barrier between its writes to a[] and this call.
__device__ __forceinline__ int block_excl_scan(int *a, int n, int *part)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = (n + 255) >> 8;
    const int b = min(tid * chunk, n), e = min(b + chunk, n);
    int s = 0;
    for (int i = b; i < e; i++) s += a[i];
    int incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    const int p0 = part[0], p1 = part[1], p2 = part[2], p3 = part[3];
    int run = incl - s + (wave > 0 ? p0 : 0) + (wave > 1 ? p1 : 0) + (wave > 2 ? p2 : 0);
    for (int i = b; i < e; i++) {
        const int t = a[i];
        a[i] = run;
        run += t;
    }
    __syncthreads();
    return p0 + p1 + p2 + p3;
}

__device__ __forceinline__ int oct_quadrant(const OctNode &b, int x, int y, int &midx, int &midy)
{
    midx = b.x0 + ((b.x1 - b.x0 + 1) >> 1);  // ceil(float(UR.x-UL.x)/2), ORBextractor.cc:641
    midy = b.y0 + ((b.y1 - b.y0 + 1) >> 1);
    return (x < midx) ? ((y < midy) ? 0 : 2) : ((y < midy) ? 1 : 3);
}

__device__ __forceinline__ OctNode oct_child(const OctNode &b, int q, int midx, int midy)
{
    OctNode c;
    c.x0 = (q & 1) ? midx : b.x0;
    c.x1 = (q & 1) ? b.x1 : midx;
    c.y0 = (q & 2) ? midy : b.y0;
    c.y1 = (q & 2) ? b.y1 : midy;
    return c;
}

// Double-buffered node arrays are addressed as base + buffer * NC (a runtime-indexed array of pointers
// would live in scratch memory).
template <typename T>
struct OctBuf2 {
    T *base;
    int stride;
    __device__ __forceinline__ T *operator[](int b) const { return base + b * stride; }
};

struct OctLds {
    OctBuf2<int> cnt;
    OctBuf2<int> seq;
    OctBuf2<OctNode> box;
    OctBuf2<uint8_t> noMore;
    int *child;   // [4*NC] child counts, then child positions in the new list
    int *newPos;  // [NC]
    int *scanA;   // [SC]
    int *scanB;   // [NC]
    int *order;   // [NC]
    unsigned *best;
    int *part;    // [256]
    int *vars;    // [16]
};

__host__ __device__ inline size_t oct_lds_bytes(int NC, int SC)
{
    return (size_t)NC * (2 * 4 + 2 * 4 + 2 * 8 + 4 * 4 + 4 + 4 + 4 + 4) + (size_t)((2 * NC + 15) & ~15) +
           (size_t)SC * 4 + 256 * 4 + 16 * 4 + 64;
}

__device__ inline OctLds oct_carve(unsigned char *base, int NC, int SC)
{
    OctLds L;
    int *ip = reinterpret_cast<int *>(base);
    L.cnt.base = ip; L.cnt.stride = NC; ip += 2 * NC;
    L.seq.base = ip; L.seq.stride = NC; ip += 2 * NC;
    L.box.base = reinterpret_cast<OctNode *>(ip); L.box.stride = NC; ip += 4 * NC;
    L.child = ip; ip += 4 * NC;
    L.newPos = ip; ip += NC;
    L.scanA = ip; ip += SC;
    L.scanB = ip; ip += NC;
    L.order = ip; ip += NC;
    L.best = reinterpret_cast<unsigned *>(ip); ip += NC;
    L.part = ip; ip += 256;
    L.vars = ip; ip += 16;
    uint8_t *bp = reinterpret_cast<uint8_t *>(ip);
    L.noMore.base = bp; L.noMore.stride = NC;
    return L;
}

// vars[] slots
enum { V_SIZE = 0, V_NEXPAND = 1, V_RSTAR = 2, V_NCAND = 3 };

// grid = frames * nLevels, block = 256, dynamic LDS = oct_lds_bytes(NC, SC).
//
// Candidate state (packed point, node index, quadrant) lives in REGISTERS when the level has at most
// 256 * kOctRegPts candidates (thread t owns candidates t, t + 256, ...): the subdivision passes then
// touch only LDS.  Larger levels run the same code with the state in global scratch.
constexpr int kOctRegPts = 8;

template <bool kRegs>
__device__ __forceinline__ void octree_body(const OctLds &L, const LevelGeom &lg, int level, int n, uint32_t *__restrict__ P,
                                            uint16_t *__restrict__ nodeIdx, uint8_t *__restrict__ quad,
                                            amos_keypoint *__restrict__ out, int *__restrict__ lvCountOut)
{
    const int tid = threadIdx.x;
    const int N = lg.quota;
    uint32_t rp[kOctRegPts];
    int rnode[kOctRegPts], rquad[kOctRegPts];
#define OCT_POINTS(k, i) _Pragma("unroll") for (int k = 0, i = tid; k < (kRegs ? kOctRegPts : (n + 255) / 256); k++, i += 256) if (i < n)
#define OCT_P(k, i) (kRegs ? rp[k] : P[i])
#define OCT_NODE(k, i) (kRegs ? rnode[k] : (int)nodeIdx[i])
#define OCT_SET_NODE(k, i, v) do { if (kRegs) rnode[k] = (v); else nodeIdx[i] = (uint16_t)(v); } while (0)
#define OCT_QUAD(k, i) (kRegs ? rquad[k] : (int)quad[i])
#define OCT_SET_QUAD(k, i, v) do { if (kRegs) rquad[k] = (v); else quad[i] = (uint8_t)(v); } while (0)
    if (kRegs) {
        OCT_POINTS(k, i) rp[k] = P[i];
    }

    // ---- 2. root nodes, ORBextractor.cc:718-786
    const int spanX = lg.maxBX - kMinBorder, spanY = lg.maxBY - kMinBorder;
    const int nIni = lg.nIni;
    const float hX = __fdiv_rn((float)spanX, (float)nIni);
    for (int i = tid; i < nIni; i += 256) L.cnt[0][i] = 0;
    __syncthreads();
    OCT_POINTS(k, i) {
        const int x = OCT_P(k, i) & 0xfff;
        int idx = (int)__fdiv_rn((float)x, hX);
        idx = min(max(idx, 0), nIni - 1);
        OCT_SET_NODE(k, i, idx);
        atomicAdd(&L.cnt[0][idx], 1);
    }
    __syncthreads();
    if (tid == 0) {
        int size = 0;
        for (int i = 0; i < nIni; i++) {
            const int c = L.cnt[0][i];
            L.newPos[i] = size;
            if (c > 0) {
                OctNode b;
                b.x0 = (short)(int)__fmul_rn(hX, (float)i);
                b.x1 = (short)(int)__fmul_rn(hX, (float)(i + 1));
                b.y0 = 0;
                b.y1 = (short)spanY;
                L.box[1][size] = b;
                L.cnt[1][size] = c;
                L.seq[1][size] = i;
                L.noMore[1][size] = (c == 1);
                size++;
            }
        }
        L.vars[V_SIZE] = size;
    }
    __syncthreads();
    OCT_POINTS(k, i) OCT_SET_NODE(k, i, L.newPos[OCT_NODE(k, i)]);
    int cur = 1;
    int size = L.vars[V_SIZE];
    int seqBase = nIni;
    __syncthreads();

    // ---- 3. subdivision, ORBextractor.cc:800-1021
    bool finalPhase = false;
    for (;;) {
        const int prevSize = size;
        const int nxt = cur ^ 1;
        // children per quadrant of every multi-point node
        for (int i = tid; i < 4 * size; i += 256) L.child[i] = 0;
        if (tid == 0) { L.vars[V_NEXPAND] = 0; L.vars[V_RSTAR] = 0x7fffffff; L.vars[V_NCAND] = 0; }
        __syncthreads();
        OCT_POINTS(k, i) {
            const int nd = OCT_NODE(k, i);
            if (!L.noMore[cur][nd]) {
                const uint32_t p = OCT_P(k, i);
                int mx, my;
                const int q = oct_quadrant(L.box[cur][nd], p & 0xfff, (p >> 12) & 0xfff, mx, my);
                OCT_SET_QUAD(k, i, q);
                atomicAdd(&L.child[nd * 4 + q], 1);
            }
        }
        __syncthreads();

        int nDiv;  // how many candidates get divided in this pass, in processing order
        if (!finalPhase) {
            // processing order = list order; every multi-point node is divided
            for (int i = tid; i < size; i += 256) L.order[i] = i;
            nDiv = size;  // (single-point nodes contribute no children)
            __syncthreads();
        } else {
            // candidates = multi-point nodes, processed in descending (count, creation order)
            for (int i = tid; i < size; i += 256) {
                if (!L.noMore[cur][i]) {
                    const int ci = L.cnt[cur][i], si = L.seq[cur][i];
                    int rank = 0;
                    for (int j = 0; j < size; j++) {
                        if (L.noMore[cur][j]) continue;
                        const int cj = L.cnt[cur][j];
                        rank += (cj > ci) || (cj == ci && L.seq[cur][j] > si);
                    }
                    L.order[rank] = i;
                    atomicAdd(&L.vars[V_NCAND], 1);
                }
            }
            __syncthreads();
            const int m = L.vars[V_NCAND];
            // growth of the list along the processing order; stop right after reaching N (:1003)
            for (int r = tid; r < m; r += 256) {
                const int *ch = &L.child[L.order[r] * 4];
                L.scanA[r] = (ch[0] > 0) + (ch[1] > 0) + (ch[2] > 0) + (ch[3] > 0) - 1;
            }
            __syncthreads();
            for (int r = tid; r < m; r += 256) L.scanB[r] = L.scanA[r];
            __syncthreads();
            block_excl_scan(L.scanA, m, L.part);
            for (int r = tid; r < m; r += 256)
                if (prevSize + L.scanA[r] + L.scanB[r] >= N) atomicMin(&L.vars[V_RSTAR], r);
            __syncthreads();
            const int rstar = L.vars[V_RSTAR];
            nDiv = rstar == 0x7fffffff ? m : rstar + 1;
        }

        // creation order of the children along the processing order
        for (int r = tid; r < nDiv; r += 256) {
            const int i = L.order[r];
            int nz = 0;
            if (!L.noMore[cur][i]) {
                const int *ch = &L.child[i * 4];
                nz = (ch[0] > 0) + (ch[1] > 0) + (ch[2] > 0) + (ch[3] > 0);
            }
            L.scanA[r] = nz;
        }
        // survivors (not divided) keep their relative order behind the new children
        for (int i = tid; i < size; i += 256) L.scanB[i] = 1;
        __syncthreads();
        for (int r = tid; r < nDiv; r += 256)
            if (!L.noMore[cur][L.order[r]]) L.scanB[L.order[r]] = 0;
        __syncthreads();
        const int K = block_excl_scan(L.scanA, nDiv, L.part);
        const int M = block_excl_scan(L.scanB, size, L.part);
        // build the new list
        for (int i = tid; i < size; i += 256) {
            // scanB[i] is exclusive; a survivor is one whose own flag was 1
            const bool survivor = (i + 1 < size ? L.scanB[i + 1] : M) != L.scanB[i];
            if (survivor) {
                const int p = K + L.scanB[i];
                L.box[nxt][p] = L.box[cur][i];
                L.cnt[nxt][p] = L.cnt[cur][i];
                L.seq[nxt][p] = L.seq[cur][i];
                L.noMore[nxt][p] = L.noMore[cur][i];
                L.newPos[i] = p;
            } else {
                L.newPos[i] = -1;
            }
        }
        for (int r = tid; r < nDiv; r += 256) {
            const int i = L.order[r];
            if (L.noMore[cur][i]) continue;
            int ci = L.scanA[r];
            const OctNode b = L.box[cur][i];
            int mx, my;
            oct_quadrant(b, 0, 0, mx, my);
            const int c0 = L.child[i * 4], c1 = L.child[i * 4 + 1], c2 = L.child[i * 4 + 2], c3 = L.child[i * 4 + 3];
            int nex = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int cq = q == 0 ? c0 : q == 1 ? c1 : q == 2 ? c2 : c3;
                if (cq > 0) {
                    const int p = K - 1 - ci;
                    L.box[nxt][p] = oct_child(b, q, mx, my);
                    L.cnt[nxt][p] = cq;
                    L.seq[nxt][p] = seqBase + ci;
                    L.noMore[nxt][p] = (cq == 1);
                    L.child[i * 4 + q] = p;
                    nex += cq > 1;
                    ci++;
                }
            }
            if (nex) atomicAdd(&L.vars[V_NEXPAND], nex);
        }
        __syncthreads();
        OCT_POINTS(k, i) {
            const int nd = OCT_NODE(k, i);
            const int np = L.newPos[nd];
            OCT_SET_NODE(k, i, np >= 0 ? np : L.child[nd * 4 + OCT_QUAD(k, i)]);
        }
        const int nToExpand = L.vars[V_NEXPAND];
        size = K + M;
        seqBase += K;
        cur = nxt;
        __syncthreads();
        if (size >= N || size == prevSize) break;  // :917, :1014
        if (!finalPhase) {
            if (size + nToExpand * 3 > N) finalPhase = true;  // :936
        }
    }

    // ---- 4. best response per node (first in input order on ties), ORBextractor.cc:1024-1046,
    //         and the coordinate / octave / size fix-up of :1175-1190
    for (int i = tid; i < size; i += 256) L.best[i] = 0;
    __syncthreads();
    OCT_POINTS(k, i) {
        const unsigned key = ((OCT_P(k, i) >> 24) << 20) | (unsigned)(0xfffff - i);
        atomicMax(&L.best[OCT_NODE(k, i)], key);
    }
    __syncthreads();
    for (int i = tid; i < size; i += 256) {
        const uint32_t p = P[0xfffff - (L.best[i] & 0xfffff)];
        amos_keypoint kp;
        kp.x = (float)((int)(p & 0xfff) + kMinBorder);
        kp.y = (float)((int)((p >> 12) & 0xfff) + kMinBorder);
        kp.size = lg.patchSize;
        kp.angle = -1.f;
        kp.response = (float)(p >> 24);
        kp.octave = level;
        kp.class_id = -1;
        out[i] = kp;
    }
    if (tid == 0) *lvCountOut = size;
#undef OCT_POINTS
#undef OCT_P
#undef OCT_NODE
#undef OCT_SET_NODE
#undef OCT_QUAD
#undef OCT_SET_QUAD
}

__global__ __launch_bounds__(256) void k_octree(const Geom *__restrict__ g, const Cell *__restrict__ cells,
                                               const int *__restrict__ slotCount,
                                               const uint32_t *__restrict__ slots, uint32_t *__restrict__ pts,
                                               uint16_t *__restrict__ nodeOf, uint8_t *__restrict__ quadOf,
                                               int *__restrict__ candCount, amos_keypoint *__restrict__ lvKps,
                                               int *__restrict__ lvCount, int NC, int SC)
{
    AMOS_SET_LATENCY_PRIO();
    extern __shared__ __align__(16) unsigned char oct_smem[];
    const OctLds L = oct_carve(oct_smem, NC, SC);
    const int tid = threadIdx.x;
    const int frame = blockIdx.x / g->nLevels, level = blockIdx.x - frame * g->nLevels;
    const LevelGeom &lg = g->lv[level];

    // ---- 1. candidates of the level in the reference's order: cells row-major, FAST order inside.
    // A wave copies a run of cells: lanes cover the slots of 64 / 16 cells... simple form: each thread
    // copies the slots of its cells, the per-cell counts are small (a few per cell).
    const int *sc = slotCount + (size_t)frame * g->totalCells + lg.cellStart;
    for (int c = tid; c < lg.nCells; c += 256) L.scanA[c] = sc[c];
    __syncthreads();
    const int n = block_excl_scan(L.scanA, lg.nCells, L.part);
    uint32_t *P = pts + (size_t)frame * g->ptsTotal + lg.ptsOff;
    for (int c = tid; c < lg.nCells; c += 256) {
        const int cnt = sc[c], off = L.scanA[c];
        const uint32_t *s = slots + (size_t)frame * g->slotTotal + cells[lg.cellStart + c].slotOff;
        for (int k = 0; k < cnt; k++) P[off + k] = s[k];
    }
    if (tid == 0) candCount[frame * g->nLevels + level] = n;
    __threadfence_block();
    __syncthreads();  // P[] is complete for this work-group

    uint16_t *nodeIdx = nodeOf + (size_t)frame * g->ptsTotal + lg.ptsOff;
    uint8_t *quad = quadOf + (size_t)frame * g->ptsTotal + lg.ptsOff;
    amos_keypoint *out = lvKps + (size_t)frame * g->kpLevelTotal + lg.kpOff;
    int *cntOut = lvCount + frame * g->nLevels + level;
    if (n <= 256 * kOctRegPts)
        octree_body<true>(L, lg, level, n, P, nodeIdx, quad, out, cntOut);
    else
        octree_body<false>(L, lg, level, n, P, nodeIdx, quad, out, cntOut);
}

// ---------------------------------------------------------------------------------------------
// a6  computeOrientation / IC_Angle (ORBextractor.cc:108-161, 618-632) + cv::fastAtan2.
// One wave per keypoint; lanes 0-31 take row v, lanes 32-63 row v+1; integer moments are reduced
// across the wave with DPP-style shuffles, lane 0 evaluates the float32 polynomial.
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = __fmul_rn(0.9997878412794807f, scale), p3 = __fmul_rn(-0.3258083974640975f, scale);
    const float p5 = __fmul_rn(0.1555786518463281f, scale), p7 = __fmul_rn(-0.04432655554792128f, scale);
    const float eps = 2.2204460492503131e-16f;  // (float)DBL_EPSILON
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

__device__ __forceinline__ int level_of_slot(const Geom *g, int slot)
{
    int level = -1;
    for (int l = 0; l < g->nLevels; l++)
        if (slot >= g->lv[l].kpOff && slot < g->lv[l].kpOff + g->lv[l].nodeCap) level = l;
    return level;
}

// grid = xcd_grid(ceil(kpLevelTotal/16), frames), block = 256 = 16 keypoints: SIXTEEN LANES PER
// KEYPOINT.  The 31 x 31 patch is 62 half rows of 16 bytes (columns -15 .. 0 and +1 .. +16); lane j takes
// half (j & 1) of rows (j >> 1) + 8 q, q = 0..3, as four unaligned 16-byte loads issued together; a
// per-(row, half) table in LDS holds, per byte, the weight u + 16 inside the circular patch (0 outside)
// and a 0/1 mask, so a dword costs two v_dot4_u32_u8:
//   m10 = sum (u + 16) p - 16 sum p,   m01 = sum v p.
__device__ __forceinline__ int group16_sum(int v)
{
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 16);
    return v;
}

__global__ __launch_bounds__(256) void k_orient(const uint8_t *__restrict__ pyr, const Geom *__restrict__ g,
                                               amos_keypoint *__restrict__ lvKps, const int *__restrict__ lvCount, int nFrames)
{
    AMOS_SET_LATENCY_PRIO();
    // wt[row][half][0] = weights u + 16 of the 16 bytes of that half row (0 outside the circle),
    // wt[row][half][1] = 0/1 masks; row = v + 15, row 31 (v = 16) does not exist: all zero
    __shared__ uint4 wt[32][2][2];
    {
        const int row = threadIdx.x >> 3, half = (threadIdx.x >> 2) & 1, d = threadIdx.x & 3;
        const int v = row - kHalfPatch;
        const int um = v <= kHalfPatch ? c_umax[v < 0 ? -v : v] : -1;
        unsigned wu = 0, ones = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int u = -kHalfPatch + 16 * half + 4 * d + b;
            const int au = u < 0 ? -u : u;
            if (au <= um) {
                wu |= (unsigned)(u + 16) << (8 * b);
                ones |= 1u << (8 * b);
            }
        }
        reinterpret_cast<unsigned *>(&wt[row][half][0])[d] = wu;
        reinterpret_cast<unsigned *>(&wt[row][half][1])[d] = ones;
    }
    __syncthreads();
    const int j = threadIdx.x & 15;
    int frame, chunk;
    if (!xcd_frame_chunk(blockIdx.x, (g->kpLevelTotal + 15) >> 4, nFrames, frame, chunk)) return;
    const int slot = chunk * 16 + (threadIdx.x >> 4);
    const int level = slot < g->kpLevelTotal ? level_of_slot(g, slot) : -1;
    bool active = level >= 0;
    const LevelGeom &lg = g->lv[active ? level : 0];
    active = active && (slot - lg.kpOff < lvCount[frame * g->nLevels + (active ? level : 0)]);
    amos_keypoint *kp = lvKps + (size_t)frame * g->kpLevelTotal + (active ? slot : 0);
    const int cx = active ? __float2int_rn(kp->x) : kEdge, cy = active ? __float2int_rn(kp->y) : kEdge;
    const int stride = lg.stride;
    const int half = j & 1, rsub = j >> 1;
    // four 16-byte pieces per lane (vector-memory instructions are the scarce resource, not bytes)
    const uint8_t *col = level_origin(pyr, g, frame, active ? level : 0) + (ptrdiff_t)(cy - kHalfPatch) * stride + cx - kHalfPatch + 16 * half;
    uint4 px[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int row = min(rsub + 8 * q, 2 * kHalfPatch);
        __builtin_memcpy(&px[q], col + __mul24(row, stride), 16);  // unaligned global_load_dwordx4
    }
    unsigned acc10 = 0, sum1 = 0;
    int m01 = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int row = rsub + 8 * q;
        const uint4 w = wt[row][half][0], o = wt[row][half][1];
        const uint4 p = px[q];
        acc10 = __builtin_amdgcn_udot4(p.x, w.x, acc10, false);
        acc10 = __builtin_amdgcn_udot4(p.y, w.y, acc10, false);
        acc10 = __builtin_amdgcn_udot4(p.z, w.z, acc10, false);
        acc10 = __builtin_amdgcn_udot4(p.w, w.w, acc10, false);
        unsigned s1 = __builtin_amdgcn_udot4(p.x, o.x, 0u, false);
        s1 = __builtin_amdgcn_udot4(p.y, o.y, s1, false);
        s1 = __builtin_amdgcn_udot4(p.z, o.z, s1, false);
        s1 = __builtin_amdgcn_udot4(p.w, o.w, s1, false);
        sum1 += s1;
        m01 += (row - kHalfPatch) * (int)s1;
    }
    int m10 = (int)acc10 - 16 * (int)sum1;
    m10 = group16_sum(m10);
    m01 = group16_sum(m01);
    if (active && j == 0) kp->angle = fast_atan2_deg((float)m01, (float)m10);
}

// ---------------------------------------------------------------------------------------------
// a9 (first half)  cv::GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101), 8-bit fixed-point path
// (SURVEY A.2): dst = (sum_ij k_i k_j p + 32768) >> 16 with taps {18,34,48,56,48,34,18}/256.
// The device planes already carry the 19-px reflect-101 border, so the 3-px halo is read directly.
//
// A thread owns an 8-pixel-wide column strip and marches down kBlurStrip rows (no LDS).  The kernel is
// sized by instruction issue (one vector-memory instruction per 16 cycles and CU, one VALU per cycle):
//   * per row ONE unaligned 16-byte buffer load (pixels x-4 .. x+11) and, per output row, one 8-byte
//     store; eight rows of loads are issued together;
//   * horizontal 7-tap sums: the twelve byte-shifted dwords of the window (v_alignbyte, three are
//     free) feed 2 x v_dot4_u32_u8 per pixel;
//   * vertical: the sums (< 2^16) of rows (2i, 2i+1) are kept as NON-overlapping u16 pairs in a ring of
//     four registers per pixel; an output row is 3 x v_dot2_u32_u16 + 1 mad (even rows) or
//     4 x v_dot2_u32_u16 (odd rows, taps shifted by one), exact integer (sum + 32768) >> 16.
// Work items are linearised as (strip, group) inside a level so that consecutive lanes touch
// consecutive 8-byte pieces whatever the level width.  grid = (ceil(items/256), frames).
constexpr int kBlurStrip = 64;

__global__ __launch_bounds__(256) void k_blur(const uint8_t *__restrict__ pyr, uint8_t *__restrict__ blur,
                                             const Geom *__restrict__ g, int nFrames)
{
    typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
    typedef unsigned uint4v __attribute__((ext_vector_type(4)));
    typedef unsigned uint2v __attribute__((ext_vector_type(2)));
    int frame, chunk;  // XCD-aware: vertically adjacent strips (six shared halo rows) of a frame run on one XCD
    if (!xcd_frame_chunk(blockIdx.x, (g->blurItems + 255) >> 8, nFrames, frame, chunk)) return;
    const int item = chunk * 256 + threadIdx.x;
    const bool active = item < g->blurItems;
    int level = 0;
    for (int l = 1; l < g->nLevels; l++)
        if (active && item >= g->lv[l].blurItemStart) level = l;
    const LevelGeom &lg = g->lv[level];
    const int local = active ? item - lg.blurItemStart : 0;
    const int strip = local / lg.blurGroups;
    const int x = (local - strip * lg.blurGroups) * 8, y0 = strip * kBlurStrip;
    const int stride = lg.stride, planeRows = lg.h + 2 * kEdge;
    // One descriptor per frame (wave-uniform: a wave may straddle two levels, so the level's plane offset
    // travels in the per-lane byte offset instead).  Offsets count from the frame's first pyramid byte.
    const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(pyr) + (size_t)frame * g->frameBytes, 0, (int)g->frameBytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(blur + (size_t)frame * g->frameBytes, 0, (int)g->frameBytes, 0x00020000);
    const int colIn = lg.planeOff + x - 4 + kPadLeft, colOut = lg.planeOff + x + kPadLeft;
    const int lastOff = __mul24(planeRows - 1, stride) + colIn;   // last plane row (y = h + 18)
    int inOff = __mul24(y0 - 3 + kEdge, stride) + colIn;          // row y0 - 3 >= -3
    int outOff = __mul24(y0 + kEdge, stride) + colOut;
    int rowsLeft = active ? min(kBlurStrip, lg.h - y0) : 0;        // output rows still to store
    const unsigned KA = 18u | (34u << 8) | (48u << 16) | (56u << 24);  // taps of pixels x-3 .. x
    const unsigned KB = 48u | (34u << 8) | (18u << 16);                // taps of pixels x+1 .. x+3
    const ushort2v E0 = {18, 34}, E1 = {48, 56}, E2 = {48, 34};                 // even row: + 18 * current
    const ushort2v O0 = {0, 18}, O1 = {34, 48}, O2 = {56, 48}, O3 = {34, 18};    // odd row
    unsigned ring[8][4];  // ring[c][i % 4] = h[2i] | h[2i+1] << 16
    unsigned even[8];     // h of the last even row
#pragma unroll 1
    for (int rb = 0; rb < kBlurStrip + 6; rb += 8) {
        uint4v W[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            W[i] = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(src, inOff, 0, 0));
            inOff = min(inOff + stride, lastOff);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const unsigned d[4] = {W[i].x, W[i].y, W[i].z, W[i].w};
            unsigned A[13];  // A[o] = window bytes o .. o+3, o = 1 .. 12
#pragma unroll
            for (int o = 1; o <= 12; o++) A[o] = (o & 3) == 0 ? d[o >> 2] : __builtin_amdgcn_alignbyte(d[(o >> 2) + 1], d[o >> 2], o & 3);
            unsigned h[8];
#pragma unroll
            for (int c = 0; c < 8; c++) h[c] = __builtin_amdgcn_udot4(A[c + 1], KA, __builtin_amdgcn_udot4(A[c + 5], KB, 0u, false), false);
            // rows are numbered r = rb + i from the strip's first halo row; (rb is a multiple of 8, so
            // parity and ring slots depend on i only)
            unsigned acc[8];
            if ((i & 1) == 0) {
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    even[c] = h[c];
                    // rows r-6 .. r-1 are pairs (r-6)/2, (r-4)/2, (r-2)/2
                    unsigned a = __umul24(h[c], 18u) + 32768u;
                    a = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, ring[c][((i + 8 - 6) / 2) & 3]), E0, a, false);
                    a = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, ring[c][((i + 8 - 4) / 2) & 3]), E1, a, false);
                    a = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, ring[c][((i + 8 - 2) / 2) & 3]), E2, a, false);
                    acc[c] = a;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    ring[c][((i - 1) / 2) & 3] = even[c] | (h[c] << 16);
                    // row r-6 is the high half of pair (r-7)/2, then pairs (r-5)/2, (r-3)/2, (r-1)/2
                    unsigned a = 32768u;
                    a = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, ring[c][((i + 8 - 7) / 2) & 3]), O0, a, false);
                    a = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, ring[c][((i + 8 - 5) / 2) & 3]), O1, a, false);
                    a = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, ring[c][((i + 8 - 3) / 2) & 3]), O2, a, false);
                    a = __builtin_amdgcn_udot2(__builtin_bit_cast(ushort2v, ring[c][((i - 1) / 2) & 3]), O3, a, false);
                    acc[c] = a;
                }
            }
            if (rb + i >= 6 && rowsLeft > 0) {  // output row y0 + r - 6
                // byte 2 of each accumulator is the pixel
                uint2v o;
                o.x = __builtin_amdgcn_perm(acc[1], acc[0], 0x0c0c0602u) | __builtin_amdgcn_perm(acc[3], acc[2], 0x06020c0cu);
                o.y = __builtin_amdgcn_perm(acc[5], acc[4], 0x0c0c0602u) | __builtin_amdgcn_perm(acc[7], acc[6], 0x06020c0cu);
                __builtin_amdgcn_raw_buffer_store_b64(o, dst, outOff, 0, 0);  // row pitch leaves room past w
                outOff += stride;
                rowsLeft--;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// sincosf exactly as glibc >= 2.28 evaluates it for |x| < 120 (ARM optimized-routines algorithm:
// double-precision polynomials, no FMA).  The host oracle holds the same restatement and is
// compared bit for bit with the host libm over every float in [0, 2*pi].
__device__ __forceinline__ void glibc_sincosf(float y, float &sinp, float &cosp)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    const unsigned top = (__float_as_uint(y) >> 20) & 0x7ff;
    double x = (double)y;
    int n = 0;
    double sgnc = 1.0;  // table[1] negates the cosine coefficients
    if (top < ((__float_as_uint(0x1.921FB6p-1f) >> 20) & 0x7ff)) {
        if (top < ((__float_as_uint(0x1p-12f) >> 20) & 0x7ff)) {
            sinp = y;
            cosp = 1.0f;
            return;
        }
    } else {
        const double r = __dmul_rn(x, hpi_inv);
        n = ((int)r + 0x800000) >> 24;
        x = __dsub_rn(x, __dmul_rn((double)n, hpi));
        const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        if (n & 2) sgnc = -1.0;
        // sincosf_poly(x * s, x * x, ...)
        const double x2r = __dmul_rn(x, x);
        x = __dmul_rn(x, s);
        const double x2 = x2r;
        const double x4 = __dmul_rn(x2, x2), x3 = __dmul_rn(x2, x);
        const double c2 = __dadd_rn(sgnc * C3, __dmul_rn(x2, sgnc * C4));
        const double s1 = __dadd_rn(S2, __dmul_rn(x2, S3));
        const double c1 = __dadd_rn(sgnc * C0, __dmul_rn(x2, sgnc * C1));
        const double x5 = __dmul_rn(x3, x2), x6 = __dmul_rn(x4, x2);
        const double sv = __dadd_rn(x, __dmul_rn(x3, S1));
        const double cv = __dadd_rn(c1, __dmul_rn(x4, sgnc * C2));
        const float sres = (float)__dadd_rn(sv, __dmul_rn(x5, s1));
        const float cres = (float)__dadd_rn(cv, __dmul_rn(x6, c2));
        if (n & 1) { sinp = cres; cosp = sres; } else { sinp = sres; cosp = cres; }
        return;
    }
    {
        const double x2 = __dmul_rn(x, x);
        const double x4 = __dmul_rn(x2, x2), x3 = __dmul_rn(x2, x);
        const double c2 = __dadd_rn(C3, __dmul_rn(x2, C4));
        const double s1 = __dadd_rn(S2, __dmul_rn(x2, S3));
        const double c1 = __dadd_rn(C0, __dmul_rn(x2, C1));
        const double x5 = __dmul_rn(x3, x2), x6 = __dmul_rn(x4, x2);
        const double sv = __dadd_rn(x, __dmul_rn(x3, S1));
        const double cv = __dadd_rn(c1, __dmul_rn(x4, C2));
        sinp = (float)__dadd_rn(sv, __dmul_rn(x5, s1));
        cosp = (float)__dadd_rn(cv, __dmul_rn(x6, c2));
    }
}

// ---------------------------------------------------------------------------------------------
// a9 (second half) + a10  computeDescriptors / computeOrbDescriptor (ORBextractor.cc:173-227,
// 1525-1540) and the level-0 rescale + concatenation of ProcessDesp (:1747-1820).
// SIXTEEN LANES PER KEYPOINT (16 keypoints per work-group).  The group first stages the keypoint's
// 37-row window of the blurred level (rotated pattern offsets are <= 18 in magnitude) in LDS with
// 111 unaligned 16-byte loads (7 per lane) instead of 512 scattered byte gathers; then in step i lane j
// evaluates point pair 16 i + j from LDS; one __ballot per step yields, for each of the wave's four
// keypoints, the 16-bit descriptor word i, which lane i of the group keeps and finally stores
// (16 x 2 B = one 32-byte row).  The pattern sits in LDS as floats (one ds_read_b128 per pair).
constexpr int kDescR = 18;          // |rotated offset| <= sqrt(13^2 + 13^2) = 18.4 -> 18 after rounding
constexpr int kDescRowBytes = 48;   // columns -18 .. +29 as three 16-byte pieces
constexpr int kDescRows = 2 * kDescR + 1;

#ifndef AMOS_DESC_KPS
#define AMOS_DESC_KPS 16
#endif
constexpr int kDescKps = AMOS_DESC_KPS;  // keypoints per work-group (16 lanes each): 28 KB of patches + 4 KB of pattern in LDS

__global__ __launch_bounds__(kDescKps * 16) void k_describe(const uint8_t *__restrict__ blur, const Geom *__restrict__ g,
                                                           const amos_keypoint *__restrict__ lvKps,
                                                           const int *__restrict__ lvCount,
                                                           amos_keypoint *__restrict__ outKps, uint8_t *__restrict__ outDesc,
                                                           int *__restrict__ outCount, int nFrames)
{
    AMOS_SET_LATENCY_PRIO();
    __shared__ float4 pat[256];
    __shared__ uint4 patch[kDescKps][kDescRows * 3];
    int frame, chunk;
    if (!xcd_frame_chunk(blockIdx.x, (g->kpLevelTotal + kDescKps - 1) / kDescKps, nFrames, frame, chunk)) return;
    for (int e = threadIdx.x; e < 256; e += kDescKps * 16) {
        const signed char *p = &c_pattern[e * 4];
        pat[e] = float4{(float)p[0], (float)p[1], (float)p[2], (float)p[3]};
    }
    const int lane = threadIdx.x & 63, j = threadIdx.x & 15, sub = lane >> 4, grp = threadIdx.x >> 4;
    const int slot = chunk * kDescKps + grp;
    const int *cnt = lvCount + frame * g->nLevels;
    if (slot == 0 && j == 0) {
        int total = 0;
        for (int l = 0; l < g->nLevels; l++) total += cnt[l];
        outCount[frame] = total;
    }
    const int level = slot < g->kpLevelTotal ? level_of_slot(g, slot) : -1;
    bool active = level >= 0;
    const LevelGeom &lg = g->lv[active ? level : 0];
    const int i = slot - lg.kpOff;
    active = active && i < cnt[active ? level : 0];
    int dstIdx = i;
    for (int l = 0; l < (active ? level : 0); l++) dstIdx += cnt[l];
    active = active && dstIdx < g->kpCap;
    // only the three fields the descriptor needs live through the kernel (the whole 28-byte record kept in registers ended up partly
    // in scratch memory); lane 0 of the keypoint reads the record again when it writes it out
    const amos_keypoint *kpSrc = lvKps + (size_t)frame * g->kpLevelTotal + (active ? slot : 0);
    struct { float x, y, angle; } kp = {kpSrc->x, kpSrc->y, kpSrc->angle};
    if (!active) { kp.x = kp.y = (float)(kEdge + kDescR); kp.angle = 0.f; }
    const int stride = lg.stride;
    const uint8_t *corner = level_origin(blur, g, frame, active ? level : 0) + (ptrdiff_t)(__float2int_rn(kp.y) - kDescR) * stride +
                            (__float2int_rn(kp.x) - kDescR);
#pragma unroll
    for (int it = 0; it < (kDescRows * 3 + 15) / 16; it++) {
        const int piece = it * 16 + j;
        if (piece < kDescRows * 3) {
            const int r = piece / 3, c = piece - 3 * r;
            uint4 v;
            __builtin_memcpy(&v, corner + __mul24(r, stride) + 16 * c, 16);  // unaligned global_load_dwordx4
            patch[grp][piece] = v;
        }
    }
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float a, b;
    glibc_sincosf(__fmul_rn(kp.angle, factorPI), b, a);
    __syncthreads();
    const uint8_t *center = reinterpret_cast<const uint8_t *>(&patch[grp][0]) + kDescR * kDescRowBytes + kDescR;
    unsigned myword = 0;
#pragma unroll
    for (int step = 0; step < 16; step++) {
        const float4 pt = pat[16 * step + j];
        const int r0 = __float2int_rn(__fmaf_rn(pt.x, b, __fmul_rn(pt.y, a)));
        const int c0 = __float2int_rn(__fmaf_rn(pt.x, a, -__fmul_rn(pt.y, b)));
        const int r1 = __float2int_rn(__fmaf_rn(pt.z, b, __fmul_rn(pt.w, a)));
        const int c1 = __float2int_rn(__fmaf_rn(pt.z, a, -__fmul_rn(pt.w, b)));
        const int t0 = center[__mul24(r0, kDescRowBytes) + c0];
        const int t1 = center[__mul24(r1, kDescRowBytes) + c1];
        const unsigned long long bal = __ballot(t0 < t1);
        const unsigned word = (unsigned)(bal >> (16 * sub)) & 0xffffu;
        if (j == step) myword = word;
    }
    if (active) {
        reinterpret_cast<uint16_t *>(outDesc + ((size_t)frame * g->kpCap + dstIdx) * 32)[j] = (uint16_t)myword;
        if (j == 0) {
            amos_keypoint o = *kpSrc;
            if (level != 0) {  // keypoint->pt *= scale, ORBextractor.cc:1804-1813
                o.x = __fmul_rn(o.x, lg.scale);
                o.y = __fmul_rn(o.y, lg.scale);
            }
            outKps[(size_t)frame * g->kpCap + dstIdx] = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// a8  ORBextractor::MovingKeyPoints (ORBextractor.cc:1688-1745): dilate then erode of the mask with
// the 31x31 MORPH_ELLIPSE element, then the per-keypoint gate.
__constant__ int c_ellipse_dx[31];  // half-width of every element row, host-computed

// The element's row half-widths (checked against c_ellipse_dx on the host at create time): row |dy| of the 31 x 31 ellipse
// spans columns -kMorphDx[|dy|] .. +kMorphDx[|dy|].  Ten distinct widths, nested: a wider window's maximum dominates a
// narrower one's, so
//     dilate(y, x) = max_dy  H_{dx(dy)}(y + dy, x),      H_r(y, x) = max_{|j| <= r} src(y, x + j)
// and the ten H_r of a pixel come from ONE outward sweep over its row (31 loads, 15 v_max3) instead of one sweep per
// element row.  Phase 1 stages the tile (neutral value outside the image, OpenCV's morphologyDefaultBorderValue),
// phase 2 writes the nine planes H_5 .. H_15 of every tile row (H_0 is the tile itself), phase 3 takes for every output
// pixel the 31 values (one per element row, each from its row's plane): 46 KB of LDS, about 90 LDS loads and 45 vector
// instructions per output pixel against 700 + 700 for the direct form.
// grid = (ceil(w/64), ceil(h/32), frames), block = 256.
__device__ constexpr int kMorphDx[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 7, 5, 0};   // by |dy|
__device__ constexpr int kMorphPlane[16] = {8, 8, 8, 8, 7, 7, 7, 6, 6, 5, 4, 3, 2, 1, 0, -1};           // plane of that width; -1 = the tile
constexpr int kMorphTileW = 64, kMorphTileH = 32, kMorphRows = kMorphTileH + 30, kMorphPitch = 96;

template <bool kDilate>
__global__ __launch_bounds__(256) void k_morph31(const uint8_t *__restrict__ src, size_t srcFrameStride, int srcStride,
                                                uint8_t *__restrict__ dst, size_t dstFrameStride, int stride, int w, int h)
{
    __shared__ __align__(16) uint8_t tile[kMorphRows][kMorphPitch];
    __shared__ __align__(16) uint8_t plane[9][kMorphRows][kMorphTileW];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * kMorphTileW, y0 = blockIdx.y * kMorphTileH;
    src += (size_t)blockIdx.z * srcFrameStride;
    dst += (size_t)blockIdx.z * dstFrameStride;
    const int neutral = kDilate ? 0 : 255;
    auto pick = [](int a, int b, int c) { return kDilate ? max(a, max(b, c)) : min(a, min(b, c)); };  // v_max3 / v_min3
    __shared__ int sLo, sHi;
    if (tid == 0) { sLo = 255; sHi = 0; }
    __syncthreads();
    int lo = 255, hi = 0;  // range of the IMAGE pixels under the tile and its 15-pixel border
    for (int idx = tid; idx < kMorphRows * (kMorphTileW + 30); idx += 256) {
        const int r = idx / (kMorphTileW + 30), c = idx - r * (kMorphTileW + 30);
        const int y = y0 - 15 + r, x = x0 - 15 + c;
        const bool in = y >= 0 && y < h && x >= 0 && x < w;
        const int v = in ? src[(size_t)y * srcStride + x] : neutral;
        tile[r][c] = (uint8_t)v;
        lo = in ? min(lo, v) : lo;
        hi = in ? max(hi, v) : hi;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        lo = min(lo, __shfl_xor(lo, d, 64));
        hi = max(hi, __shfl_xor(hi, d, 64));
    }
    if ((tid & 63) == 0) { atomicMin(&sLo, lo); atomicMax(&sHi, hi); }
    __syncthreads();
    const int lx = tid & 63;
    // A tile whose whole neighbourhood holds ONE value keeps it (a window's maximum / minimum over equal values; every window contains its own
    // image pixel, so the border's neutral value never wins): person masks are mostly such tiles -- all background, or all inside a silhouette
    // -- and they skip the two window phases.
    if (sLo >= sHi) {
        const int v0 = sLo;
        for (int ly = tid >> 6; ly < kMorphTileH; ly += 4) {
            const int x = x0 + lx, y = y0 + ly;
            if (x < w && y < h) dst[(size_t)y * stride + x] = (uint8_t)v0;
        }
        return;
    }
    for (int r = tid >> 6; r < kMorphRows; r += 4) {  // phase 2: the nested horizontal windows of (r, lx)
        const uint8_t *c = &tile[r][lx + 15];
        int m = pick(c[0], pick(c[-1], c[1], c[-2]), pick(c[2], c[-3], c[3]));
        m = pick(m, pick(c[-4], c[4], c[-5]), c[5]);
        plane[0][r][lx] = (uint8_t)m;                                   // H_5
        m = pick(m, pick(c[-6], c[6], c[-7]), c[7]);
        plane[1][r][lx] = (uint8_t)m;                                   // H_7
        m = pick(m, pick(c[-8], c[8], c[-9]), c[9]);
        plane[2][r][lx] = (uint8_t)m;                                   // H_9
#pragma unroll
        for (int k = 10; k <= 15; k++) {                                // H_10 .. H_15
            m = pick(m, c[-k], c[k]);
            plane[k - 7][r][lx] = (uint8_t)m;
        }
    }
    __syncthreads();
    for (int ly = tid >> 6; ly < kMorphTileH; ly += 4) {  // phase 3: one value per element row
        const int x = x0 + lx, y = y0 + ly;
        int acc = pick(plane[8][ly + 15][lx], tile[ly][lx + 15], tile[ly + 30][lx + 15]);  // dy = 0 (H_15), dy = -15 / +15 (H_0)
#pragma unroll
        for (int dy = 1; dy <= 14; dy++) acc = pick(acc, plane[kMorphPlane[dy]][ly + 15 - dy][lx], plane[kMorphPlane[dy]][ly + 15 + dy][lx]);
        if (x < w && y < h) dst[(size_t)y * stride + x] = (uint8_t)acc;
    }
}

// One work-group per frame walks the levels in order; kept keypoints are compacted in place
// (order preserved), removed ones appended to `removed` in the reference's order.
__global__ __launch_bounds__(256) void k_gate(const Geom *__restrict__ g, amos_keypoint *__restrict__ lvKps,
                                             int *__restrict__ lvCount, const uint8_t *__restrict__ closedBase,
                                             size_t closedFrameStride, int maskStride, const double *__restrict__ labels,
                                             int labelStride,
                                             const int *__restrict__ centerIds, int nCenters,
                                             const int *__restrict__ rm, int nRm,
                                             amos_keypoint *__restrict__ removed, int *__restrict__ nRemoved,
                                             int *__restrict__ errFlag)
{
    __shared__ int wkeep[4], wrem[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frame = blockIdx.x;
    const uint8_t *closed = closedBase + (size_t)frame * closedFrameStride;
    int remBase = 0;
    for (int level = 0; level < g->nLevels; level++) {
        const LevelGeom &lg = g->lv[level];
        amos_keypoint *kps = lvKps + (size_t)frame * g->kpLevelTotal + lg.kpOff;
        const int n = lvCount[frame * g->nLevels + level];
        const float scale = level != 0 ? lg.scale : 1.f;
        int keepBase = 0;
        for (int i0 = 0; i0 < n; i0 += 256) {
            const int i = i0 + tid;
            amos_keypoint kp;
            bool valid = i < n, drop = false;
            if (valid) {
                kp = kps[i];
                const int ix = (int)__fmul_rn(kp.x, scale), iy = (int)__fmul_rn(kp.y, scale);
                if (ix < 0 || iy < 0 || ix >= g->W || iy >= g->H) {
                    atomicOr(errFlag, 1);
                } else {
                    if (labels) {
                        const long ci = (long)(labels[(size_t)iy * labelStride + ix] - 1);
                        if (ci < 0 || ci >= nCenters) atomicOr(errFlag, 2);
                        else {
                            const int id = centerIds[ci];
                            if (id < 0 || id >= nRm) atomicOr(errFlag, 2);
                            else if (rm[id] == 1) drop = true;
                        }
                    }
                    if (closed[(size_t)iy * maskStride + ix] != 0) drop = true;
                }
            }
            const unsigned long long bk = __ballot(valid && !drop), br = __ballot(valid && drop);
            if (lane == 0) { wkeep[wave] = __popcll(bk); wrem[wave] = __popcll(br); }
            __syncthreads();  // every thread has read its keypoint before anyone overwrites the list
            int ko = keepBase, ro = remBase;
            for (int w = 0; w < wave; w++) { ko += wkeep[w]; ro += wrem[w]; }
            const unsigned long long below = (1ull << lane) - 1ull;
            if (valid && !drop) kps[ko + __popcll(bk & below)] = kp;
            if (valid && drop) removed[(size_t)frame * g->kpLevelTotal + ro + __popcll(br & below)] = kp;
            keepBase += wkeep[0] + wkeep[1] + wkeep[2] + wkeep[3];
            remBase += wrem[0] + wrem[1] + wrem[2] + wrem[3];
            __syncthreads();
        }
        if (tid == 0) lvCount[frame * g->nLevels + level] = keepBase;
    }
    if (tid == 0) nRemoved[frame] = remBase;
}

// cv::undistortPoints(pts, pts, K, distCoef, Mat(), K) for one point, as Frame::UndistortKeyPoints and
// Frame::ComputeImageBounds call it (Frame.cc:1052-1118, 1121-1170): OpenCV 4.5 cvUndistortPointsInternal
// with its default criteria (exactly 5 fixed-point iterations, no epsilon test), double arithmetic,
// k = (k1, k2, p1, p2, k3) and the remaining coefficients zero, R = I, P = K.  Shared by the device
// kernel and the host-side image bounds.
__host__ __device__ inline void undistort_point(float u, float v, double fx, double fy, double cx, double cy, const double (&k)[5],
                                                float &xo, float &yo)
{
    const double ifx = 1. / fx, ify = 1. / fy;
    double x = ((double)u - cx) * ifx, y = ((double)v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = 1. / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) {  // OpenCV gives up and returns the normalised input point
            x = ((double)u - cx) * ifx;
            y = ((double)v - cy) * ify;
            break;
        }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    xo = (float)(fx * x + cx);  // RR = K * I; ww = 1
    yo = (float)(fy * y + cy);
}

struct UndistortArgs {
    double fx, fy, cx, cy;
    double k[5];
    int identity;  // mDistCoef[0] == 0: mvKeysUn = mvKeys (Frame.cc:1057-1061)
};

// Frame::UndistortKeyPoints for every keypoint of the batch result.  grid = (ceil(kpCap/256), frames).
__global__ __launch_bounds__(256) void k_undistort(const Geom *__restrict__ g, const amos_keypoint *__restrict__ outKps,
                                                  const int *__restrict__ outCount, const UndistortArgs a,
                                                  amos_keypoint *__restrict__ kpsUn)
{
    const int i = blockIdx.x * 256 + threadIdx.x, frame = blockIdx.y;
    if (i >= outCount[frame] || i >= g->kpCap) return;
    const size_t o = (size_t)frame * g->kpCap + i;
    amos_keypoint kp = outKps[o];
    if (!a.identity) undistort_point(kp.x, kp.y, a.fx, a.fy, a.cx, a.cy, a.k, kp.x, kp.y);
    kpsUn[o] = kp;
}

// Frame::ComputeStereoFromRGBD (Frame.cc:1576-1615) + the grid cell of PosInGrid (Frame.cc:1007-1030),
// one thread per keypoint of the batch result: the depth is looked up at the DISTORTED keypoint (mvKeys),
// uRight and the grid cell come from the undistorted one (mvKeysUn; kpsUn == nullptr: the same).
// grid = (ceil(kpCap/256), frames).
__global__ __launch_bounds__(256) void k_rgbd_glue(const Geom *__restrict__ g, const amos_keypoint *__restrict__ outKps,
                                                  const amos_keypoint *__restrict__ kpsUn,
                                                  const int *__restrict__ outCount, const uint8_t *__restrict__ depth,
                                                  int depthIsU16, float depthFactor, size_t depthFrameStride,
                                                  size_t depthRowStride, float mbf, float minX, float minY,
                                                  float gridWInv, float gridHInv, float *__restrict__ uRight,
                                                  float *__restrict__ depthOut, int *__restrict__ gridCell)
{
    const int i = blockIdx.x * 256 + threadIdx.x, frame = blockIdx.y;
    if (i >= outCount[frame] || i >= g->kpCap) return;
    const size_t o = (size_t)frame * g->kpCap + i;
    const amos_keypoint kp = outKps[o];
    const amos_keypoint kpU = kpsUn ? kpsUn[o] : kp;
    const int u = (int)kp.x, v = (int)kp.y;  // imDepth.at<float>(v, u) with float arguments
    float d = -1.f;
    if (depth != nullptr && u >= 0 && v >= 0 && u < g->W && v < g->H) {
        const uint8_t *row = depth + (size_t)frame * depthFrameStride + (size_t)v * depthRowStride;
        d = depthIsU16 ? __fmul_rn((float)reinterpret_cast<const uint16_t *>(row)[u], depthFactor)
                       : reinterpret_cast<const float *>(row)[u];
    }
    const bool ok = d > 0;
    if (depthOut) depthOut[o] = ok ? d : -1.f;
    if (uRight) uRight[o] = ok ? __fsub_rn(kpU.x, __fdiv_rn(mbf, d)) : -1.f;
    const int px = (int)roundf(__fmul_rn(__fsub_rn(kpU.x, minX), gridWInv));
    const int py = (int)roundf(__fmul_rn(__fsub_rn(kpU.y, minY), gridHInv));
    gridCell[o] = (px < 0 || px >= AMOS_FRAME_GRID_COLS || py < 0 || py >= AMOS_FRAME_GRID_ROWS) ? -1 : px * AMOS_FRAME_GRID_ROWS + py;
}

// Unpacks the compacted candidates of one level into amos_keypoint records (parity tests only).
__global__ void k_unpack_candidates(const uint32_t *__restrict__ pts, int n, amos_keypoint *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = pts[i];
    amos_keypoint kp;
    kp.x = (float)(p & 0xfff);
    kp.y = (float)((p >> 12) & 0xfff);
    kp.size = 7.f;
    kp.angle = -1.f;
    kp.response = (float)(p >> 24);
    kp.octave = 0;
    kp.class_id = -1;
    out[i] = kp;
}

}  // namespace amos

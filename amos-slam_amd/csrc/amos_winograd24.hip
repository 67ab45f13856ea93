// amos_winograd24.hip -- the stride-1 3 x 3 convolutions of the mask network (a15, yolact.py:47-200, 265-400, backbone.py:60-200) as Winograd
// F(2 x 4, 3 x 3) on the fp32 MFMA units: the vertical direction as F(2, 3) (four positions per two output rows), the horizontal one as
// F(4, 3) (six positions per four output columns) -- 24 multiplies per 2 x 4 outputs and channel pair, 3.0 per output against 4.0 for
// F(2 x 2) (amos_winograd.hip) and 9.0 for the direct convolution.  Float32 in, float32 accumulate.
//
//   Y (2 x 4) = A2^T [ sum_c (G2 g G4^T) . (B2^T d B4) ] A4          d: 4 x 6 input tile (pad 1), g: 3 x 3 filter
//   B2^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]      G2 = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]        A2^T = [1 1 1 0; 0 1 -1 -1]
//   B4^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G4   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]    A4^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
//
// Why this shape and not F(4 x 4): 24 positions divide over the 8 waves of a work-group (3 each, two waves per SIMD, the MFMA load of the
// four SIMDs balanced) and their accumulators for 32 tiles x 64 output channels are 96 registers per lane; the 36 positions of F(4 x 4)
// need 9 or 12 waves (an unbalanced SIMD, or 3 waves per SIMD at 170 registers for 96 accumulators + fragments + transform) and either
// half the output channels per work-group or 295 KB of accumulators per CU.
//
// Structure (the F(2 x 2) kernel's, amos_winograd.hip, with these differences): one work-group = 512 threads = 8 waves owns 32 consecutive
// tiles (2 x 4 outputs each: the same 256 output pixels) x 64 output channels x all 24 positions.  Wave w multiplies positions 3w .. 3w + 2
// = row w >> 1 of the 4 x 6 position grid, columns 3 (w & 1) .. + 2 -- and the SAME wave makes exactly these rows of B2^T d B4 for all 32
// tiles (lane = tile x channel quad), so a V tile is private to its wave (only the raw patch is shared: one work-group barrier per stage).
//   U (G2 g G4^T, k_winograd24_weights, MFMA fragment order) goes from global memory straight into the owning wave's registers;
//   X: the raw patch [4 rows][columns][8 channels] by LDS-DMA (buffer_load_dwordx4 ... lds), out-of-image pixels as out-of-range offsets
//      (zeros); the tiles of a row segment share their columns (column = 4 x tile + c, c = 0 .. 5);
//   V: thread (tile, channel quad) reads 2 rows x 5 columns of the patch, 44 multiply-adds, three ds_write_b128;
//   a stage (8 input channels) = positions 0 and 1 (16 MFMAs) | barrier | position 2 (8 MFMAs), fragments and the next V tile made under
//   the MFMAs, U fetched one and a half stages ahead, the raw patch a whole stage ahead behind a counted vmcnt.
// Epilogue: every wave applies A4 to its half row (four partial sums per tile and channel), the partials meet in LDS (two rounds of 32
// output channels), every thread finishes one output row of a tile for four channels: A2^T, bias (+ residual), ReLU, 16-byte stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "../../include/amos_frontend.h"
#include "amos_common.h"

namespace amos {

constexpr int kW24Tiles = 32;     // tiles per work-group (one 32-row MFMA block)
constexpr int kW24Cout = 64;      // output channels per work-group (two 32-column MFMA blocks)
constexpr int kW24K = 8;          // input channels per stage
constexpr int kW24Threads = 512;
constexpr int kW24Pos = 24;       // 4 x 6 positions
#ifndef AMOS_W24_GROUP
#define AMOS_W24_GROUP 16
#endif
constexpr int kW24Group = AMOS_W24_GROUP;                 // m blocks of an XCD that run one n tile before the next (amos_winograd.hip, map 1); 16: 2 - 3 % faster than
                                                          // 1 .. 8 on the 256-channel layers and the least L2 fetch traffic (2.24 GB per launch of proto_net[8] against 2.56; tools/r4_w24_map.sh)
constexpr int kW24StageU = kW24Pos * kW24Cout * kW24K;    // floats of one stage's U image (48 KB)
constexpr int kW24PosV = kW24Tiles * kW24K;               // floats of one position's V block (1 KB)
constexpr int kW24StageV = kW24Pos * kW24PosV;            // floats of one stage's V tile (24 KB)
constexpr int kW24RawCols = 256;                          // columns of the raw patch (32 tiles in up to 32 one-tile segments: 32 x 6 = 192)
constexpr int kW24RawRow = kW24RawCols * kW24K;           // floats of one patch row
constexpr int kW24StageR = 4 * kW24RawRow;                // floats of one raw patch (32 KB)
#ifndef AMOS_W24_XAHEAD
#define AMOS_W24_XAHEAD 1  /* stages between the request of a raw patch and the barrier that needs it: 1 (two patch buffers) or 2 (three, experiment) */
#endif
constexpr int kW24RawBufs = AMOS_W24_XAHEAD + 1;
constexpr int kW24ExchangeFloats = 8 * 4 * 32 * 32;       // the epilogue's exchange image (128 KB)
constexpr int kW24LoopFloats = 2 * kW24StageV + kW24RawBufs * kW24StageR;  // 2 V tiles + 2 (3) raw patches: 112 (144) KB
constexpr int kW24LdsFloats = kW24ExchangeFloats > kW24LoopFloats ? kW24ExchangeFloats : kW24LoopFloats;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct W24Args {
    const float *x, *u, *bias, *res;
    float *y;
    int B, H, W, C, N;          // frames, image size, input channels, output channels
    int tilesX, tilesY, tilesPerImage, totalTiles;
    int mBlocks, nTiles, stages, relu;
    unsigned xBytes;            // size of x in bytes (the buffer descriptor's range)
    // Layouts.  x: byte strides of a pixel and of a stage (8 input channels): channels-last [b][h][w][C] has (4 C, 32), the channel-blocked
    // form [b][C / 8][h][w][8] has (32, 32 H W) -- there the 32 bytes a stage reads of a pixel sit beside the same stage's 32 bytes of the next
    // pixel, so a patch row is ONE contiguous run of whole cache lines used once, where channels-last hands every 128-byte line to four
    // different stages (and L1 has lost it by the time the next one asks).  y / res: outBlocked selects [b][N / 8][h][w][8].
    unsigned xPixelBytes, xStageBytes;
    int outBlocked;
    int virtualBlocks;          // ids of the (tile block, channel tile) map; the persistent form walks them with a stride of gridDim.x
};

// k * a + b per component as one fused multiply-add each (the file is compiled with -ffp-contract=off)
__device__ __forceinline__ f32x4 w24_fma(float k, f32x4 a, f32x4 b)
{
    return f32x4{__builtin_fmaf(k, a.x, b.x), __builtin_fmaf(k, a.y, b.y), __builtin_fmaf(k, a.z, b.z), __builtin_fmaf(k, a.w, b.w)};
}

// V block of one position: element (row = tile, k) at float offset row * 8 + 4 * ((k >> 2) ^ f(row)) + (k & 3)  (amos_winograd.hip)
__device__ __host__ __forceinline__ int w24_swz(int row, int half) { return row * 8 + 4 * (half ^ (((row >> 2) ^ (row >> 3)) & 1)); }
// raw patch: 16-byte unit of (column, channel quad) inside a patch row -- blocks of 32 columns, inside a block [column % 4][column / 4][quad]:
// one LDS-DMA instruction fills a block lane-linearly with lane pairs fetching the two quads of one pixel (32 contiguous bytes), and the
// transform's reads (lanes = consecutive tiles x both quads: every FOURTH column) are 32-byte pieces at a 32-byte pitch: conflict-free
// (with F(2 x 2)'s [parity][column / 2] order the pitch would be 64 bytes: two-way bank conflicts on all ten reads)
__device__ __forceinline__ int w24_raw_unit(int col, int quad) { return (col >> 5) * 64 + (col & 3) * 16 + ((col & 31) >> 2) * 2 + quad; }

// Weights [cout][3][3][cin] (a channels-last Conv2d weight) -> U = G2 g G4^T in the order the MFMA's B fragments are read: for every
// (cout tile nt, stage s, position p = 6 a + b, 32-channel block j) 64 lanes x 4 floats, lane l = U_p[nt * 64 + j * 32 + (l & 31)][s * 8 + 4 (l >> 5) ..+3];
// sums in double, one rounding.
__global__ __launch_bounds__(256) void k_winograd24_weights(const float *__restrict__ w, float *__restrict__ u, int cin, int cout)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cin * cout) return;
    const int c = idx % cin, n = idx / cin;
    double g[3][3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) g[a][b] = (double)w[((size_t)(n * 3 + a) * 3 + b) * cin + c];
    double t[4][3];  // G2 g
    for (int b = 0; b < 3; b++) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5 * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5 * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const int stages = cin / kW24K, nt = n / kW24Cout, nr = n % kW24Cout, s = c / kW24K, k = c % kW24K;
    float *img = u + (size_t)(nt * stages + s) * kW24StageU + (size_t)((nr >> 5) * 64 + (k >> 2) * 32 + (nr & 31)) * 4 + (k & 3);
    for (int a = 0; a < 4; a++) {
        const double t0 = t[a][0], t1 = t[a][1], t2 = t[a][2];
        const double r[6] = {t0 / 4.0, -(t0 + t1 + t2) / 6.0, -(t0 - t1 + t2) / 6.0, t0 / 24.0 + t1 / 12.0 + t2 / 6.0, t0 / 24.0 - t1 / 12.0 + t2 / 6.0, t2};
        for (int b = 0; b < 6; b++) img[(size_t)(a * 6 + b) * 512] = (float)r[b];
    }
}

// The whole work of one wave, compiled once per half of the position row (kHalf = wave & 1): the two halves combine the patch columns
// differently, and a run-time branch on that inside a stage would split the stage's basic block -- the interleaving of the transform
// with the MFMAs (sched_group_barrier) only works inside one.  Every wave of a work-group passes the same barriers in either copy.
//
// kPersist (round 5): ONE work-group per CU walks the ids blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x is a multiple of 8, so an id's XCD
// is the group's).  A CU holds one group at a time (128 KB of LDS, 512 x 256 registers), so with one id per group nothing overlaps a group's
// prologue -- two raw patches and four U fragments fetched from L2 / HBM, then the first transform: ~10 k of a group's ~140 k cycles -- or its
// epilogue.  Here the LAST stages of an id request the NEXT id's first two raw patches in the very slots the steady state uses for
// "stage s + 3" (the buffers are free by then) and the epilogue requests its four prologue U fragments half way, so they travel under the last
// MFMAs and the epilogue; what is left between two ids is one wait, the first transform and two barriers.  For that the epilogue's exchange
// image may not overlay the raw patches: it is 64 KB (the V tiles + the 16 KB tail of the allocation) and takes four rounds of 16 tiles x 32
// channels instead of two of 32 x 32; same sums in the same order, same bits.
//
// kCB (round 5): 32-channel blocks of output per work-group -- 2 (64 output channels, the large launches) or 1.  One frame per pass leaves
// the 256-channel layers at 69 x 69 twenty tile blocks: 80 work-groups of 64 channels on 256 CUs.  With 32 channels per work-group there are
// 160, each with half the MFMAs (the transform is done twice as often: the price of filling the chip); the channel tile nt then addresses
// block nt & 1 of the 64-channel slice nt >> 1 of U.
template <int kHalf, bool kPersist, int kCB>
__device__ __forceinline__ void w24_run(const W24Args &a, float *smem24)
{
    static_assert(kCB == 2 || !kPersist, "the persistent form is written for 64 output channels per work-group");
    constexpr int half = kHalf;
#define AMOS_W24_V(buf) (smem24 + (buf) * kW24StageV)
#define AMOS_W24_R(buf) (smem24 + 2 * kW24StageV + (buf) * kW24StageR)
    // id -> (m block, n tile): an XCD (ids are dealt round-robin over the 8 XCDs) owns a contiguous run of m blocks and walks it in groups of
    // kW24Group blocks, n tile after n tile (amos_winograd.hip, map 1: the group's patches stay in the XCD's L2 for the next n tile, and
    // half the XCD's CUs share one weight slice at a time)
    const int perXcd = (a.mBlocks + 7) >> 3;
    auto decode = [&](int id, int &mbOut, int &ntOut) -> bool {
        const int xcd = id & 7, seq = id >> 3;
        const int per = kW24Group * a.nTiles, grp = seq / per, in = seq - grp * per;
        const int mbLocal = grp * kW24Group + in % kW24Group;
        ntOut = in / kW24Group;
        mbOut = xcd * perXcd + mbLocal;
        return mbLocal < perXcd && mbOut < a.mBlocks;
    };
    auto next_valid = [&](int id, int &mbOut, int &ntOut) -> int {  // the first valid id >= id of this group's walk, or -1
        for (; id < a.virtualBlocks; id += (int)gridDim.x)
            if (decode(id, mbOut, ntOut)) return id;
        return -1;
    };
    int mb = 0, nt = 0;
    int id = kPersist ? next_valid((int)blockIdx.x, mb, nt) : (decode((int)blockIdx.x, mb, nt) ? (int)blockIdx.x : -1);
    if (id < 0) return;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);

    // ---- geometry of an id's tile run: segments of tiles of one tile row.  This thread is (a) the loader of raw-patch column
    // `col` (block `wave`, lane order [column % 4][column / 4][quad]) for all four patch rows and (b) the transformer of tile `tl`, channel quad
    // `quad`, position row `prow`, position columns 3 * half .. + 2.
    const int quad = lane & 1, tl = lane >> 1, prow = wave >> 1;
    const int col = 32 * wave + 4 * ((lane >> 1) & 7) + (lane >> 4);  // lane l fills unit l of its block: w24_raw_unit(col, lane & 1) == 64 * wave + l
    // xo[r]: loader: byte offset of (patch row r, column col, channel quad) in x, or the buffer's size (zeros) when there is no such pixel;
    // cbase: transformer: first patch column of tile tl
    auto geometry = [&](int mbi, int (&xo)[4], int &cbase) {
        const int T0 = mbi * kW24Tiles, nT = min(kW24Tiles, a.totalTiles - T0);
        int b = T0 / a.tilesPerImage, rem = T0 - b * a.tilesPerImage, ty = rem / a.tilesX, tx = rem - ty * a.tilesX;
        cbase = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) xo[r] = (int)a.xBytes;
        for (int t0 = 0, cb = 0; t0 < nT;) {  // (b, ty, tx) = the segment's first tile, t0 its index in the run, cb its first column
            const int n = min(a.tilesX - tx, nT - t0);
            if (col >= cb && col < cb + 4 * n + 2) {
                const int lc = col - cb, tloc = min(lc >> 2, n - 1), ix = 4 * (tx + tloc) - 1 + (lc - 4 * tloc);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int iy = 2 * ty - 1 + r;
                    if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                        xo[r] = (int)((unsigned)b * (unsigned)(a.H * a.W) * (unsigned)a.C * 4u + (unsigned)(iy * a.W + ix) * a.xPixelBytes + 16u * quad);
                }
            }
            if (tl >= t0 && tl < t0 + n) cbase = cb + 4 * (tl - t0);
            t0 += n;
            cb += 4 * n + 2;
            tx = 0;
            if (++ty == a.tilesY) { ty = 0; b++; }
        }
        if (tl >= nT) cbase = 0;  // tiles past the end of the tensor: any valid patch position (their results are not stored)
    };
    int xoff[4], colBase = 0;
    geometry(mb, xoff, colBase);
    int xoffN[4] = {0, 0, 0, 0}, colBaseN = 0, mbN = 0, ntN = 0, idN = -1;  // the next id of a persistent group (tail of the stage loop)
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.xBytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *LdsPtr;
    // the transformer's ten reads: rows (r0, r1) of B2^T d row `prow` (0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3), columns colBase + half + c, c = 0 .. 4
    const int r0 = prow == 0 ? 0 : (prow == 2 ? 2 : 1), r1 = prow == 0 ? 2 : (prow == 1 ? 2 : (prow == 2 ? 1 : 3));
    int rsrc0[5];  // float offsets inside a raw patch of (row r0, column colBase + half + c); row r1 is (r1 - r0) * kW24RawRow further
#define AMOS_W24_SET_RSRC() { _Pragma("unroll") for (int c = 0; c < 5; c++) rsrc0[c] = r0 * kW24RawRow + w24_raw_unit(colBase + half + c, quad) * 4; }
    AMOS_W24_SET_RSRC()
    const int rdelta = (r1 - r0) * kW24RawRow;
    const float sgn = prow == 1 ? 1.f : -1.f;                 // the row's second term is added (row 1) or subtracted
    const int vdst = (3 * wave) * kW24PosV + w24_swz(tl, quad);  // + p * kW24PosV for position 3 * wave + p
    // U fragments of this wave's three positions: [position of the triple][cout block] x 16 bytes per lane and stage
    const float *const ubase = a.u + (size_t)(wave * 3) * 512 + lane * 4;
    // (kCB == 1: U of 64-channel slice nt >> 1; its 32-channel block nt & 1 is picked in AMOS_W24_FETCH_UP)
    const float *usrc = ubase + (size_t)(kCB == 1 ? nt >> 1 : nt) * a.stages * kW24StageU, *usrcN = usrc;
    const int ujb = kCB == 1 ? nt & 1 : 0;

#ifdef AMOS_W24_EXP_NOX  /* timing experiments (results are wrong): tools/w24_variants.sh */
#define AMOS_W24_FETCH_XO(xo, s, buf) {}
#else
#define AMOS_W24_FETCH_XO(xo, s, buf)                                                                                                \
    {                                                                                                                                \
        _Pragma("unroll") for (int r = 0; r < 4; r++)                                                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (LdsPtr)(AMOS_W24_R(buf) + r * kW24RawRow + wave * 256), 16, xo[r], (s) * a.xStageBytes, 0, 0); \
    }
#endif
#define AMOS_W24_FETCH_X(s, buf) AMOS_W24_FETCH_XO(xoff, s, buf)
#ifdef AMOS_W24_U_NT   /* experiment: the weight stream with the non-temporal hint */
#define AMOS_W24_ULOAD(p) __builtin_nontemporal_load(p)
#else
#define AMOS_W24_ULOAD(p) (*(p))
#endif
#ifdef AMOS_W24_EXP_NOU
#define AMOS_W24_FETCH_UP(fb, up, s, p) { _Pragma("unroll") for (int j = 0; j < kCB; j++) fb[j] = f32x4{(float)(s), (float)lane, 1.f, (float)(p)}; }
#else
#define AMOS_W24_FETCH_UP(fb, up, s, p)                                                                                              \
    {                                                                                                                                \
        _Pragma("unroll") for (int j = 0; j < kCB; j++)                                                                              \
            fb[j] = AMOS_W24_ULOAD(reinterpret_cast<const f32x4 *>((up) + (size_t)(s) * kW24StageU + ((p) * 2 + j + ujb) * 256));    \
    }
#endif
#define AMOS_W24_FETCH_U(fb, s, p) AMOS_W24_FETCH_UP(fb, usrc, s, p)
    // raw patch (buffer rb) -> this thread's half row of B2^T d B4 -> the wave's V blocks (buffer vb).  s_c = d[r0][c] +- d[r1][c] for the five
    // columns; half 0: V0 = 4 s0 - 5 s2 + s4, V1 = (s4 - 4 s2) + (s3 - 4 s1), V2 = (s4 - 4 s2) - (s3 - 4 s1);
    // half 1 (s = t1 .. t5): V3 = (s3 - s1) + 2 (s2 - s0), V4 = (s3 - s1) - 2 (s2 - s0), V5 = 4 s0 - 5 s2 + s4.
#define AMOS_W24_SCOL(c)                                                                                                             \
    ([&]() -> f32x4 {                                                                                                                \
        const f32x4 d0 = *reinterpret_cast<const f32x4 *>(rp + rsrc0[c]), d1 = *reinterpret_cast<const f32x4 *>(rp + rsrc0[c] + rdelta); \
        return f32x4{__builtin_fmaf(sgn, d1.x, d0.x), __builtin_fmaf(sgn, d1.y, d0.y), __builtin_fmaf(sgn, d1.z, d0.z),              \
                     __builtin_fmaf(sgn, d1.w, d0.w)};  /* exact: sgn = +-1 */                                                       \
    }())
#ifdef AMOS_W24_EXP_NOT
#define AMOS_W24_TRANSFORM(rb, vb) {}
#else
#define AMOS_W24_TRANSFORM(rb, vb)                                                                                                   \
    {                                                                                                                                \
        const float *rp = AMOS_W24_R(rb);                                                                                            \
        float *vd = AMOS_W24_V(vb) + vdst;                                                                                           \
        const f32x4 s0 = AMOS_W24_SCOL(0), s2 = AMOS_W24_SCOL(2), s4 = AMOS_W24_SCOL(4);                                             \
        const f32x4 e = w24_fma(4.f, s0, w24_fma(-5.f, s2, s4));   /* the end position of the half row */                            \
        const f32x4 s1 = AMOS_W24_SCOL(1), s3 = AMOS_W24_SCOL(3);                                                                    \
        if (half == 0) {                                                                                                             \
            const f32x4 p = w24_fma(-4.f, s2, s4), q = w24_fma(-4.f, s1, s3);                                                        \
            *reinterpret_cast<f32x4 *>(vd) = e;                                                                                      \
            *reinterpret_cast<f32x4 *>(vd + kW24PosV) = p + q;                                                                       \
            *reinterpret_cast<f32x4 *>(vd + 2 * kW24PosV) = p - q;                                                                   \
        } else {                                                                                                                     \
            const f32x4 p = s3 - s1, q = s2 - s0;                                                                                    \
            *reinterpret_cast<f32x4 *>(vd) = w24_fma(2.f, q, p);                                                                     \
            *reinterpret_cast<f32x4 *>(vd + kW24PosV) = w24_fma(-2.f, q, p);                                                         \
            *reinterpret_cast<f32x4 *>(vd + 2 * kW24PosV) = e;                                                                       \
        }                                                                                                                            \
    }
#endif

    f32x16 acc[3][kCB];  // [position of the triple][cout block]
#define AMOS_W24_ZERO_ACC()                                                              \
    _Pragma("unroll") for (int p = 0; p < 3; p++)                                        \
        _Pragma("unroll") for (int j = 0; j < kCB; j++)                                  \
            _Pragma("unroll") for (int r = 0; r < 16; r++) acc[p][j][r] = 0.f;
    AMOS_W24_ZERO_ACC()
    const int foff = w24_swz(lane & 31, lane >> 5);  // this lane's 4 floats inside a position's block
    f32x4 fa0, fa1, fa2, fbE0[kCB], fbE1[kCB], fbE2[kCB], fbO0[kCB], fbO1[kCB], fbO2[kCB];
#define AMOS_W24_LDFRAG(fa, buf, p) fa = *reinterpret_cast<const f32x4 *>(AMOS_W24_V(buf) + (wave * 3 + (p)) * kW24PosV + foff);
#define AMOS_W24_MFMAS(fa, fb, p)                                                                    \
    _Pragma("unroll") for (int j = 0; j < kCB; j++) {                                                \
        acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb[j].x, acc[p][j], 0, 0, 0);         \
        acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb[j].y, acc[p][j], 0, 0, 0);         \
        acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb[j].z, acc[p][j], 0, 0, 0);         \
        acc[p][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb[j].w, acc[p][j], 0, 0, 0);         \
    }
// behind each of the next `n` MFMAs: up to `dr` LDS reads, `va` vector instructions, `dw` LDS writes, `vm` vector-memory requests
#define AMOS_W24_INTERLEAVE(n, dr, va, dw, vm)                                            \
    _Pragma("unroll") for (int q = 0; q < (n); q++) {                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                \
        if (dr) __builtin_amdgcn_sched_group_barrier(0x100, dr, 0);                       \
        if (va) __builtin_amdgcn_sched_group_barrier(0x002, va, 0);                       \
        if (dw) __builtin_amdgcn_sched_group_barrier(0x200, dw, 0);                       \
        if (vm) __builtin_amdgcn_sched_group_barrier(0x020, vm, 0);                       \
    }
    // One stage s (V buffer and raw-patch buffer s & 1):
    //   first part (16 MFMAs): positions 0 and 1 multiply (position 0's A fragment was read in the second part of stage s - 1); the A fragments
    //     of positions 1 and 2 are read; the wave's V blocks of stage s + 1 are made from the raw patch of stage s + 1 (complete since the
    //     barrier of stage s - 1) into the other V buffer; U(s + 1, position 2) is requested into the other parity's registers;
    //   barrier: this wave's pieces of the raw patch of stage s + 2 (requested a whole stage ago) have landed -- vmcnt(2): the two U loads of
    //     the first part are the only younger vector-memory operations; nobody reads the raw patch of stage s + 1 any more;
    //   second part (8 MFMAs): position 2 multiplies; position 0's A fragment of stage s + 1 is read; U(s + 2, positions 0 and 1) is requested
    //     into the registers this stage's first part has finished with; the raw patch of stage s + 3 is requested into the buffer the
    //     transform has just released.
#ifdef AMOS_W24_EXP_NOBAR
#define AMOS_W24_BARRIER(kVm) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
#define AMOS_W24_BARRIER(kVm) asm volatile("s_waitcnt vmcnt(" #kVm ") lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#ifndef AMOS_W24_VM_EVERY
#define AMOS_W24_VM_EVERY 2  /* MFMAs between two vector-memory requests of a stage's second part */
#endif
#ifndef AMOS_W24_SPLIT
#define AMOS_W24_SPLIT 1  /* 1: position 0 | barrier | positions 1, 2 (3 % faster);  0: positions 0, 1 | barrier | position 2 */
#endif
#if AMOS_W24_SPLIT == 0
#define AMOS_W24_STAGE(s, vb, fbC0, fbC1, fbC2, fbN2, kNext, kNext2, kNext3, kVm)         \
    {                                                                                     \
        AMOS_W24_LDFRAG(fa1, vb, 1);                                                      \
        AMOS_W24_LDFRAG(fa2, vb, 2);                                                      \
        if (kNext) AMOS_W24_TRANSFORM((vb) ^ 1, (vb) ^ 1);                                \
        if (kNext) AMOS_W24_FETCH_U(fbN2, (s) + 1, 2);                                    \
        AMOS_W24_MFMAS(fa0, fbC0, 0);                                                     \
        AMOS_W24_MFMAS(fa1, fbC1, 1);                                                     \
        /* all twelve LDS reads under the first three MFMAs, their consumers from the fourth on: no LDS latency between two MFMAs */ \
        AMOS_W24_INTERLEAVE(3, 4, 0, 0, 0);                                               \
        if (kNext) {                                                                      \
            AMOS_W24_INTERLEAVE(8, 0, 5, 0, 0);   /* column sums and outputs */           \
            AMOS_W24_INTERLEAVE(3, 0, 2, 1, 0);   /* three writes */                      \
            AMOS_W24_INTERLEAVE(2, 0, 0, 0, 1);   /* U of position 2 */                   \
        }                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                               \
        __builtin_amdgcn_sched_barrier(0);                                                \
        AMOS_W24_BARRIER(kVm)                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if (kNext) AMOS_W24_LDFRAG(fa0, (vb) ^ 1, 0);                                     \
        if (kNext2) AMOS_W24_FETCH_U(fbC0, (s) + 2, 0);                                   \
        if (kNext2) AMOS_W24_FETCH_U(fbC1, (s) + 2, 1);                                   \
        if (kNext3) AMOS_W24_FETCH_X((s) + 3, (vb) ^ 1);                                  \
        AMOS_W24_MFMAS(fa2, fbC2, 2);                                                     \
        AMOS_W24_INTERLEAVE(1, 1, 0, 0, 0);                                               \
        AMOS_W24_INTERLEAVE(7, 0, 0, 0, 2);                                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                \
        __builtin_amdgcn_sched_barrier(0);                                                \
    }
#define AMOS_W24_STAGE_EVEN(s, n1, n2, n3, vm) AMOS_W24_STAGE(s, 0, fbE0, fbE1, fbE2, fbO2, n1, n2, n3, vm)
#define AMOS_W24_STAGE_ODD(s, n1, n2, n3, vm) AMOS_W24_STAGE(s, 1, fbO0, fbO1, fbO2, fbE2, n1, n2, n3, vm)
#define AMOS_W24_PROLOGUE_U() { AMOS_W24_FETCH_U(fbE0, 0, 0); AMOS_W24_FETCH_U(fbE1, 0, 1); AMOS_W24_FETCH_U(fbE2, 0, 2); AMOS_W24_FETCH_U(fbO0, 1, 0); AMOS_W24_FETCH_U(fbO1, 1, 1); }
#define AMOS_W24_LOOP_VM 2
#define AMOS_W24_LOOP_VM_CB1 0

#else
/* the stage with every request's source spelt out: (kT) transform + next fragment; (kU1, u1, s1) U of positions 1, 2 of stage s1 from u1;
   (kU0, u0, s0) U of position 0 of stage s0 from u0; (kX, xo, sx) the raw patch of stage sx from xo -- the steady state and the tail, where
   a persistent group's requests go to its NEXT id */
#define AMOS_W24_STAGE_G(vb, fbC0, fbC1, fbC2, fbN1, fbN2, kT, kU1, u1, s1, kU0, u0, s0, kX, xo, sx, kVm) \
    {                                                                                     \
        AMOS_W24_LDFRAG(fa1, vb, 1);                                                      \
        AMOS_W24_LDFRAG(fa2, vb, 2);                                                      \
        if (kT) AMOS_W24_TRANSFORM(AMOS_W24_RB_READ(vb), (vb) ^ 1);                       \
        if (kU1) AMOS_W24_FETCH_UP(fbN1, u1, s1, 1);                                      \
        if (kU1) AMOS_W24_FETCH_UP(fbN2, u1, s1, 2);                                      \
        AMOS_W24_MFMAS(fa0, fbC0, 0);                                                     \
        AMOS_W24_INTERLEAVE(3, 4, 0, 0, 0);                                               \
        if (kT) {                                                                         \
            AMOS_W24_INTERLEAVE(2, 0, 12, 0, 1);                                          \
            AMOS_W24_INTERLEAVE(2, 0, 10, 1, 1);                                          \
            AMOS_W24_INTERLEAVE(1, 0, 4, 1, 0);                                           \
        }                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                \
        __builtin_amdgcn_sched_barrier(0);                                                \
        AMOS_W24_BARRIER(kVm)                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if (kT) AMOS_W24_LDFRAG(fa0, (vb) ^ 1, 0);                                        \
        if (kU0) AMOS_W24_FETCH_UP(fbC0, u0, s0, 0);                                      \
        if (kX) AMOS_W24_FETCH_XO(xo, sx, AMOS_W24_RB_READ(vb));                          \
        AMOS_W24_MFMAS(fa1, fbC1, 1);                                                     \
        AMOS_W24_MFMAS(fa2, fbC2, 2);                                                     \
        AMOS_W24_INTERLEAVE(1, 1, 0, 0, 0);                                               \
        _Pragma("unroll") for (int q = 0; q < 6; q++) {                                   \
            __builtin_amdgcn_sched_group_barrier(0x008, AMOS_W24_VM_EVERY, 0);            \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                            \
        }                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                               \
        __builtin_amdgcn_sched_barrier(0);                                                \
        AMOS_W24_RB_ADVANCE();                                                            \
    }
#define AMOS_W24_STAGE(s, vb, fbC0, fbC1, fbC2, fbN1, fbN2, kNext, kNext2, kNext3, kVm)   \
    {                                                                                     \
        AMOS_W24_LDFRAG(fa1, vb, 1);                                                      \
        AMOS_W24_LDFRAG(fa2, vb, 2);                                                      \
        if (kNext) AMOS_W24_TRANSFORM(AMOS_W24_RB_READ(vb), (vb) ^ 1);                                \
        if (kNext) AMOS_W24_FETCH_U(fbN1, (s) + 1, 1);                                    \
        if (kNext) AMOS_W24_FETCH_U(fbN2, (s) + 1, 2);                                    \
        AMOS_W24_MFMAS(fa0, fbC0, 0);                                                     \
        /* all twelve LDS reads under the first three MFMAs, their consumers from the fourth on: no LDS latency between two MFMAs */ \
        AMOS_W24_INTERLEAVE(3, 4, 0, 0, 0);                                               \
        if (kNext) {                                                                      \
            AMOS_W24_INTERLEAVE(2, 0, 12, 0, 1);                                          \
            AMOS_W24_INTERLEAVE(2, 0, 10, 1, 1);                                          \
            AMOS_W24_INTERLEAVE(1, 0, 4, 1, 0);                                           \
        }                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                \
        __builtin_amdgcn_sched_barrier(0);                                                \
        AMOS_W24_BARRIER(kVm)                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if (kNext) AMOS_W24_LDFRAG(fa0, (vb) ^ 1, 0);                                     \
        if (kNext2) AMOS_W24_FETCH_U(fbC0, (s) + 2, 0);                                   \
        if (kNext3) AMOS_W24_FETCH_X((s) + 2 + AMOS_W24_XAHEAD, AMOS_W24_RB_READ(vb));                                  \
        AMOS_W24_MFMAS(fa1, fbC1, 1);                                                     \
        AMOS_W24_MFMAS(fa2, fbC2, 2);                                                     \
        AMOS_W24_INTERLEAVE(1, 1, 0, 0, 0);                                               \
        /* the six vector-memory requests one per two MFMAs: all eight waves pass here together, and 48 KB requested within a few   \
           hundred cycles fill the CU's memory pipeline -- the waves then wait to ISSUE, with their MFMAs behind */               \
        _Pragma("unroll") for (int q = 0; q < 6; q++) {                                   \
            __builtin_amdgcn_sched_group_barrier(0x008, AMOS_W24_VM_EVERY, 0);            \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                            \
        }                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                               \
        __builtin_amdgcn_sched_barrier(0);                                                \
        AMOS_W24_RB_ADVANCE();                                                            \
    }
#define AMOS_W24_STAGE_EVEN(s, n1, n2, n3, vm) AMOS_W24_STAGE(s, 0, fbE0, fbE1, fbE2, fbO1, fbO2, n1, n2, n3, vm)
#define AMOS_W24_STAGE_ODD(s, n1, n2, n3, vm) AMOS_W24_STAGE(s, 1, fbO0, fbO1, fbO2, fbE1, fbE2, n1, n2, n3, vm)
#define AMOS_W24_STAGE_EVEN_G(...) AMOS_W24_STAGE_G(0, fbE0, fbE1, fbE2, fbO1, fbO2, __VA_ARGS__)
#define AMOS_W24_STAGE_ODD_G(...) AMOS_W24_STAGE_G(1, fbO0, fbO1, fbO2, fbE1, fbE2, __VA_ARGS__)
#define AMOS_W24_PROLOGUE_U() { AMOS_W24_FETCH_U(fbE0, 0, 0); AMOS_W24_FETCH_U(fbE1, 0, 1); AMOS_W24_FETCH_U(fbE2, 0, 2); AMOS_W24_FETCH_U(fbO0, 1, 0); }
#if AMOS_W24_XAHEAD == 1
#define AMOS_W24_LOOP_VM 4
#define AMOS_W24_LOOP_VM_CB1 2
#define AMOS_W24_RB_READ(vb) ((vb) ^ 1)   /* the raw patch of stage s + 1 sits in buffer (s + 1) & 1 */
#define AMOS_W24_RB_ADVANCE()
#else
#define AMOS_W24_LOOP_VM 14               /* younger than the patch request the barrier needs: 4 + 2 U and 4 X of the stage before, 4 U of this one */
#define AMOS_W24_LOOP_VM_CB1 0
#define AMOS_W24_RB_READ(vb) rbuf         /* ... in buffer (s + 1) % 3, a wave-uniform run-time index */
#define AMOS_W24_RB_ADVANCE() rbuf = rbuf == 2 ? 0 : rbuf + 1;
#endif
#endif

    // prologue: raw patches of stages 0 and 1; U of stage 0 (three positions) and of stage 1 (positions 0 and 1)
    AMOS_W24_FETCH_X(0, 0);
    AMOS_W24_FETCH_X(1, 1);
#if AMOS_W24_XAHEAD == 2
    AMOS_W24_FETCH_X(2, 2);
#endif
    AMOS_W24_PROLOGUE_U();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    AMOS_W24_TRANSFORM(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    AMOS_W24_LDFRAG(fa0, 0, 0);
#if AMOS_W24_XAHEAD == 1
    AMOS_W24_FETCH_X(2, 0);
#else
    AMOS_W24_FETCH_X(3, 0);
    int rbuf = 1;  // buffer of the raw patch of stage s + 1
#endif
#if AMOS_W24_SPLIT == 0 || AMOS_W24_XAHEAD != 1
    static_assert(!kPersist, "the persistent form is written for the default stage shape");
#endif
    for (;;) {  // one trip per id (a persistent group: until its walk ends)
    int s = 0;
    for (; s + 4 < a.stages; s += 2) {  // two stages per trip: the register names follow the stage parity
        if (kCB == 2) {
            AMOS_W24_STAGE_EVEN(s, true, true, true, AMOS_W24_LOOP_VM);
            AMOS_W24_STAGE_ODD(s + 1, true, true, true, AMOS_W24_LOOP_VM);
        } else {  // (half as many U loads are younger than the raw patch the barrier waits for)
            AMOS_W24_STAGE_EVEN(s, true, true, true, AMOS_W24_LOOP_VM_CB1);
            AMOS_W24_STAGE_ODD(s + 1, true, true, true, AMOS_W24_LOOP_VM_CB1);
        }
    }
    // the last four stages (the stage count is even and at least four: amos_mask_winograd_supported): less and less left to request for
    // THIS id -- a persistent group fills the free slots with the first requests of its next one
    bool haveNext = false;
    if (kPersist) {
        idN = next_valid(id + (int)gridDim.x, mbN, ntN);
        haveNext = idN >= 0;  // uniform over the work-group
        if (haveNext) {
            geometry(mbN, xoffN, colBaseN);
            usrcN = ubase + (size_t)ntN * a.stages * kW24StageU;
        }
    }
#if AMOS_W24_SPLIT == 1 && AMOS_W24_XAHEAD == 1
    if (kPersist) {
        //                 transform | U 1,2 of stage  | U 0 of stage      | raw patch of stage       | younger requests the barrier lets be
        AMOS_W24_STAGE_EVEN_G(true,    true, usrc, s + 1, true, usrc, s + 2,  true, xoff, s + 3,           4);
        AMOS_W24_STAGE_ODD_G(true,     true, usrc, s + 2, true, usrc, s + 3,  haveNext, xoffN, 0,          4);
        AMOS_W24_STAGE_EVEN_G(true,    true, usrc, s + 3, false, usrc, 0,     haveNext, xoffN, 1,          8);   // (the next id's first patch + this part's four U loads)
        AMOS_W24_STAGE_ODD_G(false,    false, usrc, 0,    false, usrc, 0,     false, xoffN, 0,             63);  // (nothing of this id is awaited any more)
    } else
#endif
    {
        AMOS_W24_STAGE_EVEN(s, true, true, AMOS_W24_XAHEAD == 1, 0);
        AMOS_W24_STAGE_ODD(s + 1, true, true, false, 0);
        AMOS_W24_STAGE_EVEN(s + 2, true, false, false, 0);
        AMOS_W24_STAGE_ODD(s + 3, false, false, false, 0);
    }

    // ---- epilogue.  This wave holds M[pr][pc] for pr = wave >> 1, pc = 3 * half + p.  Along A4's columns (M A4)[pr][j] = sum_pc M[pr][pc] A4^T[j][pc]:
    // half 0 contributes (M0 + M1 + M2, M1 - M2, M1 + M2, M1 - M2), half 1 (M3 + M4, 2 (M3 - M4), 4 (M3 + M4), 8 (M3 - M4) + M5).  Exchange image:
    // [wave][j][tile 32][cout 32] floats, one cout block of 32 per round (persistent form: [wave][j][tile 16][cout 32], 16 tiles of a cout
    // block per round, in the V tiles' LDS + the allocation's last 16 KB: the raw patches hold the next id's requests).  Accumulator register r
    // of lane l is tile row (r & 3) + 8 (r >> 2) + 4 (l >> 5), channel l & 31 of its 32 x 32 block.
#ifdef AMOS_W24_EXP_NOEPI
    if (acc[0][0][0] == 12345.f && acc[1][1][3] == 5.f && acc[2][0][7] == 1.f) a.y[t] = acc[0][1][1] + acc[1][0][2] + acc[2][1][3];  // keeps the accumulators alive
    return;
#endif
    // (raw barriers: a __syncthreads() would also wait for the next id's requests in flight)
#define AMOS_W24_EPI_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    AMOS_W24_EPI_BARRIER()  // every wave is done with the V tiles (and, one id per group, with the raw patches)
    constexpr int kRounds = kPersist ? 4 : kCB, kRoundTiles = kPersist ? 16 : 32;
    // finishing thread: channel quad of the round, tile of the round, output row of the tile (persistent form: + which two of the four pixels)
    const int oq = t & 7, otl = kPersist ? (t >> 3) & 15 : (t >> 3) & 31, oy = kPersist ? (t >> 7) & 1 : t >> 8, jp = kPersist ? t >> 8 : 0;
    auto exch = [&](int w) -> float * {  // wave w's part of the exchange image
        return kPersist ? smem24 + (w < 6 ? w * (4 * 16 * 32) : (2 * kW24StageV + 2 * kW24StageR) + (w - 6) * (4 * 16 * 32)) : smem24 + w * (4 * 32 * 32);
    };
#pragma unroll
    for (int rd = 0; rd < kRounds; rd++) {
        const int jb = kPersist ? rd >> 1 : rd, th = kPersist ? rd & 1 : 0;
        if (rd) AMOS_W24_EPI_BARRIER()  // the previous round's readers are done
        // the next id's four prologue U fragments: requested once half of the accumulators are dead (with them live through the whole
        // epilogue the kernel spilled), two rounds of exchange + stores ahead of their use
        if (kPersist && rd == 2 && haveNext) {
            AMOS_W24_FETCH_UP(fbE0, usrcN, 0, 0); AMOS_W24_FETCH_UP(fbE1, usrcN, 0, 1); AMOS_W24_FETCH_UP(fbE2, usrcN, 0, 2); AMOS_W24_FETCH_UP(fbO0, usrcN, 1, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (kPersist && (r >> 3) != th) continue;  // this round's 16 tiles: rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5) with r >> 3 == th
            const float m0 = acc[0][jb][r], m1 = acc[1][jb][r], m2 = acc[2][jb][r];
            float c0, c1, c2, c3;
            if (half == 0) {
                const float dlt = m1 - m2, sm = m1 + m2;
                c0 = m0 + sm; c1 = dlt; c2 = sm; c3 = dlt;
            } else {
                const float dlt = m0 - m1, sm = m0 + m1;
                c0 = sm; c1 = 2.f * dlt; c2 = 4.f * sm; c3 = 8.f * dlt + m2;
            }
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) - (kPersist ? 16 * th : 0), cc = lane & 31;
            float *dst = exch(wave) + row * 32 + cc;
            dst[0] = c0;
            dst[1 * kRoundTiles * 32] = c1;
            dst[2 * kRoundTiles * 32] = c2;
            dst[3 * kRoundTiles * 32] = c3;
        }
        AMOS_W24_EPI_BARRIER()
        // S[pr][j] = (M A4)[pr][j];  Y[0][j] = S[0][j] + S[1][j] + S[2][j];  Y[1][j] = S[1][j] - S[2][j] - S[3][j]
        const int n0 = (kCB == 1 ? nt * 32 : nt * kW24Cout + jb * 32) + 4 * oq;
        const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 yv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (kPersist && (j >> 1) != jp) continue;  // (persistent form: this thread finishes two of the four pixels)
            f32x4 sp[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int pr = oy + k;  // rows 0, 1, 2 for the upper output row, 1, 2, 3 for the lower one
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(exch(pr * 2 + 0) + (j * kRoundTiles + otl) * 32 + 4 * oq);
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(exch(pr * 2 + 1) + (j * kRoundTiles + otl) * 32 + 4 * oq);
                sp[k] = lo + hi;
            }
            yv[j] = oy == 0 ? (sp[0] + sp[1]) + sp[2] : (sp[0] - sp[1]) - sp[2];
        }
        const int T = mb * kW24Tiles + (kPersist ? 16 * th : 0) + otl;
        if (T < a.totalTiles) {
            const int b = T / a.tilesPerImage, rem = T - b * a.tilesPerImage, ty = rem / a.tilesX, tx = rem - ty * a.tilesX;
            const int py = 2 * ty + oy;
            if (py < a.H) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (kPersist && (j >> 1) != jp) continue;
                    const int px = 4 * tx + j;
                    if (px >= a.W) continue;
                    const size_t o = a.outBlocked ? ((((size_t)b * (a.N >> 3) + (n0 >> 3)) * a.H + py) * a.W + px) * 8 + (n0 & 7)
                                                  : ((size_t)(b * a.H + py) * a.W + px) * a.N + n0;
                    f32x4 v = yv[j] + bv;
                    if (a.res) v = v + *reinterpret_cast<const f32x4 *>(a.res + o);
                    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    *reinterpret_cast<f32x4 *>(a.y + o) = v;
                }
            }
        }
    }
    if (!kPersist || !haveNext) break;
    // ---- the next id: its first two raw patches and its four prologue U fragments were requested above
    id = idN; mb = mbN; nt = ntN;
#pragma unroll
    for (int r = 0; r < 4; r++) xoff[r] = xoffN[r];
    colBase = colBaseN;
    usrc = usrcN;
    AMOS_W24_SET_RSRC()
    AMOS_W24_ZERO_ACC()
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the patches have landed; the last round's readers are done with the V tiles
    AMOS_W24_TRANSFORM(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    AMOS_W24_LDFRAG(fa0, 0, 0);
    AMOS_W24_FETCH_X(2, 0);
    }  // ids
#undef AMOS_W24_V
#undef AMOS_W24_R
}

__global__ __launch_bounds__(kW24Threads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_winograd24_conv(const W24Args a)
{
    extern __shared__ __align__(16) float smem24[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) w24_run<1, false, 2>(a, smem24);
    else w24_run<0, false, 2>(a, smem24);
}

// 32 output channels per work-group: small launches (see w24_run, kCB)
__global__ __launch_bounds__(kW24Threads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_winograd24_conv_n32(const W24Args a)
{
    extern __shared__ __align__(16) float smem24[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) w24_run<1, false, 1>(a, smem24);
    else w24_run<0, false, 1>(a, smem24);
}

// one work-group per CU walking the ids (see w24_run): grid = a multiple of 8 work-groups, at most one per CU
__global__ __launch_bounds__(kW24Threads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_winograd24_conv_persistent(const W24Args a)
{
    extern __shared__ __align__(16) float smem24[];
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) w24_run<1, true, 2>(a, smem24);
    else w24_run<0, true, 2>(a, smem24);
}

const char *w24_variant_tag()
{
    return ""
#ifdef AMOS_W24_EXP_NOX
        " AMOS_W24_EXP_NOX"
#endif
#ifdef AMOS_W24_EXP_NOU
        " AMOS_W24_EXP_NOU"
#endif
#ifdef AMOS_W24_EXP_NOT
        " AMOS_W24_EXP_NOT"
#endif
#ifdef AMOS_W24_EXP_NOBAR
        " AMOS_W24_EXP_NOBAR"
#endif
#ifdef AMOS_W24_EXP_NOEPI
        " AMOS_W24_EXP_NOEPI"
#endif
        ;
}

}  // namespace amos

using namespace amos;

// -1 automatic (by launch size), 0 never, 1 always: AMOS_W24_PERSIST in the environment is the initial value, amos_mask_winograd24_persistent_mode sets it
static int g_w24_persist = -2;
static int w24_persist_mode()
{
    if (g_w24_persist == -2) {
        const char *env = getenv("AMOS_W24_PERSIST");
        g_w24_persist = env && (env[0] == '0' || env[0] == '1') ? env[0] - '0' : -1;
    }
    return g_w24_persist;
}

static int g_w24_narrow = -2;  // -1 by launch size, 0 always 64 output channels per work-group, 1 always 32 (AMOS_W24_NARROW: initial value)
static int w24_narrow_mode()
{
    if (g_w24_narrow == -2) {
        const char *env = getenv("AMOS_W24_NARROW");
        g_w24_narrow = env && (env[0] == '0' || env[0] == '1') ? env[0] - '0' : -1;
    }
    return g_w24_narrow;
}

extern "C" {

int amos_mask_winograd24_narrow_mode(int mode)
{
    const int before = w24_narrow_mode();
    if (mode >= -1 && mode <= 1) g_w24_narrow = mode;
    return before;
}

int amos_mask_winograd24_persistent_mode(int mode)
{
    const int before = w24_persist_mode();
    if (mode >= -1 && mode <= 1) g_w24_persist = mode;
    return before;
}

size_t amos_mask_winograd24_weight_floats(int cin, int cout)
{
    return amos_mask_winograd_supported(cin, cout) == AMOS_OK ? (size_t)kW24Pos * cin * cout : 0;
}

int amos_mask_winograd24_weights_device(void *stream, const float *d_w, float *d_u, int cin, int cout)
{
    if (!d_w || !d_u || amos_mask_winograd_supported(cin, cout) != AMOS_OK) {
        set_error("amos_mask_winograd24_weights_device: invalid argument (cin %% 16 == 0, cin >= 32, cout %% 64 == 0)");
        return AMOS_ERR_INVALID;
    }
    hipLaunchKernelGGL(k_winograd24_weights, dim3((unsigned)((cin * cout + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_w, d_u, cin, cout);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_mask_winograd24_conv_device(void *stream, const float *d_x, const float *d_u, const float *d_bias, const float *d_residual, float *d_y,
                                     int batch, int h, int w, int cin, int cout, int relu)
{
    return amos_mask_winograd24_conv_layout_device(stream, d_x, d_u, d_bias, d_residual, d_y, batch, h, w, cin, cout, relu, 0, 0);
}

int amos_mask_winograd24_conv_layout_device(void *stream, const float *d_x, const float *d_u, const float *d_bias, const float *d_residual, float *d_y,
                                            int batch, int h, int w, int cin, int cout, int relu, int in_blocked, int out_blocked)
{
    const long long xBytes = (long long)batch * h * w * cin * 4;
    if (!d_x || !d_u || !d_y || batch < 1 || h < 1 || w < 1 || amos_mask_winograd_supported(cin, cout) != AMOS_OK || xBytes > 0x7fffffffLL - 4096 ||
        ((uintptr_t)d_x | (uintptr_t)d_u | (uintptr_t)d_y | (uintptr_t)d_bias | (uintptr_t)d_residual) % 16 != 0) {
        set_error("amos_mask_winograd24_conv_device: invalid argument (cin %% 16 == 0, cin >= 32, cout %% 64 == 0, input below 2 GiB, 16-byte aligned channels-last tensors)");
        return AMOS_ERR_INVALID;
    }
    static DeviceOnce ldsAttr, ldsAttrP, ldsAttrN;  // per device (amos_common.h)
    const size_t lds = (size_t)kW24LdsFloats * sizeof(float);  // 128 KB
    AMOS_HIP_CHECK(set_max_dynamic_lds(ldsAttr, reinterpret_cast<const void *>(k_winograd24_conv), (int)lds, (hipStream_t)stream));
    W24Args a;
    a.x = d_x; a.u = d_u; a.bias = d_bias; a.res = d_residual; a.y = d_y;
    a.B = batch; a.H = h; a.W = w; a.C = cin; a.N = cout;
    a.tilesX = (w + 3) / 4; a.tilesY = (h + 1) / 2;
    a.tilesPerImage = a.tilesX * a.tilesY;
    a.totalTiles = batch * a.tilesPerImage;
    a.mBlocks = (a.totalTiles + kW24Tiles - 1) / kW24Tiles;
    // small launches: 32 output channels per work-group (twice the groups, half the MFMAs each) while 64-channel groups would leave most
    // CUs without one; amos_mask_winograd24_narrow_mode forces a side (tests, probes)
    const int narrowMode = w24_narrow_mode();
    // (measured at one frame, tools/r5_small_gemm_probe.py: 256 -> 256 at 69 x 69, 80 groups of 64: 59 us, 160 of 32: 45; 256 -> 384 at 69 x 69,
    // 120 of 64: 62, 240 of 32: 81 -- its 9.4 MB of U no longer fit an XCD's L2 when every CU streams them)
    const bool narrow = narrowMode >= 0 ? narrowMode != 0 : (long long)a.mBlocks * (cout / kW24Cout) <= 100;
    a.nTiles = cout / (narrow ? 32 : kW24Cout);
    a.stages = cin / kW24K;
    a.relu = relu;
    a.xBytes = (unsigned)xBytes;
    a.xPixelBytes = in_blocked ? 32u : (unsigned)cin * 4u;
    a.xStageBytes = in_blocked ? (unsigned)(h * w) * 32u : 32u;
    a.outBlocked = out_blocked ? 1 : 0;
    // ids: 8 XCDs x groups of (kW24Group m blocks x nTiles); the last group may be partly empty (those work-groups return at once)
    const int perXcd = (a.mBlocks + 7) / 8, groups = (perXcd + kW24Group - 1) / kW24Group;
    a.virtualBlocks = groups * kW24Group * a.nTiles * 8;
    const dim3 block(kW24Threads);
    // Persistent form (one work-group per CU walking the ids, the next id's first requests under the current one's last stages and
    // epilogue); AMOS_W24_PERSIST=1 / amos_mask_winograd24_persistent_mode(1) selects it (A/B runs, tests)
    const int forced = w24_persist_mode();
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount / 8 * 8;
        if (cus < 8) cus = 256;
    }
    // MEASURED (tools/r5_w24_persist.py, 64 frames, interleaved A/B, same bits): the persistent form is 3.5 - 5 % SLOWER on the large layers
    // (256 ch at 138 x 138: 4.74 against 4.58 ms) and 36 % slower where a CU gets only a few ids (512 ch at 18 x 18): what it hides of the
    // prologue it loses to the four-round epilogue, to the wait for the last stores before the next first transform (the vector-memory counter
    // retires in order) and to a static walk instead of the dispatcher's "next id to the first free CU".  So the automatic choice is the one
    // work-group per id form; the persistent one stays selectable (mode 1) and under test.
    const bool persist = forced == 1 && a.virtualBlocks > cus && !narrow;
    if (persist) {
        AMOS_HIP_CHECK(set_max_dynamic_lds(ldsAttrP, reinterpret_cast<const void *>(k_winograd24_conv_persistent), (int)lds, (hipStream_t)stream));
        hipLaunchKernelGGL(k_winograd24_conv_persistent, dim3((unsigned)std::min(cus, a.virtualBlocks)), block, lds, (hipStream_t)stream, a);
    } else if (narrow) {
        AMOS_HIP_CHECK(set_max_dynamic_lds(ldsAttrN, reinterpret_cast<const void *>(k_winograd24_conv_n32), (int)lds, (hipStream_t)stream));
        hipLaunchKernelGGL(k_winograd24_conv_n32, dim3((unsigned)a.virtualBlocks), block, lds, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(k_winograd24_conv, dim3((unsigned)a.virtualBlocks), block, lds, (hipStream_t)stream, a);
    }
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

}  // extern "C"

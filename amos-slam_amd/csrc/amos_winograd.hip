// amos_winograd.hip -- the stride-1 3 x 3 convolutions of the mask network (a15: protonet, prediction head, FPN prediction layers,
// the bottlenecks' conv2: 70 % of the network's multiply-accumulates, yolact.py:47-200, 265-400, backbone.py:60-200) as Winograd
// F(2 x 2, 3 x 3) on the fp32 MFMA units: 16 multiplies per 2 x 2 outputs and channel pair instead of 36 (2.25 x fewer), float32 in,
// float32 accumulate (cuDNN, the reference's backend, picks a Winograd form for these float32 layers too).
//
//   Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A          d: 4 x 4 input tile (pad 1), g: 3 x 3 filter, Y: 2 x 2 outputs
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// The 16 positions of the transformed domain are 16 independent GEMMs  M_p[tile][cout] = sum_c V_p[tile][c] U_p[cout][c]  that share
// their operands' source: one work-group (512 threads, 8 waves, one per CU by its 128 KB of LDS) owns 64 consecutive tiles (= 256
// output pixels) x 64 output channels x ALL 16 positions, so that the input transform is computed once per tile and channel and the
// output transform happens on chip.  Wave w multiplies positions 2w and 2w + 1 (2 x 2 x 2 accumulators of 32 x 32: 128 registers).
// K advances 8 input channels per stage:
//   U (weights, transformed once per layer by k_winograd_weights) -- position p is multiplied by ONE wave, so its U fragments never
//     meet another wave: they go from global memory straight into that wave's registers, 16 bytes per lane and fully coalesced (the
//     weight image is stored in fragment order), two stages ahead of their use; no LDS;
//   X (input pixels) -- every pixel of the work-group's patch is fetched ONCE per stage by LDS-DMA (buffer_load_dwordx4 ... lds: a
//     gather by per-lane offset; a pixel outside the image is an out-of-range offset, for which the buffer unit delivers zeros) into
//     a raw patch [4 rows][columns][8 channels]: tiles of a row segment share their columns (column = 2 x tile + c), a run of 64 tiles
//     that wraps to the next tile row (or frame) starts a new segment.  (The first version loaded each tile's 2 x 4 pixels per
//     thread: eight times the vector-memory instructions, 32 distinct cache lines each -- a fifth of the kernel's time.)
//   V -- thread (tile, channel quad, position row) reads its 2 x 4 pixels from the raw patch, 32 additions, four ds_write_b128 into
//     the V tile [16 positions][64 tiles][8], double-buffered; the MFMA's A fragments are ds_read_b128 from it (lanes 0-31 read k = 0..3,
//     lanes 32-63 k = 4..7 of a row and feed four v_mfma_f32_32x32x2_f32).  The two 4-float halves of V row r are stored swapped when
//     ((r >> 2) ^ (r >> 3)) & 1: conflict-free for the fragment reads and for the transform's writes, without padding.
// Epilogue: every wave reduces its two positions along the row of A (two partial sums), the partials meet in LDS (two rounds of 32
// tiles), every thread finishes A^T (.) A for one tile and four channels, adds bias (+ residual), applies ReLU and stores 16 bytes
// per output pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/amos_frontend.h"
#include "amos_common.h"

namespace amos {

constexpr int kWinoTiles = 64;    // tiles per work-group (rows of the 16 GEMMs)
constexpr int kWinoCout = 64;     // output channels per work-group
constexpr int kWinoK = 8;         // input channels per stage
constexpr int kWinoThreads = 512;
#ifndef AMOS_WINO_GROUP
#define AMOS_WINO_GROUP 4
#endif
#ifndef AMOS_WINO_MAP
#define AMOS_WINO_MAP 1
#endif
constexpr int kWinoGroup = AMOS_WINO_GROUP;  // consecutive work-groups of an XCD that share a cout tile (experiments: tools/wino_variants.sh G<n>)
constexpr int kWinoStageU = 16 * kWinoCout * kWinoK;   // floats of one stage's U image (32 KB)
constexpr int kWinoStageV = 16 * kWinoTiles * kWinoK;  // floats of one stage's V tile (32 KB)
constexpr int kWinoRawCols = 256;                      // columns of the raw patch (64 tiles in up to 64 one-tile segments: 2 x 64 + 2 x 64)
constexpr int kWinoRawRow = kWinoRawCols * kWinoK;     // floats of one patch row
constexpr int kWinoStageR = 4 * kWinoRawRow;           // floats of one raw patch (32 KB)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef AMOS_WINO_EXP_TIMING  // experiment build (tools/wino_variants.sh TIMING, tools/wino_timing.py): per-wave cycle sums of the stage's phases
__device__ unsigned long long *g_wino_timing = nullptr;  // [work-group][wave][8]: first half, wait + barrier, second half, stages, prologue + loop, epilogue, start
#define AMOS_WINO_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime();
#else
#define AMOS_WINO_T(var)
#endif

struct WinoArgs {
    const float *x, *u, *bias, *res;
    float *y;
    int B, H, W, C, N;          // frames, image size, input channels, output channels
    int tilesX, tilesY, tilesPerImage, totalTiles;
    int mBlocks, nTiles, stages, relu;
    unsigned xBytes;            // size of x in bytes (the buffer descriptor's range)
};

// V tile: element (row, k) of a 64-row position block lives at float offset row * 8 + 4 * ((k >> 2) ^ f(row)) + (k & 3)
__device__ __host__ __forceinline__ int wino_swz(int row, int half) { return row * 8 + 4 * (half ^ (((row >> 2) ^ (row >> 3)) & 1)); }
// raw patch: 16-byte unit of (column, channel quad) inside a patch row -- blocks of 32 columns, inside a block [column parity][column / 2][quad]:
// one LDS-DMA instruction fills a block lane-linearly with lane pairs fetching the two quads of one pixel (32 contiguous bytes), and the
// transform's reads (lanes = consecutive tiles x both quads: every second column) are conflict-free
__device__ __forceinline__ int wino_raw_unit(int col, int quad) { return (col >> 5) * 64 + (col & 1) * 32 + ((col & 31) >> 1) * 2 + quad; }

// Weights [cout][3][3][cin] (a channels-last Conv2d weight) -> U = G g G^T in the order the MFMA's B fragments are read: for every
// (cout tile nt, stage s, position p, 32-channel block j) 64 lanes x 4 floats, lane l = U_p[nt * 64 + j * 32 + (l & 31)][s * 8 + 4 (l >> 5) ..+3];
// sums in double, one rounding.
__global__ __launch_bounds__(256) void k_winograd_weights(const float *__restrict__ w, float *__restrict__ u, int cin, int cout)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cin * cout) return;
    const int c = idx % cin, n = idx / cin;
    double g[3][3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) g[a][b] = (double)w[((size_t)(n * 3 + a) * 3 + b) * cin + c];
    double t[4][3];  // G g
    for (int b = 0; b < 3; b++) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5 * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5 * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const int stages = cin / kWinoK, nt = n / kWinoCout, nr = n % kWinoCout, s = c / kWinoK, k = c % kWinoK;
    float *img = u + (size_t)(nt * stages + s) * kWinoStageU + (size_t)((nr >> 5) * 64 + (k >> 2) * 32 + (nr & 31)) * 4 + (k & 3);
    for (int a = 0; a < 4; a++) {
        const double r0 = t[a][0], r1 = 0.5 * (t[a][0] + t[a][1] + t[a][2]), r2 = 0.5 * (t[a][0] - t[a][1] + t[a][2]), r3 = t[a][2];
        const double r[4] = {r0, r1, r2, r3};
        for (int b = 0; b < 4; b++) img[(size_t)(a * 4 + b) * 512] = (float)r[b];
    }
}

__global__ __launch_bounds__(kWinoThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_winograd_conv(const WinoArgs a)
{
    // V tiles of two stages (64 KB), raw patches of two stages (64 KB); the epilogue's exchange reuses all of it
    extern __shared__ __align__(16) float smem[];
#define AMOS_WINO_V(buf) (smem + (buf) * kWinoStageV)
#define AMOS_WINO_R(buf) (smem + 2 * kWinoStageV + (buf) * kWinoStageR)
    // id -> (m block, n tile): ids are dealt round-robin over the 8 XCDs by the hardware.  Map 1 (default): an XCD owns a contiguous run
    // of m blocks and runs the n tiles of kGroup (4) m blocks side by side -- the patch of an m block is fetched from memory once and
    // found in the XCD's L2 by the work-groups of the other n tiles and, for two of its four rows, by the next tile row's m block:
    // measured on proto_net 256 -> 256 at 138 x 138, 32 frames (tools/r4_wino_map.sh): L2 fetch traffic 1.93 GB per launch (3.1 x the
    // input tensor) against 7.92 GB (12.7 x) for map 0 with groups of 32, 2.88 against 2.95 ms.  Map 0 (round 3): m blocks dealt round-robin
    // over the XCDs, kGroup consecutive work-groups of an XCD share the n tile (the 1 MB weight slice)
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    constexpr int kGroup = kWinoGroup;
    const int per = kGroup * a.nTiles, grp = seq / per, in = seq - grp * per;
#if AMOS_WINO_MAP == 1
    // an XCD owns a CONTIGUOUS run of m blocks (consecutive tile rows share two of their four patch rows: the second one finds them in
    // this XCD's L2), and the n tiles of kGroup m blocks run side by side on it (the patch is fetched from memory by the first of them)
    const int perXcd = (a.mBlocks + 7) >> 3, mbLocal = grp * kGroup + in % kGroup;
    const int nt = in / kGroup, mb = xcd * perXcd + mbLocal;
    if (mbLocal >= perXcd || mb >= a.mBlocks) return;
#else
    const int nt = in / kGroup, mb = (grp * kGroup + in % kGroup) * 8 + xcd;
    if (mb >= a.mBlocks) return;
#endif
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    // ---- geometry of the work-group's tile run: segments of tiles of one tile row.  This thread is (a) the loader of raw-patch column
    // `col` (block `wave`, lane order [parity][half][quad]) for all four patch rows and (b) the transformer of tile `tl`, channel quad
    // `quad`, position row `prow`.
    const int quad = lane & 1, tl = (lane >> 1) + 32 * (wave & 1), prow = wave >> 1;
    const int col = 32 * wave + 2 * ((lane >> 1) & 15) + (lane >> 5);
    int xoff[4];     // loader: byte offset of (patch row r, column col, channel quad) in x, or the buffer's size (zeros) when there is no such pixel
    int colBase = 0; // transformer: first patch column of tile tl
    int nCols = 0;   // columns the run occupies (wave-uniform)
    {
        const int T0 = mb * kWinoTiles, nT = min(kWinoTiles, a.totalTiles - T0);
        int b = T0 / a.tilesPerImage, rem = T0 - b * a.tilesPerImage, ty = rem / a.tilesX, tx = rem - ty * a.tilesX;
#pragma unroll
        for (int r = 0; r < 4; r++) xoff[r] = (int)a.xBytes;
        for (int t0 = 0, cb = 0; t0 < nT;) {  // (b, ty, tx) = the segment's first tile, t0 its index in the run, cb its first column
            const int n = min(a.tilesX - tx, nT - t0);
            if (col >= cb && col < cb + 2 * n + 2) {
                const int lc = col - cb, tloc = min(lc >> 1, n - 1), ix = 2 * (tx + tloc) - 1 + (lc - 2 * tloc);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int iy = 2 * ty - 1 + r;
                    if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                        xoff[r] = (int)((((unsigned)(b * a.H + iy) * a.W + ix) * a.C + 4 * quad) * 4u);
                }
            }
            if (tl >= t0 && tl < t0 + n) colBase = cb + 2 * (tl - t0);
            t0 += n;
            cb += 2 * n + 2;
            nCols = cb;
            tx = 0;
            if (++ty == a.tilesY) { ty = 0; b++; }
        }
        if (tl >= nT) colBase = 0;  // tiles past the end of the tensor: any valid patch position (their results are not stored)
    }
    // (every wave issues its four LDS-DMA requests, also when its column block lies past the run's columns -- out-of-range offsets, zeros, no
    // memory traffic: the compiler counts vector-memory operations per code path when it places the waits for the U loads, and a branch
    // around the requests made it assume the shorter path)
    (void)nCols;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.xBytes, 0x00020000);
    typedef __attribute__((address_space(3))) void *LdsPtr;
    // the transformer's eight reads: rows (r0, r1) of B^T d B row `prow` (0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3), columns colBase + c
    const int r0 = prow == 0 ? 0 : (prow == 2 ? 2 : 1), r1 = prow == 0 ? 2 : (prow == 1 ? 2 : (prow == 2 ? 1 : 3));
    int rsrc0[4];  // float offsets inside a raw patch of (row r0, column colBase + c); row r1 is (r1 - r0) * kWinoRawRow further
#pragma unroll
    for (int c = 0; c < 4; c++) rsrc0[c] = r0 * kWinoRawRow + wino_raw_unit(colBase + c, quad) * 4;
    const int rdelta = (r1 - r0) * kWinoRawRow;
    const float sgn = prow == 1 ? 1.f : -1.f;                        // the row's second term is added (row 1) or subtracted
    const int vdst = (prow * 4 * kWinoTiles) * kWinoK + wino_swz(tl, quad);  // + j * 64 rows for position prow * 4 + j
    // U fragments of this wave's two positions: [position of the pair][cout block] x 16 bytes per lane and stage
    const float *usrc = a.u + (size_t)nt * a.stages * kWinoStageU + (size_t)(wave * 2) * 512 + lane * 4;

    // raw patch of stage s -> LDS (one LDS-DMA gather per patch row; 64 lanes x 16 bytes land lane-linearly in the row's block `wave`)
#ifdef AMOS_WINO_EXP_NOX  /* timing experiments (results are wrong): tools/wino_variants.sh */
#define AMOS_WINO_FETCH_X(s, buf) {}
#else
#define AMOS_WINO_FETCH_X(s, buf)                                                                                                    \
    {                                                                                                                                \
        _Pragma("unroll") for (int r = 0; r < 4; r++)                                                                                \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (LdsPtr)(AMOS_WINO_R(buf) + r * kWinoRawRow + wave * 256), 16, xoff[r], (s) * (kWinoK * 4), 0, 0); \
    }
#endif
    // U fragments of stage s, position p of the pair -> registers
#ifdef AMOS_WINO_EXP_NOU
#define AMOS_WINO_FETCH_U(fb, s, p) { _Pragma("unroll") for (int j = 0; j < 2; j++) fb[j] = f32x4{(float)(s), (float)lane, 1.f, (float)(p)}; }
#else
#define AMOS_WINO_FETCH_U(fb, s, p)                                                                                                  \
    {                                                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; j++)                                                                                \
            fb[j] = *reinterpret_cast<const f32x4 *>(usrc + (size_t)(s) * kWinoStageU + ((p) * 2 + j) * 256);                        \
    }
#endif
    // raw patch (buffer rb) -> this thread's row of B^T d B -> V tile (buffer vb).  Written column by column (2, 0, 1, 3) so that few
    // values are alive at a time: t_c = d[r0][c] +- d[r1][c];  V_0 = t0 - t2, V_1 = t1 + t2, V_2 = t2 - t1, V_3 = t1 - t3.
#define AMOS_WINO_TCOL(c)                                                                                                            \
    ([&]() -> f32x4 {                                                                                                                \
        const f32x4 d0 = *reinterpret_cast<const f32x4 *>(rp + rsrc0[c]), d1 = *reinterpret_cast<const f32x4 *>(rp + rsrc0[c] + rdelta); \
        return f32x4{__builtin_fmaf(sgn, d1.x, d0.x), __builtin_fmaf(sgn, d1.y, d0.y), __builtin_fmaf(sgn, d1.z, d0.z),              \
                     __builtin_fmaf(sgn, d1.w, d0.w)};  /* exact: sgn = +-1 */                                                       \
    }())
#ifdef AMOS_WINO_EXP_NOT
#define AMOS_WINO_TRANSFORM(rb, vb) {}
#else
#define AMOS_WINO_TRANSFORM(rb, vb)                                                                                                  \
    {                                                                                                                                \
        const float *rp = AMOS_WINO_R(rb);                                                                                           \
        float *vd = AMOS_WINO_V(vb) + vdst;                                                                                          \
        const f32x4 t2 = AMOS_WINO_TCOL(2);                                                                                          \
        const f32x4 t0 = AMOS_WINO_TCOL(0);                                                                                          \
        *reinterpret_cast<f32x4 *>(vd) = t0 - t2;                                                                                    \
        const f32x4 t1 = AMOS_WINO_TCOL(1);                                                                                          \
        *reinterpret_cast<f32x4 *>(vd + 1 * kWinoTiles * kWinoK) = t1 + t2;                                                          \
        *reinterpret_cast<f32x4 *>(vd + 2 * kWinoTiles * kWinoK) = t2 - t1;                                                          \
        const f32x4 t3 = AMOS_WINO_TCOL(3);                                                                                          \
        *reinterpret_cast<f32x4 *>(vd + 3 * kWinoTiles * kWinoK) = t1 - t3;                                                          \
    }
#endif

    f32x16 acc[2][2][2];  // [position of the pair][tile block][cout block]
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[p][i][j][r] = 0.f;
    const int foff = wino_swz(lane & 31, lane >> 5);  // this lane's 4 floats inside a 32-row block
    // A fragments: one register set per position of the pair, so the LDS reads of one position are in flight under the MFMAs of the
    // other; B fragments: [stage parity][position of the pair][cout block], loaded two stages (position 0) / one and a half (position 1) ahead
    f32x4 fa0[2], fa1[2], fbE0[2], fbE1[2], fbO0[2], fbO1[2];
#define AMOS_WINO_LDFRAG(fa, buf, p)                                                                                                  \
    {                                                                                                                                 \
        const int pos = wave * 2 + (p);                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; i++) fa[i] = *reinterpret_cast<const f32x4 *>(AMOS_WINO_V(buf) + (pos * kWinoTiles + i * 32) * kWinoK + foff); \
    }
#define AMOS_WINO_MFMAS(fa, fb, p)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; i++) _Pragma("unroll") for (int j = 0; j < 2; j++) {           \
        acc[p][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[p][i][j], 0, 0, 0);       \
        acc[p][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[p][i][j], 0, 0, 0);       \
        acc[p][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[p][i][j], 0, 0, 0);       \
        acc[p][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[p][i][j], 0, 0, 0);       \
    }
// `each` instructions of the named kind behind each of the next `n` MFMAs
#define AMOS_WINO_INTERLEAVE(mask, n, each)                                               \
    _Pragma("unroll") for (int q = 0; q < (n); q++) {                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                \
        __builtin_amdgcn_sched_group_barrier(mask, each, 0);                              \
    }
    // One stage s (V buffer and raw-patch buffer s & 1) = two halves of 16 MFMAs around ONE barrier.
    //   first half: position 0 multiplies (its A fragments were read in the second half of stage s - 1); position 1's A fragments are
    //     read; the V tile of stage s + 1 is made from the raw patch of stage s + 1 (complete since the barrier of stage s - 1) into the
    //     other V buffer; U(s + 1, position 1) is requested into the other parity's position-1 registers (last read in the second
    //     half of stage s - 1);
    //   barrier: V(s + 1) is complete; this wave's pieces of the raw patch of stage s + 2 (requested a whole stage ago) have landed --
    //     vmcnt(2): the only vector-memory operations that may be younger than that request are the four U loads since, of which the
    //     compiler may have moved two ahead of it; nobody reads V(s) or the raw patch of stage s + 1 any more;
    //   second half: position 1 multiplies; position 0's A fragments of stage s + 1 are read; the raw patch of stage s + 3 is requested
    //     into the buffer the transform has just released; U(s + 2, position 0) is requested into the registers position 0 of this
    //     stage has just finished with (same stage parity).
    // No LDS latency and no global latency sits between two MFMAs.  kNext / kNext2 / kNext3: stages s + 1 / s + 2 / s + 3 exist.
#ifdef AMOS_WINO_EXP_NOBAR
#define AMOS_WINO_BARRIER(kVm) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
#define AMOS_WINO_BARRIER(kVm) asm volatile("s_waitcnt vmcnt(" #kVm ") lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#ifdef AMOS_WINO_EXP_TIMING
#define AMOS_WINO_TSUM tsum0 += tb - ta; tsum1 += tc - tb; tsum2 += td - tc; tsum3 += 1;
#else
#define AMOS_WINO_TSUM
#endif
#define AMOS_WINO_STAGE(s, vb, fbC0, fbC1, fbN1, kNext, kNext2, kNext3, kVm)              \
    {                                                                                     \
        AMOS_WINO_T(ta)                                                                   \
        AMOS_WINO_LDFRAG(fa1, vb, 1);                                                     \
        if (kNext) AMOS_WINO_TRANSFORM((vb) ^ 1, (vb) ^ 1);                               \
        if (kNext) AMOS_WINO_FETCH_U(fbN1, (s) + 1, 1);                                   \
        AMOS_WINO_MFMAS(fa0, fbC0, 0);                                                    \
        AMOS_WINO_INTERLEAVE(0x100, 2, 1);  /* position 1's A fragments */                \
        if (kNext) {                                                                      \
            AMOS_WINO_INTERLEAVE(0x100, 2, 2);  /* columns 2 and 0 */                     \
            AMOS_WINO_INTERLEAVE(0x002, 1, 6);                                            \
            AMOS_WINO_INTERLEAVE(0x200, 1, 1);                                            \
            AMOS_WINO_INTERLEAVE(0x100, 1, 2);  /* column 1 */                            \
            AMOS_WINO_INTERLEAVE(0x002, 1, 6);                                            \
            AMOS_WINO_INTERLEAVE(0x200, 2, 1);                                            \
            AMOS_WINO_INTERLEAVE(0x100, 1, 2);  /* column 3 */                            \
            AMOS_WINO_INTERLEAVE(0x002, 1, 4);                                            \
            AMOS_WINO_INTERLEAVE(0x200, 1, 1);                                            \
            AMOS_WINO_INTERLEAVE(0x020, 2, 1);                                            \
        }                                                                                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                               \
        __builtin_amdgcn_sched_barrier(0);                                                \
        AMOS_WINO_T(tb)                                                                   \
        AMOS_WINO_BARRIER(kVm)                                                            \
        AMOS_WINO_T(tc)                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if (kNext) AMOS_WINO_LDFRAG(fa0, (vb) ^ 1, 0);                                    \
        if (kNext3) AMOS_WINO_FETCH_X((s) + 3, (vb) ^ 1);                                 \
        if (kNext2) AMOS_WINO_FETCH_U(fbC0, (s) + 2, 0);                                  \
        AMOS_WINO_MFMAS(fa1, fbC1, 1);                                                    \
        AMOS_WINO_INTERLEAVE(0x100, 2, 1);                                                \
        AMOS_WINO_INTERLEAVE(0x020, 6, 1);                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                               \
        __builtin_amdgcn_sched_barrier(0);                                                \
        AMOS_WINO_T(td)                                                                   \
        AMOS_WINO_TSUM                                                                    \
    }
#define AMOS_WINO_STAGE_EVEN(s, n1, n2, n3, vm) AMOS_WINO_STAGE(s, 0, fbE0, fbE1, fbO1, n1, n2, n3, vm)
#define AMOS_WINO_STAGE_ODD(s, n1, n2, n3, vm) AMOS_WINO_STAGE(s, 1, fbO0, fbO1, fbE1, n1, n2, n3, vm)

#ifdef AMOS_WINO_EXP_TIMING
    unsigned long long tsum0 = 0, tsum1 = 0, tsum2 = 0, tsum3 = 0;
    const unsigned long long tstart = __builtin_amdgcn_s_memtime();
#endif
    // prologue: raw patches of stages 0 and 1; U of stage 0 (both positions) and of stage 1 (position 0)
    AMOS_WINO_FETCH_X(0, 0);
    AMOS_WINO_FETCH_X(1, 1);
    AMOS_WINO_FETCH_U(fbE0, 0, 0);
    AMOS_WINO_FETCH_U(fbE1, 0, 1);
    AMOS_WINO_FETCH_U(fbO0, 1, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    AMOS_WINO_TRANSFORM(0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    AMOS_WINO_LDFRAG(fa0, 0, 0);
    AMOS_WINO_FETCH_X(2, 0);
    int s = 0;
    for (; s + 4 < a.stages; s += 2) {  // two stages per trip: the register names follow the stage parity
        AMOS_WINO_STAGE_EVEN(s, true, true, true, 2);
        AMOS_WINO_STAGE_ODD(s + 1, true, true, true, 2);
    }
    // the last four stages (the stage count is even and at least four: amos_mask_winograd_supported): less and less left to request.
    // (One straight-line tail: with a tail per stage count the register allocator spills the accumulators.)
    AMOS_WINO_STAGE_EVEN(s, true, true, true, 0);
    AMOS_WINO_STAGE_ODD(s + 1, true, true, false, 0);
    AMOS_WINO_STAGE_EVEN(s + 2, true, false, false, 0);
    AMOS_WINO_STAGE_ODD(s + 3, false, false, false, 0);
#undef AMOS_WINO_STAGE_EVEN
#undef AMOS_WINO_STAGE_ODD
#ifdef AMOS_WINO_EXP_TIMING
    const unsigned long long tloop = __builtin_amdgcn_s_memtime();
#endif
#undef AMOS_WINO_LDFRAG
#undef AMOS_WINO_MFMAS
#undef AMOS_WINO_INTERLEAVE
#undef AMOS_WINO_STAGE
#undef AMOS_WINO_BARRIER
#undef AMOS_WINO_FETCH_U
#undef AMOS_WINO_FETCH_X
#undef AMOS_WINO_TRANSFORM
#undef AMOS_WINO_TCOL
#undef AMOS_WINO_V
#undef AMOS_WINO_R

    // ---- epilogue.  Position (pr, pc) = (wave >> 1, 2 (wave & 1) + p).  Along A's columns: (M A)[pr][0] = M0 + M1 + M2, [1] = M1 - M2 - M3,
    // so this wave's pair contributes c0 = M0 + M1, c1 = M1 (pc = 0, 1) or c0 = M2, c1 = -M2 - M3 (pc = 2, 3).  Exchange image:
    // [wave][c][tile 32][cout 64] floats, one tile block of 32 per round.  Accumulator register r of lane l is tile row
    // (r & 3) + 8 (r >> 2) + 4 (l >> 5), channel l & 31 of its 32 x 32 block.
    const bool upper = (wave & 1) != 0;
    const int oq = t & 15, otl = t >> 4;  // finishing thread: channel quad and tile of the round
    const int n0 = nt * kWinoCout + 4 * oq;
    const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4 *>(a.bias + n0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; i++) {
        if (i) __syncthreads();  // the previous round's readers are done
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float m0 = acc[0][i][j][r], m1 = acc[1][i][j][r];
                const float c0 = upper ? m0 : m0 + m1, c1 = upper ? -m0 - m1 : m1;
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = j * 32 + (lane & 31);
                smem[((wave * 2 + 0) * 32 + row) * kWinoCout + col] = c0;
                smem[((wave * 2 + 1) * 32 + row) * kWinoCout + col] = c1;
            }
        __syncthreads();
        // Y[0][j] = sum over pr = 0, 1, 2 of (M A)[pr][j];  Y[1][j] = (M A)[1][j] - (M A)[2][j] - (M A)[3][j]
        f32x4 ma[4][2];
#pragma unroll
        for (int pr = 0; pr < 4; pr++)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(&smem[(((pr * 2 + 0) * 2 + c) * 32 + otl) * kWinoCout + 4 * oq]);
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(&smem[(((pr * 2 + 1) * 2 + c) * 32 + otl) * kWinoCout + 4 * oq]);
                ma[pr][c] = lo + hi;
            }
        const int T = mb * kWinoTiles + i * 32 + otl;
        if (T < a.totalTiles) {
            const int b = T / a.tilesPerImage, rem = T - b * a.tilesPerImage, ty = rem / a.tilesX, tx = rem - ty * a.tilesX;
#pragma unroll
            for (int oy = 0; oy < 2; oy++)
#pragma unroll
                for (int ox = 0; ox < 2; ox++) {
                    const int py = 2 * ty + oy, px = 2 * tx + ox;
                    if (py >= a.H || px >= a.W) continue;
                    f32x4 v = oy == 0 ? (ma[0][ox] + ma[1][ox]) + ma[2][ox] : (ma[1][ox] - ma[2][ox]) - ma[3][ox];
                    const size_t o = ((size_t)(b * a.H + py) * a.W + px) * a.N + n0;
                    v = v + bv;
                    if (a.res) v = v + *reinterpret_cast<const f32x4 *>(a.res + o);
                    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    *reinterpret_cast<f32x4 *>(a.y + o) = v;
                }
        }
    }
#ifdef AMOS_WINO_EXP_TIMING
    if (g_wino_timing && lane == 0 && blockIdx.x < 4096) {
        unsigned long long *o = g_wino_timing + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = tsum0; o[1] = tsum1; o[2] = tsum2; o[3] = tsum3; o[4] = tloop - tstart; o[5] = __builtin_amdgcn_s_memtime() - tloop; o[6] = tstart;
    }
#endif
}

}  // namespace amos

using namespace amos;

extern "C" {

#ifdef AMOS_WINO_EXP_TIMING
int amos_mask_winograd_timing_buffer(unsigned long long *d_buf)  // 4096 x 8 x 8 values, or NULL
{
    AMOS_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_wino_timing), &d_buf, sizeof(d_buf)));
    return AMOS_OK;
}
#endif

int amos_mask_winograd_supported(int cin, int cout)
{
    // an even number of 8-channel stages, at least four (the kernel's pipeline is unrolled by two with a four-stage tail)
    return (cin >= 4 * kWinoK && cin % (2 * kWinoK) == 0 && cout >= kWinoCout && cout % kWinoCout == 0) ? AMOS_OK : AMOS_ERR_INVALID;
}

size_t amos_mask_winograd_weight_floats(int cin, int cout)
{
    return amos_mask_winograd_supported(cin, cout) == AMOS_OK ? (size_t)16 * cin * cout : 0;
}

int amos_mask_winograd_weights_device(void *stream, const float *d_w, float *d_u, int cin, int cout)
{
    if (!d_w || !d_u || amos_mask_winograd_supported(cin, cout) != AMOS_OK) {
        set_error("amos_mask_winograd_weights_device: invalid argument (cin %% 16 == 0, cin >= 32, cout %% 64 == 0)");
        return AMOS_ERR_INVALID;
    }
    hipLaunchKernelGGL(k_winograd_weights, dim3((unsigned)((cin * cout + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_w, d_u, cin, cout);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_mask_winograd_conv_device(void *stream, const float *d_x, const float *d_u, const float *d_bias, const float *d_residual, float *d_y,
                                   int batch, int h, int w, int cin, int cout, int relu)
{
    const long long xBytes = (long long)batch * h * w * cin * 4;
    if (!d_x || !d_u || !d_y || batch < 1 || h < 1 || w < 1 || amos_mask_winograd_supported(cin, cout) != AMOS_OK || xBytes > 0x7fffffffLL - 4096 ||
        ((uintptr_t)d_x | (uintptr_t)d_u | (uintptr_t)d_y | (uintptr_t)d_bias | (uintptr_t)d_residual) % 16 != 0) {
        set_error("amos_mask_winograd_conv_device: invalid argument (cin %% 16 == 0, cin >= 32, cout %% 64 == 0, input below 2 GiB, 16-byte aligned channels-last tensors)");
        return AMOS_ERR_INVALID;
    }
    static DeviceOnce ldsAttr;  // per device (amos_common.h)
    const size_t lds = (size_t)2 * (kWinoStageV + kWinoStageR) * sizeof(float);  // two V tiles + two raw patches = 128 KB
    AMOS_HIP_CHECK(set_max_dynamic_lds(ldsAttr, reinterpret_cast<const void *>(k_winograd_conv), (int)lds, (hipStream_t)stream));
    WinoArgs a;
    a.x = d_x; a.u = d_u; a.bias = d_bias; a.res = d_residual; a.y = d_y;
    a.B = batch; a.H = h; a.W = w; a.C = cin; a.N = cout;
    a.tilesX = (w + 1) / 2; a.tilesY = (h + 1) / 2;
    a.tilesPerImage = a.tilesX * a.tilesY;
    a.totalTiles = batch * a.tilesPerImage;
    a.mBlocks = (a.totalTiles + kWinoTiles - 1) / kWinoTiles;
    a.nTiles = cout / kWinoCout;
    a.stages = cin / kWinoK;
    a.relu = relu;
    a.xBytes = (unsigned)xBytes;
    // ids: 8 XCDs x groups of (kWinoGroup m blocks x nTiles); the last group may be partly empty (those work-groups return at once)
    const int perXcd = (a.mBlocks + 7) / 8, groups = (perXcd + kWinoGroup - 1) / kWinoGroup;
    const dim3 grid((unsigned)(groups * kWinoGroup * a.nTiles * 8)), block(kWinoThreads);
    hipLaunchKernelGGL(k_winograd_conv, grid, block, lds, (hipStream_t)stream, a);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

}  // extern "C"

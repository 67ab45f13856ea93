// amos_winograd.hip -- the stride-1 3 x 3 convolutions of the mask network (a15: protonet, prediction head, FPN prediction layers,
// the bottlenecks' conv2: 70 % of the network's multiply-accumulates, yolact.py:47-200, 265-400, backbone.py:60-200) as Winograd
// F(2 x 2, 3 x 3) on the fp32 MFMA units: 16 multiplies per 2 x 2 outputs and channel pair instead of 36 (2.25 x fewer), float32 in,
// float32 accumulate (cuDNN, the reference's backend, picks a Winograd form for these float32 layers too).
//
//   Y = A^T [ sum_c (G g G^T) . (B^T d B) ] A          d: 4 x 4 input tile (pad 1), g: 3 x 3 filter, Y: 2 x 2 outputs
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// The 16 positions of the transformed domain are 16 independent GEMMs  M_p[tile][cout] = sum_c V_p[tile][c] U_p[cout][c]  that share
// their operands' source: one work-group (512 threads, 8 waves, one per CU by its 128 KB of LDS) owns 64 tiles (= 256 output pixels)
// x 64 output channels x ALL 16 positions, so that the input transform is computed once per tile and channel and the output
// transform happens on chip.  Wave w multiplies positions 2w and 2w + 1 (2 x 2 x 2 accumulators of 32 x 32: 128 registers).  K advances
// 8 input channels per stage through double-buffered LDS:
//   U (weights, transformed once per layer by k_winograd_weights into the exact LDS image of every (cout tile, stage): 32 KB
//     contiguous) arrives by LDS-DMA (global_load_lds_dwordx4, no registers, 4 instructions per wave and stage);
//   V: thread (tile, channel quad, position row) reads the 2 x 4 input pixels its row of B^T d B needs as 16-byte buffer loads (a
//     pixel outside the image is an out-of-range offset: the buffer unit returns zeros, no select), 32 additions, four ds_write_b128.
// LDS rows are 8 floats; the two 4-float halves of row r are stored swapped when (r >> 3) & 1, which makes the ds_read_b128 fragment
// reads conflict-free without padding (lanes 0-31 read k = 0..3, lanes 32-63 k = 4..7 of a row and feed four v_mfma_f32_32x32x2_f32).
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
constexpr int kWinoStageU = 16 * kWinoCout * kWinoK;   // floats of one stage's U image (32 KB)
constexpr int kWinoStageV = 16 * kWinoTiles * kWinoK;  // floats of one stage's V image (32 KB)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WinoArgs {
    const float *x, *u, *bias, *res;
    float *y;
    int B, H, W, C, N;          // frames, image size, input channels, output channels
    int tilesX, tilesY, tilesPerImage, totalTiles;
    int mBlocks, nTiles, stages, relu;
    unsigned xBytes;            // size of x in bytes (the buffer descriptor's range)
};

// element (row, k) of an 8-float LDS row block lives at float offset row * 8 + 4 * ((k >> 2) ^ ((row >> 3) & 1)) + (k & 3)
__device__ __host__ __forceinline__ int wino_swz(int row, int half) { return row * 8 + 4 * (half ^ ((row >> 3) & 1)); }

// Weights [cout][3][3][cin] (a channels-last Conv2d weight) -> U images: for every (cout tile nt, stage s) 16 x 64 rows of 8 floats,
// image[((nt * stages + s) * 16 + p) * 64 + n][swizzled k] = (G g G^T)[p] of filter (nt * 64 + n, s * 8 + k); sums in double, one rounding.
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
    float *img = u + ((size_t)(nt * stages + s) * 16) * kWinoCout * kWinoK;
    for (int a = 0; a < 4; a++) {
        const double r0 = t[a][0], r1 = 0.5 * (t[a][0] + t[a][1] + t[a][2]), r2 = 0.5 * (t[a][0] - t[a][1] + t[a][2]), r3 = t[a][2];
        const double r[4] = {r0, r1, r2, r3};
        for (int b = 0; b < 4; b++) img[(size_t)(a * 4 + b) * kWinoCout * kWinoK + wino_swz(nr, k >> 2) + (k & 3)] = (float)r[b];
    }
}

__global__ __launch_bounds__(kWinoThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_winograd_conv(const WinoArgs a)
{
    // two stages of (U, V), 128 KB; the epilogue's exchange reuses all of it
    extern __shared__ __align__(16) float smem[];
#define AMOS_WINO_U(buf) (smem + (buf) * (kWinoStageU + kWinoStageV))
#define AMOS_WINO_V(buf) (smem + (buf) * (kWinoStageU + kWinoStageV) + kWinoStageU)
    // id -> (m block, n tile): ids are dealt round-robin over the 8 XCDs; on an XCD, 32 consecutive work-groups (one per CU) share the
    // n tile, i.e. the 1 MB weight slice that stays in that XCD's L2, and walk 32 m blocks; the next 32 take the next n tile
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    constexpr int kGroup = 32;
    const int per = kGroup * a.nTiles, grp = seq / per, in = seq - grp * per;
    const int nt = in / kGroup, mb = (grp * kGroup + in % kGroup) * 8 + xcd;
    if (mb >= a.mBlocks) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;

    // ---- this thread's share of the input transform: tile, channel quad, row of B^T d B
    const int quad = t & 1, tl = (t >> 1) & 63, prow = t >> 7;
    const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)a.xBytes, 0x00020000);
    int xoff[8];  // byte offsets of the 2 x 4 pixels (channel quad included), or the buffer's size for a pixel outside the image
    {
        const int T = min(mb * kWinoTiles + tl, a.totalTiles - 1);
        const int b = T / a.tilesPerImage, rem = T - b * a.tilesPerImage, ty = rem / a.tilesX, tx = rem - ty * a.tilesX;
        // rows of d this position row combines: B^T row 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
        const int r0 = prow == 0 ? 0 : (prow == 2 ? 2 : 1), r1 = prow == 0 ? 2 : (prow == 1 ? 2 : (prow == 2 ? 1 : 3));
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int iy = 2 * ty - 1 + (i ? r1 : r0);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int ix = 2 * tx - 1 + c;
                const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                xoff[i * 4 + c] = inside ? (int)((((unsigned)(b * a.H + iy) * a.W + ix) * a.C + 4 * quad) * 4u) : (int)a.xBytes;
            }
        }
    }
    const float sgn = prow == 1 ? 1.f : -1.f;                        // the row's second term is added (row 1) or subtracted
    const int vdst = (prow * 4 * kWinoTiles) * kWinoK + wino_swz(tl, quad);  // + j * 64 rows for position prow * 4 + j
    // ---- U by LDS-DMA: wave w moves pieces 4w .. 4w + 3 of the stage's 32 (1 KB each: 64 lanes x 16 bytes, lane-linear)
    const float *usrc = a.u + (size_t)nt * a.stages * kWinoStageU + (size_t)(wave * 4) * 256 + lane * 4;
    typedef __attribute__((address_space(3))) void *LdsPtr;
    typedef const __attribute__((address_space(1))) void *GlobalPtr;

    f32x4 raw[8];
#define AMOS_WINO_FETCH_U(s, buf)                                                                                                    \
    {                                                                                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; j++)                                                                                \
            __builtin_amdgcn_global_load_lds((GlobalPtr)(usrc + (size_t)(s) * kWinoStageU + j * 256), (LdsPtr)(AMOS_WINO_U(buf) + (wave * 4 + j) * 256), 16, 0, 0); \
    }
#define AMOS_WINO_FETCH_X(s)                                                                                                         \
    {                                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 8; i++)                                                                                \
            raw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xoff[i], (s) * (kWinoK * 4), 0));         \
    }
#define AMOS_WINO_STASH(buf)                                                                                                         \
    {                                                                                                                                \
        f32x4 tt[4];                                                                                                                 \
        _Pragma("unroll") for (int c = 0; c < 4; c++) tt[c] = f32x4{__builtin_fmaf(sgn, raw[4 + c].x, raw[c].x), __builtin_fmaf(sgn, raw[4 + c].y, raw[c].y), __builtin_fmaf(sgn, raw[4 + c].z, raw[c].z), \
                                                                          __builtin_fmaf(sgn, raw[4 + c].w, raw[c].w)};  /* exact: sgn = +-1 */                                             \
        float *vd = AMOS_WINO_V(buf) + vdst;                                                                                                  \
        *reinterpret_cast<f32x4 *>(vd) = tt[0] - tt[2];                                                                              \
        *reinterpret_cast<f32x4 *>(vd + 1 * kWinoTiles * kWinoK) = tt[1] + tt[2];                                                    \
        *reinterpret_cast<f32x4 *>(vd + 2 * kWinoTiles * kWinoK) = tt[2] - tt[1];                                                    \
        *reinterpret_cast<f32x4 *>(vd + 3 * kWinoTiles * kWinoK) = tt[1] - tt[3];                                                    \
    }

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
    // fragments: one register set per position of the pair, so the LDS reads of one position are in flight under the MFMAs of the other
    f32x4 fa0[2], fb0[2], fa1[2], fb1[2];
#define AMOS_WINO_LDFRAG(fa, fb, buf, p)                                                                                              \
    {                                                                                                                                 \
        const int pos = wave * 2 + (p);                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 2; i++) fa[i] = *reinterpret_cast<const f32x4 *>(AMOS_WINO_V(buf) + (pos * kWinoTiles + i * 32) * kWinoK + foff); \
        _Pragma("unroll") for (int j = 0; j < 2; j++) fb[j] = *reinterpret_cast<const f32x4 *>(AMOS_WINO_U(buf) + (pos * kWinoCout + j * 32) * kWinoK + foff);  \
    }
#define AMOS_WINO_MFMAS(fa, fb, p)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; i++) _Pragma("unroll") for (int j = 0; j < 2; j++) {           \
        acc[p][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[p][i][j], 0, 0, 0);       \
        acc[p][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[p][i][j], 0, 0, 0);       \
        acc[p][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[p][i][j], 0, 0, 0);       \
        acc[p][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[p][i][j], 0, 0, 0);       \
    }
// one instruction of the named kind behind each of the next `n` MFMAs
#define AMOS_WINO_INTERLEAVE(mask, n, each)                                               \
    _Pragma("unroll") for (int q = 0; q < (n); q++) {                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                \
        __builtin_amdgcn_sched_group_barrier(mask, each, 0);                              \
    }
    // One stage = two halves of 16 MFMAs.  First half: position 0 of the pair multiplies (its fragments were read in the second half
    // of the stage before), position 1's fragments are read, the next stage's V tile is made from the pixels requested a whole stage
    // earlier and written to the other buffer, and the pixels of the stage after that are requested (HBM latency is about three
    // quarters of a stage); then the barrier: the other buffer is complete (this wave's U pieces of it have landed: vmcnt leaves only
    // the eight pixel loads just issued in flight) and nobody reads this stage's buffer any more, since every wave holds position 1's
    // fragments.  Second half: position 1 multiplies, the next stage's position-0 fragments are read from the other buffer, and the
    // U tile of the stage after that is requested by LDS-DMA into the buffer this stage just released.  No LDS latency and no global
    // latency sits between two MFMAs.  (A raw s_barrier: __syncthreads() would drain the pixel loads as well.)
#define AMOS_WINO_STAGE(s, buf, kNext, kNext2)                                            \
    {                                                                                     \
        AMOS_WINO_LDFRAG(fa1, fb1, buf, 1);                                               \
        if (kNext) AMOS_WINO_STASH((buf) ^ 1);                                            \
        if (kNext2) AMOS_WINO_FETCH_X((s) + 2);                                           \
        AMOS_WINO_MFMAS(fa0, fb0, 0);                                                     \
        AMOS_WINO_INTERLEAVE(0x100, 4, 1);                                                \
        if (kNext) {                                                                      \
            AMOS_WINO_INTERLEAVE(0x002, 6, 6);                                            \
            AMOS_WINO_INTERLEAVE(0x200, 4, 1);                                            \
        }                                                                                 \
        if (kNext2) AMOS_WINO_INTERLEAVE(0x020, 2, 4);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                               \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if (kNext2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if (kNext) AMOS_WINO_LDFRAG(fa0, fb0, (buf) ^ 1, 0);                              \
        if (kNext2) AMOS_WINO_FETCH_U((s) + 2, buf);                                      \
        AMOS_WINO_MFMAS(fa1, fb1, 1);                                                     \
        AMOS_WINO_INTERLEAVE(0x100, 4, 1);                                                \
        if (kNext2) AMOS_WINO_INTERLEAVE(0x020, 4, 1);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                               \
        __builtin_amdgcn_sched_barrier(0);                                                \
    }

    AMOS_WINO_FETCH_U(0, 0);
    AMOS_WINO_FETCH_X(0);
    AMOS_WINO_STASH(0);
    if (a.stages > 1) AMOS_WINO_FETCH_X(1);
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (with one stage only: the eight loads above do not exist and this waits for everything)
    AMOS_WINO_LDFRAG(fa0, fb0, 0, 0);
    if (a.stages > 1) AMOS_WINO_FETCH_U(1, 1);
    int s = 0;
    for (; s + 2 < a.stages; s++) {
        const int buf = s & 1;
        AMOS_WINO_STAGE(s, buf, true, true);
    }
    if (s + 1 < a.stages) {  // the last but one: nothing left to request
        const int buf = s & 1;
        AMOS_WINO_STAGE(s, buf, true, false);
        s++;
    }
    {
        const int buf = s & 1;
        AMOS_WINO_STAGE(s, buf, false, false);
    }
#undef AMOS_WINO_LDFRAG
#undef AMOS_WINO_MFMAS
#undef AMOS_WINO_INTERLEAVE
#undef AMOS_WINO_STAGE
#undef AMOS_WINO_FETCH_U
#undef AMOS_WINO_FETCH_X
#undef AMOS_WINO_U
#undef AMOS_WINO_V
#undef AMOS_WINO_STASH

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
}

}  // namespace amos

using namespace amos;

extern "C" {

int amos_mask_winograd_supported(int cin, int cout)
{
    return (cin >= kWinoK && cin % kWinoK == 0 && cout >= kWinoCout && cout % kWinoCout == 0) ? AMOS_OK : AMOS_ERR_INVALID;
}

size_t amos_mask_winograd_weight_floats(int cin, int cout)
{
    return amos_mask_winograd_supported(cin, cout) == AMOS_OK ? (size_t)16 * cin * cout : 0;
}

int amos_mask_winograd_weights_device(void *stream, const float *d_w, float *d_u, int cin, int cout)
{
    if (!d_w || !d_u || amos_mask_winograd_supported(cin, cout) != AMOS_OK) {
        set_error("amos_mask_winograd_weights_device: invalid argument (cin %% 8 == 0, cout %% 64 == 0)");
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
        set_error("amos_mask_winograd_conv_device: invalid argument (cin %% 8 == 0, cout %% 64 == 0, input below 2 GiB, 16-byte aligned channels-last tensors)");
        return AMOS_ERR_INVALID;
    }
    static bool attrSet = false;
    const size_t lds = (size_t)2 * (kWinoStageU + kWinoStageV) * sizeof(float);
    if (!attrSet) {
        AMOS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_winograd_conv), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attrSet = true;
    }
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
    // ids: 8 XCDs x groups of (32 m blocks x nTiles); the last group may be partly empty (those work-groups return at once)
    const int perXcd = (a.mBlocks + 7) / 8, groups = (perXcd + 31) / 32;
    const dim3 grid((unsigned)(groups * 32 * a.nTiles * 8)), block(kWinoThreads);
    hipLaunchKernelGGL(k_winograd_conv, grid, block, lds, (hipStream_t)stream, a);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

}  // extern "C"

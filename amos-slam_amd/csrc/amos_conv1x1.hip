// amos_conv1x1.hip -- the 1 x 1 convolutions of the mask network (a15: the ResNet-50 bottlenecks' conv1 / conv3 / downsample and the
// FPN laterals of yolact.py / backbone.py, 36 of the network's 75 convolutions) as ONE fp32 MFMA GEMM with the
// bias, the residual and the ReLU in its epilogue.
//
// Channels-last activations make a 1 x 1 convolution a plain GEMM:  Y[m][n] = sum_k X[row(m)][k] * W[n][k]  with m = output pixel
// (batch, y, x), k = input channel (contiguous), n = output channel; a stride only changes row(m).  MIOpen's fp32 implicit-GEMM
// kernels reach 44 - 96 TFLOP/s on these shapes (tools/conv_fuse_probe.py) and leave the bias / residual / ReLU to a second pass over
// the output (amos_mask_bias_act_device, 12 % of the mask pass); here the epilogue is free and the low-K layers (64 -> 256 at
// 138 x 138: 25 FLOP per byte) run at the memory rate.
//
// Kernel: work-group = 4 waves, wave tile 64 x 64 (2 x 2 v_mfma_f32_32x32x2_f32 accumulators, 64 VGPRs), work-group tile 128 x 128
// (128 x 64 with 32 x 64 wave tiles when the output channels are not a multiple of 128).  K advances 32 at a time through double-buffered LDS: every thread fetches its 16-byte pieces
// of the next X and W tiles into registers before the MFMAs of the current stage and writes them to the other buffer after, one
// barrier per stage.  LDS rows are 36 floats apart, so the sixteen rows a ds_read_b128 phase touches start 4 banks apart: conflict-free.
// The sum over k is order-free, so lanes 0-31 take k = 8j .. 8j+3 and lanes 32-63 k = 8j+4 .. 8j+7 of a row with ONE ds_read_b128 and
// feed four MFMAs from it (MFMA t multiplies k = 8j+t of the low half with k = 8j+4+t of the high half, for X and W alike).
// Work-group ids are dealt so that the tiles of one row block (all n for the same m) run on the same XCD back to back: X is read from
// HBM once and from that XCD's L2 afterwards.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "../../include/amos_frontend.h"
#include "amos_common.h"

namespace amos {

constexpr int kGemmBK = 32;                 // k per stage
constexpr int kGemmPitch = kGemmBK + 4;     // floats between LDS rows

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __attribute__((aligned(16))) const float g_conv_zeros[4] = {0.f, 0.f, 0.f, 0.f};  // what a tap outside the image reads

struct ConvGemmArgs {
    const float *x, *w, *bias, *res;
    float *y;
    int M, N, K;                 // output pixels, output channels, input channels (the GEMM's k runs over taps x K)
    int outW, outHW, inW, inH;   // row(m): b = m / outHW, (oy, ox) of the rest; tap (dy, dx) reads input pixel (oy * stride - pad + dy, ox * stride - pad + dx)
    int stride, relu, mTiles, nTiles;
    int kh, kw, pad;             // kTaps == false: 1, 1, 0
    // split-K (small launches: one frame per pass): blockIdx.y = split, each split runs stagesPerSplit stages of the k loop and leaves its
    // tile in `partial`; the LAST work-group to arrive at a tile (tile counter) adds the splits' tiles in split order -- the same bits
    // whichever group that is -- and runs the epilogue.  splits == 1: none of this.
    int splits, stagesPerSplit;
    float *partial;              // [splits][mTiles * nTiles][BM * BN]
    unsigned *counters;          // [mTiles * nTiles], zero before the launch and after it
};

// kTaps: a kh x kw convolution as an implicit GEMM.  The weight is [cout][kh][kw][cin] (a channels-last Conv2d weight), so the W tile
// of stage s is simply 32 more floats along each row; the X tile of a stage belongs to ONE tap (cin % 32 == 0): row m reads input
// pixel (oy * stride - pad + dy, ox * stride - pad + dx), or zeros outside the image.
template <int WM, int WN, int TI, bool kTaps>  // waves along m and n; a wave's tile is 32 TI x 64
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_conv_gemm(const ConvGemmArgs a)  // LDS admits two groups per CU: 256 registers
{
    constexpr int BM = 32 * TI * WM, BN = 64 * WN;
    constexpr int XP = BM / 32, WP = BN / 32;  // 16-byte pieces per thread and stage
    constexpr int kStage = (BM + BN) * kGemmPitch;  // floats of one stage: the X tile, then the W tile
    static_assert(2 * kStage >= BM * BN, "the epilogue stages the output tile in the same LDS");
    __shared__ __align__(16) float smem[2 * kStage];
    float(*Xs)[kStage] = reinterpret_cast<float(*)[kStage]>(smem);
    float(*Ws)[kStage] = reinterpret_cast<float(*)[kStage]>(smem + BM * kGemmPitch);
    // id -> (m tile, n tile): ids are dealt round-robin over the 8 XCDs; within an XCD consecutive work-groups walk the n tiles of one m tile
    // (split-K launches are small: there blockIdx.x IS the tile, so that consecutive tiles go to consecutive XCDs -- three m tiles would
    // otherwise put all work on three of the eight -- and the splits of a tile, blockIdx.y, still share one)
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int nt = a.splits > 1 ? (int)blockIdx.x % a.nTiles : seq % a.nTiles;
    const int mt = a.splits > 1 ? (int)blockIdx.x / a.nTiles : (seq / a.nTiles) * 8 + xcd;
    if (mt >= a.mTiles) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int pc = t & 7, pr = t >> 3;  // piece column (4 floats), first row of this thread's pieces
    const float *xsrc[XP];
    int iy0[XP], ix0[XP];  // kTaps: input coordinates of tap (0, 0)
#pragma unroll
    for (int i = 0; i < XP; i++) {
        const int m = min(mt * BM + pr + 32 * i, a.M - 1);  // rows past the end repeat the last one; their results are not stored
        iy0[i] = ix0[i] = 0;
        if (!kTaps && a.stride == 1) {
            xsrc[i] = a.x + (size_t)m * a.K + 4 * pc;
        } else {
            const int b = m / a.outHW, rem = m - b * a.outHW, oy = rem / a.outW, ox = rem - oy * a.outW;
            iy0[i] = oy * a.stride - a.pad;
            ix0[i] = ox * a.stride - a.pad;
            // (for a border row this points outside the image; it is only dereferenced at taps that fall inside)
            xsrc[i] = a.x + (((ptrdiff_t)b * a.inH + iy0[i]) * a.inW + ix0[i]) * a.K + 4 * pc;
        }
    }
    const int kcPerTap = a.K / kGemmBK, wRow = a.K * a.kh * a.kw;  // stages per tap; floats of one weight row
    const float *wsrc[WP];
#pragma unroll
    for (int j = 0; j < WP; j++) wsrc[j] = a.w + (size_t)(nt * BN + pr + 32 * j) * wRow + 4 * pc;
    static_assert(XP == 4 && (WP == 2 || WP == 4), "the staging registers below are named one by one");
    float4 x0, x1, x2, x3, w0, w1, w2 = {}, w3 = {};  // (arrays of these end up in scratch once the scheduling fences below are in place)
    // kTaps: the tap / channel-block of the NEXT fetch, advanced after every fetch (stages are fetched in order); a row whose tap falls
    // outside the image reads 16 bytes of zeros instead (a select on the ADDRESS: the load itself stays unconditional, so the stage
    // remains one scheduling region)
    const int stagesAll = (a.K / kGemmBK) * a.kh * a.kw;
    const int sBegin = a.splits > 1 ? (int)blockIdx.y * a.stagesPerSplit : 0;
    const int sEnd = a.splits > 1 ? min(sBegin + a.stagesPerSplit, stagesAll) : stagesAll;
    int fKc = 0, fDy = 0, fDx = 0;
    if (kTaps && sBegin > 0) {  // the tap / channel block of this split's first stage
        const int kcPerTap0 = a.K / kGemmBK, tap = sBegin / kcPerTap0;
        fKc = sBegin - tap * kcPerTap0;
        fDy = tap / a.kw;
        fDx = tap - fDy * a.kw;
    }
    // (the select is made on integers and the result read through a global-address-space pointer: a select of two C++ pointers of
    // different provenance becomes a FLAT load, which also counts as an LDS operation and would be waited for at the barrier)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) f32x4 *GlobalF4;
    const uintptr_t zeroAddr = (uintptr_t)g_conv_zeros;
#define AMOS_GEMM_TAPSRC(i) \
    ((GlobalF4)(((unsigned)(iy0[i] + fDy) < (unsigned)a.inH && (unsigned)(ix0[i] + fDx) < (unsigned)a.inW) ? (uintptr_t)(xsrc[i] + tapOff) : zeroAddr))
#define AMOS_GEMM_FETCH(stage)                                                                                               \
    {                                                                                                                        \
        const int k0 = (stage) * kGemmBK;                                                                                    \
        if (kTaps) {                                                                                                         \
            const ptrdiff_t tapOff = (ptrdiff_t)(fDy * a.inW + fDx) * a.K + fKc * kGemmBK;                                   \
            x0 = __builtin_bit_cast(float4, *AMOS_GEMM_TAPSRC(0));                                                               \
            x1 = __builtin_bit_cast(float4, *AMOS_GEMM_TAPSRC(1));                                                               \
            x2 = __builtin_bit_cast(float4, *AMOS_GEMM_TAPSRC(2));                                                               \
            x3 = __builtin_bit_cast(float4, *AMOS_GEMM_TAPSRC(3));                                                               \
            fKc++;                                                                                                           \
            const bool tapDone = fKc == kcPerTap;                                                                            \
            fKc = tapDone ? 0 : fKc;                                                                                         \
            fDx += tapDone ? 1 : 0;                                                                                          \
            const bool rowDone = fDx == a.kw;                                                                                \
            fDx = rowDone ? 0 : fDx;                                                                                         \
            fDy += rowDone ? 1 : 0;                                                                                          \
        } else {                                                                                                             \
            x0 = *reinterpret_cast<const float4 *>(xsrc[0] + k0);                                                            \
            x1 = *reinterpret_cast<const float4 *>(xsrc[1] + k0);                                                            \
            x2 = *reinterpret_cast<const float4 *>(xsrc[2] + k0);                                                            \
            x3 = *reinterpret_cast<const float4 *>(xsrc[3] + k0);                                                            \
        }                                                                                                                    \
        w0 = *reinterpret_cast<const float4 *>(wsrc[0] + k0);                                                                \
        w1 = *reinterpret_cast<const float4 *>(wsrc[1] + k0);                                                                \
        if (WP == 4) {                                                                                                       \
            w2 = *reinterpret_cast<const float4 *>(wsrc[WP - 2] + k0);                                                       \
            w3 = *reinterpret_cast<const float4 *>(wsrc[WP - 1] + k0);                                                       \
        }                                                                                                                    \
    }
#define AMOS_GEMM_STASH(buf)                                                                          \
    {                                                                                                 \
        float *xs = &Xs[buf][pr * kGemmPitch + 4 * pc], *ws = &Ws[buf][pr * kGemmPitch + 4 * pc];     \
        *reinterpret_cast<float4 *>(xs) = x0;                                                         \
        *reinterpret_cast<float4 *>(xs + 32 * kGemmPitch) = x1;                                       \
        *reinterpret_cast<float4 *>(xs + 64 * kGemmPitch) = x2;                                       \
        *reinterpret_cast<float4 *>(xs + 96 * kGemmPitch) = x3;                                       \
        *reinterpret_cast<float4 *>(ws) = w0;                                                         \
        *reinterpret_cast<float4 *>(ws + 32 * kGemmPitch) = w1;                                       \
        if (WP == 4) {                                                                                \
            *reinterpret_cast<float4 *>(ws + 64 * kGemmPitch) = w2;                                   \
            *reinterpret_cast<float4 *>(ws + 96 * kGemmPitch) = w3;                                   \
        }                                                                                             \
    }
    // fragments: two register sets, so the LDS reads of k-step kk + 1 are in flight under the MFMAs of k-step kk
    float4 fa0[TI], fb0[2], fa1[TI], fb1[2];
#define AMOS_GEMM_LDFRAG(fa, fb, buf, kk)                                                                                                        \
    {                                                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < TI; i++) fa[i] = *reinterpret_cast<const float4 *>(&Xs[buf][xoff + i * 32 * kGemmPitch + 8 * (kk)]); \
        _Pragma("unroll") for (int j = 0; j < 2; j++) fb[j] = *reinterpret_cast<const float4 *>(&Ws[buf][woff + j * 32 * kGemmPitch + 8 * (kk)]);  \
    }
#define AMOS_GEMM_MFMAS(fa, fb)                                                                           \
    _Pragma("unroll") for (int i = 0; i < TI; i++) _Pragma("unroll") for (int j = 0; j < 2; j++) {        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);           \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);           \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);           \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);           \
    }
// scheduling pattern of one phase: one memory instruction of the named kind behind each of the first `n` MFMAs, then the rest of the MFMAs
#define AMOS_GEMM_INTERLEAVE(mask, n)                                                     \
    _Pragma("unroll") for (int q = 0; q < (n); q++) {                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                \
        __builtin_amdgcn_sched_group_barrier(mask, 1, 0);                                 \
    }
    constexpr int kFragReads = TI + 2, kMfmas = 8 * TI;
    // One stage = four k-steps of 8.  The LDS reads of step kk + 1 go out under the MFMAs of step kk; tile s + 1 (in the staging
    // registers since the stage before) is written to the other buffer under step 1, tile s + 2 is requested under step 2; the
    // barrier sits between steps 2 and 3 -- by then this wave holds the fragments of step 3 in registers, so after the barrier
    // nobody reads this stage's buffer any more and the next stage may overwrite it, and the first fragments of the next stage
    // (other buffer, complete at the barrier) are read under the MFMAs of step 3: no LDS latency is exposed at the stage boundary.
#define AMOS_GEMM_STAGE(s, buf, kStash, kFetch, kNext)                                    \
    {                                                                                     \
        AMOS_GEMM_LDFRAG(fa1, fb1, buf, 1);                                               \
        AMOS_GEMM_MFMAS(fa0, fb0);                                                        \
        AMOS_GEMM_INTERLEAVE(0x100, kFragReads);                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, kMfmas, 0);                           \
        __builtin_amdgcn_sched_barrier(0);                                                \
        AMOS_GEMM_LDFRAG(fa0, fb0, buf, 2);                                               \
        if (kStash) AMOS_GEMM_STASH((buf) ^ 1);                                           \
        AMOS_GEMM_MFMAS(fa1, fb1);                                                        \
        AMOS_GEMM_INTERLEAVE(0x100, kFragReads);                                          \
        AMOS_GEMM_INTERLEAVE(0x200, XP + WP);                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, kMfmas, 0);                           \
        __builtin_amdgcn_sched_barrier(0);                                                \
        AMOS_GEMM_LDFRAG(fa1, fb1, buf, 3);                                               \
        if (kFetch) AMOS_GEMM_FETCH((s) + 2);                                             \
        AMOS_GEMM_MFMAS(fa0, fb0);                                                        \
        AMOS_GEMM_INTERLEAVE(0x100, kFragReads);                                          \
        AMOS_GEMM_INTERLEAVE(0x020, XP + WP);                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, kMfmas, 0);                           \
        __builtin_amdgcn_sched_barrier(0);                                                \
        __syncthreads();                                                                  \
        if (kNext) AMOS_GEMM_LDFRAG(fa0, fb0, (buf) ^ 1, 0);                              \
        AMOS_GEMM_MFMAS(fa1, fb1);                                                        \
        AMOS_GEMM_INTERLEAVE(0x100, kFragReads);                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, kMfmas, 0);                           \
        __builtin_amdgcn_sched_barrier(0);                                                \
    }
    f32x16 acc[TI][2];
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int xoff = (wm * 32 * TI + (lane & 31)) * kGemmPitch + 4 * (lane >> 5);
    const int woff = (wn * 64 + (lane & 31)) * kGemmPitch + 4 * (lane >> 5);
    // stages [sBegin, sEnd) of the k loop (all of them without split-K); `s` counts from 0 for the buffer parity, the fetches name the
    // absolute stage
    const int stages = sEnd - sBegin;
    AMOS_GEMM_FETCH(sBegin);
    AMOS_GEMM_STASH(0);
    if (stages > 1) AMOS_GEMM_FETCH(sBegin + 1);
    __syncthreads();
    AMOS_GEMM_LDFRAG(fa0, fb0, 0, 0);
    int s = 0;
    for (; s + 2 < stages; s++) {
        const int buf = s & 1;
        AMOS_GEMM_STAGE(sBegin + s, buf, true, true, true);
    }
    if (s + 1 < stages) {  // the last but one: nothing left to request
        const int buf = s & 1;
        AMOS_GEMM_STAGE(sBegin + s, buf, true, false, true);
        s++;
    }
    {
        const int buf = s & 1;
        AMOS_GEMM_STAGE(sBegin + s, buf, false, false, false);
    }
#undef AMOS_GEMM_FETCH
#undef AMOS_GEMM_TAPSRC
#undef AMOS_GEMM_STASH
#undef AMOS_GEMM_LDFRAG
#undef AMOS_GEMM_MFMAS
#undef AMOS_GEMM_INTERLEAVE
#undef AMOS_GEMM_STAGE
    // epilogue.  Accumulator register r of a 32 x 32 tile is row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31: the tile goes
    // through LDS (row-major, BN floats per row; a half-wave writes 32 consecutive floats) so that every thread then handles
    // 16-byte pieces of output rows: bias, residual and ReLU on float4, one coalesced residual load and one store per piece.
    __syncthreads();  // every wave is done with the last stage's operands
#pragma unroll
    for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                smem[(wm * 32 * TI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * BN + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
    __syncthreads();
    constexpr int kCols = BN / 4, kRowsPerPass = 256 / kCols, kPasses = BM / kRowsPerPass;  // float4 columns; rows per sweep of the group
    const int oc = t % kCols, orow = t / kCols;
    const int n0 = nt * BN + 4 * oc;
    if (a.splits > 1) {
        // this split's tile -> partial[split][tile] (tile-contiguous: every store instruction writes whole lines); the last group to arrive
        // sums the splits in order into the LDS tile and falls through to the ordinary epilogue.
        // Visibility.  The 8 XCDs of the chip have an L2 each, so a device-wide release / acquire pair (__threadfence) writes an L2 back and
        // invalidates one: ~50 us per launch, measured -- more than the launches this path serves.  All splits of a tile share blockIdx.x,
        // and work-groups are dealt to XCDs by linear id modulo 8 (gridDim.x is a multiple of 8), so they run on ONE XCD and meet in ITS
        // L2: stores are write-through to it (complete at vmcnt(0)) and the reader's loads miss its L1.  That the
        // dealing really is what this assumes is CHECKED by every group (the XCC_ID hardware register against blockIdx.x % 8); a group that
        // finds itself elsewhere falls back to the device-wide fences and says so in the tile counter, and the reader then does too.
        __shared__ unsigned sArrived;
        const int tile = mt * a.nTiles + nt, nTilesAll = a.mTiles * a.nTiles;
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;  // HW_REG_XCC_ID[3:0]
        const bool home = xcc == (blockIdx.x & 7u);
        float *mine = a.partial + ((size_t)blockIdx.y * nTilesAll + tile) * (BM * BN);
#pragma unroll
        for (int q = 0; q < kPasses; q++) {
            const int e = (q * kRowsPerPass + orow) * BN + 4 * oc;
            *reinterpret_cast<float4 *>(mine + e) = *reinterpret_cast<const float4 *>(&smem[e]);
        }
        if (home) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tile is in this XCD's L2
        else __threadfence();                                        // ... or written back for everybody
        __syncthreads();
        if (t == 0) sArrived = atomicAdd(&a.counters[tile], home ? 1u : 0x10001u);  // low half: arrivals; high half: groups away from home
        __syncthreads();
        const unsigned arrived = sArrived;
        if ((arrived & 0xffffu) != (unsigned)(a.splits - 1)) return;
        if (!home || (arrived >> 16) != 0) __threadfence();
#pragma unroll
        for (int q = 0; q < kPasses; q++) {
            const int e = (q * kRowsPerPass + orow) * BN + 4 * oc;
            float4 sum = {0.f, 0.f, 0.f, 0.f};
            for (int sp = 0; sp < a.splits; sp++) {
                // (plain loads: this CU's L1 was invalidated when the kernel started and has never held these lines since -- nobody reads a
                // partial tile but its last group -- so they come from the L2 the splits wrote to; an agent-scope load would go PAST that L2)
                typedef float f32x4v __attribute__((ext_vector_type(4)));
                const f32x4v v = *reinterpret_cast<const volatile f32x4v *>(a.partial + ((size_t)sp * nTilesAll + tile) * (BM * BN) + e);
                sum.x = sp ? sum.x + v.x : v.x; sum.y = sp ? sum.y + v.y : v.y; sum.z = sp ? sum.z + v.z : v.z; sum.w = sp ? sum.w + v.w : v.w;
            }
            *reinterpret_cast<float4 *>(&smem[e]) = sum;  // (each thread rewrites exactly the pieces it reads below)
        }
        if (t == 0) atomicExch(&a.counters[tile], 0u);  // for the next launch
    }
    const float4 bv = a.bias ? *reinterpret_cast<const float4 *>(a.bias + n0) : float4{0.f, 0.f, 0.f, 0.f};
    // (round 5, tools/gemm_bench.hip: a work-group ALONE on a CU keeps 88 - 97 % of the two-group rate on the compute-bound layers -- one wave per
    // SIMD nearly fills the MFMA pipe in the main loop -- and spends 10 of its 24 us per tile outside the main loop on 256 -> 1024 + residual at
    // 35 x 35.  Requesting the whole residual tile, 16 loads per thread, before the accumulators' trip through LDS was SLOWER, 410 -> 422 us on
    // that layer and 658 -> 686 on 64 -> 256 at 138 x 138: 64 KB of requests in one burst delay the stage fetches of the CU's other group;
    // requesting it in two halves during the last two stages of the main loop, which issue no stage fetch: 412 / 653, no change.  What a tile
    // costs outside its main loop is the epilogue's write of the 64 KB tile: M 78 400, N 1 024 at k = 32 / 64 / 256 takes 127 / 168 / 384 us --
    // 37 us per stage, 89 % of the pipe, on top of 90 us that no k amortises and the CU's second group does not hide.)
    constexpr int kBatch = 8;  // residual loads in flight per thread
#pragma unroll
    for (int p0 = 0; p0 < kPasses; p0 += kBatch) {
        float4 rv[kBatch];
#pragma unroll
        for (int q = 0; q < kBatch; q++) {
            const int m = mt * BM + (p0 + q) * kRowsPerPass + orow;
            rv[q] = (a.res && m < a.M) ? *reinterpret_cast<const float4 *>(a.res + (size_t)m * a.N + n0) : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < kBatch; q++) {
            const int row = (p0 + q) * kRowsPerPass + orow, m = mt * BM + row;
            if (m >= a.M) continue;
            float4 v = *reinterpret_cast<const float4 *>(&smem[row * BN + 4 * oc]);
            v.x = (v.x + bv.x) + rv[q].x; v.y = (v.y + bv.y) + rv[q].y; v.z = (v.z + bv.z) + rv[q].z; v.w = (v.w + bv.w) + rv[q].w;
            if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4 *>(a.y + (size_t)m * a.N + n0) = v;
        }
    }
}

}  // namespace amos

using namespace amos;

// 128 x 128 tiles unless they would leave the chip short of work: below 1 024 work-groups the 128 x 64 shape (twice as many,
// lower per-group efficiency) balances the 256 CUs better -- measured on the 35 x 35 and 18 x 18 layers (tools/conv1x1_probe.py with
// the narrow shape forced: 1024 -> 256 at 35 x 35 0.212 -> 0.187 ms, 2048 -> 512 at 18 x 18 0.259 -> 0.209 ms; the large layers lose 5 - 10 %).
// amos_mask_conv_tile_mode forces a shape (tests, experiments); AMOS_GEMM_NARROW=0 / 1 in the environment is its initial value, read once.
static int g_gemm_tile_mode = -2;  // -2: environment not read yet; -1 automatic; 0 wide; 1 narrow

static int gemm_tile_mode()
{
    if (g_gemm_tile_mode == -2) {
        const char *env = getenv("AMOS_GEMM_NARROW");
        g_gemm_tile_mode = env && (env[0] == '0' || env[0] == '1') ? env[0] - '0' : -1;
    }
    return g_gemm_tile_mode;
}

static bool gemm_wide_tiles(long long M, int cout)
{
    const int mode = gemm_tile_mode();
    const long long wideGroups = ((M + 127) / 128) * (cout / 128);
    const bool narrow = mode >= 0 ? mode == 1 : wideGroups < 1024;
    return cout % 128 == 0 && !narrow;
}

extern "C" {

int amos_mask_conv_tile_mode(int mode)
{
    const int before = gemm_tile_mode();
    if (mode >= -1 && mode <= 1) g_gemm_tile_mode = mode;
    return before;
}

// the kernel amos_mask_conv_device launches for this shape, as rocprofv3 names it
int amos_mask_conv_kernel_name(int batch, int in_h, int in_w, int cin, int cout, int kh, int kw, int stride, int pad, char *name, int name_len)
{
    if (!name || name_len < 48 || batch < 1 || amos_mask_conv_supported(cin, cout, kh, kw, stride, pad) != AMOS_OK) {
        set_error("amos_mask_conv_kernel_name: invalid argument");
        return AMOS_ERR_INVALID;
    }
    const int oh = (in_h + 2 * pad - kh) / stride + 1, ow = (in_w + 2 * pad - kw) / stride + 1;
    const bool wide = gemm_wide_tiles((long long)batch * oh * ow, cout), taps = kh * kw > 1 || pad > 0;
    snprintf(name, (size_t)name_len, "amos::k_conv_gemm<%s, %s>", wide ? "2, 2, 2" : "4, 1, 1", taps ? "true" : "false");
    return AMOS_OK;
}

// 0 = this shape is served by amos_mask_conv_device, otherwise AMOS_ERR_INVALID (the caller keeps its library convolution)
int amos_mask_conv_supported(int cin, int cout, int kh, int kw, int stride, int pad)
{
    return (cin >= kGemmBK && cin % kGemmBK == 0 && cout >= 64 && cout % 64 == 0 && stride >= 1 && stride <= 4 && kh >= 1 && kh <= 7 && kw >= 1 && kw <= 7 &&
            pad >= 0 && pad < kh && pad < kw) ? AMOS_OK : AMOS_ERR_INVALID;
}

// Split-K plan of a launch: small launches (one frame per pass: 3 - 150 work-groups on 256 CUs, up to 72 stages each) are cut along k so
// that about a chip's worth of work-groups runs, every split keeping at least four stages.  1 = no split.
static const int kSplitCounterBytes = 16384;  // 4 096 tile counters at the head of the workspace
static int gemm_splits(long long M, int cout, int stages, bool wide)
{
    static const int minTiles = getenv("AMOS_GEMM_SPLIT_BELOW") ? atoi(getenv("AMOS_GEMM_SPLIT_BELOW")) : 192;   // experiment knobs
    static const int target = getenv("AMOS_GEMM_SPLIT_TARGET") ? atoi(getenv("AMOS_GEMM_SPLIT_TARGET")) : 256;
    const long long tiles = ((M + 127) / 128) * (cout / (wide ? 128 : 64));
    if (tiles >= minTiles || stages < 8 || tiles > kSplitCounterBytes / 4) return 1;
    long long s = (target + tiles - 1) / tiles;
    s = std::min<long long>(s, stages / 4);
    s = std::min<long long>(s, 16);
    return (int)std::max<long long>(s, 1);
}

size_t amos_mask_conv_workspace_bytes(int batch, int in_h, int in_w, int cin, int cout, int kh, int kw, int stride, int pad)
{
    if (batch < 1 || in_h < 1 || in_w < 1 || amos_mask_conv_supported(cin, cout, kh, kw, stride, pad) != AMOS_OK || in_h + 2 * pad < kh || in_w + 2 * pad < kw) return 0;
    const int oh = (in_h + 2 * pad - kh) / stride + 1, ow = (in_w + 2 * pad - kw) / stride + 1;
    const long long M = (long long)batch * oh * ow;
    const bool wide = gemm_wide_tiles(M, cout);
    const int stages = (cin / kGemmBK) * kh * kw, splits = gemm_splits(M, cout, stages, wide);
    if (splits <= 1) return 0;
    const long long tiles = ((M + 127) / 128) * (cout / (wide ? 128 : 64));
    return (size_t)kSplitCounterBytes + (size_t)splits * tiles * 128 * (wide ? 128 : 64) * sizeof(float);
}

int amos_mask_conv_device(void *stream, const float *d_x, const float *d_w, const float *d_bias, const float *d_residual, float *d_y, int batch,
                          int in_h, int in_w, int cin, int cout, int kh, int kw, int stride, int pad, int relu)
{
    return amos_mask_conv_ws_device(stream, d_x, d_w, d_bias, d_residual, d_y, batch, in_h, in_w, cin, cout, kh, kw, stride, pad, relu, nullptr, 0);
}

int amos_mask_conv_ws_device(void *stream, const float *d_x, const float *d_w, const float *d_bias, const float *d_residual, float *d_y, int batch,
                             int in_h, int in_w, int cin, int cout, int kh, int kw, int stride, int pad, int relu, void *d_workspace, size_t workspace_bytes)
{
    if (!d_x || !d_w || !d_y || batch < 1 || in_h < 1 || in_w < 1 || amos_mask_conv_supported(cin, cout, kh, kw, stride, pad) != AMOS_OK ||
        in_h + 2 * pad < kh || in_w + 2 * pad < kw || ((uintptr_t)d_x | (uintptr_t)d_w | (uintptr_t)d_y | (uintptr_t)d_bias | (uintptr_t)d_residual) % 16 != 0) {
        set_error("amos_mask_conv_device: invalid argument (cin %% 32 == 0, cout %% 64 == 0, kernel <= 7 x 7, pad < kernel, stride 1..4, 16-byte aligned channels-last tensors)");
        return AMOS_ERR_INVALID;
    }
    const int oh = (in_h + 2 * pad - kh) / stride + 1, ow = (in_w + 2 * pad - kw) / stride + 1;
    const long long M = (long long)batch * oh * ow;
    if (M > 0x7fffffffLL / 4 || (long long)batch * in_h * in_w * cin > 0x7fffffffffLL) { set_error("amos_mask_conv_device: tensor too large"); return AMOS_ERR_INVALID; }
    ConvGemmArgs a;
    a.x = d_x; a.w = d_w; a.bias = d_bias; a.res = d_residual; a.y = d_y;
    a.M = (int)M; a.N = cout; a.K = cin;
    a.outW = ow; a.outHW = oh * ow; a.inW = in_w; a.inH = in_h;
    a.stride = stride; a.relu = relu; a.kh = kh; a.kw = kw; a.pad = pad;
    const bool wide = gemm_wide_tiles(M, cout), taps = kh * kw > 1 || pad > 0;
    const int BM = 128, BN = wide ? 128 : 64;
    a.mTiles = (int)((M + BM - 1) / BM);
    a.nTiles = cout / BN;
    a.splits = 1;
    a.stagesPerSplit = 0;
    a.partial = nullptr;
    a.counters = nullptr;
    if (d_workspace) {  // split-K when the plan for this shape says so and the caller brought the scratch for it
        const int stages = (cin / kGemmBK) * kh * kw, splits = gemm_splits(M, cout, stages, wide);
        const size_t need = (size_t)kSplitCounterBytes + (size_t)splits * a.mTiles * a.nTiles * BM * BN * sizeof(float);
        if (splits > 1) {
            if (workspace_bytes < need || (uintptr_t)d_workspace % 16 != 0) { set_error("amos_mask_conv_ws_device: workspace of %zu bytes, %zu needed (16-byte aligned)", workspace_bytes, need); return AMOS_ERR_CAPACITY; }
            a.stagesPerSplit = (stages + splits - 1) / splits;
            a.splits = (stages + a.stagesPerSplit - 1) / a.stagesPerSplit;  // no empty split
            a.counters = (unsigned *)d_workspace;
            a.partial = (float *)((uint8_t *)d_workspace + kSplitCounterBytes);
        }
    }
    const dim3 grid(a.splits > 1 ? (unsigned)((a.mTiles * a.nTiles + 7) / 8 * 8) : (unsigned)(((a.mTiles + 7) / 8) * 8 * a.nTiles), (unsigned)a.splits), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (wide && taps) hipLaunchKernelGGL((k_conv_gemm<2, 2, 2, true>), grid, block, 0, st, a);
    else if (wide) hipLaunchKernelGGL((k_conv_gemm<2, 2, 2, false>), grid, block, 0, st, a);
    else if (taps) hipLaunchKernelGGL((k_conv_gemm<4, 1, 1, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_conv_gemm<4, 1, 1, false>), grid, block, 0, st, a);
    AMOS_HIP_CHECK(hipGetLastError());
    return AMOS_OK;
}

int amos_mask_conv1x1_supported(int cin, int cout, int stride) { return amos_mask_conv_supported(cin, cout, 1, 1, stride, 0); }

int amos_mask_conv1x1_device(void *stream, const float *d_x, const float *d_w, const float *d_bias, const float *d_residual, float *d_y, int batch,
                             int in_h, int in_w, int cin, int cout, int stride, int relu)
{
    return amos_mask_conv_device(stream, d_x, d_w, d_bias, d_residual, d_y, batch, in_h, in_w, cin, cout, 1, 1, stride, 0, relu);
}

}  // extern "C"

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
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int nt = seq % a.nTiles, mt = (seq / a.nTiles) * 8 + xcd;
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
    int fKc = 0, fDy = 0, fDx = 0;
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
    const int stages = kcPerTap * a.kh * a.kw;
    AMOS_GEMM_FETCH(0);
    AMOS_GEMM_STASH(0);
    if (stages > 1) AMOS_GEMM_FETCH(1);
    __syncthreads();
    AMOS_GEMM_LDFRAG(fa0, fb0, 0, 0);
    int s = 0;
    for (; s + 2 < stages; s++) {
        const int buf = s & 1;
        AMOS_GEMM_STAGE(s, buf, true, true, true);
    }
    if (s + 1 < stages) {  // the last but one: nothing left to request
        const int buf = s & 1;
        AMOS_GEMM_STAGE(s, buf, true, false, true);
        s++;
    }
    {
        const int buf = s & 1;
        AMOS_GEMM_STAGE(s, buf, false, false, false);
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
    const float4 bv = a.bias ? *reinterpret_cast<const float4 *>(a.bias + n0) : float4{0.f, 0.f, 0.f, 0.f};
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

int amos_mask_conv_device(void *stream, const float *d_x, const float *d_w, const float *d_bias, const float *d_residual, float *d_y, int batch,
                          int in_h, int in_w, int cin, int cout, int kh, int kw, int stride, int pad, int relu)
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
    const dim3 grid((unsigned)(((a.mTiles + 7) / 8) * 8 * a.nTiles)), block(256);
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

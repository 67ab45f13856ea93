// Micro-benchmark (gfx950): the vector-ALU ISSUE RATE of the byte / packed-16-bit instructions the ORB kernels are made of, in
// wave-instructions per clock per CU, at 1 / 2 / 4 / 8 resident waves per SIMD on every CU.  Eight independent dependency chains
// per wave (x0..x7), so a chain's own latency is hidden from 1 wave per SIMD on if the pipeline is <= 8 issue slots deep.
// Cycles: every wave stamps s_memtime (shader clock; MI355X_MICROARCH.md constants table) and s_memrealtime (constant 100 MHz) at its
// start and end; the host takes the span from the earliest start to the latest end on the chip-wide 100 MHz counter (the VALU
// arbiter favours the oldest wave, so one wave's own elapsed time is NOT the time its SIMD was busy) and converts it to shader
// clocks with the clock actually held (a wave's own shader ticks per 100 MHz tick).  A 30 ms warm-up launch first, so that the clocks have ramped.
// Build: hipcc -O2 --offload-arch=gfx950 -o valu_issue_rate valu_issue_rate.hip ; prints a markdown table (profiles/r04_valu_issue.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define R8(OP)                                                                                                                    \
    OP("%[x0]") OP("%[x1]") OP("%[x2]") OP("%[x3]") OP("%[x4]") OP("%[x5]") OP("%[x6]") OP("%[x7]")
#define OPS_add(x) "v_add_u32 " x ", " x ", %[y]\n"
#define OPS_pkmax(x) "v_pk_max_u16 " x ", " x ", %[y]\n"
#define OPS_pkadd(x) "v_pk_add_u16 " x ", " x ", %[y]\n"
#define OPS_pksubc(x) "v_pk_sub_u16 " x ", " x ", %[y] clamp\n"
#define OPS_pkmin(x) "v_pk_min_u16 " x ", " x ", %[y]\n"
#define OPS_perm(x) "v_perm_b32 " x ", " x ", %[y], %[z]\n"
#define OPS_alignb(x) "v_alignbyte_b32 " x ", " x ", %[y], 3\n"
#define OPS_dot4(x) "v_dot4_u32_u8 " x ", " x ", %[y], %[z]\n"
#define OPS_dot2(x) "v_dot2_u32_u16 " x ", " x ", %[y], %[z]\n"
#define OPS_min3(x) "v_min3_u32 " x ", " x ", %[y], %[z]\n"
#define OPS_sdwa(x) "v_add_u32_sdwa " x ", " x ", %[y] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define OPS_sdwamax(x) "v_max_u16_sdwa " x ", " x ", %[y] dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_0\n"
#define OPS_lshlor(x) "v_lshl_or_b32 " x ", " x ", 3, %[y]\n"
#define OPS_bfe(x) "v_bfe_u32 " x ", " x ", 8, 8\n"
#define OPS_and(x) "v_and_b32 " x ", " x ", %[y]\n"
#define OPS_cmpcnd(x) "v_cmp_gt_u32 vcc, " x ", %[y]\n v_cndmask_b32 " x ", " x ", %[z], vcc\n"
#define OPS_mad24(x) "v_mad_u32_u24 " x ", " x ", %[y], %[z]\n"
#define OPS_sad(x) "v_sad_u8 " x ", " x ", %[y], %[z]\n"
#define OPS_fma(x) "v_fma_f32 " x ", " x ", %[y], %[z]\n"
#define OPS_max16(x) "v_max_u16 " x ", " x ", %[y]\n"
#define OPS_xor(x) "v_xor_b32 " x ", " x ", %[y]\n"
#define OPS_lshl(x) "v_lshlrev_b32 " x ", 1, " x "\n"
// the FAST necessary test's mix: alignbyte, perm, and, packed min / max / saturating sub (9 instructions, 4-byte and 8-byte encodings)
#define OPS_mix(x) "v_alignbyte_b32 " x ", " x ", %[y], 1\n v_pk_min_u16 " x ", " x ", %[y]\n v_pk_max_u16 " x ", " x ", %[z]\n v_and_b32 " x ", " x ", %[y]\n v_pk_sub_u16 " x ", " x ", %[z] clamp\n v_perm_b32 " x ", " x ", %[y], %[z]\n v_pk_max_u16 " x ", " x ", %[y]\n v_or_b32 " x ", " x ", %[z]\n v_pk_min_u16 " x ", " x ", %[z]\n"
#define OPS_sub(x) "v_sub_u32 " x ", " x ", %[y]\n"
#define OPS_maxu32(x) "v_max_u32 " x ", " x ", %[y]\n"
#define OPS_or3(x) "v_or3_b32 " x ", " x ", %[y], %[z]\n"
#define OPS_add3(x) "v_add3_u32 " x ", " x ", %[y], %[z]\n"
#define OPS_andor(x) "v_and_or_b32 " x ", " x ", %[y], %[z]\n"
#define OPS_cnd(x) "v_cndmask_b32 " x ", " x ", %[y], vcc\n"
#define OPS_cmp(x) "v_cmp_gt_u32 vcc, " x ", %[y]\n"
#define OPS_addu16(x) "v_add_u16 " x ", " x ", %[y]\n"
#define OPS_mul24(x) "v_mul_u32_u24 " x ", " x ", %[y]\n"
#define OPS_lshr(x) "v_lshrrev_b32 " x ", 1, " x "\n"
#define OPS_mov(x) "v_mov_b32 " x ", %[y]\n"
#define OPS_bcnt(x) "v_bcnt_u32_b32 " x ", " x ", %[y]\n"

enum Op { ADD, PKMAX, PKMIN, PKADD, PKSUBC, PERM, ALIGNB, DOT4, DOT2, MIN3, SDWA, SDWAMAX, LSHLOR, BFE, AND, CMPCND, MAD24, SAD, FMA, BCNT, MAX16, XOR, LSHL, MIX, SUB, MAXU32, OR3, ADD3, ANDOR, CND, CMP, ADDU16, MUL24, LSHR, MOV, N_OPS };
static const char *kNames[N_OPS] = {"v_add_u32", "v_pk_max_u16", "v_pk_min_u16", "v_pk_add_u16", "v_pk_sub_u16 clamp", "v_perm_b32", "v_alignbyte_b32", "v_dot4_u32_u8",
                                    "v_dot2_u32_u16", "v_min3_u32", "v_add_u32_sdwa (byte select)", "v_max_u16_sdwa (byte sel, word dst preserve)", "v_lshl_or_b32",
                                    "v_bfe_u32", "v_and_b32", "v_cmp_gt_u32 + v_cndmask_b32 (pair)", "v_mad_u32_u24", "v_sad_u8", "v_fma_f32", "v_bcnt_u32_b32", "v_max_u16 (VOP2)", "v_xor_b32 (VOP2)", "v_lshlrev_b32 (VOP2)", "FAST mix: alignbyte, perm, and, or, 5 packed min / max / sub (9 per chain step)", "v_sub_u32 (VOP2)", "v_max_u32 (VOP2)", "v_or3_b32", "v_add3_u32", "v_and_or_b32", "v_cndmask_b32 (VOP2, vcc)", "v_cmp_gt_u32 (to vcc)", "v_add_u16 (VOP2)", "v_mul_u32_u24 (VOP2)", "v_lshrrev_b32 (VOP2)", "v_mov_b32"};
static const int kInstrPerOp[N_OPS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 9, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned long long *cyc, int iters)
{
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned x0 = threadIdx.x, x1 = 7 + threadIdx.x, x2 = 9, x3 = 11 * threadIdx.x, x4 = 5, x5 = 77, x6 = threadIdx.x ^ 0x55, x7 = 123;
    unsigned y = 0x01030205u + threadIdx.x, z = 0x07060504u;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#define BODY(M) asm volatile(R8(M) R8(M) R8(M) R8(M) : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5), [x6] "+v"(x6), [x7] "+v"(x7) : [y] "v"(y), [z] "v"(z) : "vcc");
        if (OP == ADD) { BODY(OPS_add) }
        if (OP == PKMAX) { BODY(OPS_pkmax) }
        if (OP == PKMIN) { BODY(OPS_pkmin) }
        if (OP == PKADD) { BODY(OPS_pkadd) }
        if (OP == PKSUBC) { BODY(OPS_pksubc) }
        if (OP == PERM) { BODY(OPS_perm) }
        if (OP == ALIGNB) { BODY(OPS_alignb) }
        if (OP == DOT4) { BODY(OPS_dot4) }
        if (OP == DOT2) { BODY(OPS_dot2) }
        if (OP == MIN3) { BODY(OPS_min3) }
        if (OP == SDWA) { BODY(OPS_sdwa) }
        if (OP == SDWAMAX) { BODY(OPS_sdwamax) }
        if (OP == LSHLOR) { BODY(OPS_lshlor) }
        if (OP == BFE) { BODY(OPS_bfe) }
        if (OP == AND) { BODY(OPS_and) }
        if (OP == CMPCND) { BODY(OPS_cmpcnd) }
        if (OP == MAD24) { BODY(OPS_mad24) }
        if (OP == SAD) { BODY(OPS_sad) }
        if (OP == FMA) { BODY(OPS_fma) }
        if (OP == BCNT) { BODY(OPS_bcnt) }
        if (OP == MAX16) { BODY(OPS_max16) }
        if (OP == SUB) { BODY(OPS_sub) }
        if (OP == MAXU32) { BODY(OPS_maxu32) }
        if (OP == OR3) { BODY(OPS_or3) }
        if (OP == ADD3) { BODY(OPS_add3) }
        if (OP == ANDOR) { BODY(OPS_andor) }
        if (OP == CND) { BODY(OPS_cnd) }
        if (OP == CMP) { BODY(OPS_cmp) }
        if (OP == ADDU16) { BODY(OPS_addu16) }
        if (OP == MUL24) { BODY(OPS_mul24) }
        if (OP == LSHR) { BODY(OPS_lshr) }
        if (OP == MOV) { BODY(OPS_mov) }
        if (OP == XOR) { BODY(OPS_xor) }
        if (OP == LSHL) { BODY(OPS_lshl) }
        if (OP == MIX) { asm volatile(R8(OPS_mix) : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5), [x6] "+v"(x6), [x7] "+v"(x7) : [y] "v"(y), [z] "v"(z)); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x % 64 == 0) {
        unsigned long long *c = cyc + (size_t)(blockIdx.x * 4 + threadIdx.x / 64) * 5;
        c[0] = t0; c[1] = t1; c[2] = r0; c[3] = r1; c[4] = xcc & 0xf;
    }
}

struct Res { double perClk, ghz, oneWave; };

template <int OP>
static Res run(unsigned *out, unsigned long long *cyc, int wavesPerSimd, int cus)
{
    const int blocks = cus * wavesPerSimd, iters = 16000 / wavesPerSimd + 2000;  // a block = 4 waves = one wave on each SIMD of a CU
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        (void)hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h((size_t)blocks * 4 * 5);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    const double instr = (OP == MIX ? 8.0 : 32.0) * iters * kInstrPerOp[OP];  // wave-instructions per wave
    // s_memrealtime is one chip-wide 100 MHz counter: the launch's span is max end - min start over ALL waves; s_memtime counters of
    // different XCDs are not comparable, so the shader clock comes from each wave's own (s_memtime span) / (s_memrealtime span)
    std::vector<double> ratio, own;
    unsigned long long r0 = ~0ull, r1 = 0;
    for (size_t w = 0; w < (size_t)blocks * 4; w++) {
        const unsigned long long *c = &h[w * 5];
        r0 = std::min(r0, c[2]); r1 = std::max(r1, c[3]);
        ratio.push_back((double)(c[1] - c[0]) / (double)(c[3] - c[2]));
        own.push_back((double)(c[1] - c[0]));
    }
    std::sort(ratio.begin(), ratio.end()); std::sort(own.begin(), own.end());
    const double ticksPer100MHz = ratio[ratio.size() / 2];
    Res r;
    r.ghz = ticksPer100MHz * 0.1;
    r.perClk = (double)blocks * 4 * instr / cus / ((double)(r1 - r0) * ticksPer100MHz);
    r.oneWave = instr / own[own.size() / 2];  // the median wave's own instructions per clock (its share of its SIMD; the arbiter favours old waves)
    return r;
}

template <int OP>
static void row(unsigned *out, unsigned long long *cyc, int cus)
{
    printf("| `%s` |", kNames[OP]);
    for (int w : {1, 2, 4, 8}) {
        Res r = run<OP>(out, cyc, w, cus);
        printf(" %.2f (%.2f GHz; median wave alone %.3f / clk) |", r.perClk, r.ghz, r.oneWave);
    }
    printf("\n");
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned *out; unsigned long long *cyc;
    (void)hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    (void)hipMalloc(&cyc, (size_t)cus * 8 * 4 * 5 * 8);
    for (int rep = 0; rep < 12; rep++) hipLaunchKernelGGL((k<FMA>), dim3(cus * 8), dim3(256), 0, 0, out, cyc, 4000);  // warm-up: clocks ramp
    (void)hipDeviceSynchronize();
    printf("device: %s, %d CUs, clock %d MHz\n\n", p.gcnArchName, cus, p.clockRate / 1000);
    printf("wave-instructions per shader clock per CU (span earliest wave start .. latest wave end); in brackets the clock held during "
           "the launch (s_memtime / s_memrealtime) and the median wave's own instructions per clock\n\n");
    printf("| instruction | 1 wave / SIMD | 2 waves / SIMD | 4 waves / SIMD | 8 waves / SIMD |\n|---|---|---|---|---|\n");
    row<ADD>(out, cyc, cus); row<SUB>(out, cyc, cus); row<AND>(out, cyc, cus); row<XOR>(out, cyc, cus); row<LSHL>(out, cyc, cus); row<LSHR>(out, cyc, cus); row<MOV>(out, cyc, cus);
    row<MAXU32>(out, cyc, cus); row<MAX16>(out, cyc, cus); row<ADDU16>(out, cyc, cus); row<MUL24>(out, cyc, cus); row<CND>(out, cyc, cus); row<CMP>(out, cyc, cus);
    row<OR3>(out, cyc, cus); row<ADD3>(out, cyc, cus); row<ANDOR>(out, cyc, cus); row<LSHLOR>(out, cyc, cus); row<BFE>(out, cyc, cus); row<MIN3>(out, cyc, cus);
    row<PKMAX>(out, cyc, cus); row<PKMIN>(out, cyc, cus); row<PKADD>(out, cyc, cus); row<PKSUBC>(out, cyc, cus);
    row<PERM>(out, cyc, cus); row<ALIGNB>(out, cyc, cus); row<DOT4>(out, cyc, cus); row<DOT2>(out, cyc, cus); row<SAD>(out, cyc, cus);
    row<SDWA>(out, cyc, cus); row<SDWAMAX>(out, cyc, cus); row<CMPCND>(out, cyc, cus); row<MAD24>(out, cyc, cus); row<BCNT>(out, cyc, cus); row<FMA>(out, cyc, cus); row<MIX>(out, cyc, cus);
    return 0;
}

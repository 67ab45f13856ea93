// Micro-benchmark (gfx950): the vector-ALU ISSUE RATE of the byte / packed-16-bit instructions the ORB kernels are made of, in
// wave-instructions per clock per CU, at 1 / 2 / 4 / 8 resident waves per SIMD on every CU.  Eight independent dependency chains
// per wave (x0..x7), so a chain's own latency is hidden from 1 wave per SIMD on if the pipeline is <= 8 issue slots deep.
// Cycles from s_memtime inside the kernel (shader clock; MI355X_MICROARCH.md constants table), wall time from HIP events beside it.
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
#define OPS_bcnt(x) "v_bcnt_u32_b32 " x ", " x ", %[y]\n"

enum Op { ADD, PKMAX, PKMIN, PKADD, PKSUBC, PERM, ALIGNB, DOT4, DOT2, MIN3, SDWA, SDWAMAX, LSHLOR, BFE, AND, CMPCND, MAD24, SAD, FMA, BCNT, N_OPS };
static const char *kNames[N_OPS] = {"v_add_u32", "v_pk_max_u16", "v_pk_min_u16", "v_pk_add_u16", "v_pk_sub_u16 clamp", "v_perm_b32", "v_alignbyte_b32", "v_dot4_u32_u8",
                                    "v_dot2_u32_u16", "v_min3_u32", "v_add_u32_sdwa (byte select)", "v_max_u16_sdwa (byte sel, word dst preserve)", "v_lshl_or_b32",
                                    "v_bfe_u32", "v_and_b32", "v_cmp_gt_u32 + v_cndmask_b32 (pair)", "v_mad_u32_u24", "v_sad_u8", "v_fma_f32", "v_bcnt_u32_b32"};
static const int kInstrPerOp[N_OPS] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1};

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned *out, unsigned long long *cyc, int iters)
{
    unsigned x0 = threadIdx.x, x1 = 7 + threadIdx.x, x2 = 9, x3 = 11 * threadIdx.x, x4 = 5, x5 = 77, x6 = threadIdx.x ^ 0x55, x7 = 123;
    unsigned y = 0x01030205u + threadIdx.x, z = 0x07060504u;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
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
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

struct Res { double perClkCycles, perClkWall; };

template <int OP>
static Res run(unsigned *out, unsigned long long *cyc, int wavesPerSimd, int cus)
{
    const int blocks = cus * wavesPerSimd, iters = 4000;  // a block = 4 waves = one wave on each SIMD of a CU (the dispatcher fills CUs round-robin)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double instr = 32.0 * iters * kInstrPerOp[OP];                         // wave-instructions per wave
    Res r;
    r.perClkCycles = 4.0 * wavesPerSimd * instr / med;                           // per CU: 4 SIMDs x waves, over the median wave's own cycles
    r.perClkWall = (double)blocks * 4 * instr / cus / (ms * 1e-3 * 2.4e9);       // whole launch at a nominal 2.4 GHz (includes launch ramp and clock droop)
    return r;
}

template <int OP>
static void row(unsigned *out, unsigned long long *cyc, int cus)
{
    printf("| `%s` |", kNames[OP]);
    for (int w : {1, 2, 4, 8}) {
        Res r = run<OP>(out, cyc, w, cus);
        printf(" %.2f (%.2f) |", r.perClkCycles, r.perClkWall);
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
    (void)hipMalloc(&cyc, (size_t)cus * 8 * 4 * 8);
    printf("device: %s, %d CUs, clock %d MHz\n\n", p.gcnArchName, cus, p.clockRate / 1000);
    printf("wave-instructions per clock per CU: from the median wave's own s_memtime cycles (in brackets: from the launch's wall time at a nominal 2.4 GHz)\n\n");
    printf("| instruction | 1 wave / SIMD | 2 waves / SIMD | 4 waves / SIMD | 8 waves / SIMD |\n|---|---|---|---|---|\n");
    row<ADD>(out, cyc, cus); row<AND>(out, cyc, cus); row<LSHLOR>(out, cyc, cus); row<BFE>(out, cyc, cus); row<MIN3>(out, cyc, cus);
    row<PKMAX>(out, cyc, cus); row<PKMIN>(out, cyc, cus); row<PKADD>(out, cyc, cus); row<PKSUBC>(out, cyc, cus);
    row<PERM>(out, cyc, cus); row<ALIGNB>(out, cyc, cus); row<DOT4>(out, cyc, cus); row<DOT2>(out, cyc, cus); row<SAD>(out, cyc, cus);
    row<SDWA>(out, cyc, cus); row<SDWAMAX>(out, cyc, cus); row<CMPCND>(out, cyc, cus); row<MAD24>(out, cyc, cus); row<BCNT>(out, cyc, cus); row<FMA>(out, cyc, cus);
    return 0;
}

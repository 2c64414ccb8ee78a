// valu_issue_table.hip — sustained issue cost (cycles per wave-instruction per SIMD) of the instruction
// classes the SPH kernels are made of, on gfx950, at 1 and 8 waves per SIMD.  The "VALU issue ceiling" that
// bench.py's roofline.valu_issue refers to is priced with this table.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_issue_table.hip -o /tmp/valu_issue_table
// Every loop body is 64 instructions over 8 independent registers (dependency distance 8), inline asm so the
// compiler cannot fold or re-associate anything.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define KERNEL(NAME, INSTR_PER_X, ASM)                                                                            \
    __global__ __launch_bounds__(256) void NAME(unsigned* out, unsigned long long* cyc, int iters, float sa, unsigned su) { \
        unsigned r[8];                                                                                            \
        for (int j = 0; j < 8; ++j) r[j] = 0x3f800000u + threadIdx.x * 8 + j;                                     \
        float va = sa + (float)threadIdx.x; unsigned vu = su + threadIdx.x;                                       \
        const unsigned long long t0 = clock64();                                                                  \
        for (int i = 0; i < iters; ++i) {                                                                         \
            asm volatile(ASM                                                                                      \
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) \
                         : "s"(sa), "s"(su), "v"(va), "v"(vu)                                                     \
                         : "vcc");                                                                                \
        }                                                                                                         \
        const unsigned long long t1 = clock64();                                                                  \
        unsigned acc = 0;                                                                                         \
        for (int j = 0; j < 8; ++j) acc ^= r[j];                                                                  \
        out[blockIdx.x * 256 + threadIdx.x] = acc;                                                                \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                          \
    }

#define X_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %8\n\t"
#define X_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n\t"
#define X_ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n\t"
#define X_ADDU(i) "v_add_u32 %" #i ", %" #i ", %9\n\t"
#define X_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n\t"
#define X_AND(i) "v_and_b32 %" #i ", %9, %" #i "\n\t"
#define X_MOV(i) "v_mov_b32 %" #i ", %9\n\t"
#define X_MINU(i) "v_min_u32 %" #i ", %" #i ", %9\n\t"
#define X_FFBH(i) "v_ffbh_u32 %" #i ", %" #i "\n\t"
#define X_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n\t"
#define X_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n\t"
#define X_RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n\t"
#define X_CMPSEL(i) "v_cmp_gt_f32 vcc, %8, %" #i "\n\tv_cndmask_b32 %" #i ", %" #i ", %11, vcc\n\t"
#define X_CMPADDC(i) "v_cmp_nlt_f32 vcc, %8, %" #i "\n\tv_addc_co_u32 %" #i ", vcc, %" #i ", %" #i ", vcc\n\t"
#define X_CMPU_SEL2(i) "v_cmp_gt_u32 vcc, %9, %" #i "\n\tv_cndmask_b32 %" #i ", %" #i ", %11, vcc\n\tv_cndmask_b32 %" #i ", %11, %" #i ", vcc\n\t"
#define X_PKFMA(i) ""   /* placeholder: packed forms below use register pairs */
#define X_MADU64(i) "v_mul_lo_u32 %" #i ", %" #i ", %9\n\t"
#define X_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %9\n\t"
#define X_CVT(i) "v_cvt_f32_u32 %" #i ", %" #i "\n\t"
#define X_FLOOR(i) "v_floor_f32 %" #i ", %" #i "\n\t"
#define X_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 5\n\t"
#define X_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %8\n\t"
#define X_XOR3(i) "v_xad_u32 %" #i ", %" #i ", %9, %9\n\t"
#define X_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %9, %9\n\t"
#define X_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %9\n\t"
#define X_READLANE(i) "v_readfirstlane_b32 s20, %" #i "\n\t"

KERNEL(k_fma, 1, BODY64(X_FMA))
KERNEL(k_mul, 1, BODY64(X_MUL))
KERNEL(k_add, 1, BODY64(X_ADD))
KERNEL(k_addu, 1, BODY64(X_ADDU))
KERNEL(k_lshl, 1, BODY64(X_LSHL))
KERNEL(k_and, 1, BODY64(X_AND))
KERNEL(k_mov, 1, BODY64(X_MOV))
KERNEL(k_minu, 1, BODY64(X_MINU))
KERNEL(k_ffbh, 1, BODY64(X_FFBH))
KERNEL(k_rcp, 1, BODY64(X_RCP))
KERNEL(k_sqrt, 1, BODY64(X_SQRT))
KERNEL(k_rsq, 1, BODY64(X_RSQ))
KERNEL(k_cmpsel, 2, BODY64(X_CMPSEL))
KERNEL(k_cmpaddc, 2, BODY64(X_CMPADDC))
KERNEL(k_cmpu_sel2, 3, BODY64(X_CMPU_SEL2))
KERNEL(k_mullo, 1, BODY64(X_MADU64))
KERNEL(k_mulhi, 1, BODY64(X_MULHI))
KERNEL(k_cvt, 1, BODY64(X_CVT))
KERNEL(k_floor, 1, BODY64(X_FLOOR))
KERNEL(k_bfe, 1, BODY64(X_BFE))
KERNEL(k_max3, 1, BODY64(X_MAX3))
KERNEL(k_xad, 1, BODY64(X_XOR3))
KERNEL(k_add3, 1, BODY64(X_ADD3))
KERNEL(k_lshladd, 1, BODY64(X_LSHLADD))

// packed f32: 4 register pairs
__global__ __launch_bounds__(256) void k_pkfma(unsigned* out, unsigned long long* cyc, int iters, float sa, unsigned su) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[4];
    for (int j = 0; j < 4; ++j) p[j] = f2{1.0f + threadIdx.x, 2.0f + j};
    const f2 a = {sa, sa};
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#define PK(i) "v_pk_fma_f32 %" #i ", %" #i ", %4, %4\n\t"
        asm volatile(PK(0) PK(1) PK(2) PK(3) PK(0) PK(1) PK(2) PK(3) PK(0) PK(1) PK(2) PK(3) PK(0) PK(1) PK(2) PK(3)
                     PK(0) PK(1) PK(2) PK(3) PK(0) PK(1) PK(2) PK(3) PK(0) PK(1) PK(2) PK(3) PK(0) PK(1) PK(2) PK(3)
                     PK(0) PK(1) PK(2) PK(3) PK(0) PK(1) PK(2) PK(3) PK(0) PK(1) PK(2) PK(3) PK(0) PK(1) PK(2) PK(3)
                     PK(0) PK(1) PK(2) PK(3) PK(0) PK(1) PK(2) PK(3) PK(0) PK(1) PK(2) PK(3) PK(0) PK(1) PK(2) PK(3)
                     : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3])
                     : "v"(a));
    }
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = __float_as_uint(p[0].x + p[1].y + p[2].x + p[3].y);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kfn)(unsigned*, unsigned long long*, int, float, unsigned);

static void run(const char* name, kfn k, int per_x, int blocks, unsigned* d, unsigned long long* dc, FILE* csv) {
    const int iters = 2048;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, dc, iters, 1.0001f, 3u);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, dc, iters, 1.0001f, 3u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    static unsigned long long hc[8192];
    hipMemcpy(hc, dc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < blocks; ++i) mean += (double)hc[i];
    mean /= blocks;
    const double instr_per_wave = (double)iters * 64 * per_x;
    const double waves_per_simd = blocks * 4.0 / 1024.0;
    // in-kernel: a wave's elapsed shader cycles / its instructions / co-resident waves on its SIMD
    const double cyc_in = mean / instr_per_wave / (waves_per_simd < 1 ? 1 : waves_per_simd);
    const double winstr = blocks * 4.0 * instr_per_wave;
    const double cyc_wall = 1024.0 * 2.4e9 / (winstr / (ms * 1e-3));
    printf("%-34s waves/SIMD=%4.1f  %.3f ms  in-kernel %.2f cyc/instr/SIMD   wall@2.4GHz %.2f\n", name, waves_per_simd, ms, cyc_in,
           cyc_wall);
    if (csv) fprintf(csv, "%s,%.1f,%.4f,%.3f,%.3f\n", name, waves_per_simd, ms, cyc_in, cyc_wall);
}

int main(int argc, char** argv) {
    unsigned* d;
    unsigned long long* dc;
    hipMalloc(&d, 256 * 8192 * 4);
    hipMalloc(&dc, 8192 * 8);
    FILE* csv = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (csv) fprintf(csv, "instruction,waves_per_simd,ms,cycles_per_instr_per_simd_in_kernel,cycles_per_instr_per_simd_wall_2.4GHz\n");
    struct { const char* n; kfn k; int per; } T[] = {
        {"v_fma_f32", k_fma, 1}, {"v_mul_f32", k_mul, 1}, {"v_add_f32", k_add, 1}, {"v_pk_fma_f32", k_pkfma, 1},
        {"v_max3_f32", k_max3, 1}, {"v_floor_f32", k_floor, 1}, {"v_cvt_f32_u32", k_cvt, 1},
        {"v_add_u32", k_addu, 1}, {"v_add3_u32", k_add3, 1}, {"v_lshl_add_u32", k_lshladd, 1}, {"v_xad_u32", k_xad, 1},
        {"v_lshlrev_b32", k_lshl, 1}, {"v_and_b32", k_and, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_mov_b32", k_mov, 1},
        {"v_min_u32", k_minu, 1}, {"v_ffbh_u32", k_ffbh, 1}, {"v_mul_lo_u32", k_mullo, 1}, {"v_mul_hi_u32", k_mulhi, 1},
        {"v_rcp_f32", k_rcp, 1}, {"v_sqrt_f32", k_sqrt, 1}, {"v_rsq_f32", k_rsq, 1},
        {"v_cmp_gt_f32+v_cndmask (per instr)", k_cmpsel, 2}, {"v_cmp_nlt_f32+v_addc_co (per instr)", k_cmpaddc, 2},
        {"v_cmp_gt_u32+2*v_cndmask (per instr)", k_cmpu_sel2, 3},
    };
    for (int blocks : {256, 2048}) {      // 1 wave per SIMD, 8 waves per SIMD
        for (auto& t : T) run(t.n, t.k, t.per, blocks, d, dc, csv);
        printf("\n");
    }
    if (csv) fclose(csv);
    return 0;
}

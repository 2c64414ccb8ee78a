// valu_peak.hip — micro-benchmark: sustained issue rate of plain and packed f32 VALU on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    const f2 pa = {a, a}, pb = {b, b};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
                p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
                p0 = __builtin_elementwise_fma(p0, pa, pb); p1 = __builtin_elementwise_fma(p1, pa, pb);
                p2 = __builtin_elementwise_fma(p2, pa, pb); p3 = __builtin_elementwise_fma(p3, pa, pb);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // mul + add, non-fused (what -ffp-contract=off code looks like)
                x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x0 = x0 + b; x1 = x1 + b; x2 = x2 + b; x3 = x3 + b;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE>
void run(const char* name, int blocks, float* d, int instr_per_iter) {
    const int iters = 4096;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double waves = (double)blocks * 4, winstr = waves * iters * instr_per_iter;
    printf("%-28s blocks=%5d  %.3f ms  %.1f G wave-instr/s  -> %.2f cycles/instr/SIMD @2.4GHz\n", name, blocks, ms,
           winstr / ms / 1e6, 1024.0 * 2.4e9 / (winstr / (ms * 1e-3)));
}
int main() {
    float* d; hipMalloc(&d, 256 * 8192 * 4);
    for (int blocks : {1024, 2048, 8192}) {
        run<0>("v_fma_f32 x32/iter", blocks, d, 32);
        run<1>("v_pk_fma_f32 x32/iter", blocks, d, 32);
        run<2>("v_mul+v_add x32/iter", blocks, d, 32);
    }
    return 0;
}

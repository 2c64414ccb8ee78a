// div_peak.hip — sustained rate of correctly-rounded f32 divide / sqrt on gfx950 (what bounds k_force).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE, int ILP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a) {
    float x[ILP];
#pragma unroll
    for (int u = 0; u < ILP; ++u) x[u] = 1.0f + threadIdx.x * 0.001f + u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            if (MODE == 0) x[u] = a / x[u] + 1.5f;                 // IEEE divide (+1 add)
            else if (MODE == 1) x[u] = __builtin_sqrtf(x[u]) + 2.0f;   // IEEE sqrt (+1 add)
            else x[u] = x[u] * a + 1.5f;                           // mul + add
        }
    }
    float s = 0;
#pragma unroll
    for (int u = 0; u < ILP; ++u) s += x[u];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int ILP>
void run(const char* name, int blocks, float* d) {
    const int iters = 2048;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, ILP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.37f);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, ILP>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.37f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * 4 * iters * ILP;   // wave-level ops
    printf("%-22s ILP=%d blocks=%5d %.3f ms  %.1f G wave-ops/s  -> %.1f SIMD-cycles per wave-op @2.1GHz\n", name, ILP, blocks, ms,
           ops / ms / 1e6, 1024.0 * 2.1e9 / (ops / (ms * 1e-3)));
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 8192 * 4);
    for (int blocks : {2048, 8192}) {
        run<0, 1>("fdiv (IEEE)", blocks, d); run<0, 2>("fdiv (IEEE)", blocks, d); run<0, 4>("fdiv (IEEE)", blocks, d); run<0, 8>("fdiv (IEEE)", blocks, d);
        run<1, 1>("sqrt (IEEE)", blocks, d); run<1, 4>("sqrt (IEEE)", blocks, d);
        run<2, 4>("mul+add", blocks, d);
    }
    return 0;
}

// cycles per quotient: plain IEEE divide vs shared-reciprocal variants (4 numerators per denominator)
#include <hip/hip_runtime.h>
#include <cstdio>
// EXPERIMENT RECORD (round 1): sharing the reciprocal refinement of hipcc's IEEE f32 division
// between quotients with one denominator is bit-identical to `/` (checked on 2^22 x 4 operand
// sets incl. raw bit patterns and specials) and costs 30 instead of 45 SIMD-cycles per quotient
// in this isolated loop — but inside k_force it was SLOWER (1.11-1.52 ms vs 1.04 ms): the guard /
// fallback either adds branches that stop two neighbours' chains from interleaving, or doubles
// the body and costs occupancy.  The product therefore uses plain `/`.
struct SharedRcp { float ds, r1; };
__device__ __forceinline__ SharedRcp rcp_shared(float n0, float d) {
    bool flag; SharedRcp R;
    R.ds = __builtin_amdgcn_div_scalef(n0, d, false, &flag);
    const float r0 = __builtin_amdgcn_rcpf(R.ds);
    const float e = __builtin_fmaf(-R.ds, r0, 1.0f);
    R.r1 = __builtin_fmaf(e, r0, r0);
    return R;
}
__device__ __forceinline__ float div_shared(const SharedRcp& R, float n, float d) {
    bool f0, f1;
    const float ds = __builtin_amdgcn_div_scalef(n, d, false, &f0);
    if (__float_as_uint(ds) != __float_as_uint(R.ds)) return n / d;
    const float ns = __builtin_amdgcn_div_scalef(n, d, true, &f1);
    float q = ns * R.r1;
    float e = __builtin_fmaf(-ds, q, ns);
    q = __builtin_fmaf(e, R.r1, q);
    e = __builtin_fmaf(-ds, q, ns);
    q = __builtin_amdgcn_div_fmasf(e, R.r1, q, f1);
    return __builtin_amdgcn_div_fixupf(q, d, n);
}
__device__ __forceinline__ float div_shared_nocheck(const SharedRcp& R, float n, float d) {
    bool f1;
    const float ns = __builtin_amdgcn_div_scalef(n, d, true, &f1);
    float q = ns * R.r1;
    float e = __builtin_fmaf(-R.ds, q, ns);
    q = __builtin_fmaf(e, R.r1, q);
    e = __builtin_fmaf(-R.ds, q, ns);
    q = __builtin_amdgcn_div_fmasf(e, R.r1, q, f1);
    return __builtin_amdgcn_div_fixupf(q, d, n);
}
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a) {
    float x0 = 1.0f + threadIdx.x * 0.001f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, d = 1.5f + threadIdx.x * 0.01f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { x0 = x0 / d + a; x1 = x1 / d + a; x2 = x2 / d + a; x3 = x3 / d + a; }
        else if (MODE == 1) { const SharedRcp R = rcp_shared(x0, d); x0 = div_shared(R, x0, d) + a; x1 = div_shared(R, x1, d) + a; x2 = div_shared(R, x2, d) + a; x3 = div_shared(R, x3, d) + a; }
        else { const SharedRcp R = rcp_shared(x0, d); x0 = div_shared_nocheck(R, x0, d) + a; x1 = div_shared_nocheck(R, x1, d) + a; x2 = div_shared_nocheck(R, x2, d) + a; x3 = div_shared_nocheck(R, x3, d) + a; }
        d += 0.001f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3;
}
template <int MODE> void run(const char* name, float* d) {
    const int iters = 2048, blocks = 8192;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.37f);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.37f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * 4 * iters * 4;
    printf("%-28s %.3f ms -> %.1f SIMD-cycles per quotient @2.1GHz\n", name, ms, 1024.0 * 2.1e9 / (ops / (ms * 1e-3)));
}
int main() { float* d; (void)hipMalloc(&d, 256 * 8192 * 4); run<0>("plain / (x4 same denom)", d); run<1>("div_shared checked", d); run<2>("div_shared unchecked", d); return 0; }

// EXPERIMENT (round 1): lean correctly-rounded reciprocal / square root for the guarded ranges of
// the force pass.  Enumerates EVERY f32 in the range and compares with hipcc's correctly rounded
// 1.0f/b and __builtin_sqrtf(x); prints mismatch counts and the first failing bit patterns.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/unary_exact.hip -o /tmp/unary_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float rcp_nr1(float b) {
    float y = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
__device__ __forceinline__ float rcp_nr2(float b) {
    float y = rcp_nr1(b);
    const float e = __builtin_fmaf(-b, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
// v_sqrt_f32 then pick among s-1ulp, s, s+1ulp by the sign of the exact residuals (the core of
// hipcc's expansion, without its denormal scaling and special-case selects)
__device__ __forceinline__ float sqrt_fix(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float edn = __builtin_fmaf(-dn, s, x), eup = __builtin_fmaf(-up, s, x);
    float r = edn <= 0.0f ? dn : s;
    r = eup > 0.0f ? up : r;
    return r;
}
// Newton/Markstein: s' = RN(s + (x - s*s) * h), h ~ 1/(2s)
__device__ __forceinline__ float sqrt_nm(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, h, s);
}
template <int WHICH>
__global__ __launch_bounds__(256) void k(uint32_t lo_bits, uint32_t hi_bits, unsigned long long* bad, uint32_t* first) {
    const uint32_t stride = gridDim.x * 256;
    uint32_t n = 0;
    for (uint64_t b = (uint64_t)lo_bits + blockIdx.x * 256 + threadIdx.x; b <= hi_bits; b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        float got, ref;
        if (WHICH == 0) { got = rcp_nr1(x); ref = __fdiv_rn(1.0f, x); }
        else if (WHICH == 1) { got = rcp_nr2(x); ref = __fdiv_rn(1.0f, x); }
        else if (WHICH == 2) { got = sqrt_fix(x); ref = __builtin_sqrtf(x); }
        else { got = sqrt_nm(x); ref = __builtin_sqrtf(x); }
        if (__float_as_uint(got) != __float_as_uint(ref)) {
            ++n;
            const uint32_t slot = atomicAdd(first, 1u);
            if (slot < 8u) first[1 + slot] = (uint32_t)b;
        }
    }
    if (n) atomicAdd(bad, (unsigned long long)n);
}
template <int WHICH> void run(const char* name, float lo, float hi) {
    unsigned long long* bad; uint32_t* first;
    (void)hipMalloc((void**)&bad, 8); (void)hipMalloc((void**)&first, 64);
    (void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0, 64);
    uint32_t lb, hb; memcpy(&lb, &lo, 4); memcpy(&hb, &hi, 4);
    hipLaunchKernelGGL(k<WHICH>, dim3(8192), dim3(256), 0, 0, lb, hb, bad, first);
    unsigned long long h; uint32_t f[16];
    (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(f, first, 64, hipMemcpyDeviceToHost);
    printf("%-34s [%g, %g]: %u inputs, %llu mismatches", name, lo, hi, hb - lb + 1, h);
    for (uint32_t i = 0; i < (f[0] < 8 ? f[0] : 8); ++i) printf(" 0x%08x", f[1 + i]);
    printf("\n");
}
int main() {
    run<0>("rcp: v_rcp + 1 Newton step", 0x1p-20f, 0x1p20f);
    run<1>("rcp: v_rcp + 2 Newton steps", 0x1p-20f, 0x1p20f);
    run<2>("sqrt: v_sqrt + +-1ulp residual fix", 0x1p-40f, 0x1p40f);
    run<3>("sqrt: v_sqrt + Newton/Markstein", 0x1p-40f, 0x1p40f);
    return 0;
}

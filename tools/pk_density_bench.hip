// pk_density_bench.hip — the 2D density pass's candidate loop in isolation, plain f32 against packed f32
// (v_pk_add_f32 / v_pk_mul_f32), at 8 waves per SIMD: does the packed form buy issue slots on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize tools/pk_density_bench.hip -o /tmp/pk_density_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float add_rn(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float term1(float h2, float poly6, float mass, float2 me, float2 q) {
    const float dx = q.x - me.x, dy = q.y - me.y;
    const float r2 = dx * dx + dy * dy;
    float kern = 0.0f;
    if (!(r2 > h2)) { const float d = h2 - r2; kern = poly6 * d * d * d; }
    return mass * kern;
}
__device__ __forceinline__ f2 term2(float h2, float poly6, float mass, f2 me, float2 qa, float2 qb) {
    const f2 da = f2{qa.x, qa.y} - me, db = f2{qb.x, qb.y} - me;
    const f2 sa = da * da, sb = db * db;
    const f2 r2 = {add_rn(sa.x, sa.y), add_rn(sb.x, sb.y)};
    const f2 diff = f2{h2, h2} - r2;
    f2 kern = ((f2{poly6, poly6} * diff) * diff) * diff;
    kern.x = r2.x > h2 ? 0.0f : kern.x;
    kern.y = r2.y > h2 ? 0.0f : kern.y;
    return f2{mass, mass} * kern;
}
template <int PACKED>
__global__ __launch_bounds__(256) void k(const float2* __restrict__ p, float* out, float h2, float poly6, float mass, int n, int reps) {
    __shared__ float2 s[1024];
    for (int j = threadIdx.x; j < 1024; j += 256) s[j] = p[j];
    __syncthreads();
    const float2 m = p[threadIdx.x];
    float rho = 0;
    for (int r = 0; r < reps; ++r) {
        const float2* sp = s + ((r * 7 + (threadIdx.x >> 6)) & 255);
        if (PACKED) {
            const f2 me = {m.x, m.y};
            for (int k = 0; k + 4 <= n; k += 4) {
                const f2 a = term2(h2, poly6, mass, me, sp[k], sp[k + 1]);
                const f2 b = term2(h2, poly6, mass, me, sp[k + 2], sp[k + 3]);
                rho += a.x; rho += a.y; rho += b.x; rho += b.y;
            }
        } else {
            for (int k = 0; k + 4 <= n; k += 4) {
                const float t0 = term1(h2, poly6, mass, m, sp[k]), t1 = term1(h2, poly6, mass, m, sp[k + 1]);
                const float t2 = term1(h2, poly6, mass, m, sp[k + 2]), t3 = term1(h2, poly6, mass, m, sp[k + 3]);
                rho += t0; rho += t1; rho += t2; rho += t3;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = rho;
}
int main() {
    float2* p; float* o;
    hipMalloc(&p, 2048 * sizeof(float2)); hipMalloc(&o, 8192 * 256 * 4);
    float2 h[2048];
    for (int i = 0; i < 2048; ++i) h[i] = make_float2((i % 37) * 0.01f, (i % 53) * 0.013f);
    hipMemcpy(p, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {2048, 4096}) for (int pk = 0; pk < 2; ++pk) for (int rep = 0; rep < 2; ++rep) {
        const int n = 96, reps = 200;
        hipEventRecord(a);
        if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, p, o, 0.04f, 3.0f, 1.0f, n, reps);
        else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, p, o, 0.04f, 3.0f, 1.0f, n, reps);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        float r0; hipMemcpy(&r0, o + 77, 4, hipMemcpyDeviceToHost);
        printf("blocks %d packed %d: %.3f ms  (%.3f ns per candidate per wave-slot)  out %08x\n", blocks, pk, ms,
               ms * 1e6 / ((double)blocks * 4 * n * reps / 1024.0), *(unsigned*)&r0);
    }
    return 0;
}

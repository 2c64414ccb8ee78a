// kernels_sort.hip — the reference's bitonic network (sort.wgsl:27-51, schedule
// src/simulation.rs:323-347) on 8-byte (key<<32 | source index) pairs.
//
// The reference issues S(S+1)/2 full-array dispatches over 32-byte records.  The
// network is data-oblivious and compares keys only (strict `>`, so equal keys
// never swap), hence sorting pairs and gathering the payload afterwards yields
// the bit-identical arrangement.  Here every step whose compare distance fits a
// workgroup tile runs out of LDS:
//   * k_bitonic_local<INIT>: stages 0..LOG_T-1 entirely inside one tile;
//   * k_bitonic_global:      one step with block size > tile (HBM pass);
//   * k_bitonic_local<TAIL>: the remaining steps of a stage (distance T/2..1).
// Elements at index >= n do not exist in the reference (`if index_high >=
// num_values return`, sort.wgsl:39-41); pairs touching them are skipped.
#include "fs_device.h"
#include "fs_kernels.h"

namespace fsd {

#define SORT_LOG_T 12
#define SORT_T (1u << SORT_LOG_T)
#define SORT_THREADS 256

__device__ __forceinline__ void cmpx_lds(u64* s, uint32_t lo, uint32_t hi, uint32_t base, uint32_t n) {
    if (base + hi >= n) return;
    const u64 a = s[lo], b = s[hi];
    if ((uint32_t)(a >> 32) > (uint32_t)(b >> 32)) { s[lo] = b; s[hi] = a; }
}

// One step over the tile held in LDS.  sh = stage - step (gw = 1 << sh).
__device__ __forceinline__ void local_step(u64* s, uint32_t sh, bool flip, uint32_t base, uint32_t n) {
    const uint32_t gw = 1u << sh;
#pragma unroll
    for (uint32_t m = 0; m < SORT_T / 2 / SORT_THREADS; ++m) {
        const uint32_t p = threadIdx.x + m * SORT_THREADS;
        const uint32_t lo = ((p >> sh) << (sh + 1)) | (p & (gw - 1));
        const uint32_t hi = flip ? (lo ^ ((gw << 1) - 1u)) : (lo | gw);
        cmpx_lds(s, lo, hi, base, n);
    }
    __syncthreads();
}

template <bool INIT>
__global__ __launch_bounds__(SORT_THREADS) void k_bitonic_local(u64* __restrict__ pairs, uint32_t n,
                                                                uint32_t num_stages) {
    __shared__ u64 s[SORT_T];
    const uint32_t base = blockIdx.x * SORT_T;
#pragma unroll
    for (uint32_t m = 0; m < SORT_T / SORT_THREADS; ++m) {
        const uint32_t j = threadIdx.x + m * SORT_THREADS;
        s[j] = (base + j < n) ? pairs[base + j] : ~0ull;
    }
    __syncthreads();
    if (INIT) {
        for (uint32_t stage = 0; stage < num_stages; ++stage) {
            local_step(s, stage, true, base, n);                       // step 0: mirrored compare
            for (uint32_t step = 1; step <= stage; ++step) local_step(s, stage - step, false, base, n);
        }
    } else {
        for (int sh = SORT_LOG_T - 1; sh >= 0; --sh) local_step(s, (uint32_t)sh, false, base, n);
    }
#pragma unroll
    for (uint32_t m = 0; m < SORT_T / SORT_THREADS; ++m) {
        const uint32_t j = threadIdx.x + m * SORT_THREADS;
        if (base + j < n) pairs[base + j] = s[j];
    }
}

// One global step: thread p handles the pair (lo, hi) exactly as sort.wgsl:29-50.
__global__ __launch_bounds__(256) void k_bitonic_global(u64* __restrict__ pairs, uint32_t n, uint32_t sh, int flip,
                                                        uint32_t num_pairs) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= num_pairs) return;
    const uint32_t gw = 1u << sh;
    const uint32_t lo = ((p >> sh) << (sh + 1)) | (p & (gw - 1));
    const uint32_t hi = flip ? (lo ^ ((gw << 1) - 1u)) : (lo | gw);
    if (hi >= n) return;
    const u64 a = pairs[lo], b = pairs[hi];
    if ((uint32_t)(a >> 32) > (uint32_t)(b >> 32)) { pairs[lo] = b; pairs[hi] = a; }
}

int launch_bitonic_sort(hipStream_t st, u64* pairs, uint32_t n) {
    if (n <= 1) return 0;
    uint32_t p2 = 1, S = 0;
    while (p2 < n) { p2 <<= 1; ++S; }
    const uint32_t tiles = (n + SORT_T - 1) / SORT_T;
    int launches = 0;
    const uint32_t init_stages = S < SORT_LOG_T ? S : SORT_LOG_T;
    hipLaunchKernelGGL(k_bitonic_local<true>, dim3(tiles), dim3(SORT_THREADS), 0, st, pairs, n, init_stages);
    ++launches;
    const uint32_t num_pairs = p2 / 2;
    for (uint32_t stage = SORT_LOG_T; stage < S; ++stage) {
        // steps whose block (2 << sh) exceeds the tile run in HBM
        for (uint32_t step = 0; stage - step >= SORT_LOG_T; ++step) {
            hipLaunchKernelGGL(k_bitonic_global, dim3((num_pairs + 255) / 256), dim3(256), 0, st, pairs, n,
                               stage - step, step == 0 ? 1 : 0, num_pairs);
            ++launches;
        }
        hipLaunchKernelGGL(k_bitonic_local<false>, dim3(tiles), dim3(SORT_THREADS), 0, st, pairs, n, 0u);
        ++launches;
    }
    return launches;
}

}  // namespace fsd

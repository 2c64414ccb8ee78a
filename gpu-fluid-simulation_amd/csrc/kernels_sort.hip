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
#include <stdlib.h>

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

// M consecutive global steps of one stage in ONE pass, register-blocked: a thread owns the
// 2^M elements whose indices differ only in bits [a-M+1, a] (a = stage - first step), loads
// them (each load is a coalesced 512-B wave segment: consecutive lanes = consecutive
// columns), runs the M compare-exchange steps in VGPRs and stores them back.  No LDS, no
// barriers; HBM/MALL traffic per M steps = one read + one write of the pair array.
//
// FLIP: the first step of a stage compares x with its mirror x ^ (2^(a+1)-1)
// (sort.wgsl:32-36, `group_height - 2*h`).  In "virtual" indices v (upper-half rows read
// from p = v ^ (2^a - 1)) the mirror step is a plain distance-2^a step; the later steps of
// the pass act on upper-half rows in reversed physical order, so the compare is reversed
// there.  Indices >= n hold a never-moving sentinel (see file header).
template <int M, bool FLIP>
__global__ __launch_bounds__(256) void k_bitonic_strided(u64* __restrict__ pairs, uint32_t n, uint32_t a,
                                                         uint32_t num_threads) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= num_threads) return;
    constexpr int R = 1 << M;
    const uint32_t low = a - (uint32_t)M + 1u;
    const uint32_t vbase = ((g >> low) << (a + 1u)) | (g & ((1u << low) - 1u));
    const uint32_t mirror = (1u << a) - 1u;
    u64 x[R];
    u64 changed = 0;   // bit r set when x[r] took part in a swap: untouched elements are not stored
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t v = vbase | ((uint32_t)r << low);
        const uint32_t p = (FLIP && (r >> (M - 1))) ? (v ^ mirror) : v;
        x[r] = p < n ? pairs[p] : ~0ull;
    }
#pragma unroll
    for (int b = M - 1; b >= 0; --b) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r & (1 << b)) continue;
            const int r1 = r | (1 << b);
            const bool rev = FLIP && b < M - 1 && (r >> (M - 1));   // upper half after the mirror step
            const uint32_t klo = (uint32_t)((rev ? x[r1] : x[r]) >> 32);
            const uint32_t khi = (uint32_t)((rev ? x[r] : x[r1]) >> 32);
            if (klo > khi) {
                const u64 t = x[r]; x[r] = x[r1]; x[r1] = t;
                changed |= (1ull << r) | (1ull << r1);
            }
        }
    }
    if (changed == 0) return;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t v = vbase | ((uint32_t)r << low);
        const uint32_t p = (FLIP && (r >> (M - 1))) ? (v ^ mirror) : v;
        if ((changed >> r) & 1ull) pairs[p] = x[r];   // a sentinel (p >= n) never swaps, so p < n here
    }
}

template <int M>
static void launch_strided(hipStream_t st, u64* pairs, uint32_t n, uint32_t a, bool flip, uint32_t p2) {
    const uint32_t threads = p2 >> M;
    const dim3 grid((threads + 255u) / 256u), block(256);
    if (flip) hipLaunchKernelGGL((k_bitonic_strided<M, true>), grid, block, 0, st, pairs, n, a, threads);
    else hipLaunchKernelGGL((k_bitonic_strided<M, false>), grid, block, 0, st, pairs, n, a, threads);
}

static int sort_mmax() {
    static int m = [] {
        const char* e = getenv("FS_SORT_MMAX");
        int v = e ? atoi(e) : 5;
        return v < 1 ? 1 : (v > 6 ? 6 : v);
    }();
    return m;
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
    const int mmax = sort_mmax();
    for (uint32_t stage = SORT_LOG_T; stage < S; ++stage) {
        // steps whose block (2 << sh) exceeds the tile: sh = stage .. SORT_LOG_T, in passes of <= mmax steps
        const int gsteps = (int)(stage - SORT_LOG_T + 1);
        const int npass = (gsteps + mmax - 1) / mmax;
        uint32_t a = stage;
        for (int ps = 0; ps < npass; ++ps) {
            const int m = gsteps / npass + (ps < gsteps % npass ? 1 : 0);
            const bool flip = ps == 0;
            switch (m) {
                case 1: launch_strided<1>(st, pairs, n, a, flip, p2); break;
                case 2: launch_strided<2>(st, pairs, n, a, flip, p2); break;
                case 3: launch_strided<3>(st, pairs, n, a, flip, p2); break;
                case 4: launch_strided<4>(st, pairs, n, a, flip, p2); break;
                case 5: launch_strided<5>(st, pairs, n, a, flip, p2); break;
                default: launch_strided<6>(st, pairs, n, a, flip, p2); break;
            }
            a -= (uint32_t)m;
            ++launches;
        }
        hipLaunchKernelGGL(k_bitonic_local<false>, dim3(tiles), dim3(SORT_THREADS), 0, st, pairs, n, 0u);
        ++launches;
    }
    return launches;
}

}  // namespace fsd

// kernels_csort.hip — FS_SORT_COUNTING: O(N) cell counting sort, hand-written.
// NOT the reference's algorithm: it yields the STABLE arrangement (ascending source index
// inside a cell) instead of the bitonic network's tie order, so floats agree with the
// reference path only to summation-order tolerance; keys and cell starts are identical.
// Deterministic (no dependence on atomic arrival order):
//   k_cs_hist      key per particle, per-cell histogram (atomics only count)
//   k_cs_scan_*    exclusive scan of the histogram = dense cell-start table `cs`
//   k_cs_scatter   slot = cs[key] + arrival ticket          (order inside a cell arbitrary)
//   k_cs_fixup     each slot ranks its source index inside its cell segment and writes the
//                  (key, src) pair at cs[key] + rank          (order inside a cell = by source)
// The pairs then feed the same k_reorder / density / force chain as the bitonic mode.
#include "fs_device.h"
#include "fs_kernels.h"

namespace fsd {

#define CS_BLOCK 256
#define CS_ITEMS 16
#define CS_TILE (CS_BLOCK * CS_ITEMS)

// Consecutive particles mostly share a cell (the input is the previous step's cell order), so a
// wave would hit the same counter several times.  Runs of equal keys in adjacent lanes are
// combined: the run's first lane issues ONE atomic for the whole run.
struct WaveRun { uint32_t head_lane, offset, length; bool is_head; };
__device__ __forceinline__ WaveRun wave_run(uint32_t key, bool active) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t prev = __shfl_up(key, 1);
    const bool prev_active = __shfl_up(active ? 1 : 0, 1) != 0;
    const bool head = active && (lane == 0 || !prev_active || prev != key);
    const unsigned long long heads = __ballot(head), act = __ballot(active);
    WaveRun r;
    r.is_head = head;
    const unsigned long long upto = heads & (~0ull >> (63u - lane));            // heads at lanes <= mine
    r.head_lane = upto ? 63u - (uint32_t)__clzll(upto) : lane;
    r.offset = lane - r.head_lane;
    const unsigned long long after = (lane == 63u) ? 0ull : ((heads | ~act) & (~0ull << (lane + 1u)));
    // run ends at the next head or the first inactive lane after me
    const uint32_t end = after ? (uint32_t)__ffsll((long long)after) - 1u : 64u;
    r.length = end - r.head_lane;
    return r;
}

__global__ __launch_bounds__(CS_BLOCK) void k_cs_hist(StepParams P, const float2* __restrict__ pos,
                                                      const float2* __restrict__ vel, uint32_t* __restrict__ key_out,
                                                      uint32_t* __restrict__ hist, uint32_t* __restrict__ gap_counter) {
    const uint32_t i = blockIdx.x * CS_BLOCK + threadIdx.x;
    if (i == 0) *gap_counter = 0;
    const bool active = i < P.n;
    uint32_t key = 0;
    if (active) { key = cell_of_point(P, predict_pos(P, pos[i], vel[i])); key_out[i] = key; }
    const uint32_t k = key < P.ncell ? key : P.ncell - 1u;
    const WaveRun r = wave_run(k, active);
    if (r.is_head) atomicAdd(&hist[k], r.length);
}

__global__ __launch_bounds__(CS_BLOCK) void k_cs_scan_reduce(const uint32_t* __restrict__ in, uint32_t count,
                                                             uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t s[CS_BLOCK / 64];
    const uint32_t base = blockIdx.x * CS_TILE;
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < CS_ITEMS; ++k) {
        const uint32_t j = base + k * CS_BLOCK + threadIdx.x;
        if (j < count) sum += in[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if ((threadIdx.x & 63u) == 0) s[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (uint32_t w = 0; w < CS_BLOCK / 64; ++w) t += s[w]; block_sums[blockIdx.x] = t; }
}

__global__ __launch_bounds__(CS_BLOCK) void k_cs_scan_sums(uint32_t* __restrict__ block_sums, uint32_t nblocks) {
    __shared__ uint32_t s[CS_BLOCK];
    const uint32_t chunk = (nblocks + CS_BLOCK - 1) / CS_BLOCK, b0 = threadIdx.x * chunk;
    uint32_t acc = 0;
    for (uint32_t b = b0; b < b0 + chunk && b < nblocks; ++b) acc += block_sums[b];
    s[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t run = 0; for (uint32_t t = 0; t < CS_BLOCK; ++t) { const uint32_t v = s[t]; s[t] = run; run += v; } }
    __syncthreads();
    uint32_t run = s[threadIdx.x];
    for (uint32_t b = b0; b < b0 + chunk && b < nblocks; ++b) { const uint32_t v = block_sums[b]; block_sums[b] = run; run += v; }
}

// Exclusive scan of one tile with its block offset; thread t owns CS_ITEMS consecutive items.
__global__ __launch_bounds__(CS_BLOCK) void k_cs_scan_apply(const uint32_t* __restrict__ in, uint32_t count,
                                                            const uint32_t* __restrict__ block_offs,
                                                            uint32_t* __restrict__ out) {
    __shared__ uint32_t s[CS_BLOCK];
    const uint32_t base = blockIdx.x * CS_TILE + threadIdx.x * CS_ITEMS;
    uint32_t v[CS_ITEMS], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < CS_ITEMS; ++k) { v[k] = (base + k < count) ? in[base + k] : 0u; sum += v[k]; }
    s[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 1; o < CS_BLOCK; o <<= 1) {          // Hillis-Steele inclusive scan of the 256 sums
        const uint32_t add = threadIdx.x >= o ? s[threadIdx.x - o] : 0u;
        __syncthreads();
        s[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = block_offs[blockIdx.x] + s[threadIdx.x] - sum;
#pragma unroll
    for (uint32_t k = 0; k < CS_ITEMS; ++k) { if (base + k < count) out[base + k] = run; run += v[k]; }
}

__global__ __launch_bounds__(CS_BLOCK) void k_cs_scatter(uint32_t n, uint32_t ncell, const uint32_t* __restrict__ key,
                                                         const uint32_t* __restrict__ cs, uint32_t* __restrict__ cursor,
                                                         uint32_t* __restrict__ slot_src) {
    const uint32_t i = blockIdx.x * CS_BLOCK + threadIdx.x;
    const bool active = i < n;
    const uint32_t kk = active ? key[i] : 0u;
    const uint32_t k = kk < ncell ? kk : ncell - 1u;
    const WaveRun r = wave_run(k, active);
    uint32_t base = 0;
    if (r.is_head) base = cs[k] + atomicAdd(&cursor[k], r.length);    // one ticket block per run
    base = __shfl(base, r.head_lane);
    if (active) slot_src[base + r.offset] = i;
}

__global__ __launch_bounds__(CS_BLOCK) void k_cs_fixup(uint32_t n, uint32_t ncell, const uint32_t* __restrict__ key,
                                                       const uint32_t* __restrict__ cs,
                                                       const uint32_t* __restrict__ slot_src, u64* __restrict__ pairs) {
    const uint32_t p = blockIdx.x * CS_BLOCK + threadIdx.x;
    if (p >= n) return;
    const uint32_t src = slot_src[p];
    const uint32_t kk = key[src];
    const uint32_t k = kk < ncell ? kk : ncell - 1u;
    const uint32_t lo = cs[k], hi = cs[k + 1u];
    uint32_t rank = 0;
    for (uint32_t q = lo; q < hi; ++q) rank += slot_src[q] < src ? 1u : 0u;
    pairs[lo + rank] = ((u64)kk << 32) | (u64)src;
}

// ---- slab mode: keys come from the (key<<32 | slot) pairs, DEAD slots are skipped -------------
__global__ __launch_bounds__(CS_BLOCK) void k_cs_hist_pairs(uint32_t cap, uint32_t ncell, const u64* __restrict__ pairs,
                                                            uint32_t* __restrict__ key_out, uint32_t* __restrict__ hist,
                                                            uint32_t* __restrict__ gap_counter) {
    const uint32_t i = blockIdx.x * CS_BLOCK + threadIdx.x;
    if (i == 0) *gap_counter = 0;
    uint32_t key = FS_DEAD_KEY;
    if (i < cap) { key = (uint32_t)(pairs[i] >> 32); key_out[i] = key; }
    const bool active = key != FS_DEAD_KEY;
    const uint32_t k = key < ncell ? key : ncell - 1u;
    const WaveRun r = wave_run(k, active);
    if (r.is_head) atomicAdd(&hist[k], r.length);
}

__global__ __launch_bounds__(CS_BLOCK) void k_cs_scatter_live(uint32_t cap, uint32_t ncell, const uint32_t* __restrict__ key,
                                                              const uint32_t* __restrict__ cs, uint32_t* __restrict__ cursor,
                                                              uint32_t* __restrict__ slot_src, uint32_t* __restrict__ n_live_out) {
    const uint32_t i = blockIdx.x * CS_BLOCK + threadIdx.x;
    if (i == 0) *n_live_out = cs[ncell];                 // total of the histogram = live particles
    const uint32_t kk = i < cap ? key[i] : FS_DEAD_KEY;
    const bool active = kk != FS_DEAD_KEY;
    const uint32_t k = kk < ncell ? kk : ncell - 1u;
    const WaveRun r = wave_run(k, active);
    uint32_t base = 0;
    if (r.is_head) base = cs[k] + atomicAdd(&cursor[k], r.length);
    base = __shfl(base, r.head_lane);
    if (active) slot_src[base + r.offset] = i;
}

__global__ __launch_bounds__(CS_BLOCK) void k_cs_fixup_live(uint32_t cap, uint32_t ncell, const uint32_t* __restrict__ key,
                                                            const uint32_t* __restrict__ cs,
                                                            const uint32_t* __restrict__ slot_src, u64* __restrict__ pairs) {
    const uint32_t p = blockIdx.x * CS_BLOCK + threadIdx.x;
    if (p >= cap) return;
    const uint32_t n_live = cs[ncell];
    if (p >= n_live) { pairs[p] = ((u64)FS_DEAD_KEY << 32) | (u64)p; return; }
    const uint32_t src = slot_src[p];
    const uint32_t kk = key[src];
    const uint32_t k = kk < ncell ? kk : ncell - 1u;
    const uint32_t lo = cs[k], hi = cs[k + 1u];
    uint32_t rank = 0;
    for (uint32_t q = lo; q < hi; ++q) rank += slot_src[q] < src ? 1u : 0u;
    pairs[lo + rank] = ((u64)kk << 32) | (u64)src;
}

void launch_counting_sort_pairs(hipStream_t st, uint32_t cap, uint32_t ncell, u64* pairs, uint32_t* cs, uint32_t* scratch,
                                uint32_t* gap_counter, uint32_t* n_live_out) {
    const uint32_t count = ncell + 1u;
    uint32_t* hist = scratch;
    uint32_t* cursor = hist + count;
    uint32_t* key = cursor + ncell;
    uint32_t* slot_src = key + cap;
    uint32_t* sums = slot_src + cap;
    const uint32_t nblocks = (count + CS_TILE - 1) / CS_TILE;
    const dim3 grid((cap + CS_BLOCK - 1) / CS_BLOCK), block(CS_BLOCK);
    (void)hipMemsetAsync(hist, 0, ((size_t)count + ncell) * sizeof(uint32_t), st);
    hipLaunchKernelGGL(k_cs_hist_pairs, grid, block, 0, st, cap, ncell, pairs, key, hist, gap_counter);
    hipLaunchKernelGGL(k_cs_scan_reduce, dim3(nblocks), block, 0, st, hist, count, sums);
    hipLaunchKernelGGL(k_cs_scan_sums, dim3(1), block, 0, st, sums, nblocks);
    hipLaunchKernelGGL(k_cs_scan_apply, dim3(nblocks), block, 0, st, hist, count, sums, cs);
    hipLaunchKernelGGL(k_cs_scatter_live, grid, block, 0, st, cap, ncell, key, cs, cursor, slot_src, n_live_out);
    hipLaunchKernelGGL(k_cs_fixup_live, grid, block, 0, st, cap, ncell, key, cs, slot_src, pairs);
}

size_t counting_sort_scratch_words(uint32_t n, uint32_t ncell) {
    const size_t nblocks = ((size_t)ncell + 1 + CS_TILE - 1) / CS_TILE;
    // hist (ncell+1) | cursor (ncell) | key (n) | slot_src (n) | block sums
    return ((size_t)ncell + 1) + ncell + (size_t)n * 2 + nblocks + 16;
}

void launch_counting_sort(hipStream_t st, const StepParams& P, const float2* pos, const float2* vel, u64* pairs,
                          uint32_t* cs, uint32_t* scratch, uint32_t* gap_counter) {
    const uint32_t n = P.n, ncell = P.ncell, count = ncell + 1u;
    uint32_t* hist = scratch;
    uint32_t* cursor = hist + count;
    uint32_t* key = cursor + ncell;
    uint32_t* slot_src = key + n;
    uint32_t* sums = slot_src + n;
    const uint32_t nblocks = (count + CS_TILE - 1) / CS_TILE;
    (void)hipMemsetAsync(hist, 0, ((size_t)count + ncell) * sizeof(uint32_t), st);      // hist + cursor
    hipLaunchKernelGGL(k_cs_hist, dim3((n + CS_BLOCK - 1) / CS_BLOCK), dim3(CS_BLOCK), 0, st, P, pos, vel, key, hist,
                       gap_counter);
    hipLaunchKernelGGL(k_cs_scan_reduce, dim3(nblocks), dim3(CS_BLOCK), 0, st, hist, count, sums);
    hipLaunchKernelGGL(k_cs_scan_sums, dim3(1), dim3(CS_BLOCK), 0, st, sums, nblocks);
    hipLaunchKernelGGL(k_cs_scan_apply, dim3(nblocks), dim3(CS_BLOCK), 0, st, hist, count, sums, cs);
    hipLaunchKernelGGL(k_cs_scatter, dim3((n + CS_BLOCK - 1) / CS_BLOCK), dim3(CS_BLOCK), 0, st, n, ncell, key, cs,
                       cursor, slot_src);
    hipLaunchKernelGGL(k_cs_fixup, dim3((n + CS_BLOCK - 1) / CS_BLOCK), dim3(CS_BLOCK), 0, st, n, ncell, key, cs,
                       slot_src, pairs);
}

}  // namespace fsd

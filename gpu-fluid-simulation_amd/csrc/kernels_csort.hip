// kernels_csort.hip — FS_SORT_COUNTING: O(N) cell counting sort, hand-written.
// NOT the reference's algorithm: it yields the STABLE arrangement (ascending source index
// inside a cell) instead of the bitonic network's tie order, so floats agree with the
// reference path only to summation-order tolerance; keys and cell starts are identical.
// Deterministic (no dependence on atomic arrival order):
// (atomics only hand out tickets; the final order inside a cell is by source index) — kernels below.
// The pairs then feed the same density / force chain as the bitonic mode.
#include <hip/hip_ext.h>

#include "fs_device.h"
#include "fs_kernels.h"
#include "fs_scan.h"

namespace fsd {

#define CS_BLOCK 256
#define CS_ITEMS 16
#define CS_TILE (CS_BLOCK * CS_ITEMS)
#ifndef CS_RANK_MAX
#define CS_RANK_MAX 2048u    // longest cell segment k_cs_fixreorder ranks with its serial loop; longer ones are sorted (cs_sort_segment)
#endif

// ---- pipeline (round 3): 4 launches, no memset, atomics only in the histogram -------------------------------
//   k_cs_hist        predict + key per particle; per-cell histogram with wave-aggregated atomics.  The value the
//                    atomic returns is the particle's ARRIVAL TICKET inside its cell: stored beside the key
//                    (kt[i] = key << 32 | ticket), so the scatter needs no second atomic pass and no cursor array.
//   k_scan_lookback  exclusive scan of the histogram = dense cell-start table `cs`, ONE launch (decoupled
//                    look-back, fs_scan.h); zeroes the histogram behind itself (invariant: all-zero between steps).
//   k_cs_scatter     slot_src[cs[key] + ticket] = i                  (order inside a cell = arrival order)
//   k_cs_fixreorder  each slot ranks its source index inside its cell segment -> final position d = cs[key] + rank
//                    (order inside a cell = by source index: deterministic), and does the WHOLE reorder pass for d
//                    right there: payload gather, predicted position, start_indices, safe-operand bit, (key, src) pair.
// The "safe operand" words (fs_device.h) are written by ballot in k_reorder; here a thread does not own a whole
// wave of sorted positions, so the words are preset to all-ones by k_cs_hist and unsafe particles clear their bit.

__global__ __launch_bounds__(CS_BLOCK) void k_cs_hist(StepParams P, const float2* __restrict__ pos,
                                                      const float2* __restrict__ vel, u64* __restrict__ kt,
                                                      uint32_t* __restrict__ hist, uint32_t* __restrict__ gap_counter,
                                                      unsigned long long* __restrict__ safe) {
    const uint32_t i = blockIdx.x * CS_BLOCK + threadIdx.x;
    if (i == 0) *gap_counter = 0;
    if (i < (P.n + 63u) / 64u) safe[i] = ~0ull;
    const bool active = i < P.n;
    uint32_t key = 0;
    if (active) key = cell_of_point(P, predict_pos(P, pos[i], vel[i]));
    const uint32_t k = key < P.ncell ? key : P.ncell - 1u;
    const WaveRun r = wave_run(k, active);
    uint32_t base = 0;
    if (r.is_head) base = atomicAdd(&hist[k], r.length);            // one ticket block per run
    base = __shfl(base, r.head_lane);
    if (active) kt[i] = ((u64)key << 32) | (u64)(base + r.offset);
}

// Exclusive scan of in[0, count) -> out, one launch.  Tile = SCAN_TILE items per workgroup (1024 threads x 16: the
// look-back's prefix frontier advances ~128 tiles per global-memory round trip, so tiles must be FEW — 641 for the
// 10.5 M cells of the 16 M scene — for the chain to hide under the streaming), thread t owns 16 consecutive items
// (four 16-byte loads).  `in` is zeroed behind the read; `in`, `out` 16-byte aligned.  total_out (may be null)
// receives the grand total.
#define SCAN_BLOCK 1024
#define SCAN_ITEMS 16
#define SCAN_TILE (SCAN_BLOCK * SCAN_ITEMS)
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_lookback(uint32_t* __restrict__ in, uint32_t count,
                                                              uint32_t* __restrict__ out, u64* __restrict__ state,
                                                              uint32_t* __restrict__ ticket, uint32_t epoch,
                                                              uint32_t* __restrict__ total_out) {
    __shared__ uint32_t s_wave[SCAN_BLOCK / 64];
    __shared__ uint32_t s_bid, s_excl;
    if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t bid = s_bid, nblocks = gridDim.x;
    if (bid == nblocks - 1u && threadIdx.x == 0) *ticket = 0u;      // every ticket of this launch has been handed out
    const uint32_t base = bid * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], sum = 0;
    if (base + SCAN_ITEMS <= count) {
#pragma unroll
        for (uint32_t k = 0; k < SCAN_ITEMS; k += 4) {
            const uint4 q = *reinterpret_cast<const uint4*>(in + base + k);
            v[k] = q.x; v[k + 1] = q.y; v[k + 2] = q.z; v[k + 3] = q.w;
        }
#pragma unroll
        for (uint32_t k = 0; k < SCAN_ITEMS; k += 4) *reinterpret_cast<uint4*>(in + base + k) = make_uint4(0u, 0u, 0u, 0u);
    } else {
#pragma unroll
        for (uint32_t k = 0; k < SCAN_ITEMS; ++k) v[k] = (base + k < count) ? in[base + k] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < SCAN_ITEMS; ++k) if (base + k < count) in[base + k] = 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < SCAN_ITEMS; ++k) sum += v[k];
    // inclusive scan of the thread sums: wave scan + the 16 wave totals
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if ((int)lane >= o) inc += t; }
    if (lane == 63u) s_wave[w] = inc;
    __syncthreads();
    uint32_t wave_off = 0, tile_total = 0;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_BLOCK / 64; ++k) { const uint32_t t = s_wave[k]; if (k < w) wave_off += t; tile_total += t; }
    const u64 tag = (u64)(epoch & 0x3FFFFFFFu);
    if (w == 0) {
        if (bid == 0) {
            if (lane == 0) { lb_store(state, (LB_FLAG_PREFIX << 62) | (tag << 32) | (u64)tile_total); s_excl = 0u; }
        } else {
            if (lane == 0) lb_store(state + bid, (LB_FLAG_AGG << 62) | (tag << 32) | (u64)tile_total);
            const u64 ex = lookback_exclusive<32>(state, bid, tag, [](u64 a, u64 b) { return (a + b) & 0xFFFFFFFFull; });
            if (lane == 0) {
                lb_store(state + bid, (LB_FLAG_PREFIX << 62) | (tag << 32) | ((ex + tile_total) & 0xFFFFFFFFull));
                s_excl = (uint32_t)ex;
            }
        }
    }
    __syncthreads();
    uint32_t run = s_excl + wave_off + inc - sum;
    if (base + SCAN_ITEMS <= count) {
#pragma unroll
        for (uint32_t k = 0; k < SCAN_ITEMS; k += 4) {
            uint4 q;
            q.x = run; run += v[k]; q.y = run; run += v[k + 1]; q.z = run; run += v[k + 2]; q.w = run; run += v[k + 3];
            *reinterpret_cast<uint4*>(out + base + k) = q;
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < SCAN_ITEMS; ++k) { if (base + k < count) out[base + k] = run; run += v[k]; }
    }
    if (total_out && base < count && base + SCAN_ITEMS >= count) *total_out = run;   // the thread that owns item count - 1
}

// Four slots per thread, 256 apart: the three dependent loads of a slot (kt -> cs -> store address) of all four are in flight
// together (the kernel is latency-bound at a slab rank's size).
#define SC_ITEMS 4
// n_dev (may be null): a device-side slot count below `n` (the boundary strips of an overlapped slab step use a prefix of their
// slot array whose length only the device knows); the grid still covers `n`.
__global__ __launch_bounds__(CS_BLOCK) void k_cs_scatter(uint32_t n, uint32_t ncell, const u64* __restrict__ kt,
                                                         const uint32_t* __restrict__ cs, uint32_t* __restrict__ slot_src,
                                                         const uint32_t* __restrict__ n_dev,
                                                         unsigned long long* __restrict__ safe_preset) {
    if (n_dev) { const uint32_t m = *n_dev; n = m < n ? m : n; }
    if (blockIdx.x * (CS_BLOCK * SC_ITEMS) >= n) return;
    // slab handles: the "safe operand" words of this block's slots start as all-ones, k_cs_fixreorder (next in the stream) clears
    // the unsafe particles' bits.  (Done here rather than in k_slab_pack: in an edge-first step the pack of tick t + 1 may run
    // beside the edge columns' density launch of tick t, which still reads the words of tick t.)
    if (safe_preset && threadIdx.x < CS_BLOCK * SC_ITEMS / 64u) safe_preset[blockIdx.x * (CS_BLOCK * SC_ITEMS / 64u) + threadIdx.x] = ~0ull;
    const uint32_t i0 = blockIdx.x * (CS_BLOCK * SC_ITEMS) + threadIdx.x;
    u64 e[SC_ITEMS];
#pragma unroll
    for (int it = 0; it < SC_ITEMS; ++it) {
        const uint32_t i = i0 + (uint32_t)it * CS_BLOCK;
        e[it] = i < n ? kt[i] : ((u64)FS_DEAD_KEY << 32);
    }
    uint32_t start[SC_ITEMS];
#pragma unroll
    for (int it = 0; it < SC_ITEMS; ++it) {
        const uint32_t kk = (uint32_t)(e[it] >> 32);
        start[it] = kk == FS_DEAD_KEY ? 0u : cs[kk < ncell ? kk : ncell - 1u];     // DEAD: slab mode, an empty slot
    }
#pragma unroll
    for (int it = 0; it < SC_ITEMS; ++it)
        if ((uint32_t)(e[it] >> 32) != FS_DEAD_KEY) slot_src[start[it] + (uint32_t)e[it]] = i0 + (uint32_t)it * CS_BLOCK;
}

// ---- cells too large for the serial rank loop (round 4) ----------------------------------------------------------------
// k_cs_fixreorder ranks a slot inside its cell segment of slot_src with a serial loop: O(m^2) for a cell of m particles — fine
// for the tens a cell holds, a multi-second kernel for 1e5 coincident particles.  A segment longer than CS_RANK_MAX
// is therefore SORTED in place, by the workgroup that holds the cell's first slot (at most one such cell starts in a
// workgroup: the next cell start is more than CS_RANK_MAX slots away): the reference's own ascending flip / disperse network
// on the 28-bit source indices (distinct values; positions past the segment act as +inf, which an ascending-only network
// never moves), all 256 threads, global memory, a barrier per pass — O(m log^2 m), ~1 ms for 1e5.  It then sets bit 31 of
// the segment's first entry (source indices are < 2^28); every thread of that cell — in this workgroup or in a later one —
// waits for the bit and takes slot p's particle as it now stands there: rank = p - lo, the stable order.  The wait cannot
// deadlock: the sorting workgroup has the lowest index of all that hold slots of the cell, so it was dispatched no later,
// and its own duty precedes any wait of its threads.  (Bounded all the same, like the sort's stand-by barrier.)
#define CS_SORTED_FLAG 0x80000000u
#define CS_SRC_MASK 0x0FFFFFFFu
__device__ __forceinline__ void cs_cmpx(uint32_t* v, uint32_t m, uint32_t a, uint32_t b) {      // a < b
    if (b < m) {
        const uint32_t x = v[a], y = v[b];
        if (x > y) { v[a] = y; v[b] = x; }
    }
}
__device__ __forceinline__ void cs_sort_segment(uint32_t* v, uint32_t m) {      // all threads of the workgroup
    uint32_t p2 = 1;
    while (p2 < m) p2 <<= 1;
    for (uint32_t h = 1; h < p2; h <<= 1) {
        for (uint32_t t = threadIdx.x; t < (p2 >> 1); t += CS_BLOCK) {           // flip: i <-> mirror inside blocks of 2h
            const uint32_t q = (t / h) * 2u * h, r = t % h;
            cs_cmpx(v, m, q + r, q + 2u * h - 1u - r);
        }
        __syncthreads();
        for (uint32_t hh = h >> 1; hh >= 1u; hh >>= 1) {                          // disperse: distance hh
            for (uint32_t t = threadIdx.x; t < (p2 >> 1); t += CS_BLOCK) {
                const uint32_t q = (t / hh) * 2u * hh, r = t % hh;
                cs_cmpx(v, m, q + r, q + r + hh);
            }
            __syncthreads();
        }
    }
}

// Fused rank fix-up + reorder pass (k_reorder<false> / k_slab_reorder<false> of round 2).  SLAB: `n` = slot capacity,
// the live count is cs[ncell]; slots past it become DEAD pairs.
template <bool SLAB>
__global__ __launch_bounds__(CS_BLOCK) void k_cs_fixreorder(StepParams P, uint32_t n, const u64* __restrict__ kt,
                                                            const uint32_t* __restrict__ cs,
                                                            uint32_t* slot_src, u64* __restrict__ pairs,
                                                            const float2* __restrict__ pos_in, const float2* __restrict__ vel_in,
                                                            float2* __restrict__ pos_s, float2* __restrict__ vel_s,
                                                            float2* __restrict__ pred_s, uint32_t* __restrict__ key_s,
                                                            unsigned char* __restrict__ owned, uint32_t* __restrict__ start_ref,
                                                            unsigned long long* __restrict__ safe, uint32_t* __restrict__ force_defer,
                                                            uint32_t* __restrict__ force_work_count,
                                                            const uint32_t* __restrict__ n_dev) {
    const uint32_t p = blockIdx.x * CS_BLOCK + threadIdx.x;
    if (n_dev) { const uint32_t m = *n_dev; n = m < n ? m : n; }     // device-side slot count (see k_cs_scatter)
    __shared__ uint32_t s_seg[2];
    if (threadIdx.x == 0) {                      // the force pass's worklists of this step (same block size and count)
        force_defer[2u * blockIdx.x] = 0u;
        force_defer[2u * blockIdx.x + 1u] = 0u;
        if (blockIdx.x == 0) { force_work_count[0] = 0u; force_work_count[1] = 0u; }
        s_seg[1] = 0u;
    }
    bool act = p < n;
    if (SLAB && act) {
        const uint32_t n_live = cs[P.ncell];
        if (p >= n_live) { pairs[p] = ((u64)FS_DEAD_KEY << 32) | (u64)p; owned[p] = 0; act = false; }
    }
    uint32_t src = 0, key = FS_DEAD_KEY, lo = 0, hi = 0;
    if (act) {
        // (while another workgroup sorts the segment, an entry is always SOME member of it: enough to find the cell)
        src = __hip_atomic_load(&slot_src[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & CS_SRC_MASK;
        key = (uint32_t)(kt[src] >> 32);
        const uint32_t k = key < P.ncell ? key : P.ncell - 1u;
        lo = cs[k]; hi = cs[k + 1u];
    }
    const bool big = act && hi - lo > CS_RANK_MAX;
    __syncthreads();                             // s_seg cleared
    if (big && p == lo) { s_seg[0] = lo; s_seg[1] = hi; }
    __syncthreads();
    if (s_seg[1] != 0u) {                        // block-uniform: this workgroup holds the first slot of a large cell
        cs_sort_segment(slot_src + s_seg[0], s_seg[1] - s_seg[0]);
        if (threadIdx.x == 0) {
            __threadfence();
            atomicOr(&slot_src[s_seg[0]], CS_SORTED_FLAG);
        }
    }
    uint32_t rank = 0;
    if (big) {
        uint32_t spins = 0;
        while (!(__hip_atomic_load(&slot_src[lo], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) & CS_SORTED_FLAG)) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 26)) break;     // (cannot happen: see above; the cell would keep a valid but unordered arrangement)
        }
        src = __hip_atomic_load(&slot_src[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & CS_SRC_MASK;
        rank = p - lo;
    } else if (act) {
        for (uint32_t q = lo; q < hi; ++q) rank += (slot_src[q] & CS_SRC_MASK) < src ? 1u : 0u;
    }
    if (!act) return;
    const uint32_t d = lo + rank;
    pairs[d] = ((u64)key << 32) | (u64)src;
    const float2 ps = pos_in[src];
    const float2 v = vel_in[src];
    pos_s[d] = ps;
    vel_s[d] = v;
    const float2 pd = predict_pos(P, ps, v);     // same expression as the key generation -> same bits
    pred_s[d] = pd;
    if (key_s) key_s[d] = key;
    if (!kin_safe(pd, v)) atomicAnd(&safe[d >> 6], ~(1ull << (d & 63u)));   // rare; words preset to all-ones
    if (SLAB) {
        uint32_t cxl, cy;
        key_to_local(P, key, &cxl, &cy);
        const int32_t cxg = (int32_t)cxl + P.col_origin;
        owned[d] = (cxg >= (int32_t)P.own_lo && cxg < (int32_t)P.own_hi) ? 1 : 0;
    }
    if (rank == 0u && key < P.ncell) {           // first particle of its cell: compute.wgsl:49-55 (index 0 skipped)
        if (d != 0u || !P.ref_quirks) start_ref[key] = d;
    }
}

static inline size_t even(size_t w) { return (w + 1u) & ~(size_t)1u; }
struct CsLayout { uint32_t *hist, *slot_src, *ticket; u64 *kt, *state; uint32_t scan_blocks; };
static CsLayout cs_layout(uint32_t* scratch, uint32_t n, uint32_t ncell) {
    // hist (ncell + 1) | kt (n x u64) | slot_src (n) | scan state (u64 per tile) | tickets
    CsLayout L;
    size_t o = 0;
    L.hist = scratch; o = even((size_t)ncell + 1u);
    L.kt = (u64*)(scratch + o); o += 2 * (size_t)n;
    L.slot_src = scratch + o; o = even(o + n);
    L.scan_blocks = (uint32_t)(((size_t)ncell + 1u + SCAN_TILE - 1) / SCAN_TILE);
    L.state = (u64*)(scratch + o); o += 2 * (size_t)L.scan_blocks;
    L.ticket = scratch + o;
    return L;
}
// `ncell_max`: the largest table the handle will ever scan (slab handles move their window).
size_t counting_sort_scratch_words(uint32_t n, uint32_t ncell_max) {
    const size_t tiles = ((size_t)ncell_max + 1u + SCAN_TILE - 1) / SCAN_TILE;
    return even((size_t)ncell_max + 1u) + 2 * (size_t)n + even(n) + 2 * tiles + 16;
}
// which words of the scratch must be zero when a handle is created: the histogram and the ticket counters (all of it is simplest)
u64* counting_sort_kt(uint32_t* scratch, uint32_t n, uint32_t ncell) { return cs_layout(scratch, n, ncell).kt; }
uint32_t* counting_sort_hist(uint32_t* scratch) { return scratch; }

// Slab mode: kt / hist were filled by k_slab_pack + k_slab_unpack (kernels_slab.hip).
void launch_counting_sort_pairs(hipStream_t st, uint32_t cap, uint32_t ncell, uint32_t ncell_alloc, uint32_t* cs, uint32_t* scratch,
                                uint32_t* n_live_out, uint32_t epoch, const uint32_t* n_dev, unsigned long long* safe_preset) {
    const CsLayout L = cs_layout(scratch, cap, ncell_alloc);
    const uint32_t count = ncell + 1u, tiles = (count + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_lookback, dim3(tiles), dim3(SCAN_BLOCK), 0, st, L.hist, count, cs, L.state, L.ticket, epoch, n_live_out);
    hipLaunchKernelGGL(k_cs_scatter, dim3((cap + CS_BLOCK * SC_ITEMS - 1) / (CS_BLOCK * SC_ITEMS)), dim3(CS_BLOCK), 0, st, cap, ncell, L.kt, cs, L.slot_src, n_dev, safe_preset);
}
uint32_t* counting_sort_slot_src(uint32_t* scratch, uint32_t n, uint32_t ncell_alloc) { return cs_layout(scratch, n, ncell_alloc).slot_src; }
void launch_counting_reorder_slab(hipStream_t st, const StepParams& P, uint32_t cap, uint32_t ncell_alloc, uint32_t* scratch, u64* pairs,
                                  const uint32_t* cs, const float2* pos_in, const float2* vel_in, float2* pos_s, float2* vel_s,
                                  float2* pred_s, uint32_t* key_s, unsigned char* owned, uint32_t* start_ref,
                                  unsigned long long* safe, uint32_t* force_defer, uint32_t* force_work_count,
                                  const uint32_t* n_dev, hipEvent_t done) {
    const CsLayout L = cs_layout(scratch, cap, ncell_alloc);
    if (done) {   // the kernel's own completion signal is the event: no separate barrier packet in the stream (hipEventRecord costs
                  // the following kernel ~6 us of idle queue)
        hipExtLaunchKernelGGL(k_cs_fixreorder<true>, dim3((cap + CS_BLOCK - 1) / CS_BLOCK), dim3(CS_BLOCK), 0, st, nullptr, done, 0, P, cap,
                              L.kt, cs, L.slot_src, pairs, pos_in, vel_in, pos_s, vel_s, pred_s, key_s, owned, start_ref, safe, force_defer,
                              force_work_count, n_dev);
        return;
    }
    hipLaunchKernelGGL(k_cs_fixreorder<true>, dim3((cap + CS_BLOCK - 1) / CS_BLOCK), dim3(CS_BLOCK), 0, st, P, cap, L.kt, cs,
                       L.slot_src, pairs, pos_in, vel_in, pos_s, vel_s, pred_s, key_s, owned, start_ref, safe, force_defer, force_work_count,
                       n_dev);
}

void launch_counting_sort(hipStream_t st, const StepParams& P, const float2* pos, const float2* vel, uint32_t* cs,
                          uint32_t* scratch, uint32_t* gap_counter, unsigned long long* safe, uint32_t epoch) {
    const uint32_t n = P.n, ncell = P.ncell, count = ncell + 1u;
    const CsLayout L = cs_layout(scratch, n, ncell);
    const dim3 grid((n + CS_BLOCK - 1) / CS_BLOCK), block(CS_BLOCK);
    hipLaunchKernelGGL(k_cs_hist, grid, block, 0, st, P, pos, vel, L.kt, L.hist, gap_counter, safe);
    hipLaunchKernelGGL(k_scan_lookback, dim3((count + SCAN_TILE - 1) / SCAN_TILE), dim3(SCAN_BLOCK), 0, st, L.hist, count, cs, L.state, L.ticket,
                       epoch, (uint32_t*)nullptr);
    hipLaunchKernelGGL(k_cs_scatter, dim3((n + CS_BLOCK * SC_ITEMS - 1) / (CS_BLOCK * SC_ITEMS)), block, 0, st, n, ncell, L.kt, cs, L.slot_src, (const uint32_t*)nullptr, (unsigned long long*)nullptr);
}
void launch_counting_reorder(hipStream_t st, const StepParams& P, uint32_t* scratch, u64* pairs, const uint32_t* cs,
                             const float2* pos_in, const float2* vel_in, float2* pos_s, float2* vel_s, float2* pred_s,
                             uint32_t* key_s, uint32_t* start_ref, unsigned long long* safe, uint32_t* force_defer,
                             uint32_t* force_work_count) {
    const CsLayout L = cs_layout(scratch, P.n, P.ncell);
    hipLaunchKernelGGL(k_cs_fixreorder<false>, dim3((P.n + CS_BLOCK - 1) / CS_BLOCK), dim3(CS_BLOCK), 0, st, P, P.n, L.kt, cs,
                       L.slot_src, pairs, pos_in, vel_in, pos_s, vel_s, pred_s, key_s, (unsigned char*)nullptr, start_ref, safe,
                       force_defer, force_work_count, (const uint32_t*)nullptr);
}

}  // namespace fsd

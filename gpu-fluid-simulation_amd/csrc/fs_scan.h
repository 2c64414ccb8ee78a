// fs_scan.h — device helpers shared by the counting sort (kernels_csort.hip) and the slab pack (kernels_slab.hip):
// wave-level run aggregation for histogram atomics, and the single-pass "decoupled look-back" prefix sum over
// workgroups (one launch instead of reduce / scan-of-sums / apply; Merrill & Garland's chained scan).
//
// Look-back protocol.  Workgroups take their logical index from an atomic ticket, so a workgroup only ever waits
// for workgroups that are already running (no dependence on the hardware's dispatch order).  Each publishes ONE
// 64-bit word: [63:62] flag (1 = "aggregate of my tile", 2 = "inclusive prefix up to and including my tile"),
// an epoch, and the value(s).  The epoch is the launch's sequence number: words left over from earlier launches
// never match, so the state array is never cleared.  A waiting workgroup's first wave reads 64 predecessors per
// trip (one per lane) and sums aggregates back to the nearest published prefix.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fsd {

typedef unsigned long long u64;

// Runs of equal keys in adjacent lanes are combined: the run's first lane issues ONE atomic for the whole run
// (the input is the previous step's cell order, so consecutive particles mostly share a cell).
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

__device__ __forceinline__ u64 lb_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lb_store(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

#define LB_FLAG_AGG 1ull
#define LB_FLAG_PREFIX 2ull

// Generic look-back over words of the form (flag << 62) | (tag << TAG_SHIFT) | payload, payload < 2^TAG_SHIFT.
// `Add` combines payloads (plain + for one counter; two saturating fields for the slab pack).
// Called by the FIRST WAVE of workgroup `bid` (all 64 lanes); returns the exclusive prefix of tiles [0, bid) in every lane.
template <int TAG_SHIFT, class Add>
__device__ __forceinline__ u64 lookback_exclusive(const u64* __restrict__ state, uint32_t bid, u64 tag, Add add) {
    const uint32_t lane = threadIdx.x & 63u;
    const u64 tag_mask = ((1ull << (62 - TAG_SHIFT)) - 1ull) << TAG_SHIFT;
    const u64 want = (tag << TAG_SHIFT) & tag_mask;
    const u64 pay_mask = (1ull << TAG_SHIFT) - 1ull;
    u64 excl = 0;
    int pos = (int)bid - 1;
    while (pos >= 0) {
        const int j = pos - (int)lane;
        u64 w;
        uint32_t flag;
        for (;;) {                                                     // until lanes up to the first prefix are all ready
            w = j >= 0 ? lb_load(state + j) : ((LB_FLAG_PREFIX << 62) | want);   // before tile 0: prefix 0
            flag = ((w & tag_mask) == want) ? (uint32_t)(w >> 62) : 0u;
            const unsigned long long notready = __ballot(flag == 0u);
            const unsigned long long prefix = __ballot(flag == (uint32_t)LB_FLAG_PREFIX);
            const unsigned long long upto = prefix ? ((prefix & (0ull - prefix)) << 1) - 1ull : ~0ull;   // lanes 0 .. first prefix lane
            if ((notready & upto) == 0ull) {
                // sum payloads of lanes 0 .. first prefix lane (all 64 when none is a prefix)
                u64 v = (((1ull << lane) & upto) != 0ull) ? (w & pay_mask) : 0ull;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v = add(v, (u64)__shfl_xor((unsigned long long)v, o));
                excl = add(excl, v);
                if (prefix) return excl;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        pos -= 64;
    }
    return excl;
}

}  // namespace fsd

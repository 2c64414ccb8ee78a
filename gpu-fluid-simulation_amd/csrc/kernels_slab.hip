// kernels_slab.hip — multi-GPU slab mode (NOT in the reference, which is
// single-device).  A rank owns the global cell columns [own_lo, own_hi) and keeps a
// fixed-capacity local array whose slots are either live particles or DEAD (key
// 0xFFFFFFFF).  Everything a step needs to know about counts lives on the device, so a
// step is enqueued without any host synchronisation:
//
//   slots [0, main)            particles carried over from the last step (owned + old ghosts)
//   slots [main, main+R)       records received from the left neighbour this step
//   slots [main+R, main+2R)    records received from the right neighbour this step
//
//   k_slab_classify  predict + global column; old ghosts and leavers become DEAD; flags the
//                    records each neighbour needs (migrants and the 2-column ghost halo)
//   k_slab_scan / k_slab_scatter   stable (slot-order) compaction into fixed-size messages
//                    [16-B header | R x {pos, vel}] — deterministic, no atomics
//   k_slab_unpack    received records -> slots, key from the recomputed predicted position
//   (bitonic sort over all slots: DEAD keys end up last)
//   k_slab_reorder   as k_reorder, plus live count, owned flags
//   k_density / k_force run unchanged on the local window (ghosts are not advanced)
#include "fs_device.h"
#include "fs_kernels.h"

namespace fsd {

#define SL_BLOCK 256

__device__ __forceinline__ uint32_t slab_key(const StepParams& P, float2 pred, uint32_t* cx_global) {
    uint32_t cx, cy;
    xy_of_point(P, pred, &cx, &cy);
    *cx_global = cx;
    const int32_t lo = (int32_t)P.own_lo - 2, hi = (int32_t)P.own_hi + 2;   // owned + 2 ghost columns per side
    if ((int32_t)cx < lo || (int32_t)cx >= hi || cy >= P.grid_h) return FS_DEAD_KEY;
    return cy * P.grid_w + (uint32_t)((int32_t)cx - P.col_origin);
}

__global__ __launch_bounds__(SL_BLOCK) void k_slab_classify(StepParams P, uint32_t main_slots, int has_left,
                                                            int has_right, const float2* __restrict__ pos,
                                                            const float2* __restrict__ vel,
                                                            const unsigned char* __restrict__ owned,
                                                            u64* __restrict__ pairs, unsigned char* __restrict__ flags,
                                                            uint2* __restrict__ blockcnt,
                                                            uint32_t* __restrict__ counters,
                                                            uint32_t* __restrict__ gap_counter) {
    __shared__ uint32_t s_cnt[2 * (SL_BLOCK / 64)];
    const uint32_t i = blockIdx.x * SL_BLOCK + threadIdx.x;
    if (i == 0) *gap_counter = 0;          // cell-table worklist of this step (k_slab_reorder / counting sort)
    const uint32_t n_prev = *P.n_live;
    unsigned char f = 0;
    if (i < main_slots) {
        uint32_t key = FS_DEAD_KEY;
        if (i < n_prev && owned[i]) {
            const float2 pr = predict_pos(P, pos[i], vel[i]);
            uint32_t cxg;
            key = slab_key(P, pr, &cxg);
            if (has_left && cxg < P.own_lo + 2u) f |= 1;
            if (has_right && cxg + 2u >= P.own_hi) f |= 2;
            // a leaver must land inside the neighbour's slab and not in ITS far halo: checked by the receiver
            if (!has_left && cxg < P.own_lo) atomicAdd(&counters[2], 1u);    // left the domain partition
            if (!has_right && cxg >= P.own_hi) atomicAdd(&counters[2], 1u);
        }
        pairs[i] = ((u64)key << 32) | (u64)i;
        flags[i] = f;
    } else if (i < n_prev && owned[i]) {
        // Slot capacity exceeded: the last step left more live records than main slots, and this owned
        // particle sits where the incoming messages will be unpacked.  It cannot be carried over —
        // count it (fs_slab_counters.overflow must stay 0; the driver raises on it).
        atomicAdd(&counters[3], 1u);
    }
    const unsigned long long mL = __ballot(f & 1), mR = __ballot(f & 2);
    const uint32_t w = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) { s_cnt[2 * w] = __popcll(mL); s_cnt[2 * w + 1] = __popcll(mR); }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, b = 0;
        for (uint32_t k = 0; k < SL_BLOCK / 64; ++k) { a += s_cnt[2 * k]; b += s_cnt[2 * k + 1]; }
        blockcnt[blockIdx.x] = make_uint2(a, b);
    }
}

struct SlabHeader { uint32_t count, overflow, pad0, pad1; };

// Exclusive scan of the per-block counts.  One workgroup of 1024 threads; every thread owns a
// contiguous chunk and reads it with independent 16-byte loads (two counts each), so the ~10^4-10^5
// counts of a slab cost a few microseconds instead of a serial walk.
#define SCAN_THREADS 1024
__global__ __launch_bounds__(SCAN_THREADS) void k_slab_scan(const uint2* __restrict__ blockcnt, uint32_t nblocks,
                                                            uint2* __restrict__ blockoff, SlabHeader* hdr_left,
                                                            SlabHeader* hdr_right, uint32_t R,
                                                            uint32_t* __restrict__ counters) {
    __shared__ uint2 s_sum[SCAN_THREADS];
    __shared__ uint2 s_wave[SCAN_THREADS / 64];
    uint32_t chunk = (nblocks + SCAN_THREADS - 1) / SCAN_THREADS;
    chunk = (chunk + 1u) & ~1u;                                   // even, so chunks start 16-byte aligned
    const uint32_t b0 = threadIdx.x * chunk;
    const uint32_t b1 = b0 + chunk < nblocks ? b0 + chunk : nblocks;
    uint2 acc = make_uint2(0, 0);
    uint32_t b = b0;
    for (; b + 8u <= b1; b += 8u) {                               // 4 independent uint4 loads in flight
        const uint4 v0 = *reinterpret_cast<const uint4*>(blockcnt + b), v1 = *reinterpret_cast<const uint4*>(blockcnt + b + 2);
        const uint4 v2 = *reinterpret_cast<const uint4*>(blockcnt + b + 4), v3 = *reinterpret_cast<const uint4*>(blockcnt + b + 6);
        acc.x += v0.x + v0.z + v1.x + v1.z + v2.x + v2.z + v3.x + v3.z;
        acc.y += v0.y + v0.w + v1.y + v1.w + v2.y + v2.w + v3.y + v3.w;
    }
    for (; b < b1; ++b) { acc.x += blockcnt[b].x; acc.y += blockcnt[b].y; }
    // exclusive scan of the 1024 chunk sums: wave scan + scan of the 16 wave totals
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint2 inc = acc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t tx = __shfl_up(inc.x, o), ty = __shfl_up(inc.y, o);
        if ((int)lane >= o) { inc.x += tx; inc.y += ty; }
    }
    if (lane == 63u) s_wave[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint2 run = make_uint2(0, 0);
        for (uint32_t k = 0; k < SCAN_THREADS / 64; ++k) { const uint2 v = s_wave[k]; s_wave[k] = run; run.x += v.x; run.y += v.y; }
        if (hdr_left) { hdr_left->count = run.x < R ? run.x : R; hdr_left->overflow = run.x > R; }
        if (hdr_right) { hdr_right->count = run.y < R ? run.y : R; hdr_right->overflow = run.y > R; }
        if (run.x > R || run.y > R) atomicAdd(&counters[3], 1u);
    }
    __syncthreads();
    uint2 run = make_uint2(s_wave[w].x + inc.x - acc.x, s_wave[w].y + inc.y - acc.y);
    (void)s_sum;
    for (b = b0; b < b1; ++b) {
        const uint2 v = blockcnt[b];
        blockoff[b] = run;
        run.x += v.x; run.y += v.y;
    }
}

__global__ __launch_bounds__(SL_BLOCK) void k_slab_scatter(uint32_t main_slots, const float2* __restrict__ pos,
                                                           const float2* __restrict__ vel,
                                                           const unsigned char* __restrict__ flags,
                                                           const uint2* __restrict__ blockoff,
                                                           float4* __restrict__ rec_left,
                                                           float4* __restrict__ rec_right, uint32_t R) {
    __shared__ uint32_t s_w[2 * (SL_BLOCK / 64)];
    const uint32_t i = blockIdx.x * SL_BLOCK + threadIdx.x;
    const unsigned char f = i < main_slots ? flags[i] : 0;
    const unsigned long long mL = __ballot(f & 1), mR = __ballot(f & 2);
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (lane == 0) { s_w[2 * w] = __popcll(mL); s_w[2 * w + 1] = __popcll(mR); }
    __syncthreads();
    uint32_t wl = 0, wr = 0;
    for (uint32_t k = 0; k < w; ++k) { wl += s_w[2 * k]; wr += s_w[2 * k + 1]; }
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const uint2 off = blockoff[blockIdx.x];
    if (f) {
        const float2 p = pos[i], v = vel[i];
        const float4 rec = make_float4(p.x, p.y, v.x, v.y);
        if (f & 1) { const uint32_t d = off.x + wl + __popcll(mL & below); if (d < R && rec_left) rec_left[d] = rec; }
        if (f & 2) { const uint32_t d = off.y + wr + __popcll(mR & below); if (d < R && rec_right) rec_right[d] = rec; }
    }
}

__global__ __launch_bounds__(SL_BLOCK) void k_slab_unpack(StepParams P, uint32_t main_slots, uint32_t R,
                                                          const SlabHeader* __restrict__ hdr_left,
                                                          const float4* __restrict__ rec_left,
                                                          const SlabHeader* __restrict__ hdr_right,
                                                          const float4* __restrict__ rec_right,
                                                          float2* __restrict__ pos, float2* __restrict__ vel,
                                                          u64* __restrict__ pairs, uint32_t* __restrict__ counters) {
    const uint32_t j = blockIdx.x * SL_BLOCK + threadIdx.x;
    if (j >= 2u * R) return;
    const bool right = j >= R;
    const uint32_t jj = right ? j - R : j;
    const SlabHeader* hdr = right ? hdr_right : hdr_left;
    const float4* rec = right ? rec_right : rec_left;
    uint32_t cnt = 0;
    if (hdr) { cnt = hdr->count < R ? hdr->count : R; if (jj == 0 && hdr->overflow) atomicAdd(&counters[3], 1u); }
    const uint32_t slot = main_slots + j;
    uint32_t key = FS_DEAD_KEY;
    if (jj < cnt) {
        const float4 r = rec[jj];
        const float2 p = make_float2(r.x, r.y), v = make_float2(r.z, r.w);
        pos[slot] = p;
        vel[slot] = v;
        uint32_t cxg;
        key = slab_key(P, predict_pos(P, p, v), &cxg);
        if (key == FS_DEAD_KEY) atomicAdd(&counters[2], 1u);           // travelled farther than slab + halo
        // a migrant that lands in my FAR halo zone would have been needed by my other neighbour too
        if (!right && cxg + 2u >= P.own_hi && cxg < P.own_hi) atomicAdd(&counters[4], 1u);
        if (right && cxg < P.own_lo + 2u && cxg >= P.own_lo) atomicAdd(&counters[4], 1u);
    }
    pairs[slot] = ((u64)key << 32) | (u64)slot;
}

// k_reorder for slab mode: DEAD slots are skipped, the live count and the owned flags are
// produced here.  `cap` = number of slots sorted.
template <bool FILL>
__global__ __launch_bounds__(SL_BLOCK) void k_slab_reorder(StepParams P, uint32_t cap, const u64* __restrict__ pairs,
                                                           const float2* __restrict__ pos_in,
                                                           const float2* __restrict__ vel_in,
                                                           float2* __restrict__ pos_s, float2* __restrict__ vel_s,
                                                           float2* __restrict__ pred_s, uint32_t* __restrict__ key_s,
                                                           unsigned char* __restrict__ owned,
                                                           uint32_t* __restrict__ cs, uint32_t* __restrict__ start_ref,
                                                           GapEntry* __restrict__ work, uint32_t* __restrict__ counter,
                                                           uint32_t work_cap, uint32_t* __restrict__ n_live_out,
                                                           unsigned long long* __restrict__ safe, uint32_t* __restrict__ force_defer,
                                                           uint32_t* __restrict__ force_work_count) {
    const uint32_t i = blockIdx.x * SL_BLOCK + threadIdx.x;
    if (threadIdx.x == 0) {                      // the force pass's worklists of this step (same block size)
        force_defer[2u * blockIdx.x] = 0u;
        force_defer[2u * blockIdx.x + 1u] = 0u;
        if (blockIdx.x == 0) { force_work_count[0] = 0u; force_work_count[1] = 0u; }
    }
    if (i >= cap) return;
    const u64 pr = pairs[i];
    const uint32_t key = (uint32_t)(pr >> 32);
    const uint32_t prev = i ? (uint32_t)(pairs[i - 1] >> 32) : 0u;
    if (key == FS_DEAD_KEY) {
        if (FILL) {   // bitonic path: the live count and the table come from here
            if (i == 0) { *n_live_out = 0; fill_cells(cs, 0u, P.ncell + 1u, 0u, work, counter, work_cap); }
            else if (prev != FS_DEAD_KEY) *n_live_out = i;
        }
        owned[i] = 0;
        return;
    }
    const uint32_t src = (uint32_t)pr;
    const float2 p = pos_in[src];
    const float2 v = vel_in[src];
    pos_s[i] = p;
    vel_s[i] = v;
    const float2 pd = predict_pos(P, p, v);
    pred_s[i] = pd;
    key_s[i] = key;
    {   // fs_device.h "safe operand" classification (finished by k_density): one 64-bit word per wave
        const unsigned long long sb = __builtin_amdgcn_ballot_w64(kin_safe(pd, v));   // lanes that returned above: 0
        if ((threadIdx.x & 63u) == 0u) safe[i >> 6] = sb;
    }
    const uint32_t cy = key / P.grid_w;
    const int32_t cxg = (int32_t)(key - cy * P.grid_w) + P.col_origin;
    owned[i] = (cxg >= (int32_t)P.own_lo && cxg < (int32_t)P.own_hi) ? 1 : 0;

    const uint32_t kc = key < P.ncell ? key : P.ncell;
    if (i == 0) {
        if (key < P.ncell) start_ref[key] = 0;
        if (FILL) fill_cells(cs, 0u, kc + 1u, 0u, work, counter, work_cap);
    } else if (key != prev) {
        if (key < P.ncell) start_ref[key] = i;
        const uint32_t pc = prev < P.ncell ? prev : P.ncell;
        if (FILL) fill_cells(cs, pc + 1u, kc + 1u, i, work, counter, work_cap);
    }
    if (FILL) {
        const bool last = (i + 1 == cap) || ((uint32_t)(pairs[i + 1] >> 32) == FS_DEAD_KEY);
        if (last) {
            fill_cells(cs, kc + 1u, P.ncell + 1u, i + 1u, work, counter, work_cap);
            if (i + 1 == cap) *n_live_out = cap;
        }
    }
}

struct AosParticle { float2 position, predicted, velocity; float density; uint32_t grid; };

// AoS export with GLOBAL cell keys (so results of different ranks / a single-GPU run compare).
__global__ __launch_bounds__(SL_BLOCK) void k_slab_export(StepParams P, uint32_t cap, const float2* __restrict__ pos,
                                                          const float2* __restrict__ pred,
                                                          const float2* __restrict__ vel,
                                                          const float* __restrict__ rho,
                                                          const uint32_t* __restrict__ key,
                                                          AosParticle* __restrict__ out) {
    const uint32_t i = blockIdx.x * SL_BLOCK + threadIdx.x;
    if (i >= cap) return;
    AosParticle a;
    a.position = pos[i]; a.predicted = pred[i]; a.velocity = vel[i]; a.density = rho[i];
    const uint32_t k = key[i];
    const uint32_t cy = k / P.grid_w;
    a.grid = cy * P.grid_w_global + (uint32_t)((int32_t)(k - cy * P.grid_w) + P.col_origin);
    out[i] = a;
}

// Initial owned particles: SoA import + owned flags + local keys.
__global__ __launch_bounds__(SL_BLOCK) void k_slab_import(StepParams P, uint32_t n, uint32_t cap,
                                                          const AosParticle* __restrict__ in,
                                                          float2* __restrict__ pos, float2* __restrict__ pred,
                                                          float2* __restrict__ vel, float* __restrict__ rho,
                                                          uint32_t* __restrict__ key, unsigned char* __restrict__ owned) {
    const uint32_t i = blockIdx.x * SL_BLOCK + threadIdx.x;
    if (i >= cap) return;
    if (i < n) {
        const AosParticle a = in[i];
        pos[i] = a.position; pred[i] = a.predicted; vel[i] = a.velocity; rho[i] = a.density;
        uint32_t cxg;
        key[i] = slab_key(P, a.predicted, &cxg);
        owned[i] = 1;
    } else {
        owned[i] = 0;
        key[i] = FS_DEAD_KEY;
    }
}

// Particles per GLOBAL column among the owned columns (for re-balancing): one thread per local column.
__global__ __launch_bounds__(SL_BLOCK) void k_slab_colhist(StepParams P, const uint32_t* __restrict__ cs,
                                                           uint32_t* __restrict__ hist_global) {
    const uint32_t c = blockIdx.x * SL_BLOCK + threadIdx.x;
    if (c >= P.grid_w) return;
    const int32_t cg = (int32_t)c + P.col_origin;
    if (cg < (int32_t)P.own_lo || cg >= (int32_t)P.own_hi) return;
    uint32_t sum = 0;
    for (uint32_t y = 0; y < P.grid_h; ++y) sum += cs[y * P.grid_w + c + 1] - cs[y * P.grid_w + c];
    hist_global[cg] = sum;
}

// Largest |velocity| among the owned live particles, as f32 bits (non-negative floats order like their bits):
// sizes the outer-edge margin between two re-balancing steps (multi.py).
__global__ __launch_bounds__(SL_BLOCK) void k_slab_maxspeed(const uint32_t* __restrict__ n_live,
                                                            const float2* __restrict__ vel,
                                                            const unsigned char* __restrict__ owned,
                                                            uint32_t* __restrict__ out_bits) {
    const uint32_t n = *n_live;
    float m = 0.0f;
    for (uint32_t i = blockIdx.x * SL_BLOCK + threadIdx.x; i < n; i += gridDim.x * SL_BLOCK) {
        if (!owned[i]) continue;
        const float2 v = vel[i];
        const float sp = sqrt_rn(v.x * v.x + v.y * v.y);
        if (sp > m) m = sp;                                  // NaN never wins
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63u) == 0 && m > 0.0f) atomicMax(out_bits, __float_as_uint(m));
}

// ------------------------------------------------------------------ launchers
static inline uint32_t nb(uint32_t n) { return (n + SL_BLOCK - 1) / SL_BLOCK; }

void launch_slab_maxspeed(hipStream_t st, const uint32_t* n_live, const float2* vel, const unsigned char* owned,
                          uint32_t* out_bits) {
    hipLaunchKernelGGL(k_slab_maxspeed, dim3(1024), dim3(SL_BLOCK), 0, st, n_live, vel, owned, out_bits);
}

void launch_slab_pack(hipStream_t st, const StepParams& P, uint32_t main_slots, uint32_t R, int has_left,
                      int has_right, const float2* pos, const float2* vel, const unsigned char* owned, u64* pairs,
                      unsigned char* flags, void* blockcnt, void* blockoff, void* msg_left, void* msg_right,
                      uint32_t* counters, uint32_t* gap_counter) {
    const uint32_t blocks = nb(main_slots);
    // classify covers ALL slots (P.n = capacity): slots past `main_slots` only check for stranded owned particles
    hipLaunchKernelGGL(k_slab_classify, dim3(nb(P.n > main_slots ? P.n : main_slots)), dim3(SL_BLOCK), 0, st, P, main_slots,
                       has_left, has_right, pos, vel, owned, pairs, flags, (uint2*)blockcnt, counters, gap_counter);
    SlabHeader* hl = (SlabHeader*)msg_left;
    SlabHeader* hr = (SlabHeader*)msg_right;
    hipLaunchKernelGGL(k_slab_scan, dim3(1), dim3(SCAN_THREADS), 0, st, (const uint2*)blockcnt, blocks, (uint2*)blockoff,
                       hl, hr, R, counters);
    hipLaunchKernelGGL(k_slab_scatter, dim3(blocks), dim3(SL_BLOCK), 0, st, main_slots, pos, vel, flags,
                       (const uint2*)blockoff, hl ? (float4*)(hl + 1) : nullptr, hr ? (float4*)(hr + 1) : nullptr, R);
}

void launch_slab_unpack(hipStream_t st, const StepParams& P, uint32_t main_slots, uint32_t R, const void* msg_left,
                        const void* msg_right, float2* pos, float2* vel, u64* pairs, uint32_t* counters) {
    const SlabHeader* hl = (const SlabHeader*)msg_left;
    const SlabHeader* hr = (const SlabHeader*)msg_right;
    hipLaunchKernelGGL(k_slab_unpack, dim3(nb(2 * R)), dim3(SL_BLOCK), 0, st, P, main_slots, R, hl,
                       hl ? (const float4*)(hl + 1) : nullptr, hr, hr ? (const float4*)(hr + 1) : nullptr, pos, vel,
                       pairs, counters);
}

void launch_slab_reorder(hipStream_t st, const StepParams& P, uint32_t cap, const u64* pairs, const float2* pos_in,
                         const float2* vel_in, float2* pos_s, float2* vel_s, float2* pred_s, uint32_t* key_s,
                         unsigned char* owned, uint32_t* cs, uint32_t* start_ref, void* work, uint32_t* counter,
                         uint32_t work_cap, uint32_t* n_live_out, unsigned long long* safe, uint32_t* force_defer,
                         uint32_t* force_work_count, bool cs_ready) {
    if (cs_ready) {   // counting sort: table and live count already exist
        hipLaunchKernelGGL(k_slab_reorder<false>, dim3(nb(cap)), dim3(SL_BLOCK), 0, st, P, cap, pairs, pos_in, vel_in,
                           pos_s, vel_s, pred_s, key_s, owned, cs, start_ref, (GapEntry*)work, counter, work_cap, n_live_out, safe, force_defer, force_work_count);
        return;
    }
    hipLaunchKernelGGL(k_slab_reorder<true>, dim3(nb(cap)), dim3(SL_BLOCK), 0, st, P, cap, pairs, pos_in, vel_in, pos_s,
                       vel_s, pred_s, key_s, owned, cs, start_ref, (GapEntry*)work, counter, work_cap, n_live_out, safe, force_defer, force_work_count);
    launch_fill_gaps(st, cs, work, counter, work_cap);
}

void launch_slab_export(hipStream_t st, const StepParams& P, uint32_t cap, const float2* pos, const float2* pred,
                        const float2* vel, const float* rho, const uint32_t* key, void* out) {
    hipLaunchKernelGGL(k_slab_export, dim3(nb(cap)), dim3(SL_BLOCK), 0, st, P, cap, pos, pred, vel, rho, key,
                       (AosParticle*)out);
}

void launch_slab_import(hipStream_t st, const StepParams& P, uint32_t n, uint32_t cap, const void* in, float2* pos,
                        float2* pred, float2* vel, float* rho, uint32_t* key, unsigned char* owned) {
    hipLaunchKernelGGL(k_slab_import, dim3(nb(cap)), dim3(SL_BLOCK), 0, st, P, n, cap, (const AosParticle*)in, pos,
                       pred, vel, rho, key, owned);
}

void launch_slab_colhist(hipStream_t st, const StepParams& P, const uint32_t* cs, uint32_t* hist_global) {
    hipLaunchKernelGGL(k_slab_colhist, dim3(nb(P.grid_w)), dim3(SL_BLOCK), 0, st, P, cs, hist_global);
}

size_t slab_message_bytes(uint32_t R) { return sizeof(SlabHeader) + (size_t)R * sizeof(float4); }

}  // namespace fsd

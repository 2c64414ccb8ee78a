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
//   k_slab_pack      ONE launch: predict + global column; old ghosts and leavers become DEAD; the records each neighbour
//                    needs (migrants and the 2-column ghost halo) are compacted in slot order (deterministic) into the
//                    fixed-size messages [16-B header | R x {pos, vel}] through a decoupled look-back; the key goes
//                    straight into the counting sort's histogram (its arrival ticket is stored beside it)
//   k_slab_unpack    received records -> slots, key from the recomputed predicted position (+ histogram)
//   counting sort (default): k_scan_lookback -> k_cs_scatter -> k_cs_fixreorder<true> (kernels_csort.hip; the last one is
//                    the reorder pass as well: live count, owned flags, start_indices)
//   bitonic mode:    the network over all slots (DEAD keys end up last) + k_slab_reorder
//   k_density / k_force run unchanged on the local window (ghosts are not advanced)
#include "fs_device.h"
#include "fs_kernels.h"
#include "fs_scan.h"

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

struct SlabHeader { uint32_t count, overflow, pad0, pad1; };

// The pack in TWO launches (round 2: classify -> single-workgroup scan -> scatter, three launches and a flags array):
//  k_slab_pack  * predict + global column of every carried-over owned particle; old ghosts and leavers become DEAD;
//               * COUNTING: the key goes straight into the counting sort's histogram — the atomic's return value is the
//                 particle's arrival ticket in its cell (kt[i] = key << 32 | ticket, kernels_csort.hip) — so the sort
//                 needs no pass of its own over the slots; bitonic mode: pairs[i] = key << 32 | i as before;
//               * the slots each neighbour needs (migrants + the 2-column halo) are listed per 256-slot block, in slot
//                 order (stage_l / stage_r, 256 entries per block), with the two counts in blockcnt[block];
//  k_slab_msg   one workgroup per MSG_GROUP blocks: exclusive offsets of its blocks by a wave scan + a decoupled
//               look-back over the (few) workgroups (fs_scan.h), then the listed records are gathered into the two
//               fixed-size messages in SLOT ORDER (deterministic); the last workgroup writes the two headers.
// (A look-back over the 256-slot blocks themselves — one launch — was measured first: its prefix frontier advances ~128
//  blocks per global-memory round trip, 0.17 ms for the 11 136 blocks of an 8-way rank.  The chain must be short.)
template <bool COUNTING>
__global__ __launch_bounds__(SL_BLOCK) void k_slab_pack(StepParams P, uint32_t cap, uint32_t main_slots,
                                                        int has_left, int has_right, const float2* __restrict__ pos,
                                                        const float2* __restrict__ vel,
                                                        const unsigned char* __restrict__ owned, u64* __restrict__ out /* kt or pairs */,
                                                        uint32_t* __restrict__ hist, uint2* __restrict__ blockcnt,
                                                        uint32_t* __restrict__ stage_l, uint32_t* __restrict__ stage_r,
                                                        uint32_t* __restrict__ counters, uint32_t* __restrict__ gap_counter,
                                                        unsigned long long* __restrict__ safe) {
    __shared__ uint32_t s_cnt[2 * (SL_BLOCK / 64)];
    const uint32_t i = blockIdx.x * SL_BLOCK + threadIdx.x;
    if (i == 0) *gap_counter = 0;          // cell-table worklist of this step (bitonic mode: k_slab_reorder)
    if (COUNTING && i < (cap + 63u) / 64u) safe[i] = ~0ull;             // k_cs_fixreorder clears the unsafe bits
    const uint32_t n_prev = *P.n_live;
    unsigned char f = 0;
    uint32_t key = FS_DEAD_KEY;
    if (i < main_slots) {
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
        if (!COUNTING) out[i] = ((u64)key << 32) | (u64)i;
    } else if (i < n_prev && i < cap && owned[i]) {
        // Slot capacity exceeded: the last step left more live records than main slots, and this owned
        // particle sits where the incoming messages will be unpacked.  It cannot be carried over —
        // count it (fs_slab_counters.overflow must stay 0; the driver raises on it).
        atomicAdd(&counters[3], 1u);
    }
    if (COUNTING) {
        const bool active = key != FS_DEAD_KEY;
        const uint32_t k = key < P.ncell ? key : P.ncell - 1u;
        const WaveRun r = wave_run(k, active);
        uint32_t base = 0;
        if (r.is_head) base = atomicAdd(&hist[k], r.length);
        base = __shfl(base, r.head_lane);
        if (i < main_slots) out[i] = ((u64)key << 32) | (u64)(active ? base + r.offset : 0u);
    }
    // ---- the slots each message needs, listed per block in slot order
    const unsigned long long mL = __ballot(f & 1), mR = __ballot(f & 2);
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (lane == 0) { s_cnt[2 * w] = __popcll(mL); s_cnt[2 * w + 1] = __popcll(mR); }
    __syncthreads();
    uint32_t wl = 0, wr = 0, tl = 0, tr = 0;
#pragma unroll
    for (uint32_t k = 0; k < SL_BLOCK / 64; ++k) {
        const uint32_t a = s_cnt[2 * k], b = s_cnt[2 * k + 1];
        if (k < w) { wl += a; wr += b; }
        tl += a; tr += b;
    }
    if (threadIdx.x == 0) blockcnt[blockIdx.x] = make_uint2(tl, tr);
    if (f) {
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        if (f & 1) stage_l[blockIdx.x * SL_BLOCK + wl + __popcll(mL & below)] = i;
        if (f & 2) stage_r[blockIdx.x * SL_BLOCK + wr + __popcll(mR & below)] = i;
    }
}

// Two message counters (records for the left / the right neighbour) travel through the look-back as one 40-bit payload
// of two saturating 20-bit fields: a count only matters up to R + 1 (overflow), and R < 2^20 - 2 (fs_slab_create).
#define PK_FIELD 0xFFFFFull
__device__ __forceinline__ u64 pk_add(u64 a, u64 b) {
    u64 l = (a >> 20) + (b >> 20), r = (a & PK_FIELD) + (b & PK_FIELD);
    if (l > PK_FIELD) l = PK_FIELD;
    if (r > PK_FIELD) r = PK_FIELD;
    return (l << 20) | r;
}

#define MSG_GROUP 64u        // pack blocks per k_slab_msg workgroup (one wave scans their counts)
__global__ __launch_bounds__(SL_BLOCK) void k_slab_msg(uint32_t nblocks_pack, uint32_t R, const uint2* __restrict__ blockcnt,
                                                       const uint32_t* __restrict__ stage_l, const uint32_t* __restrict__ stage_r,
                                                       const float2* __restrict__ pos, const float2* __restrict__ vel,
                                                       u64* __restrict__ state, uint32_t* __restrict__ ticket, uint32_t epoch,
                                                       SlabHeader* hdr_left, SlabHeader* hdr_right,
                                                       float4* __restrict__ rec_left, float4* __restrict__ rec_right,
                                                       uint32_t* __restrict__ counters) {
    __shared__ uint32_t s_pl[MSG_GROUP + 1], s_pr[MSG_GROUP + 1];      // exclusive prefixes of the group's block counts
    __shared__ uint32_t s_bid;
    __shared__ u64 s_excl;
    if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t bid = s_bid, ngroups = gridDim.x;
    if (bid == ngroups - 1u && threadIdx.x == 0) *ticket = 0u;          // every ticket of this launch has been handed out
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t b0 = bid * MSG_GROUP;
    const u64 tag = (u64)(epoch & 0x3FFFFFu);
    if (w == 0) {
        const uint32_t b = b0 + lane;
        const uint2 c = b < nblocks_pack ? blockcnt[b] : make_uint2(0u, 0u);
        uint32_t il = c.x, ir = c.y;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t tx = __shfl_up(il, o), ty = __shfl_up(ir, o);
            if ((int)lane >= o) { il += tx; ir += ty; }
        }
        s_pl[lane + 1u] = il; s_pr[lane + 1u] = ir;                     // inclusive -> exclusive at [lane + 1]
        if (lane == 0) { s_pl[0] = 0u; s_pr[0] = 0u; }
        const uint32_t tl = __shfl(il, 63), tr = __shfl(ir, 63);        // <= 64 * 256 each
        const u64 mine = ((u64)tl << 20) | (u64)tr;
        if (bid == 0) {
            if (lane == 0) { lb_store(state, (LB_FLAG_PREFIX << 62) | (tag << 40) | mine); s_excl = 0ull; }
        } else {
            if (lane == 0) lb_store(state + bid, (LB_FLAG_AGG << 62) | (tag << 40) | mine);
            const u64 ex = lookback_exclusive<40>(state, bid, tag, pk_add);
            if (lane == 0) { lb_store(state + bid, (LB_FLAG_PREFIX << 62) | (tag << 40) | pk_add(ex, mine)); s_excl = ex; }
        }
    }
    __syncthreads();
    const u64 ex = s_excl;
    const uint32_t offl = (uint32_t)(ex >> 20), offr = (uint32_t)(ex & PK_FIELD);
    const uint32_t tl = s_pl[MSG_GROUP], tr = s_pr[MSG_GROUP];
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const uint32_t total = side ? tr : tl, off = side ? offr : offl;
        const uint32_t* pre = side ? s_pr : s_pl;
        const uint32_t* stage = side ? stage_r : stage_l;
        float4* rec = side ? rec_right : rec_left;
        if (!rec) continue;
        for (uint32_t r = threadIdx.x; r < total; r += SL_BLOCK) {
            uint32_t lo = 0, hi = MSG_GROUP;                             // largest pb with pre[pb] <= r
            while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (pre[mid] <= r) lo = mid; else hi = mid; }
            const uint32_t slot = stage[(b0 + lo) * SL_BLOCK + (r - pre[lo])];
            const uint32_t d = off + r;
            if (d < R) { const float2 p = pos[slot], v = vel[slot]; rec[d] = make_float4(p.x, p.y, v.x, v.y); }
        }
    }
    if (bid == ngroups - 1u && threadIdx.x == 0) {                      // totals: the last group's inclusive prefix
        const uint32_t totl = offl + tl, totr = offr + tr;              // saturated at 2^20 - 1 > R
        if (hdr_left) { hdr_left->count = totl < R ? totl : R; hdr_left->overflow = totl > R; }
        if (hdr_right) { hdr_right->count = totr < R ? totr : R; hdr_right->overflow = totr > R; }
        if ((hdr_left && totl > R) || (hdr_right && totr > R)) atomicAdd(&counters[3], 1u);
    }
}

template <bool COUNTING>
__global__ __launch_bounds__(SL_BLOCK) void k_slab_unpack(StepParams P, uint32_t main_slots, uint32_t R,
                                                          const SlabHeader* __restrict__ hdr_left,
                                                          const float4* __restrict__ rec_left,
                                                          const SlabHeader* __restrict__ hdr_right,
                                                          const float4* __restrict__ rec_right,
                                                          float2* __restrict__ pos, float2* __restrict__ vel,
                                                          u64* __restrict__ out /* kt or pairs */, uint32_t* __restrict__ hist,
                                                          uint32_t* __restrict__ counters) {
    const uint32_t j = blockIdx.x * SL_BLOCK + threadIdx.x;
    const bool in_range = j < 2u * R;
    const bool right = j >= R;
    const uint32_t jj = right ? j - R : j;
    const SlabHeader* hdr = right ? hdr_right : hdr_left;
    const float4* rec = right ? rec_right : rec_left;
    uint32_t cnt = 0;
    if (in_range && hdr) { cnt = hdr->count < R ? hdr->count : R; if (jj == 0 && hdr->overflow) atomicAdd(&counters[3], 1u); }
    const uint32_t slot = main_slots + j;
    uint32_t key = FS_DEAD_KEY;
    if (in_range && jj < cnt) {
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
    if (COUNTING) {                                                     // received records join the histogram (after k_slab_pack's)
        const bool active = key != FS_DEAD_KEY;
        const uint32_t k = key < P.ncell ? key : P.ncell - 1u;
        const WaveRun r = wave_run(k, active);
        uint32_t base = 0;
        if (r.is_head) base = atomicAdd(&hist[k], r.length);
        base = __shfl(base, r.head_lane);
        if (in_range) out[slot] = ((u64)key << 32) | (u64)(active ? base + r.offset : 0u);
    } else if (in_range) {
        out[slot] = ((u64)key << 32) | (u64)slot;
    }
}

// k_reorder for slab mode: DEAD slots are skipped, the live count and the owned flags are
// produced here.  `cap` = number of slots sorted.
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
        {   // the live count and the table come from here
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
        fill_cells(cs, 0u, kc + 1u, 0u, work, counter, work_cap);
    } else if (key != prev) {
        if (key < P.ncell) start_ref[key] = i;
        const uint32_t pc = prev < P.ncell ? prev : P.ncell;
        fill_cells(cs, pc + 1u, kc + 1u, i, work, counter, work_cap);
    }
    {
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
                      int has_right, const float2* pos, const float2* vel, const unsigned char* owned, u64* out,
                      uint32_t* hist, void* blockcnt, uint32_t* stage /* 2 x capacity words */, void* state, uint32_t epoch,
                      void* msg_left, void* msg_right, uint32_t* counters, uint32_t* gap_counter, unsigned long long* safe,
                      bool counting) {
    // covers ALL slots (P.n = capacity): slots past `main_slots` only check for stranded owned particles
    const uint32_t cap = P.n > main_slots ? P.n : main_slots;
    const uint32_t blocks = nb(cap), groups = (blocks + MSG_GROUP - 1u) / MSG_GROUP;
    uint32_t* stage_l = stage;
    uint32_t* stage_r = stage + (size_t)blocks * SL_BLOCK;
    SlabHeader* hl = (SlabHeader*)msg_left;
    SlabHeader* hr = (SlabHeader*)msg_right;
    if (counting)
        hipLaunchKernelGGL(k_slab_pack<true>, dim3(blocks), dim3(SL_BLOCK), 0, st, P, cap, main_slots, has_left, has_right, pos,
                           vel, owned, out, hist, (uint2*)blockcnt, stage_l, stage_r, counters, gap_counter, safe);
    else
        hipLaunchKernelGGL(k_slab_pack<false>, dim3(blocks), dim3(SL_BLOCK), 0, st, P, cap, main_slots, has_left, has_right, pos,
                           vel, owned, out, hist, (uint2*)blockcnt, stage_l, stage_r, counters, gap_counter, safe);
    if (!hl && !hr) return;                                             // no neighbour: nothing to send
    hipLaunchKernelGGL(k_slab_msg, dim3(groups), dim3(SL_BLOCK), 0, st, blocks, R, (const uint2*)blockcnt, stage_l, stage_r, pos,
                       vel, (u64*)state, counters + 6, epoch, hl, hr, hl ? (float4*)(hl + 1) : nullptr,
                       hr ? (float4*)(hr + 1) : nullptr, counters);
}
// words of `stage` and of the look-back state launch_slab_pack needs for `cap` slots
size_t slab_stage_words(uint32_t cap) { return 2 * (size_t)nb(cap) * SL_BLOCK; }
size_t slab_msg_groups(uint32_t cap) { return (nb(cap) + MSG_GROUP - 1u) / MSG_GROUP; }

void launch_slab_unpack(hipStream_t st, const StepParams& P, uint32_t main_slots, uint32_t R, const void* msg_left,
                        const void* msg_right, float2* pos, float2* vel, u64* out, uint32_t* hist, uint32_t* counters,
                        bool counting) {
    const SlabHeader* hl = (const SlabHeader*)msg_left;
    const SlabHeader* hr = (const SlabHeader*)msg_right;
    const float4* rl = hl ? (const float4*)(hl + 1) : nullptr;
    const float4* rr = hr ? (const float4*)(hr + 1) : nullptr;
    if (counting)
        hipLaunchKernelGGL(k_slab_unpack<true>, dim3(nb(2 * R)), dim3(SL_BLOCK), 0, st, P, main_slots, R, hl, rl, hr, rr, pos, vel,
                           out, hist, counters);
    else
        hipLaunchKernelGGL(k_slab_unpack<false>, dim3(nb(2 * R)), dim3(SL_BLOCK), 0, st, P, main_slots, R, hl, rl, hr, rr, pos, vel,
                           out, hist, counters);
}

// bitonic slab mode only (the counting sort's k_cs_fixreorder<true> does the reorder itself)
void launch_slab_reorder(hipStream_t st, const StepParams& P, uint32_t cap, const u64* pairs, const float2* pos_in,
                         const float2* vel_in, float2* pos_s, float2* vel_s, float2* pred_s, uint32_t* key_s,
                         unsigned char* owned, uint32_t* cs, uint32_t* start_ref, void* work, uint32_t* counter,
                         uint32_t work_cap, uint32_t* n_live_out, unsigned long long* safe, uint32_t* force_defer,
                         uint32_t* force_work_count) {
    hipLaunchKernelGGL(k_slab_reorder, dim3(nb(cap)), dim3(SL_BLOCK), 0, st, P, cap, pairs, pos_in, vel_in, pos_s,
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

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
    return key_of_local(P, (uint32_t)((int32_t)cx - P.col_origin), cy);
}

struct SlabHeader { uint32_t count, overflow, pad0, pad1; };

// The pack in TWO launches (round 2: classify -> single-workgroup scan -> scatter, three launches and a flags array):
//  k_slab_pack  * predict + global column of every carried-over owned particle; old ghosts and leavers become DEAD;
//               * COUNTING: the key goes straight into the counting sort's histogram — the atomic's return value is the
//                 particle's arrival ticket in its cell (kt[i] = key << 32 | ticket, kernels_csort.hip) — so the sort
//                 needs no pass of its own over the slots; bitonic mode: pairs[i] = key << 32 | i as before;
//               * the slots each neighbour needs (migrants + the 2-column halo) are listed per 256-slot block, in slot
//                 order (stage_l / stage_r, 256 entries per block), with the two counts in blockcnt[block];
//  k_slab_msg   one wave per MSG_GROUP blocks: exclusive message offsets of its blocks by a wave scan + a decoupled
//               look-back over the (few) groups (fs_scan.h); the last group writes the two headers;
//  k_slab_gather one workgroup per block: its listed records -> the two fixed-size messages, in SLOT ORDER (deterministic).
// (A look-back over the 256-slot blocks themselves — one launch — was measured first: its prefix frontier advances ~128
//  blocks per global-memory round trip, 0.17 ms for the 11 136 blocks of an 8-way rank.  The chain must be short.)
// The slots of one 256-slot block each message needs (flag bit 0: left, bit 1: right), listed in slot order (stage_l / stage_r,
// 256 entries per block) with the two counts in blockcnt[block]: the input of k_slab_msg.  Whole workgroup.
__device__ __forceinline__ void block_message_lists(unsigned char f, uint32_t i, uint32_t blk, uint2* __restrict__ blockcnt,
                                                    uint32_t* __restrict__ stage_l, uint32_t* __restrict__ stage_r) {
    __shared__ uint32_t s_cnt[2 * (SL_BLOCK / 64)];
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
    if (threadIdx.x == 0) blockcnt[blk] = make_uint2(tl, tr);
    if (f) {
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        if (f & 1) stage_l[blk * SL_BLOCK + wl + __popcll(mL & below)] = i;
        if (f & 2) stage_r[blk * SL_BLOCK + wr + __popcll(mR & below)] = i;
    }
}

// overlap != 0 (the overlapped step, engine.hip): ghosts never enter the main array, so the slots past `main_slots` are not the
// unpack area of this step but hold the MIGRANTS the boundary strips received and advanced in the last one (owned flag
// set by k_strip_writeback); they are carried over like the sorted prefix.
template <bool COUNTING>
__global__ __launch_bounds__(SL_BLOCK) void k_slab_pack(StepParams P, uint32_t cap, uint32_t main_slots, int overlap, int lists,
                                                        int has_left, int has_right, const float2* __restrict__ pos,
                                                        const float2* __restrict__ vel,
                                                        const unsigned char* __restrict__ owned, u64* __restrict__ out /* kt or pairs */,
                                                        uint32_t* __restrict__ hist, uint2* __restrict__ blockcnt,
                                                        uint32_t* __restrict__ stage_l, uint32_t* __restrict__ stage_r,
                                                        uint32_t* __restrict__ counters, uint32_t* __restrict__ gap_counter,
                                                        unsigned long long* __restrict__ safe,
                                                        const uint32_t* __restrict__ key_prev, uint32_t prev_adv_lo,
                                                        uint32_t prev_adv_hi, int skip_edge) {
    const uint32_t i = blockIdx.x * SL_BLOCK + threadIdx.x;
    if (i == 0) *gap_counter = 0;          // cell-table worklist of this step (bitonic mode: k_slab_reorder)
    const uint32_t n_prev = *P.n_live;
    unsigned char f = 0;
    uint32_t key = FS_DEAD_KEY;
    if (overlap && i == 0 && n_prev > main_slots) atomicAdd(&counters[3], 1u);   // the sorted prefix ran into the migrant slots
    // lists == 0 (the messages were pre-built): where was this particle in the last step?  In the edge zone — then the edge
    // columns' chain of that step advanced it, put it into the messages and, with skip_edge, classified its slot for this step's
    // sort as well (k_slab_prepack: this launch then neither reads nor writes anything of it, and need not wait for that chain)
    bool was_edge = false, skipped = false;
    if (!lists && i < main_slots && i < n_prev && owned[i]) {
        uint32_t cxl, cy;
        key_to_local(P, key_prev[i], &cxl, &cy);
        const int32_t cg = (int32_t)cxl + P.col_origin;
        was_edge = cg >= (int32_t)P.own_lo && cg < (int32_t)P.own_hi && !(cg >= (int32_t)prev_adv_lo && cg < (int32_t)prev_adv_hi);
        skipped = skip_edge && was_edge;
    }
    if (i < main_slots || (overlap && i < cap)) {
        if ((i < main_slots ? i < n_prev : true) && owned[i] && !skipped) {
            const float2 pr = predict_pos(P, pos[i], vel[i]);
            uint32_t cxg;
            key = slab_key(P, pr, &cxg);
            if (has_left && cxg < P.own_lo + 2u) f |= 1;
            if (has_right && cxg + 2u >= P.own_hi) f |= 2;
            // a leaver must land inside the neighbour's slab and not in ITS far halo: checked by the receiver
            if (!has_left && cxg < P.own_lo) atomicAdd(&counters[2], 1u);    // left the domain partition
            if (!has_right && cxg >= P.own_hi) atomicAdd(&counters[2], 1u);
        }
        if (!COUNTING && !skipped) out[i] = ((u64)key << 32) | (u64)i;
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
        if ((i < main_slots || (overlap && i < cap)) && !skipped) out[i] = ((u64)key << 32) | (u64)(active ? base + r.offset : 0u);
    }
    if (lists) {
        block_message_lists(f, i, blockIdx.x, blockcnt, stage_l, stage_r);
    } else if (f && !was_edge) {
        // The messages of this step were built at the end of the last one, from the particles its edge-column force launch had
        // advanced by then (k_slab_prepack).  This full classification flags the same particles — unless one reached the 2-column
        // band from farther inside than the edge zone (its key of the last step says where it was): then the message that went
        // out lacks it.  Counted in far_halo (the edge zone was too narrow for its speed: fs_slab_set_boundary_cols).
        atomicAdd(&counters[4], 1u);
    }
}

// Edge-first step (engine.hip): the messages of step t+1 are built at the END of step t, as soon as the force pass
// has advanced the owned columns within `boundary_cols` of a neighboured edge (StepParams::adv_outside launch) — the exchange
// then runs beside the force pass of the interior columns and the next step's k_slab_pack.  Same classification, same lists,
// same slot order as k_slab_pack would produce at t+1 (a particle's sorted index now IS its slot then), restricted to the
// particles that launch advanced; `P` carries the window and the tick constants the next pack will use.  k_slab_pack (lists
// = 0) counts what the full classification flags and k_slab_unpack checks it against the headers: a particle that reached the
// 2-column band from farther inside than the edge zone shows up in far_halo.
__global__ __launch_bounds__(SL_BLOCK) void k_slab_prepack(StepParams P, int has_left, int has_right, int edge_walk,
                                                           const float2* __restrict__ pos, const float2* __restrict__ vel,
                                                           const unsigned char* __restrict__ owned,
                                                           const uint32_t* __restrict__ key_s, const uint32_t* __restrict__ cs,
                                                           uint2* __restrict__ blockcnt,
                                                           uint32_t* __restrict__ stage_l, uint32_t* __restrict__ stage_r,
                                                           int classify, int counting, uint32_t main_slots, u64* __restrict__ out,
                                                           uint32_t* __restrict__ hist, uint32_t* __restrict__ counters) {
    const uint32_t n = *P.n_live;
    // edge_walk (column-major ids): a small grid walks the blocks of the edge columns only (fs_device.h EdgeBlocks; k_slab_msg and
    // k_slab_gather take every other block's counts as zero); otherwise one workgroup per 256-slot block of the whole array
    EdgeBlocks E;
    E.eL = 0u; E.eR = 0u; E.nb = gridDim.x;
    if (edge_walk) E = edge_blocks(P, cs, n, 0u);
    const uint32_t count = edge_walk ? edge_block_count(E) : gridDim.x;
    for (uint32_t t = blockIdx.x; t < count; t += gridDim.x) {
        const uint32_t blk = edge_walk ? edge_block_at(E, t) : t;
        const uint32_t i = blk * SL_BLOCK + threadIdx.x;
        unsigned char f = 0;
        uint32_t key = FS_DEAD_KEY;
        bool mine = false;                                  // classify: this launch does the next pack's work for the slot
        if (i < n && owned[i]) {
            uint32_t cxl, cy;
            key_to_local(P, key_s[i], &cxl, &cy);
            const int32_t cg = (int32_t)cxl + P.col_origin;
            if (slab_advances(P, cg)) {                     // advanced already: pos / vel hold its new state
                uint32_t cxg;
                key = slab_key(P, predict_pos(P, pos[i], vel[i]), &cxg);
                if (has_left && cxg < P.own_lo + 2u) f |= 1;
                if (has_right && cxg + 2u >= P.own_hi) f |= 2;
                mine = classify && i < main_slots;
                if (mine) {                                 // exactly k_slab_pack's bookkeeping for an owned, carried-over slot
                    if (!has_left && cxg < P.own_lo) atomicAdd(&counters[2], 1u);
                    if (!has_right && cxg >= P.own_hi) atomicAdd(&counters[2], 1u);
                }
            }
        }
        if (classify) {
            if (counting) {
                const bool active = mine && key != FS_DEAD_KEY;
                const uint32_t k = key < P.ncell ? key : P.ncell - 1u;
                const WaveRun r = wave_run(k, active);
                uint32_t base = 0;
                if (r.is_head) base = atomicAdd(&hist[k], r.length);
                base = __shfl(base, r.head_lane);
                if (mine) out[i] = ((u64)key << 32) | (u64)(active ? base + r.offset : 0u);
            } else if (mine) {
                out[i] = ((u64)key << 32) | (u64)i;
            }
        }
        block_message_lists(f, i, blk, blockcnt, stage_l, stage_r);
        __syncthreads();
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
// Round 4: k_slab_msg only turns the per-block counts into per-block message offsets (blockoff) and writes the headers; the
// records are gathered by k_slab_gather, one workgroup per pack block.  (One kernel did both, each workgroup gathering the
// records of its 64 blocks: fine while the flagged slots are spread over the whole array — a few per grid row — but with
// column-major cell ids a rank's edge columns are ~80 CONSECUTIVE blocks per side, i.e. two workgroups gathered 20 000 records
// each: 50 - 67 us instead of 8.)
__global__ __launch_bounds__(64) void k_slab_msg(uint32_t nblocks_pack, uint32_t R, const uint2* __restrict__ blockcnt,
                                                 uint2* __restrict__ blockoff, u64* __restrict__ state,
                                                 uint32_t* __restrict__ ticket, uint32_t epoch, SlabHeader* hdr_left,
                                                 SlabHeader* hdr_right, uint32_t* __restrict__ counters, StepParams P,
                                                 const uint32_t* __restrict__ cs_edge) {
    __shared__ uint32_t s_bid;
    // cs_edge != null: only the edge columns' blocks were classified (k_slab_prepack with edge_walk); every other count is zero
    EdgeBlocks E;
    E.eL = 0u; E.eR = 0u; E.nb = nblocks_pack;
    if (cs_edge) E = edge_blocks(P, cs_edge, *P.n_live, 0u);
    if (threadIdx.x == 0) s_bid = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t bid = s_bid, ngroups = gridDim.x;
    if (bid == ngroups - 1u && threadIdx.x == 0) *ticket = 0u;          // every ticket of this launch has been handed out
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t b = bid * MSG_GROUP + lane;
    const u64 tag = (u64)(epoch & 0x3FFFFFu);
    const uint2 c = (b < nblocks_pack && (!cs_edge || edge_block_has(E, b))) ? blockcnt[b] : make_uint2(0u, 0u);
    uint32_t il = c.x, ir = c.y;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t tx = __shfl_up(il, o), ty = __shfl_up(ir, o);
        if ((int)lane >= o) { il += tx; ir += ty; }
    }
    const uint32_t tl = __shfl(il, 63), tr = __shfl(ir, 63);            // <= 64 * 256 each
    const u64 mine = ((u64)tl << 20) | (u64)tr;
    u64 ex = 0ull;
    if (bid == 0) {
        if (lane == 0) lb_store(state, (LB_FLAG_PREFIX << 62) | (tag << 40) | mine);
    } else {
        if (lane == 0) lb_store(state + bid, (LB_FLAG_AGG << 62) | (tag << 40) | mine);
        ex = lookback_exclusive<40>(state, bid, tag, pk_add);
        if (lane == 0) lb_store(state + bid, (LB_FLAG_PREFIX << 62) | (tag << 40) | pk_add(ex, mine));
    }
    const uint32_t offl = (uint32_t)(ex >> 20), offr = (uint32_t)(ex & PK_FIELD);
    if (b < nblocks_pack) blockoff[b] = make_uint2(offl + il - c.x, offr + ir - c.y);      // exclusive offsets of block b's records
    if (bid == ngroups - 1u && lane == 0) {                             // totals: the last group's inclusive prefix
        const uint32_t totl = offl + tl, totr = offr + tr;              // saturated at 2^20 - 1 > R
        if (hdr_left) { hdr_left->count = totl < R ? totl : R; hdr_left->overflow = totl > R; }
        if (hdr_right) { hdr_right->count = totr < R ? totr : R; hdr_right->overflow = totr > R; }
        if ((hdr_left && totl > R) || (hdr_right && totr > R)) atomicAdd(&counters[3], 1u);
    }
}

// The listed records of one pack block -> the two messages, in SLOT order (deterministic).
__global__ __launch_bounds__(SL_BLOCK) void k_slab_gather(uint32_t R, const uint2* __restrict__ blockcnt,
                                                          const uint2* __restrict__ blockoff,
                                                          const uint32_t* __restrict__ stage_l, const uint32_t* __restrict__ stage_r,
                                                          const float2* __restrict__ pos, const float2* __restrict__ vel,
                                                          float4* __restrict__ rec_left, float4* __restrict__ rec_right,
                                                          StepParams P, const uint32_t* __restrict__ cs_edge) {
    EdgeBlocks E;
    E.eL = 0u; E.eR = 0u; E.nb = gridDim.x;
    if (cs_edge) E = edge_blocks(P, cs_edge, *P.n_live, 0u);
    const uint32_t count = cs_edge ? edge_block_count(E) : gridDim.x;
    for (uint32_t w = blockIdx.x; w < count; w += gridDim.x) {
        const uint32_t blk = cs_edge ? edge_block_at(E, w) : w;
        const uint2 c = blockcnt[blk];
        if ((c.x | c.y) == 0u) continue;
        const uint2 o = blockoff[blk];
        const uint32_t t = threadIdx.x;
        if (rec_left && t < c.x && o.x + t < R) {
            const uint32_t slot = stage_l[blk * SL_BLOCK + t];
            const float2 p = pos[slot], v = vel[slot];
            rec_left[o.x + t] = make_float4(p.x, p.y, v.x, v.y);
        }
        if (rec_right && t < c.y && o.y + t < R) {
            const uint32_t slot = stage_r[blk * SL_BLOCK + t];
            const float2 p = pos[slot], v = vel[slot];
            rec_right[o.y + t] = make_float4(p.x, p.y, v.x, v.y);
        }
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
    uint32_t cxl, cy;
    key_to_local(P, key, &cxl, &cy);
    const int32_t cxg = (int32_t)cxl + P.col_origin;
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
    uint32_t cxl, cy;
    key_to_local(P, key[i], &cxl, &cy);
    a.grid = cy * P.grid_w_global + (uint32_t)((int32_t)cxl + P.col_origin);      // the reference's id (funcs.wgsl:216-218)
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
    if (P.transposed) sum = cs[(c + 1u) * P.grid_h] - cs[c * P.grid_h];       // a column is one contiguous range of cell ids
    else for (uint32_t y = 0; y < P.grid_h; ++y) sum += cs[y * P.grid_w + c + 1] - cs[y * P.grid_w + c];
    atomicAdd(&hist_global[cg], sum);       // (the buffer was zeroed; k_slab_colhist_migrants adds to the same words)
}

// Overlapped step: the migrants a rank received and advanced in the last step sit in the slots past the main ones, outside
// the sorted prefix and its cell table; `key` holds their local cell key of that step.
__global__ __launch_bounds__(SL_BLOCK) void k_slab_colhist_migrants(StepParams P, uint32_t first, uint32_t count,
                                                                    const unsigned char* __restrict__ owned,
                                                                    const uint32_t* __restrict__ key,
                                                                    uint32_t* __restrict__ hist_global) {
    const uint32_t j = blockIdx.x * SL_BLOCK + threadIdx.x;
    if (j >= count || !owned[first + j]) return;
    const uint32_t k = key[first + j];
    if (k == FS_DEAD_KEY) return;
    uint32_t cxl, cy;
    key_to_local(P, k, &cxl, &cy);
    const int32_t cg = (int32_t)cxl + P.col_origin;
    if (cg >= 0 && cg < (int32_t)P.grid_w_global) atomicAdd(&hist_global[cg], 1u);
}

// Largest |velocity| among the owned live particles, as f32 bits (non-negative floats order like their bits):
// sizes the outer-edge margin between two re-balancing steps (multi.py).
__global__ __launch_bounds__(SL_BLOCK) void k_slab_maxspeed(const uint32_t* __restrict__ n_live,
                                                            const float2* __restrict__ vel,
                                                            const unsigned char* __restrict__ owned,
                                                            uint32_t* __restrict__ out_bits, uint32_t migr_first,
                                                            uint32_t migr_count) {
    const uint32_t n = *n_live;
    float m = 0.0f;
    // the sorted prefix [0, n), then (overlapped step) the migrant slots [migr_first, migr_first + migr_count)
    for (uint32_t t = blockIdx.x * SL_BLOCK + threadIdx.x; t < n + migr_count; t += gridDim.x * SL_BLOCK) {
        const uint32_t i = t < n ? t : migr_first + (t - n);
        if (t >= n && i < n) continue;                       // (a prefix that ran into the migrant slots: counted once)
        if (!owned[i]) continue;
        const float2 v = vel[i];
        const float sp = sqrt_rn(v.x * v.x + v.y * v.y);
        if (sp > m) m = sp;                                  // NaN never wins
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(m, o); m = t > m ? t : m; }
    if ((threadIdx.x & 63u) == 0 && m > 0.0f) atomicMax(out_bits, __float_as_uint(m));
}

// ------------------------------------------------------------------ overlapped step: the boundary strips
// (engine.hip fs_slab_pack / fs_slab_step.)  Ghosts stay OUT of the main sorted array.  While the two halo messages are in
// flight the rank sorts its carried-over particles and runs density + force for the INTERIOR columns [adv_lo, adv_hi); what
// the received records can influence — the owned columns within `Z` of a slab edge — is computed afterwards on a small second
// array, the STRIP: every particle of the main array whose cell column lies in a strip window (the two ghost columns, the
// boundary columns, and two columns of interior context), plus the received records.  The strip uses the main array's own
// window and cell keys, is counting-sorted like it, and goes through the SAME k_density / k_force (StepParams::adv_outside);
// k_strip_writeback puts the results back: boundary particles to their index in the main arrays, migrants to the slot past
// the main ones that mirrors their position in the message (k_slab_pack carries them over in the next step).
//
//   k_strip_rows      (1 workgroup) per grid row and window: the main array's index range -> exclusive offsets; totals
//   k_strip_gather    one wave per (row, window): copies {pos, vel} of the range into the strip's slots, histogram + ticket
//   k_strip_unpack    the received records behind them; classification (ghost / migrant), protocol checks
//   k_strip_writeback results -> main arrays
struct StripWin { uint32_t lo0, hi0, lo1, hi1; };      // LOCAL columns [lo0, hi0) and [lo1, hi1); an empty window has lo == hi
#define STRIP_NONE 0xFFFFFFFFu

#define SR_BLOCK 1024
__global__ __launch_bounds__(SR_BLOCK) void k_strip_rows(uint32_t grid_w, uint32_t grid_h, StripWin W, uint32_t R2,
                                                         const uint32_t* __restrict__ cs, uint32_t* __restrict__ rowbase,
                                                         uint32_t* __restrict__ strip_counters) {
    __shared__ uint32_t s_wave[SR_BLOCK / 64];
    __shared__ uint32_t s_carry;
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    const uint32_t entries = 2u * grid_h;
    for (uint32_t e0 = 0; e0 < entries; e0 += SR_BLOCK) {
        const uint32_t e = e0 + threadIdx.x;
        uint32_t c = 0;
        if (e < entries) {
            const uint32_t y = e >> 1, lo = (e & 1u) ? W.lo1 : W.lo0, hi = (e & 1u) ? W.hi1 : W.hi0;
            if (lo < hi) c = cs[y * grid_w + hi] - cs[y * grid_w + lo];
        }
        uint32_t inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if ((int)lane >= o) inc += t; }
        if (lane == 63u) s_wave[w] = inc;
        __syncthreads();
        uint32_t off = s_carry, tot = 0;
#pragma unroll
        for (uint32_t k = 0; k < SR_BLOCK / 64; ++k) { const uint32_t t = s_wave[k]; if (k < w) off += t; tot += t; }
        if (e < entries) rowbase[e] = off + inc - c;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        strip_counters[1] = s_carry;                 // slots filled from the main array
        strip_counters[2] = s_carry + R2;            // slots in use once the received records sit behind them
    }
}

__global__ __launch_bounds__(SL_BLOCK) void k_strip_gather(uint32_t grid_w, uint32_t grid_h, uint32_t ncell, StripWin W,
                                                           uint32_t strip_cap, const uint32_t* __restrict__ cs,
                                                           const uint32_t* __restrict__ rowbase, const u64* __restrict__ pairs,
                                                           const float2* __restrict__ pos_s, const float2* __restrict__ vel_s,
                                                           float2* __restrict__ sp_pos, float2* __restrict__ sp_vel,
                                                           u64* __restrict__ kt, uint32_t* __restrict__ hist,
                                                           uint32_t* __restrict__ back, unsigned long long* __restrict__ safe,
                                                           uint32_t* __restrict__ counters) {
    {   // the strip's safe-operand words (k_cs_fixreorder clears the unsafe bits)
        const uint32_t words = (strip_cap + 63u) / 64u;
        for (uint32_t t = blockIdx.x * SL_BLOCK + threadIdx.x; t < words; t += gridDim.x * SL_BLOCK) safe[t] = ~0ull;
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t e = blockIdx.x * (SL_BLOCK / 64) + (threadIdx.x >> 6);       // wave-uniform
    if (e >= 2u * grid_h) return;
    const uint32_t y = e >> 1, lo = (e & 1u) ? W.lo1 : W.lo0, hi = (e & 1u) ? W.hi1 : W.hi0;
    if (lo >= hi) return;
    const uint32_t a = cs[y * grid_w + lo], c = cs[y * grid_w + hi] - a, base = rowbase[e];
    for (uint32_t k0 = 0; k0 < c; k0 += 64u) {
        const uint32_t k = k0 + lane;
        const bool active = k < c && base + k < strip_cap;
        if (k < c && !active) atomicAdd(&counters[3], 1u);           // strip capacity exceeded (never: it equals the main array's)
        uint32_t key = 0;
        if (active) key = (uint32_t)(pairs[a + k] >> 32);
        const uint32_t kc = key < ncell ? key : ncell - 1u;
        const WaveRun r = wave_run(kc, active);
        uint32_t t = 0;
        if (r.is_head) t = atomicAdd(&hist[kc], r.length);
        t = __shfl(t, r.head_lane);
        if (active) {
            const uint32_t slot = base + k;
            kt[slot] = ((u64)key << 32) | (u64)(t + r.offset);
            sp_pos[slot] = pos_s[a + k];
            sp_vel[slot] = vel_s[a + k];
            back[slot] = a + k;                      // the main arrays' sorted index: where the force pass writes
        }
    }
}

// Received records -> strip slots [n_sm + j]; `P` is the MAIN array's StepParams (same window, same keys).
__global__ __launch_bounds__(SL_BLOCK) void k_strip_unpack(StepParams P, uint32_t main_slots, uint32_t R, uint32_t strip_cap,
                                                           const SlabHeader* __restrict__ hdr_left,
                                                           const float4* __restrict__ rec_left,
                                                           const SlabHeader* __restrict__ hdr_right,
                                                           const float4* __restrict__ rec_right,
                                                           float2* __restrict__ sp_pos, float2* __restrict__ sp_vel,
                                                           u64* __restrict__ kt, uint32_t* __restrict__ hist,
                                                           uint32_t* __restrict__ back,
                                                           const uint32_t* __restrict__ strip_counters,
                                                           uint32_t* __restrict__ counters) {
    const uint32_t j = blockIdx.x * SL_BLOCK + threadIdx.x;
    const bool in_range = j < 2u * R;
    const bool right = j >= R;
    const uint32_t jj = right ? j - R : j;
    const SlabHeader* hdr = right ? hdr_right : hdr_left;
    const float4* rec = right ? rec_right : rec_left;
    uint32_t cnt = 0;
    if (in_range && hdr) { cnt = hdr->count < R ? hdr->count : R; if (jj == 0 && hdr->overflow) atomicAdd(&counters[3], 1u); }
    const uint32_t slot = strip_counters[1] + j;
    const bool room = slot < strip_cap;
    uint32_t key = FS_DEAD_KEY, dst = STRIP_NONE;
    if (in_range && jj < cnt) {
        if (!room) {
            atomicAdd(&counters[3], 1u);
        } else {
            const float4 r = rec[jj];
            const float2 p = make_float2(r.x, r.y), v = make_float2(r.z, r.w);
            sp_pos[slot] = p;
            sp_vel[slot] = v;
            uint32_t cxg;
            key = slab_key(P, predict_pos(P, p, v), &cxg);
            if (key == FS_DEAD_KEY) atomicAdd(&counters[2], 1u);           // travelled farther than slab + halo
            // a migrant that lands in my FAR halo zone would have been needed by my other neighbour too
            if (!right && cxg + 2u >= P.own_hi && cxg < P.own_hi) atomicAdd(&counters[4], 1u);
            if (right && cxg < P.own_lo + 2u && cxg >= P.own_lo) atomicAdd(&counters[4], 1u);
            // ... and one that lands within 2 columns of the interior (or in it) was not seen by the interior launch, which
            // ran while this message was in flight: the boundary zone was too narrow for its speed (fs_slab_set_boundary_cols)
            if (key != FS_DEAD_KEY && P.adv_lo < P.adv_hi) {
                if (!right && cxg + 2u >= P.adv_lo) atomicAdd(&counters[4], 1u);
                if (right && cxg < P.adv_hi + 2u) atomicAdd(&counters[4], 1u);
            }
            if (key != FS_DEAD_KEY && cxg >= P.own_lo && cxg < P.own_hi) dst = main_slots + j;   // a migrant: mine from now on
        }
    }
    const bool active = key != FS_DEAD_KEY;
    const uint32_t k = key < P.ncell ? key : P.ncell - 1u;
    const WaveRun r = wave_run(k, active);
    uint32_t base = 0;
    if (r.is_head) base = atomicAdd(&hist[k], r.length);
    base = __shfl(base, r.head_lane);
    if (in_range && room) {
        kt[slot] = ((u64)key << 32) | (u64)(active ? base + r.offset : 0u);
        back[slot] = dst;
    }
}

// Results of the strip's force pass -> the main arrays.  `P` = the strip's StepParams (adv_outside = 1).
__global__ __launch_bounds__(SL_BLOCK) void k_strip_writeback(StepParams P, uint32_t main_slots, const u64* __restrict__ sp_pairs,
                                                              const uint32_t* __restrict__ back,
                                                              const float2* __restrict__ sp_pos_out,
                                                              const float2* __restrict__ sp_vel_out,
                                                              const float2* __restrict__ sp_pred, const float* __restrict__ sp_rho,
                                                              float2* __restrict__ pos, float2* __restrict__ vel,
                                                              float2* __restrict__ pred, float* __restrict__ rho,
                                                              uint32_t* __restrict__ key, unsigned char* __restrict__ owned,
                                                              uint32_t* __restrict__ counters) {
    const uint32_t i = blockIdx.x * SL_BLOCK + threadIdx.x;
    if (i >= *P.n_live) return;
    const u64 pr = sp_pairs[i];
    const uint32_t k = (uint32_t)(pr >> 32), dst = back[(uint32_t)pr];
    if (dst == STRIP_NONE || k == FS_DEAD_KEY) return;     // a ghost record
    uint32_t cxl, cy;
    key_to_local(P, k, &cxl, &cy);
    const int32_t cg = (int32_t)cxl + P.col_origin;
    if (!slab_advances(P, cg)) {
        // interior context (advanced by the interior launch) — or a migrant that landed beyond the boundary zone: nobody
        // advanced it, it is lost (k_strip_unpack has counted it in far_halo already)
        if (dst >= main_slots) atomicAdd(&counters[2], 1u);
        return;
    }
    pos[dst] = sp_pos_out[i];
    vel[dst] = sp_vel_out[i];
    rho[dst] = sp_rho[i];
    if (dst >= main_slots) {                               // a migrant: the rest of its record, and it is carried over from now on
        pred[dst] = sp_pred[i];
        key[dst] = k;
        owned[dst] = 1;
    }
}

// ------------------------------------------------------------------ launchers
static inline uint32_t nb(uint32_t n) { return (n + SL_BLOCK - 1) / SL_BLOCK; }

void launch_slab_maxspeed(hipStream_t st, const uint32_t* n_live, const float2* vel, const unsigned char* owned,
                          uint32_t* out_bits, uint32_t migr_first, uint32_t migr_count) {
    hipLaunchKernelGGL(k_slab_maxspeed, dim3(1024), dim3(SL_BLOCK), 0, st, n_live, vel, owned, out_bits, migr_first, migr_count);
}

// offsets + headers, then the parallel gather.  blockcnt holds 2 * (blocks + 1) entries: counts, then offsets.
static void launch_slab_msg(hipStream_t st, uint32_t blocks, uint32_t groups, uint32_t R, void* blockcnt, const uint32_t* stage_l,
                            const uint32_t* stage_r, const float2* pos, const float2* vel, void* state, uint32_t epoch,
                            SlabHeader* hl, SlabHeader* hr, uint32_t* counters, const StepParams& P, const uint32_t* cs_edge = nullptr,
                            uint32_t edge_grid = 0) {
    uint2* cnt = (uint2*)blockcnt;
    uint2* off = cnt + (blocks + 1u);
    hipLaunchKernelGGL(k_slab_msg, dim3(groups), dim3(64), 0, st, blocks, R, (const uint2*)cnt, off, (u64*)state, counters + 6, epoch,
                       hl, hr, counters, P, cs_edge);
    hipLaunchKernelGGL(k_slab_gather, dim3(cs_edge ? edge_grid : blocks), dim3(SL_BLOCK), 0, st, R, (const uint2*)cnt, (const uint2*)off,
                       stage_l, stage_r, pos, vel, hl ? (float4*)(hl + 1) : nullptr, hr ? (float4*)(hr + 1) : nullptr, P, cs_edge);
}

void launch_slab_pack(hipStream_t st, const StepParams& P, uint32_t main_slots, uint32_t R, int has_left,
                      int has_right, const float2* pos, const float2* vel, const unsigned char* owned, u64* out,
                      uint32_t* hist, void* blockcnt, uint32_t* stage /* 2 x capacity words */, void* state, uint32_t epoch,
                      void* msg_left, void* msg_right, uint32_t* counters, uint32_t* gap_counter, unsigned long long* safe,
                      bool counting, bool overlap, bool lists, const uint32_t* key_prev, uint32_t prev_adv_lo, uint32_t prev_adv_hi,
                      bool skip_edge) {
    // covers ALL slots (P.n = capacity): slots past `main_slots` only check for stranded owned particles
    const uint32_t cap = P.n > main_slots ? P.n : main_slots;
    const uint32_t blocks = nb(cap), groups = (blocks + MSG_GROUP - 1u) / MSG_GROUP;
    uint32_t* stage_l = stage;
    uint32_t* stage_r = stage + (size_t)blocks * SL_BLOCK;
    SlabHeader* hl = (SlabHeader*)msg_left;
    SlabHeader* hr = (SlabHeader*)msg_right;
    if (counting)
        hipLaunchKernelGGL(k_slab_pack<true>, dim3(blocks), dim3(SL_BLOCK), 0, st, P, cap, main_slots, overlap ? 1 : 0, lists ? 1 : 0, has_left, has_right, pos,
                           vel, owned, out, hist, (uint2*)blockcnt, stage_l, stage_r, counters, gap_counter, safe, key_prev, prev_adv_lo, prev_adv_hi, skip_edge ? 1 : 0);
    else
        hipLaunchKernelGGL(k_slab_pack<false>, dim3(blocks), dim3(SL_BLOCK), 0, st, P, cap, main_slots, 0, lists ? 1 : 0, has_left, has_right, pos,
                           vel, owned, out, hist, (uint2*)blockcnt, stage_l, stage_r, counters, gap_counter, safe, key_prev, prev_adv_lo, prev_adv_hi, skip_edge ? 1 : 0);
    if ((!hl && !hr) || !lists) return;                                 // no neighbour / messages pre-built: nothing to send
    launch_slab_msg(st, blocks, groups, R, blockcnt, stage_l, stage_r, pos, vel, state, epoch, hl, hr, counters, P);
}
// words of `stage` and of the look-back state launch_slab_pack needs for `cap` slots
size_t slab_stage_words(uint32_t cap) { return 2 * (size_t)nb(cap) * SL_BLOCK; }
size_t slab_msg_groups(uint32_t cap) { return (nb(cap) + MSG_GROUP - 1u) / MSG_GROUP; }

void launch_slab_prepack(hipStream_t st, const StepParams& P_next, uint32_t cap, uint32_t R, int has_left, int has_right,
                         const float2* pos, const float2* vel, const unsigned char* owned, const uint32_t* key_s, void* blockcnt,
                         uint32_t* stage, void* state, uint32_t epoch, void* msg_left, void* msg_right, uint32_t* counters,
                         const uint32_t* cs, uint32_t edge_grid, bool classify, bool counting, uint32_t main_slots, u64* out,
                         uint32_t* hist) {
    const uint32_t blocks = nb(cap), groups = (blocks + MSG_GROUP - 1u) / MSG_GROUP;
    uint32_t* stage_l = stage;
    uint32_t* stage_r = stage + (size_t)blocks * SL_BLOCK;
    SlabHeader* hl = (SlabHeader*)msg_left;
    SlabHeader* hr = (SlabHeader*)msg_right;
    if (!hl && !hr) return;
    const bool walk = edge_grid != 0 && P_next.transposed;
    hipLaunchKernelGGL(k_slab_prepack, dim3(walk ? edge_grid : blocks), dim3(SL_BLOCK), 0, st, P_next, has_left, has_right, walk ? 1 : 0,
                       pos, vel, owned, key_s, cs, (uint2*)blockcnt, stage_l, stage_r, classify ? 1 : 0, counting ? 1 : 0, main_slots, out,
                       hist, counters);
    launch_slab_msg(st, blocks, groups, R, blockcnt, stage_l, stage_r, pos, vel, state, epoch, hl, hr, counters, P_next,
                    walk ? cs : nullptr, edge_grid);
}

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

void launch_slab_colhist(hipStream_t st, const StepParams& P, const uint32_t* cs, uint32_t* hist_global, uint32_t migr_first,
                         uint32_t migr_count, const unsigned char* owned, const uint32_t* key) {
    hipLaunchKernelGGL(k_slab_colhist, dim3(nb(P.grid_w)), dim3(SL_BLOCK), 0, st, P, cs, hist_global);
    if (migr_count)
        hipLaunchKernelGGL(k_slab_colhist_migrants, dim3(nb(migr_count)), dim3(SL_BLOCK), 0, st, P, migr_first, migr_count, owned, key,
                           hist_global);
}

size_t slab_message_bytes(uint32_t R) { return sizeof(SlabHeader) + (size_t)R * sizeof(float4); }

// ---- overlapped step: the boundary strips
void launch_strip_gather(hipStream_t st, const StepParams& P, const uint32_t win[4], uint32_t R, uint32_t strip_cap,
                         const uint32_t* cs, uint32_t* rowbase, const u64* pairs, const float2* pos_s, const float2* vel_s,
                         float2* sp_pos, float2* sp_vel, u64* kt, uint32_t* hist, uint32_t* back, unsigned long long* safe,
                         uint32_t* strip_counters, uint32_t* counters) {
    const StripWin W{win[0], win[1], win[2], win[3]};
    hipLaunchKernelGGL(k_strip_rows, dim3(1), dim3(SR_BLOCK), 0, st, P.grid_w, P.grid_h, W, 2u * R, cs, rowbase, strip_counters);
    const uint32_t waves = 2u * P.grid_h, per_block = SL_BLOCK / 64;
    hipLaunchKernelGGL(k_strip_gather, dim3((waves + per_block - 1) / per_block), dim3(SL_BLOCK), 0, st, P.grid_w, P.grid_h, P.ncell,
                       W, strip_cap, cs, rowbase, pairs, pos_s, vel_s, sp_pos, sp_vel, kt, hist, back, safe, counters);
}

void launch_strip_unpack(hipStream_t st, const StepParams& P, uint32_t main_slots, uint32_t R, uint32_t strip_cap,
                         const void* msg_left, const void* msg_right, float2* sp_pos, float2* sp_vel, u64* kt, uint32_t* hist,
                         uint32_t* back, const uint32_t* strip_counters, uint32_t* counters) {
    const SlabHeader* hl = (const SlabHeader*)msg_left;
    const SlabHeader* hr = (const SlabHeader*)msg_right;
    hipLaunchKernelGGL(k_strip_unpack, dim3(nb(2 * R)), dim3(SL_BLOCK), 0, st, P, main_slots, R, strip_cap, hl,
                       hl ? (const float4*)(hl + 1) : nullptr, hr, hr ? (const float4*)(hr + 1) : nullptr, sp_pos, sp_vel, kt, hist,
                       back, strip_counters, counters);
}

void launch_strip_writeback(hipStream_t st, const StepParams& P_strip, uint32_t main_slots, uint32_t strip_cap, const u64* sp_pairs,
                            const uint32_t* back, const float2* sp_pos_out, const float2* sp_vel_out, const float2* sp_pred,
                            const float* sp_rho, float2* pos, float2* vel, float2* pred, float* rho, uint32_t* key,
                            unsigned char* owned, uint32_t* counters) {
    hipLaunchKernelGGL(k_strip_writeback, dim3(nb(strip_cap)), dim3(SL_BLOCK), 0, st, P_strip, main_slots, sp_pairs, back, sp_pos_out,
                       sp_vel_out, sp_pred, sp_rho, pos, vel, pred, rho, key, owned, counters);
}

}  // namespace fsd

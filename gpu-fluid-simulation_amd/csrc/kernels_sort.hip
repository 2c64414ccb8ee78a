// kernels_sort.hip — the reference's bitonic network (sort.wgsl:27-51, schedule
// src/simulation.rs:323-347) on 8-byte (key<<32 | source index) pairs.
//
// The reference issues S(S+1)/2 full-array dispatches over 32-byte records.  The network is
// data-oblivious and compares keys only (strict `>`, so equal keys never swap), hence sorting
// pairs and gathering the payload afterwards yields the bit-identical arrangement.  Structure:
//   * k_bitonic_local<INIT,KEYGEN>: predict + key (compute.wgsl:8-42) fused into the tile load, then
//                                   stages 0..11 entirely inside a 4096-pair tile (registers + LDS);
//   * k_bitonic_strided<M,FLIP>:    up to M steps of a later stage whose partners lie in different
//                                   tiles, in registers (one HBM/MALL pass per M steps);
//   * k_bitonic_local<TAIL>:        the last 12 steps of a stage, inside a tile.
// Provable no-ops are skipped (per-tile dirty flags, ordered-chunk certificates): see the
// comments at k_bitonic_local and k_bitonic_strided.
// Elements at index >= n do not exist in the reference (`if index_high >= num_values return`,
// sort.wgsl:39-41); here they hold a sentinel pair that can never swap (its key is the u32
// maximum and the compare is strict), which is equivalent.
#include <stdlib.h>
#include <string.h>

#include "fs_device.h"
#include "fs_kernels.h"

namespace fsd {

#define SORT_LOG_T 12
#define SORT_T (1u << SORT_LOG_T)
#define SORT_THREADS 256
#define FS_TILE_WIDE 2u      // dirty[tile]: the packed first kernel left this tile to the 64-bit one (k_bitonic_local32)

// ------------------------------------------------------------------ tile-local kernels
// Register-blocked: 2^(12-GB) threads x E = 2^GB elements.  The 12 index bits of a tile are split in groups of GB;
// a thread holds the E elements that differ in ONE group, so GB consecutive steps run in VGPRs, and the tile is
// re-distributed through LDS between groups (instead of one LDS round trip per step).  Layout of group g (in-thread
// bits [g GB, (g+1) GB), B = g GB):   idx = (t >> B) << (B + GB) | r << B | t & (2^B - 1)
// — the top group is also the coalesced global layout (idx = r << (12-GB) | t), group 0 holds E contiguous elements.
//   GB = 4: 256 threads x 16 elements, three groups — for sorts of more than 512 tiles;
//   GB = 3: 512 threads x  8 elements, four groups: twice the waves per tile (the tile's LDS footprint bounds the
//           occupancy: four tiles per CU) for a third more LDS round trips.  Measured at 16M: first kernel 179 -> 190 us,
//           tails equal, stage 12 42 -> 38 us, sort 0.435 -> 0.445 ms.  Few tiles cannot fill the chip (1M particles: 256
//           tiles on 256 CUs) and there the shorter per-thread chains win: sort_gb().
// LDS addresses are padded (lt_pad) so that the 8-byte accesses of all layouts are bank-conflict free, or 2-way at
// worst (64 x 4-B banks; GB = 3: chosen by enumeration over the layouts and their mirrored reads).
// The mirror step of stage s is done as in k_bitonic_strided: rows with bit s set are read
// from idx ^ (2^s - 1), after which it is a plain distance-2^s step and the remaining steps
// of that round compare in reversed order on those rows.
// Elements per thread of the tile kernels, chosen per sort from the tile count (both forms are compiled): FS_SORT_GB in
// the environment pins one.
static int sort_gb(uint32_t tiles) {
    static const int env = [] { const char* e = getenv("FS_SORT_GB"); return e ? atoi(e) : 0; }();
    static const uint32_t small = [] { const char* e = getenv("FS_SORT_GB3_TILES"); return e ? (uint32_t)atoi(e) : 512u; }();
    if (env == 3 || env == 4) return env;
    // few tiles cannot fill the chip: shorter per-thread chains win there.  Sort pass, ms, 8 / 16 elements per thread:
    // 1M (256 tiles) 0.066 / 0.078, 2M 0.094 / 0.098, 4M 0.142 / 0.136, 8M 0.253 / 0.250, 16M 0.445 / 0.435
    return tiles <= small ? 3 : 4;
}
template <int GB> struct LT {
    static constexpr int E = 1 << GB;                     // elements per thread
    static constexpr int THREADS = (int)SORT_T >> GB;
    static constexpr int TOPB = SORT_LOG_T - GB;          // bit position of the top group
    static constexpr int NG = SORT_LOG_T / GB;            // groups
    static constexpr int LDS = GB == 4 ? (int)SORT_T + ((int)SORT_T >> 4) : 4384;
};

template <int GB>
__device__ __forceinline__ uint32_t lt_pad(uint32_t idx) {
    if (GB == 4) return idx + (idx >> 4);
    return idx + ((idx >> 5) << 1) + (idx >> 7);          // max 4380
}

template <int GB, int B>
__device__ __forceinline__ uint32_t lt_idx(uint32_t r, uint32_t t) {
    return ((t >> B) << (B + GB)) | (r << B) | (t & ((1u << B) - 1u));
}

__device__ __forceinline__ void lt_cx(u64& lo, u64& hi) {     // lo = physically lower element
    if ((uint32_t)(lo >> 32) > (uint32_t)(hi >> 32)) { const u64 t = lo; lo = hi; hi = t; }
}
// Packed form of the first kernel (k_bitonic_local32): one 32-bit word per element, (key - tile_min) << 12 | position in
// the tile.  key(a) > key(b)  <=>  a > (b | 0xFFF): with equal keys a <= key << 12 | 0xFFF, with key(a) > key(b)
// a >= (key(b) + 1) << 12.  Equal keys never swap, exactly as in the 64-bit form: 4 VALU instead of 5, half the LDS.
__device__ __forceinline__ void lt_cx(uint32_t& lo, uint32_t& hi) {
    if (lo > (hi | 0xFFFu)) { const uint32_t t = lo; lo = hi; hi = t; }
}

// Steps on in-thread bits TOP..0 of a group.  FLIP: the step on bit TOP is a stage's mirror step.
template <int GB, int TOP, bool FLIP, class T>
__device__ __forceinline__ void lt_round(T (&x)[1 << GB]) {
#pragma unroll
    for (int b = TOP; b >= 0; --b) {
#pragma unroll
        for (int r = 0; r < (1 << GB); ++r) {
            if (r & (1 << b)) continue;
            const int r1 = r | (1 << b);
            if (FLIP && b < TOP && ((r >> TOP) & 1)) lt_cx(x[r1], x[r]);   // reversed rows (see header)
            else lt_cx(x[r], x[r1]);
        }
    }
}

template <int GB, int B, int TOP, bool FLIP, class T>
__device__ __forceinline__ void lt_read(const T* s, T (&x)[1 << GB], uint32_t t) {
#pragma unroll
    for (int r = 0; r < (1 << GB); ++r) {
        uint32_t idx = lt_idx<GB, B>((uint32_t)r, t);
        if (FLIP && ((r >> TOP) & 1)) idx ^= (1u << (B + TOP)) - 1u;
        x[r] = s[lt_pad<GB>(idx)];
    }
}

template <int GB, int B, int TOP, bool FLIP, class T>
__device__ __forceinline__ void lt_write(T* s, const T (&x)[1 << GB], uint32_t t) {
#pragma unroll
    for (int r = 0; r < (1 << GB); ++r) {
        uint32_t idx = lt_idx<GB, B>((uint32_t)r, t);
        if (FLIP && ((r >> TOP) & 1)) idx ^= (1u << (B + TOP)) - 1u;
        s[lt_pad<GB>(idx)] = x[r];
    }
}

// A re-distribution that involves group G exchanges data between the 2^(G GB) threads that share t >> (G GB).  Up to 64
// of them that is one wave: a wave's LDS instructions execute in program order, so no workgroup barrier is needed
// there — only a compiler-level fence.
__device__ __forceinline__ void lt_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int GB, int G>
__device__ __forceinline__ void lt_sync() {
    if (G * GB > 6) __syncthreads();
    else lt_wave_sync();
}

// From group G (just written to LDS in its layout) down to group 0: the remaining plain steps of a stage or tail.
template <int GB, int G, class T>
__device__ __forceinline__ void lt_descend(T* s, T (&x)[1 << GB], uint32_t t) {
    if constexpr (G > 0) {
        lt_sync<GB, G>();
        lt_read<GB, (G - 1) * GB, GB - 1, false>(s, x, t);
        lt_round<GB, GB - 1, false>(x);
        if constexpr (G - 1 > 0) {
            lt_write<GB, (G - 1) * GB, GB - 1, false>(s, x, t);
            lt_descend<GB, G - 1>(s, x, t);
        }
    }
}

// Stage S (0..11) of the network inside a tile; on entry and exit the tile is in the group-0 layout, in registers.
template <int GB, int S, class T>
__device__ __forceinline__ void lt_stage(T* s, T (&x)[1 << GB], uint32_t t) {
    constexpr int G = S / GB, TOP = S % GB;
    if constexpr (G == 0) {
#pragma unroll
        for (int r = 0; r < (1 << GB); ++r) {
            if (r & (1 << S)) continue;
            lt_cx(x[r], x[r ^ ((2 << S) - 1)]);          // mirror inside the 2^(S+1) block
        }
        if constexpr (S > 0) lt_round<GB, (S > 0 ? S - 1 : 0), false>(x);
    } else {
        lt_write<GB, 0, GB - 1, false>(s, x, t);
        lt_sync<GB, G>();                 // group 0 -> group G (mirrored reads stay inside the 2^(S+1) block: same threads)
        lt_read<GB, G * GB, TOP, true>(s, x, t);
        lt_round<GB, TOP, true>(x);
        lt_write<GB, G * GB, TOP, true>(s, x, t);
        lt_descend<GB, G>(s, x, t);
    }
}

// The tile leaves the network in the group-0 layout (E contiguous elements per thread): stored from there, a wave's
// store instruction touches 64 different lines, 16 bytes each.  One more trip through LDS puts it into the top
// layout, whose stores are 512 contiguous bytes per wave instruction.  (A thread's group-0 positions are its own: no
// barrier before the write; the top-layout reads cross waves: one barrier after it.)
template <int GB>
__device__ __forceinline__ void lt_store(u64* __restrict__ pairs, u64* s, u64 (&x)[1 << GB], uint32_t base, uint32_t t,
                                         uint32_t n, bool in_lds = false) {
    if (!in_lds) lt_write<GB, 0, GB - 1, false>(s, x, t);
    __syncthreads();
    lt_read<GB, LT<GB>::TOPB, GB - 1, false>(s, x, t);
#pragma unroll
    for (int r = 0; r < (1 << GB); ++r) {
        const uint32_t j = ((uint32_t)r << LT<GB>::TOPB) | t;
        if (base + j < n) pairs[base + j] = x[r];
    }
}

// Late-stage plans (see k_late_cert): `*gate` holds the certificate's verdict; a launch runs when it lies in [lo, hi].
__device__ __forceinline__ bool gate_closed(const uint32_t* gate, uint32_t lo, uint32_t hi) {
    if (!gate) return false;
    const uint32_t v = *gate;
    return v < lo || v > hi;
}

// The twelve plain steps of a tail on a tile already in registers (top layout as held by thread `t1`: the caller may
// hold the tile mirrored, see k_bitonic_stage12), ending in the group-0 layout of the real thread.
template <int GB>
__device__ __forceinline__ void lt_tail_regs(u64* s, u64 (&x)[1 << GB], uint32_t t1, uint32_t t) {
    lt_round<GB, GB - 1, false>(x);
    lt_write<GB, LT<GB>::TOPB, GB - 1, false>(s, x, t1);
    lt_descend<GB, LT<GB>::NG - 1>(s, x, t);
}

// tail of a stage >= 12: plain steps on bits 11..0 of one tile; the top layout IS the coalesced global layout
template <int GB>
__device__ __forceinline__ void lt_tail(const u64* __restrict__ pairs, uint32_t n, uint32_t base, u64* s, u64 (&x)[1 << GB],
                                        uint32_t t) {
#pragma unroll
    for (int r = 0; r < (1 << GB); ++r) {
        const uint32_t j = ((uint32_t)r << LT<GB>::TOPB) | t;
        x[r] = (base + j < n) ? pairs[base + j] : ~0ull;
    }
    lt_tail_regs<GB>(s, x, t, t);
}

// `dirty[tile]` != 0 when a strided pass of the current stage swapped an element of the tile.
// A clean tile is still sorted (it was left sorted by the previous stage's tail / the init
// pass), so every compare of its tail is lower-index <= higher-index: a no-op.  Skipping it
// is therefore exact, not an approximation.
// KEYGEN (2D engine): the init pass also IS predict_next_position + create_spatial_lookup
// (compute.wgsl:8-42): it reads pos/vel and builds the (key, index) pairs on the fly instead of
// reading them — one launch and one write+read of the pair array less per step.
// KEYGEN: 0 = the pairs exist, 1 = 2D (StepParams, float2 pos / vel), 2 = 3D (KeyGen3 in the first words of the
// StepParams argument, float4 pos / vel: the expressions of sim3d.hip predict3 / cell3, bit for bit).
__device__ __forceinline__ u64 keygen3(const KeyGen3& K, const float4* __restrict__ pos, const float4* __restrict__ vel, uint32_t i) {
    const float4 p = pos[i], v = vel[i];
    float rx = p.x + v.x * K.dt, ry = p.y + v.y * K.dt, rz = p.z + v.z * K.dt;
    if (fabsf(rx) > K.bx) rx = K.bx * sign_f32(rx);
    if (fabsf(ry) > K.by) ry = K.by * sign_f32(ry);
    if (fabsf(rz) > K.bz) rz = K.bz * sign_f32(rz);
    const uint32_t cx = f32_to_u32_sat(floorf(__fdiv_rn(rx + K.bx, K.h))) + 1u;
    const uint32_t cy = f32_to_u32_sat(floorf(__fdiv_rn(ry + K.by, K.h))) + 1u;
    const uint32_t cz = f32_to_u32_sat(floorf(__fdiv_rn(rz + K.bz, K.h))) + 1u;
    return ((u64)((cz * K.gh + cy) * K.gw + cx) << 32) | (u64)i;
}
template <bool INIT, int KEYGEN, int GB>
__global__ __launch_bounds__(LT<GB>::THREADS) void k_bitonic_local(u64* __restrict__ pairs, uint32_t n,
                                                                   uint32_t num_stages, uint32_t* __restrict__ dirty,
                                                                   StepParams P, const float2* __restrict__ pos,
                                                                   const float2* __restrict__ vel,
                                                                   uint32_t* __restrict__ gap_counter,
                                                                   const uint32_t* __restrict__ gate = nullptr,
                                                                   uint32_t gate_lo = 0, uint32_t gate_hi = 0) {
    constexpr int E = LT<GB>::E;
    __shared__ u64 s[LT<GB>::LDS];
    const uint32_t base = blockIdx.x * SORT_T;
    const uint32_t t = threadIdx.x;
    if (!INIT) {
        if (gate_closed(gate, gate_lo, gate_hi)) return;   // uniform: this launch belongs to the other late-stage plan
        if (dirty[blockIdx.x] == 0) return;            // uniform: whole tile provably unchanged
    }
    u64 x[E];
    if (INIT) {
        // gate_lo == 2 with no gate (launch_bitonic_sort): this launch only serves the tiles the packed kernel
        // (k_bitonic_local32) could not take — their flag is FS_TILE_WIDE; anything else returns at once
        if (gate_lo == FS_TILE_WIDE && dirty[blockIdx.x] != FS_TILE_WIDE) return;
        if (KEYGEN != 0 && blockIdx.x == 0 && t == 0) *gap_counter = 0;      // consumed by k_reorder later in the stream
        // coalesced load, straight into LDS, then the group-0 view
#pragma unroll
        for (int r = 0; r < E; ++r) {
            const uint32_t j = ((uint32_t)r << LT<GB>::TOPB) | t;
            u64 v = ~0ull;
            if (base + j < n) {
                if (KEYGEN == 1) {
                    const uint32_t i = base + j;
                    v = ((u64)cell_of_point(P, predict_pos(P, pos[i], vel[i])) << 32) | (u64)i;
                } else if (KEYGEN == 2) {
                    v = keygen3(*reinterpret_cast<const KeyGen3*>(&P), reinterpret_cast<const float4*>(pos),
                                reinterpret_cast<const float4*>(vel), base + j);
                } else {
                    v = pairs[base + j];
                }
            }
            s[lt_pad<GB>(j)] = v;
        }
        __syncthreads();
        lt_read<GB, 0, GB - 1, false>(s, x, t);
        {   // A tile whose keys are already in order passes through the network unchanged: every compare-exchange of
            // an ascending network tests key[lower index] > key[higher index], which never holds (strict compare, so
            // equal keys stay put as well).  While the fluid still moves as a lattice whole steps change no key at
            // all, and then this kernel is key generation, one check and a store (steps 2-7, 9-11, 13-19 of the
            // 16M dam break: 190 -> ~55 us).
            int ok = 1;
#pragma unroll
            for (int r = 0; r + 1 < E; ++r) ok &= (uint32_t)(x[r] >> 32) <= (uint32_t)(x[r + 1] >> 32);
            if (t + 1u < (uint32_t)LT<GB>::THREADS) ok &= (uint32_t)(x[E - 1] >> 32) <= (uint32_t)(s[lt_pad<GB>((t + 1u) << GB)] >> 32);
            if (__syncthreads_and(ok)) {
                lt_store<GB>(pairs, s, x, base, t, n, true);   // the tile is still in LDS at its natural positions
                if (t == 0) dirty[blockIdx.x] = 0;
                return;
            }
        }
        lt_stage<GB, 0>(s, x, t);
        if (num_stages > 1) lt_stage<GB, 1>(s, x, t);
        if (num_stages > 2) lt_stage<GB, 2>(s, x, t);
        if (num_stages > 3) lt_stage<GB, 3>(s, x, t);
        if (num_stages > 4) lt_stage<GB, 4>(s, x, t);
        if (num_stages > 5) lt_stage<GB, 5>(s, x, t);
        if (num_stages > 6) lt_stage<GB, 6>(s, x, t);
        if (num_stages > 7) lt_stage<GB, 7>(s, x, t);
        if (num_stages > 8) lt_stage<GB, 8>(s, x, t);
        if (num_stages > 9) lt_stage<GB, 9>(s, x, t);
        if (num_stages > 10) lt_stage<GB, 10>(s, x, t);
        if (num_stages > 11) lt_stage<GB, 11>(s, x, t);
    } else {
        lt_tail<GB>(pairs, n, base, s, x, t);
    }
    lt_store<GB>(pairs, s, x, base, t, n);
    if (t == 0) dirty[blockIdx.x] = 0;                 // sorted again
}

// ---- the first kernel in PACKED form --------------------------------------------------------------------------------
// Stages 0..11 of a tile only ever compare keys and move (key, index) pairs INSIDE the tile, and the index is
// base + position: one 32-bit word (key - tile_min) << 12 | position carries the same information whenever the keys of
// the tile span less than 2^20 — always, for a state that was in cell order one step ago (a tile of 4096 particles covers
// ~1000 cells plus at most a few row ends; an uploaded, shuffled state does not, see FS_TILE_WIDE).  Half the LDS per
// tile (17.4 KB: 8 tiles per CU instead of 4 — the round-2 kernel sat at 3 waves per SIMD with its load, network and
// store phases adding up instead of overlapping), half the registers, half the LDS traffic, and a compare-exchange of
// 4 instructions instead of 5 (lt_cx).  The pairs are rebuilt at the store.  Same network, same strict compare on
// keys only: the arrangement — ties included — is bit for bit the 64-bit kernel's (tests/test_sort_gpu.py).
// Tiles whose keys span 2^20 or more are left untouched and flagged FS_TILE_WIDE in `dirty`; the 64-bit kernel follows
// in the stream and takes exactly those (an idle launch otherwise).
#ifndef FS_SORT32_WAVES
#define FS_SORT32_WAVES 0      // > 0: pin the register budget to that many waves per SIMD (A/B: tools/ab_variant.py)
#endif
#if FS_SORT32_WAVES > 0
#define FS_SORT32_ATTR __attribute__((amdgpu_waves_per_eu(FS_SORT32_WAVES, FS_SORT32_WAVES)))
#else
#define FS_SORT32_ATTR
#endif
template <int KEYGEN, int GB>
__global__ __launch_bounds__(LT<GB>::THREADS) FS_SORT32_ATTR void k_bitonic_local32(u64* __restrict__ pairs, uint32_t n,
                                                                     uint32_t num_stages, uint32_t* __restrict__ dirty,
                                                                     StepParams P, const float2* __restrict__ pos,
                                                                     const float2* __restrict__ vel,
                                                                     uint32_t* __restrict__ gap_counter, uint32_t wide_word) {
    constexpr int E = LT<GB>::E;
    __shared__ uint32_t s[LT<GB>::LDS];
    __shared__ uint32_t s_mm[2 * (LT<GB>::THREADS / 64)];
    const uint32_t base = blockIdx.x * SORT_T;
    const uint32_t t = threadIdx.x;
    if (KEYGEN != 0 && blockIdx.x == 0 && t == 0) *gap_counter = 0;      // consumed by k_reorder later in the stream
    uint32_t key[E];
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int r = 0; r < E; ++r) {                                        // coalesced: position j = r << TOPB | t
        const uint32_t j = ((uint32_t)r << LT<GB>::TOPB) | t;
        uint32_t k = 0xFFFFFFFFu;                                        // padding sorts last (as ~0ull does in the 64-bit form)
        if (base + j < n) {
            if (KEYGEN == 1) k = cell_of_point(P, predict_pos(P, pos[base + j], vel[base + j]));
            else if (KEYGEN == 2) k = (uint32_t)(keygen3(*reinterpret_cast<const KeyGen3*>(&P), reinterpret_cast<const float4*>(pos),
                                                         reinterpret_cast<const float4*>(vel), base + j) >> 32);
            kmin = k < kmin ? k : kmin;
            kmax = k > kmax ? k : kmax;
        }
        key[r] = k;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t a = __shfl_xor(kmin, o), b = __shfl_xor(kmax, o);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    if ((t & 63u) == 0) { s_mm[2 * (t >> 6)] = kmin; s_mm[2 * (t >> 6) + 1] = kmax; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < LT<GB>::THREADS / 64; ++w) {
        const uint32_t a = s_mm[2 * w], b = s_mm[2 * w + 1];
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    static_assert(KEYGEN == 1 || KEYGEN == 2, "the packed form builds the pairs itself: index = base + position");
    if (kmax - kmin >= (1u << 20) - 1u) {                                // uniform; 0xFFFFF is reserved for the padding
        if (t == 0) { dirty[blockIdx.x] = FS_TILE_WIDE; atomicAdd(&dirty[wide_word], 1u); }   // (counted: fs_sort_plan_info.wide_tiles)
        return;
    }
    uint32_t x[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t j = ((uint32_t)r << LT<GB>::TOPB) | t;
        const uint32_t rel = key[r] == 0xFFFFFFFFu ? 0xFFFFFu : key[r] - kmin;
        s[lt_pad<GB>(j)] = (rel << 12) | j;
    }
    __syncthreads();
    lt_read<GB, 0, GB - 1, false>(s, x, t);
    bool sorted_already;
    {   // a tile whose keys are already in order passes through the network unchanged (see k_bitonic_local)
        int ok = 1;
#pragma unroll
        for (int r = 0; r + 1 < E; ++r) ok &= (x[r] >> 12) <= (x[r + 1] >> 12);
        if (t + 1u < (uint32_t)LT<GB>::THREADS) ok &= (x[E - 1] >> 12) <= (s[lt_pad<GB>((t + 1u) << GB)] >> 12);
        sorted_already = __syncthreads_and(ok) != 0;
    }
    if (!sorted_already) {
        lt_stage<GB, 0>(s, x, t);
        if (num_stages > 1) lt_stage<GB, 1>(s, x, t);
        if (num_stages > 2) lt_stage<GB, 2>(s, x, t);
        if (num_stages > 3) lt_stage<GB, 3>(s, x, t);
        if (num_stages > 4) lt_stage<GB, 4>(s, x, t);
        if (num_stages > 5) lt_stage<GB, 5>(s, x, t);
        if (num_stages > 6) lt_stage<GB, 6>(s, x, t);
        if (num_stages > 7) lt_stage<GB, 7>(s, x, t);
        if (num_stages > 8) lt_stage<GB, 8>(s, x, t);
        if (num_stages > 9) lt_stage<GB, 9>(s, x, t);
        if (num_stages > 10) lt_stage<GB, 10>(s, x, t);
        if (num_stages > 11) lt_stage<GB, 11>(s, x, t);
        lt_write<GB, 0, GB - 1, false>(s, x, t);           // back to LDS at the natural positions
    }
    __syncthreads();
    lt_read<GB, LT<GB>::TOPB, GB - 1, false>(s, x, t);     // the coalesced layout: 512 contiguous bytes per wave store
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t j = ((uint32_t)r << LT<GB>::TOPB) | t;
        if (base + j < n) {
            pairs[base + j] = ((u64)(kmin + (x[r] >> 12)) << 32) | (u64)(base + (x[r] & 0xFFFu));
        }
    }
    if (t == 0) dirty[blockIdx.x] = 0;
}

// Stage 12 in ONE kernel: its only global step is the mirror step between the two tiles of an 8192-block, so a
// workgroup takes both tiles: A in the top layout, B read back to front (thread t holds B[4095 - (r << TOPB | t)],
// still a coalesced load) — the mirror partners then sit in the same register slot of the same thread.  After the
// compare-exchanges A's tail runs from the registers; B's registers, renamed r -> E-1 - r, ARE the top layout of thread
// THREADS-1 - t, so its tail only writes its first LDS round with that thread id.  Saves the strided pass (one read +
// write of the pair array) and a launch.  Certificate: last(A) <= first(B) (both tiles are sorted on entry) => no
// compare of the stage can swap; otherwise the pair (last(A), first(B)) itself swaps and both tails are needed.
template <int GB>
__global__ __launch_bounds__(LT<GB>::THREADS) void k_bitonic_stage12(u64* __restrict__ pairs, uint32_t n) {
    constexpr int E = LT<GB>::E;
    __shared__ u64 s[LT<GB>::LDS];
    const uint32_t t = threadIdx.x;
    const uint32_t base_a = blockIdx.x * (2u * SORT_T), base_b = base_a + SORT_T;
    if (base_b >= n) return;                           // B holds sentinels only: nothing can swap
    {
        const uint32_t last_a = (uint32_t)(pairs[base_b - 1u] >> 32), first_b = (uint32_t)(pairs[base_b] >> 32);
        if (last_a <= first_b) return;                 // uniform
    }
    u64 xa[E], xb[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint32_t j = ((uint32_t)r << LT<GB>::TOPB) | t;
        xa[r] = pairs[base_a + j];                     // base_b < n: A is complete
        const uint32_t pb = base_b + (SORT_T - 1u - j);
        xb[r] = pb < n ? pairs[pb] : ~0ull;
    }
#pragma unroll
    for (int r = 0; r < E; ++r) lt_cx(xa[r], xb[r]);               // A[j] vs B[4095 - j]: the stage's mirror step
    lt_tail_regs<GB>(s, xa, t, t);
    lt_store<GB>(pairs, s, xa, base_a, t, n);
    u64 xn[E];
#pragma unroll
    for (int r = 0; r < E; ++r) xn[r] = xb[E - 1 - r];             // natural order of thread THREADS-1 - t
    __syncthreads();                                   // A's last LDS reads are done
    lt_tail_regs<GB>(s, xn, (uint32_t)LT<GB>::THREADS - 1u - t, t);
    lt_store<GB>(pairs, s, xn, base_b, t, n);
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
//
// Exact skipping (try_skip): a workgroup covers 256 consecutive columns of its 2^M rows; each
// row's 256-element chunk lies in one tile.  If all those tiles are clean (sorted) a chunk's
// keys are bounded by its first and last element, and if the chunks are ordered
// last(chunk) <= first(next chunk) in PHYSICAL index order then every compare-exchange of
// the pass has key[lower index] <= key[higher index]: no swap can happen and the workgroup
// returns after reading 2 elements per row instead of the whole 2^M x 256 block.
// First / last key of row l (PHYSICAL order) of the 256-column chunk `chunk`, for the no-op certificate: a dirty
// row can never certify (its "range" is everything).
template <int M, bool FLIP>
__device__ __forceinline__ void strided_cert_row(const u64* __restrict__ pairs, uint32_t n, uint32_t a,
                                                 const uint32_t* __restrict__ dirty, uint32_t chunk, uint32_t l,
                                                 uint32_t* first, uint32_t* last) {
    constexpr int R = 1 << M;
    const uint32_t low = a - (uint32_t)M + 1u;
    const uint32_t mirror = (1u << a) - 1u;
    // row l in PHYSICAL order: lower half as is; with FLIP the upper half is mirrored, so its
    // rows appear in reverse order and each chunk is read back to front
    const bool upper = FLIP && (l >> (M - 1));
    const uint32_t rv = upper ? (uint32_t)(R - 1) - (l - (uint32_t)(R / 2)) : l;   // virtual row
    const uint32_t g0 = chunk * 256u, g1 = g0 + 255u;
    const uint32_t v0 = (((g0 >> low) << (a + 1u)) | (g0 & ((1u << low) - 1u))) | (rv << low);
    const uint32_t v1 = (((g1 >> low) << (a + 1u)) | (g1 & ((1u << low) - 1u))) | (rv << low);
    const uint32_t pf = upper ? (v1 ^ mirror) : v0;      // physically first / last element of the chunk
    const uint32_t pl = upper ? (v0 ^ mirror) : v1;
    const bool clean = pf >= n || dirty[pf >> SORT_LOG_T] == 0;   // past the end: sentinels, in order by definition
    const uint32_t kf = pf < n ? (uint32_t)(pairs[pf] >> 32) : 0xFFFFFFFFu;
    const uint32_t kl = pl < n ? (uint32_t)(pairs[pl] >> 32) : 0xFFFFFFFFu;
    *first = clean ? kf : 0u;
    *last = clean ? kl : 0xFFFFFFFFu;
}

template <int M, bool FLIP>
__device__ __forceinline__ void strided_body(u64* __restrict__ pairs, uint32_t n, uint32_t a, uint32_t num_threads,
                                             uint32_t* __restrict__ dirty, uint32_t g);

template <int M, bool FLIP>
__global__ __launch_bounds__(256) void k_bitonic_strided(u64* __restrict__ pairs, uint32_t n, uint32_t a,
                                                         uint32_t num_threads, uint32_t* __restrict__ dirty,
                                                         int try_skip, const uint32_t* __restrict__ gate,
                                                         uint32_t gate_lo, uint32_t gate_hi) {
    if (gate_closed(gate, gate_lo, gate_hi)) return;   // uniform: this launch belongs to the other late-stage plan
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    constexpr int R = 1 << M;
    if (try_skip) {                                    // uniform branch (kernel argument)
        __shared__ uint32_t s_first[R], s_last[R];
        __shared__ int s_skip;
        const uint32_t l = threadIdx.x;
        if (l < (uint32_t)R) strided_cert_row<M, FLIP>(pairs, n, a, dirty, blockIdx.x, l, &s_first[l], &s_last[l]);
        __syncthreads();
        if (l == 0) {
            int ok = 1;
#pragma unroll
            for (int r = 0; r < R; ++r) ok &= (s_first[r] <= s_last[r]);
#pragma unroll
            for (int r = 0; r + 1 < R; ++r) ok &= (s_last[r] <= s_first[r + 1]);
            s_skip = ok;
        }
        __syncthreads();
        if (s_skip) return;
    }
    strided_body<M, FLIP>(pairs, n, a, num_threads, dirty, g);
}

template <int M, bool FLIP>
__device__ __forceinline__ void strided_body(u64* __restrict__ pairs, uint32_t n, uint32_t a, uint32_t num_threads,
                                             uint32_t* __restrict__ dirty, uint32_t g) {
    constexpr int R = 1 << M;
    const uint32_t low = a - (uint32_t)M + 1u;
    const uint32_t mirror = (1u << a) - 1u;
    if (g >= num_threads) return;
    const uint32_t vbase = ((g >> low) << (a + 1u)) | (g & ((1u << low) - 1u));
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
        if ((changed >> r) & 1ull) {                 // a sentinel (p >= n) never swaps, so p < n here
            pairs[p] = x[r];
            dirty[p >> SORT_LOG_T] = 1u;              // this tile's tail must run
        }
    }
}

// The same pass in batch form, for the persistent stand-by kernel (k_late_fallback): a workgroup checks the
// certificates of K consecutive chunks in one round of loads (K * 2^M lanes, one row each), then runs the body on the
// chunks that failed.  Exact for any input (a chunk's body touches its own elements only).  As launches of their own
// the batch form and a tile-walking tail were measured slower than the plain kernels (profiles/r02_d_rejected.md).
template <int M>
struct StridedBatch { static constexpr int R = 1 << M; static constexpr int K = (256 / R) < 8 ? (256 / R) : 8; };

// One batch: the certificates of chunks c0 .. c0+K-1 in one round of loads, then the bodies of the chunks that failed.
// Ends with a barrier (the shared words are reusable on return).
template <int M, bool FLIP>
__device__ __forceinline__ void strided_batch(u64* pairs, uint32_t n, uint32_t a, uint32_t num_threads, uint32_t* dirty,
                                              uint32_t c0, uint32_t* s_first, uint32_t* s_last, uint32_t* s_active) {
    constexpr int R = StridedBatch<M>::R, K = StridedBatch<M>::K;
    const uint32_t nchunks = num_threads >> 8;          // whole 256-column chunks (launcher: num_threads >= 256, a power of two)
    const uint32_t l = threadIdx.x;
    if (l == 0) *s_active = 0;
    if (l < (uint32_t)(K * R) && c0 + l / (uint32_t)R < nchunks)
        strided_cert_row<M, FLIP>(pairs, n, a, dirty, c0 + l / (uint32_t)R, l % (uint32_t)R, &s_first[l], &s_last[l]);
    __syncthreads();
    if (l < (uint32_t)K && c0 + l < nchunks) {
        int ok = 1;
#pragma unroll
        for (int r = 0; r < R; ++r) ok &= (s_first[l * R + r] <= s_last[l * R + r]);
#pragma unroll
        for (int r = 0; r + 1 < R; ++r) ok &= (s_last[l * R + r] <= s_first[l * R + r + 1]);
        if (!ok) atomicOr(s_active, 1u << l);
    }
    __syncthreads();
    uint32_t act = *s_active;                           // uniform
    while (act) {
        const uint32_t k = (uint32_t)__builtin_ctz(act);
        act &= act - 1u;
        strided_body<M, FLIP>(pairs, n, a, num_threads, dirty, (c0 + k) * 256u + l);
    }
    __syncthreads();
}

template <int M>
static void launch_strided(hipStream_t st, u64* pairs, uint32_t n, uint32_t a, bool flip, uint32_t p2,
                           uint32_t* dirty, int try_skip, const uint32_t* gate = nullptr, uint32_t glo = 0, uint32_t ghi = 0) {
    const uint32_t threads = p2 >> M;
    const dim3 grid((threads + 255u) / 256u), block(256);
    if (threads < 256u) try_skip = 0;                  // the certificate assumes full 256-column workgroups
    if (flip) hipLaunchKernelGGL((k_bitonic_strided<M, true>), grid, block, 0, st, pairs, n, a, threads, dirty, try_skip, gate, glo, ghi);
    else hipLaunchKernelGGL((k_bitonic_strided<M, false>), grid, block, 0, st, pairs, n, a, threads, dirty, try_skip, gate, glo, ghi);
}

static int sort_mmax() {
    static int m = [] {
        const char* e = getenv("FS_SORT_MMAX");
        int v = e ? atoi(e) : 4;   // measured over the bench window @16M: 2: 0.90 ms, 3: 0.74, 4: 0.706, 5: 0.718, 6: 0.79
        return v < 1 ? 1 : (v > 6 ? 6 : v);
    }();
    return m;
}

// Late stages (2^stage far beyond the distance a particle's key moves in one step) are almost entirely certified
// no-ops: their cost is the launch count, so they take more steps per pass.
static int sort_mmax_late() {
    static int m = [] { const char* e = getenv("FS_SORT_MMAX_LATE"); int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > 6 ? 6 : v); }();
    return m;
}
static int sort_late_stage() {
    static int m = [] { const char* e = getenv("FS_SORT_LATE_STAGE"); return e ? atoi(e) : 18; }();
    return m;
}

static int sort_skip_stage() {
    static int m = [] { const char* e = getenv("FS_SORT_SKIP_STAGE"); return e ? atoi(e) : 12; }();
    return m;   // first stage whose strided passes try the no-op certificate (<0: never)
}

// ---------------------------------------------------------------- late stages in one shifted merge
// After stage S0-1 the array is a sequence of sorted blocks of 2^S0 = 2H elements.  Between two consecutive steps
// of the simulation a particle's key moves by a few grid rows at most, so what the remaining stages S0 .. S-1 still
// have to do is confined to a neighbourhood of the block boundaries m = b 2^S0.  If, for every boundary,
//     (C2) key[m - H - 1] <= key[m]          (the part of the left block outside the window is below the right block)
//     (C3) key[m + H]     >= key[m - 1]      (the part of the right block outside the window is above the left block)
//     (C4) key[m - 1]     <= key[m + 2H]     (the left block is below the block after the next boundary)
// then, with the windows W_b = [m - H, m + H):
//   * the elements outside all windows are in non-decreasing order over the whole array and bound every window
//     from below / above; any two windows are ordered as sets (max W_b <= min W_b+1).  By induction over the
//     network's compare-exchanges no pair with an end outside a window, or with ends in two windows, ever swaps
//     (key[lower index] <= key[higher index] holds for it; the compare is strict, ties never swap);
//   * the pairs of stages >= S0 with BOTH ends in W_b are: the mirror pairs (m-1-i, m+i) of the one stage whose
//     block centre m is (m = odd * 2^stage), and the plain steps of distance <= H/2 inside the two halves (an aligned
//     pair of distance >= H straddles no window: m is a multiple of 2H).  The halves are sorted, so the plain
//     steps are no-ops before that stage; its mirror + plain steps are a bitonic merge of the halves; afterwards the
//     window is sorted and later plain steps are no-ops again.
// Hence stages S0 .. S-1 together equal ONE merge of every window — which is stage S0-1 of the same network run on
// the array shifted by H elements (tile aligned, H >= 4096): the kernels above, a pointer offset, 2-3 launches
// instead of 3-4 per remaining stage.  k_late_cert evaluates (C2)-(C4) on the device and publishes the verdict;
// the launches of both plans are in the stream and each returns at once unless the verdict names its plan, so the
// result is the network's in every case (uploads, fast flows: the conditions fail and the per-stage plan runs).
#define SORT_NO_PLAN 255u
// plan words (dirty[sort_plan_word(n) ..], 16 of them, zero at create; [8] failure bits / [9] ticket of k_late_cert's grid):
// [0] verdict, [1] / [2] plan counters, [3] fallback barrier, [4] fallback
// barrier time-outs, [5] fit class of the last certificate, [6] calls in which the stand-by kernel had work.
// Fit class: the largest j <= 3 for which (C2), (C3) still hold with windows of H / 2^j — how much room the moves of
// this step left; the host's choice of the next steps' stage reads it (engine.hip), never the result.
// feedback (optional, host-visible): [1] stage, [2] verdict, [3] fit class, [4] time-outs, then [0] = seq.
// Round 4: a grid of small workgroups instead of one of 256 threads.  The kernel's time was never its arithmetic: at 16 M
// particles it reads 512 x 11 keys 256 KB apart — every one a TLB miss, all of them queued on ONE compute unit's address
// translation (14.6 us, profiles/r03_window_5_25_kernels.txt).  Spread over the chip the misses are taken in parallel; the
// workgroups OR their failure bits into plan[8], and the last one to arrive (ticket plan[9]) publishes the verdict.
__global__ __launch_bounds__(256) void k_late_cert(const u64* __restrict__ pairs, uint32_t n, uint32_t p2, uint32_t s0,
                                                    uint32_t* __restrict__ plan, uint32_t* __restrict__ feedback,
                                                    uint32_t seq) {
    const uint32_t H = 1u << (s0 - 1u), nb = p2 >> s0;
    uint32_t bad = 0;                    // bit 0: (C2)-(C4) fail; bits 1..3: (C2), (C3) fail with windows of H/2, H/4, H/8
    // Eleven keys per boundary, all loaded before any is compared (clamped index, sentinel selected afterwards).
    for (uint32_t b = 1u + blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += gridDim.x * blockDim.x) {
        const uint32_t m = b << s0;
        const uint32_t idx[11] = {m - 1u, m, m - H - 1u, m + H, m + 2u * H, m - (H >> 1) - 1u, m + (H >> 1),
                                  m - (H >> 2) - 1u, m + (H >> 2), m - (H >> 3) - 1u, m + (H >> 3)};
        uint32_t k[11];
#pragma unroll
        for (int j = 0; j < 11; ++j) k[j] = (uint32_t)(pairs[idx[j] < n ? idx[j] : n - 1u] >> 32);
#pragma unroll
        for (int j = 0; j < 11; ++j) k[j] = idx[j] < n ? k[j] : 0xFFFFFFFFu;      // m + 2H == p2 reads as the sentinel
        const uint32_t left_max = k[0], right_min = k[1];
        if (!(k[2] <= right_min && k[3] >= left_max && left_max <= k[4])) bad |= 1u;
        if (!(k[5] <= right_min && k[6] >= left_max)) bad |= 2u;
        if (!(k[7] <= right_min && k[8] >= left_max)) bad |= 4u;
        if (!(k[9] <= right_min && k[10] >= left_max)) bad |= 8u;
    }
    __shared__ uint32_t s_bad;
    if (threadIdx.x == 0) s_bad = 0u;
    __syncthreads();
    if (bad) atomicOr(&s_bad, bad);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_bad) atomicOr(&plan[8], s_bad);
        __threadfence();                               // the bits before the ticket
        if (atomicAdd(&plan[9], 1u) == gridDim.x - 1u) {
            __threadfence();
            const uint32_t bits = atomicExch(&plan[8], 0u);              // all workgroups' bits; both words ready for the next call
            plan[9] = 0u;
            const bool all = (bits & 1u) == 0u;
            const uint32_t verdict = all ? s0 : SORT_NO_PLAN;            // the first stage the shifted merge replaces, or none
            const uint32_t cls = !all ? 0u : !(bits & 8u) ? 3u : !(bits & 4u) ? 2u : !(bits & 2u) ? 1u : 0u;
            plan[0] = verdict;
            atomicAdd(&plan[all ? 1 : 2], 1u);         // diagnostics: calls that took the shifted / the per-stage plan
            plan[3] = 0;                                // the fallback kernel's barrier counter
            plan[5] = cls;
            if (feedback) {
                feedback[1] = s0; feedback[2] = verdict; feedback[3] = cls; feedback[4] = plan[4];
                __threadfence_system();
                feedback[0] = seq;
            }
        }
    }
}

// All workgroups of the grid have arrived `target / gridDim.x` times.  Release / acquire at agent scope around the
// counter make the passes' plain stores visible across workgroups (other XCDs' L2 included).  The spin is bounded:
// should the workgroups not all be resident (they are: the grid is far smaller than the chip) the kernel still ends,
// and the time-out is counted where the host reads it.
__device__ __forceinline__ void fallback_barrier(uint32_t* plan, uint32_t target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&plan[3], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        while (__hip_atomic_load(&plan[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 22)) { atomicAdd(&plan[4], 1u); break; }
        }
    }
    __syncthreads();
}

template <int M>
__device__ __forceinline__ void fallback_pass(bool flip, u64* pairs, uint32_t n, uint32_t a, uint32_t p2, uint32_t* dirty,
                                              uint32_t* s_first, uint32_t* s_last, uint32_t* s_active) {
    const uint32_t threads = p2 >> M, nbatch = ((threads >> 8) + StridedBatch<M>::K - 1) / StridedBatch<M>::K;
    for (uint32_t b = blockIdx.x; b < nbatch; b += gridDim.x) {
        if (flip) strided_batch<M, true>(pairs, n, a, threads, dirty, b * StridedBatch<M>::K, s_first, s_last, s_active);
        else strided_batch<M, false>(pairs, n, a, threads, dirty, b * StridedBatch<M>::K, s_first, s_last, s_active);
    }
}

// The per-stage plan for stages s0 .. S-1 in ONE launch, for the steps whose certificate fails although the host
// expected it to hold (and therefore did not put the per-stage launches into the stream): a small persistent grid
// walks the passes in order with a grid barrier between them.  Rare (the host follows the fit class with a margin,
// sort_policy.h), correct for any input, several times slower than the per-stage launches when it has real work
// (16M: ~3.5 ms against 0.25 ms of per-stage launches: every barrier is an L2 write-back and invalidate).
__global__ __launch_bounds__(256) void k_late_fallback(u64* pairs, uint32_t n, uint32_t p2, uint32_t S, uint32_t s0,
                                                       uint32_t* dirty, uint32_t* plan, uint32_t inject_timeout) {
    if (plan[0] != SORT_NO_PLAN) return;               // uniform over the grid: the shifted merge did the work
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&plan[6], 1u);                       // diagnostics: calls this kernel had to work in
        // tests only: report a time-out that did not happen (the barriers still hold, the sort stays correct), so that
        // the host's reaction — fs_step fails with FS_ERR_DEVICE from then on — has a test (tests/test_sort_gpu.py)
        if (inject_timeout) atomicAdd(&plan[4], 1u);
    }
    __shared__ u64 s[LT<4>::LDS];                       // 256 threads: the 16-element form of the tile code
    __shared__ uint32_t s_first[256], s_last[256];
    __shared__ uint32_t s_active;
    const uint32_t tiles = (n + SORT_T - 1) / SORT_T, t = threadIdx.x;
    uint32_t phase = 0;
    for (uint32_t stage = s0; stage < S; ++stage) {
        const int gsteps = (int)(stage - SORT_LOG_T + 1);
        const int npass = (gsteps + 3) / 4;
        uint32_t a = stage;
        for (int ps = 0; ps < npass; ++ps) {
            const int m = gsteps / npass + (ps < gsteps % npass ? 1 : 0);
            switch (m) {
                case 1: fallback_pass<1>(ps == 0, pairs, n, a, p2, dirty, s_first, s_last, &s_active); break;
                case 2: fallback_pass<2>(ps == 0, pairs, n, a, p2, dirty, s_first, s_last, &s_active); break;
                case 3: fallback_pass<3>(ps == 0, pairs, n, a, p2, dirty, s_first, s_last, &s_active); break;
                default: fallback_pass<4>(ps == 0, pairs, n, a, p2, dirty, s_first, s_last, &s_active); break;
            }
            a -= (uint32_t)m;
            fallback_barrier(plan, ++phase * gridDim.x);
        }
        for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
            if (dirty[tile] == 0) continue;            // uniform
            const uint32_t base = tile * SORT_T;
            u64 x[LT<4>::E];
            lt_tail<4>(pairs, n, base, s, x, t);
            lt_store<4>(pairs, s, x, base, t, n);
            if (t == 0) dirty[tile] = 0;
            __syncthreads();                           // the LDS stage is reused
        }
        fallback_barrier(plan, ++phase * gridDim.x);
    }
}

// First stage handled by the shifted merge (0: never).  Default S - 6: windows of +-2^(S-7) elements, sixteen grid rows
// of the square dam-break scenes (a row holds 2 sqrt(n) particles).  Measured at 16M (S = 24), sort pass, ms:
//   steps 10-110: none 0.729, 16: 0.618, 17: 0.641, 18: 0.648, 19: 0.666;  steps 150-250 (dense floor, fuller rows):
//   none 0.802, 16 / 17: 0.82 (the certificate fails, per-stage plan + 3 idle launches), 18: 0.718, 19: 0.740.
static int sort_fuse_stage(uint32_t S, int request) {
    static int env = [] { const char* e = getenv("FS_SORT_FUSE_STAGE"); return e ? atoi(e) : -1; }();
    const int want = request >= 0 ? request : env;
    int s0 = want >= 0 ? want : (int)S - 6;
    if (want < 0 && s0 < SORT_LOG_T + 1) s0 = SORT_LOG_T + 1;
    if (s0 < SORT_LOG_T + 1 || s0 >= (int)S) return 0;          // H must be a whole number of tiles; something must be left
    return s0;
}

uint32_t sort_tile_count(uint32_t n);
uint32_t sort_plan_word(uint32_t n) { return sort_tile_count(n) - 16u; }

// One stage >= SORT_LOG_T of the network on `pairs[0 .. n)`: its strided passes, then the tile tails.
static int launch_stage(hipStream_t st, u64* pairs, uint32_t n, uint32_t p2, uint32_t stage, uint32_t* dirty, int mmax,
                        int try_skip, const uint32_t* gate, uint32_t glo, uint32_t ghi) {
    int launches = 0;
    const uint32_t tiles = (n + SORT_T - 1) / SORT_T;
    // steps whose block (2 << sh) exceeds the tile: sh = stage .. SORT_LOG_T, in passes of <= mmax steps
    const int gsteps = (int)(stage - SORT_LOG_T + 1);
    const int npass = (gsteps + mmax - 1) / mmax;
    uint32_t a = stage;
    for (int ps = 0; ps < npass; ++ps) {
        const int m = gsteps / npass + (ps < gsteps % npass ? 1 : 0);
        const bool flip = ps == 0;
        switch (m) {
            case 1: launch_strided<1>(st, pairs, n, a, flip, p2, dirty, try_skip, gate, glo, ghi); break;
            case 2: launch_strided<2>(st, pairs, n, a, flip, p2, dirty, try_skip, gate, glo, ghi); break;
            case 3: launch_strided<3>(st, pairs, n, a, flip, p2, dirty, try_skip, gate, glo, ghi); break;
            case 4: launch_strided<4>(st, pairs, n, a, flip, p2, dirty, try_skip, gate, glo, ghi); break;
            case 5: launch_strided<5>(st, pairs, n, a, flip, p2, dirty, try_skip, gate, glo, ghi); break;
            default: launch_strided<6>(st, pairs, n, a, flip, p2, dirty, try_skip, gate, glo, ghi); break;
        }
        a -= (uint32_t)m;
        ++launches;
    }
    StepParams P0;
    memset(&P0, 0, sizeof P0);
    if (sort_gb(tiles) == 3)
        hipLaunchKernelGGL((k_bitonic_local<false, 0, 3>), dim3(tiles), dim3(LT<3>::THREADS), 0, st, pairs, n, 0u, dirty, P0,
                           (const float2*)nullptr, (const float2*)nullptr, (uint32_t*)nullptr, gate, glo, ghi);
    else
        hipLaunchKernelGGL((k_bitonic_local<false, 0, 4>), dim3(tiles), dim3(LT<4>::THREADS), 0, st, pairs, n, 0u, dirty, P0,
                           (const float2*)nullptr, (const float2*)nullptr, (uint32_t*)nullptr, gate, glo, ghi);
    return launches + 1;
}

int launch_bitonic_sort(hipStream_t st, u64* pairs, uint32_t n, uint32_t* dirty, const StepParams* keygen,
                        const float2* pos, const float2* vel, uint32_t* gap_counter, const SortPlan* plan,
                        const KeyGen3* keygen3d, const float4* pos4, const float4* vel4) {
    const int fuse_stage = plan ? plan->fuse_stage : -1;
    const bool one_fallback = plan && plan->fallback == 1;
    if (n <= 1) return 0;
    uint32_t p2 = 1, S = 0;
    while (p2 < n) { p2 <<= 1; ++S; }
    const uint32_t tiles = (n + SORT_T - 1) / SORT_T;
    int launches = 0;
    const uint32_t init_stages = S < SORT_LOG_T ? S : SORT_LOG_T;
    StepParams P0;
    memset(&P0, 0, sizeof P0);
    const int gb = sort_gb(tiles);
#define FS_LAUNCH_INIT(KG, GBV, PP, POS, VEL, GC)                                                                      \
    hipLaunchKernelGGL((k_bitonic_local<true, KG, GBV>), dim3(tiles), dim3(LT<GBV>::THREADS), 0, st, pairs, n, init_stages, \
                       dirty, PP, POS, VEL, GC, (const uint32_t*)nullptr, 0u, 0u)
    // engines' steps (the pairs are built here): the packed kernel first, then the 64-bit kernel for the tiles it flagged
    // FS_TILE_WIDE (an idle launch in a running simulation).  FS_SORT_PACKED=0: the 64-bit kernel alone, as in round 2.
    static const bool packed = [] { const char* e = getenv("FS_SORT_PACKED"); return e ? atoi(e) != 0 : true; }();
#define FS_LAUNCH_INIT32(KG, GBV, PP, POS, VEL, GC)                                                                      \
    hipLaunchKernelGGL((k_bitonic_local32<KG, GBV>), dim3(tiles), dim3(LT<GBV>::THREADS), 0, st, pairs, n, init_stages,   \
                       dirty, PP, POS, VEL, GC, sort_plan_word(n) + 7u)
#define FS_LAUNCH_INIT_WIDE(KG, GBV, PP, POS, VEL, GC)                                                                   \
    hipLaunchKernelGGL((k_bitonic_local<true, KG, GBV>), dim3(tiles), dim3(LT<GBV>::THREADS), 0, st, pairs, n, init_stages, \
                       dirty, PP, POS, VEL, GC, (const uint32_t*)nullptr, FS_TILE_WIDE, 0u)
    if (keygen) {
        if (packed) {
            if (gb == 3) { FS_LAUNCH_INIT32(1, 3, *keygen, pos, vel, gap_counter); FS_LAUNCH_INIT_WIDE(1, 3, *keygen, pos, vel, gap_counter); }
            else { FS_LAUNCH_INIT32(1, 4, *keygen, pos, vel, gap_counter); FS_LAUNCH_INIT_WIDE(1, 4, *keygen, pos, vel, gap_counter); }
            ++launches;
        } else if (gb == 3) FS_LAUNCH_INIT(1, 3, *keygen, pos, vel, gap_counter);
        else FS_LAUNCH_INIT(1, 4, *keygen, pos, vel, gap_counter);
    }
    else if (keygen3d) {
        static_assert(sizeof(KeyGen3) <= sizeof(StepParams), "KeyGen3 rides in the StepParams argument");
        memcpy(&P0, keygen3d, sizeof(KeyGen3));
        if (packed) {
            if (gb == 3) { FS_LAUNCH_INIT32(2, 3, P0, (const float2*)pos4, (const float2*)vel4, gap_counter); FS_LAUNCH_INIT_WIDE(2, 3, P0, (const float2*)pos4, (const float2*)vel4, gap_counter); }
            else { FS_LAUNCH_INIT32(2, 4, P0, (const float2*)pos4, (const float2*)vel4, gap_counter); FS_LAUNCH_INIT_WIDE(2, 4, P0, (const float2*)pos4, (const float2*)vel4, gap_counter); }
            ++launches;
        } else if (gb == 3) FS_LAUNCH_INIT(2, 3, P0, (const float2*)pos4, (const float2*)vel4, gap_counter);
        else FS_LAUNCH_INIT(2, 4, P0, (const float2*)pos4, (const float2*)vel4, gap_counter);
        memset(&P0, 0, sizeof P0);
    }
#undef FS_LAUNCH_INIT32
#undef FS_LAUNCH_INIT_WIDE
    else if (gb == 3) FS_LAUNCH_INIT(0, 3, P0, (const float2*)nullptr, (const float2*)nullptr, (uint32_t*)nullptr);
    else FS_LAUNCH_INIT(0, 4, P0, (const float2*)nullptr, (const float2*)nullptr, (uint32_t*)nullptr);
#undef FS_LAUNCH_INIT
    ++launches;                                         // leaves every tile sorted and its flag cleared
    const int skip_from = sort_skip_stage();
    const int mmax_early = sort_mmax(), mmax_late = sort_mmax_late(), late_from = sort_late_stage();
    const uint32_t s0 = (uint32_t)sort_fuse_stage(S, fuse_stage);
    static const bool fused12 = [] { const char* e = getenv("FS_SORT_FUSED12"); return e ? atoi(e) != 0 : true; }();
    uint32_t* gate = dirty + sort_plan_word(n);           // the plan words (see k_late_cert)
    for (uint32_t stage = SORT_LOG_T; stage < S; ++stage) {
        const int mmax = (int)stage >= late_from ? mmax_late : mmax_early;
        const int ts = (skip_from >= 0 && (int)stage >= skip_from) ? 1 : 0;
        if (s0 && stage == s0) {
            // verdict, then the shifted merge (runs when the verdict is 1); stages s0 .. S-1 below run when it is 0
            const uint32_t H = 1u << (s0 - 1u);
            // one boundary per thread, 16-thread workgroups: the key reads' address translations spread over the chip
            static const uint32_t cert_block = [] { const char* e = getenv("FS_SORT_CERT_BLOCK"); int v = e ? atoi(e) : 16; return (uint32_t)(v < 1 ? 1 : v > 256 ? 256 : v); }();
            const uint32_t cert_nb = p2 >> s0;
            uint32_t cert_grid = (cert_nb + cert_block - 1u) / cert_block;
            if (cert_grid > 256u) cert_grid = 256u;
            if (cert_grid < 1u) cert_grid = 1u;
            hipLaunchKernelGGL(k_late_cert, dim3(cert_grid), dim3(cert_block), 0, st, pairs, n, p2, s0, gate,
                               plan ? plan->feedback : (uint32_t*)nullptr, plan ? plan->seq : 0u);
            ++launches;
            if (n > H)
                {
                    static const int mm = [] { const char* e = getenv("FS_SORT_MMAX_SHIFTED"); int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > 6 ? 6 : v); }();
                    launches += launch_stage(st, pairs + H, n - H, p2, s0 - 1u, dirty + (H >> SORT_LOG_T), mm, 1, gate, s0, s0);
                }
            if (one_fallback) {                        // everything the certificate may still ask for, in one launch
                // one workgroup per CU at most: all of them resident whatever else the kernel shares the chip with
                static const int fb_grid = [] {
                    const char* e = getenv("FS_SORT_FALLBACK_GRID");
                    int dev = 0, cus = 64;
                    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
                    int v = e ? atoi(e) : (cus > 128 ? 128 : cus);   // 16M, 16 working calls in 110: 64: sort 1.08 ms avg, 128: 1.01, 256: 1.12
                    return v < 1 ? 1 : (v > 256 ? 256 : v);
                }();
                hipLaunchKernelGGL(k_late_fallback, dim3(fb_grid), dim3(256), 0, st, pairs, n, p2, S, s0, dirty, gate,
                                   plan && plan->inject_timeout ? 1u : 0u);
                return launches + 1;
            }
        }
        // a stage at or after the verdict's is already done: these launches then return at once (~5 us each)
        const bool gated = s0 && stage >= s0;
        if (stage == SORT_LOG_T && fused12) {          // (never gated: s0 > SORT_LOG_T)
            if (gb == 3) hipLaunchKernelGGL((k_bitonic_stage12<3>), dim3((tiles + 1u) / 2u), dim3(LT<3>::THREADS), 0, st, pairs, n);
            else hipLaunchKernelGGL((k_bitonic_stage12<4>), dim3((tiles + 1u) / 2u), dim3(LT<4>::THREADS), 0, st, pairs, n);
            ++launches;
            continue;
        }
        launches += launch_stage(st, pairs, n, p2, stage, dirty, mmax, ts, gated ? gate : nullptr, stage + 1u, SORT_NO_PLAN);
    }
    return launches;
}

uint32_t sort_tile_count(uint32_t n) {
    uint32_t p2 = 1;
    while (p2 < n) p2 <<= 1;
    return (p2 + SORT_T - 1) / SORT_T + 1u + 16u;  // tiles of the padded array (sentinel tiles included) + the late-stage plan words
}

}  // namespace fsd

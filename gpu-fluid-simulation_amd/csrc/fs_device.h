// fs_device.h — parameters and device helpers shared by the HIP kernels.
//
// Float contract: every expression below is written in the association the
// reference shaders use and the TU is compiled with -ffp-contract=off, so with
// IEEE-correct +,-,*,/,sqrt (hipcc default: correctly rounded f32 divide/sqrt)
// the kernels reproduce the CPU oracle bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fsd {

typedef unsigned long long u64;

struct ConstDiv { float c, y; int32_t ok; };   // exact division by a constant, see div_const() below

// Per-tick parameters, passed by value (replaces the 120-byte uniform buffer,
// src/uniform.rs:93-95; contents follow src/simulation.rs:470-497).
struct StepParams {
    uint32_t n;            // particle_count
    uint32_t grid_w, grid_h, ncell;
    float dt;              // delta
    float h;               // smoothing_radius
    float sqr_radius;      // h*h
    float bounds_x, bounds_y;
    float bs_x, bs_y;      // bounds * 0.5
    float mass;
    float poly6_norm;      // 4/(PI*pow(h,8)) — funcs.wgsl:76, evaluated once on the host
    float pressure_k, rest_density;
    float damping, visc_coeff;
    float spiky, visc_k;   // spiky_kernel_derivative, viscosity_kernel (simulation.rs:489-490)
    float gx, gy;
    float mouse_x, mouse_y, mouse_radius, mouse_power;
    int32_t mouse_state;
    uint32_t frame_time;
    float tex_w, tex_h;
    uint32_t tex_w_u, tex_len;
    uint32_t xcd_chunk_log2;   // xcd_block(): blocks per chunk dealt to one XCD
    int32_t tex_zero;          // host knows the force field is all zeros: the lookup of compute.wgsl:127-140 is skipped
    int32_t ref_quirks;
    int32_t fast_math;         // FS_MATH_WGSL_ULP: native rcp/sqrt in the force pass (not bit-exact)
    ConstDiv div_2h3, div_h2;  // the two constant denominators of funcs.wgsl:119 (2h^3, h^2)
    ConstDiv div_h;            // the cell size of funcs.wgsl:212-214, proven over every numerator a clamped position can give
    int32_t share_div;         // force pass: one true division per denominator + div_by_rcp (bit-identical)
    int32_t pos_by_src;        // force pass: its `pos_s` argument is the PREVIOUS state in source order (own position =
                               // pos_s[pairs[i].src]) and pos_out is a different buffer — the reorder pass then writes no
                               // sorted copy of the positions (8 B / particle less in the HBM-bound k_reorder)
    // --- slab (multi-GPU) mode: the local grid is a window of global cell columns -------------
    int32_t col_origin;        // global column of local column 0 (0 on a single GPU)
    uint32_t own_lo, own_hi;   // owned window [own_lo, own_hi) in GLOBAL columns
    uint32_t grid_w_global;
    const uint32_t* n_live;    // device-side live count (slab mode); nullptr -> n
    // --- slab mode, overlapped step (engine.hip fs_slab_pack / fs_slab_step): which owned columns THIS launch of the force
    // pass advances.  adv_outside == 0: the columns [adv_lo, adv_hi) (the interior, computed while the halo messages are in
    // flight; the serial step passes the whole owned window); adv_outside == 1: the owned columns OUTSIDE [adv_lo, adv_hi)
    // (the boundary strips, computed after the unpack).  Global columns, own_lo <= adv_lo <= adv_hi <= own_hi.
    uint32_t adv_lo, adv_hi;
    int32_t adv_outside;
    // --- cell-id layout.  Ids are v * grid_u + u: `u` runs along the axis whose three neighbour cells are consecutive ids (one
    // contiguous index range per sweep "row"), `v` along the other.  Reference layout (funcs.wgsl:216-218): u = x, v = y.
    // transposed (slab ranks with neighbours): u = y, v = x (local column) — a rank's cell COLUMNS become contiguous index
    // ranges, so the columns next to a slab edge (the halo, the edge-first force launch, the next step's messages) are a few
    // hundred whole blocks at the two ends of the sorted array instead of a few lanes of every grid row.
    uint32_t grid_u, grid_v;
    int32_t transposed;
    // --- density -> force hand-off per 256-particle block (8 words each): the block-wide candidate ranges of the three sweep rows
    // [lo0, lo1, lo2, hi0, hi1, hi2] as block_tile_bounds() reduces them from the lanes' row ranges.  Both passes sweep the
    // same rows, so the force pass reads the density pass's result (eight scalar loads) instead of repeating the reduction
    // (ballots, readlanes, an LDS round trip and a barrier).  nullptr: every pass reduces for itself.
    uint32_t* block_bounds;
};

// Slab mode: does the force pass of this launch advance a particle whose GLOBAL cell column is cg?  (Ghosts — columns outside
// the owned window — are never advanced.)
__device__ __forceinline__ bool slab_advances(const StepParams& P, int32_t cg) {
    const bool in_adv = cg >= (int32_t)P.adv_lo && cg < (int32_t)P.adv_hi;
    const bool in_own = cg >= (int32_t)P.own_lo && cg < (int32_t)P.own_hi;
    return P.adv_outside ? (in_own && !in_adv) : in_adv;
}

#define FS_DEAD_KEY 0xFFFFFFFFu   // slot holds no particle (slab mode); sorts to the end

// WGSL f32 -> u32 conversion: saturating, NaN -> 0.
__device__ __forceinline__ uint32_t f32_to_u32_sat(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

// Correctly rounded f32 sqrt.  NOT __fsqrt_rn: on ROCm 7.2 that is __ocml_native_sqrt_f32
// (bare v_sqrt_f32, ~1 ulp).  sqrtf lowers to the IEEE sequence under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt.
__device__ __forceinline__ float sqrt_rn(float x) { return __builtin_sqrtf(x); }

__device__ __forceinline__ float sign_f32(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// compute.wgsl:16-26
__device__ __forceinline__ float2 predict_pos(const StepParams& P, float2 pos, float2 vel) {
    float2 pr;
    pr.x = pos.x + vel.x * P.dt;
    pr.y = pos.y + vel.y * P.dt;
    if (fabsf(pr.x) > P.bs_x) pr.x = P.bs_x * sign_f32(pr.x);
    if (fabsf(pr.y) > P.bs_y) pr.y = P.bs_y * sign_f32(pr.y);
    return pr;
}

// funcs.wgsl:212-214.  The division by the cell size is the 3-instruction exact form when the create-time enumeration
// proved it for every numerator in [2^-60, 4 max(bs_x, bs_y)] (engine.hip prove_force_constants): positions are clamped to
// the bounds, so the numerators are 0 .. 2 bs; below 2^-60 both forms floor to 0, a NaN gives cell 0 either way.
__device__ __forceinline__ float div_const_fast(float x, float c, float y);
__device__ __forceinline__ void xy_of_point(const StepParams& P, float2 pt, uint32_t* cx, uint32_t* cy) {
    const float nx = pt.x + P.bs_x, ny = pt.y + P.bs_y;
    const float fx = floorf(P.div_h.ok ? div_const_fast(nx, P.h, P.div_h.y) : __fdiv_rn(nx, P.h));
    const float fy = floorf(P.div_h.ok ? div_const_fast(ny, P.h, P.div_h.y) : __fdiv_rn(ny, P.h));
    *cx = f32_to_u32_sat(fx) + 1u;
    *cy = f32_to_u32_sat(fy) + 1u;
}

// funcs.wgsl:206-218
__device__ __forceinline__ uint32_t cell_of_point(const StepParams& P, float2 pt) {
    uint32_t cx, cy;
    xy_of_point(P, pt, &cx, &cy);
    return cy * P.grid_w + cx;
}

// Local cell coordinates in slab mode (col_origin == 0 on a single GPU: identity).
__device__ __forceinline__ void xy_local(const StepParams& P, float2 pt, uint32_t* cx, uint32_t* cy) {
    xy_of_point(P, pt, cx, cy);
    *cx = (uint32_t)((int32_t)*cx - P.col_origin);
}
// The same as (u, v) of the handle's cell-id layout (StepParams::transposed), plus the GLOBAL cell column.
__device__ __forceinline__ void uv_local(const StepParams& P, float2 pt, uint32_t* u, uint32_t* v, int32_t* col_global) {
    uint32_t cx, cy;
    xy_of_point(P, pt, &cx, &cy);
    *col_global = (int32_t)cx;
    cx = (uint32_t)((int32_t)cx - P.col_origin);
    *u = P.transposed ? cy : cx;
    *v = P.transposed ? cx : cy;
}
__device__ __forceinline__ uint32_t key_of_local(const StepParams& P, uint32_t cx_local, uint32_t cy) {
    return P.transposed ? cx_local * P.grid_u + cy : cy * P.grid_u + cx_local;
}
// (local column, row) of a cell id
__device__ __forceinline__ void key_to_local(const StepParams& P, uint32_t key, uint32_t* cx_local, uint32_t* cy) {
    const uint32_t v = key / P.grid_u, u = key - v * P.grid_u;
    *cx_local = P.transposed ? v : u;
    *cy = P.transposed ? u : v;
}

// funcs.wgsl:129-149
__device__ __forceinline__ float rand_f32(uint32_t* st) {
    uint32_t x = *st;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    *st = x;
    return __fdiv_rn((float)x, 4294967296.0f);
}

// ---- exact division by a loop-invariant constant ---------------------------------------------
// x / c  ==  fma(r, y, q0)  with  y = RN(1/c), q0 = RN(x*y), r = fma(-q0, c, x)   (Markstein-style
// correction; 3 VALU ops instead of the ~11-op / ~36-cycle IEEE sequence).  It is correctly
// rounded for *most* (c, x); whether it is for EVERY f32 x with the simulation's actual constant
// is decided by exhaustive enumeration on the GPU when the simulation is created
// (k_verify_constdiv: all 2^32 bit patterns).  Only constants that pass use this path.
__device__ __forceinline__ float div_const_fast(float x, float c, float y) {
    const float q0 = x * y;
    const float r = __builtin_fmaf(-q0, c, x);
    return __builtin_fmaf(r, y, q0);
}
__device__ __forceinline__ float div_const(const ConstDiv& K, float x) {
    return K.ok ? div_const_fast(x, K.c, K.y) : __fdiv_rn(x, K.c);     // K.ok is uniform (kernel argument)
}

// ---- exact quotients that share a denominator -------------------------------------------------
// With y == RN(1/b) (ONE true division), RN(a/b) == fma(fma(-q0, b, a), y, q0), q0 = RN(a*y), for
// every normal a, b as long as no intermediate leaves the normal range: power-of-two scaling is
// exact and RN is sign-symmetric, so the claim only depends on the two mantissas, and all 2^46
// mantissa pairs were enumerated on the GPU against hipcc's correctly rounded `/` — zero
// mismatches (tools/div_markstein.hip x; profiles/r01_f_div_by_rcp_exhaustive.txt).  The residual
// a - b*q0 is a multiple of 2^(ea-47), hence the callers' guards: 2^-20 <= b <= 2^20 and
// 2^-60 <= |a| <= 2^60 or a == 0.  For a == -0 the quotient is +0 instead of -0; every use adds the
// quotient (or a product with it) to an accumulator that starts at +0, where the sign of a zero
// term cannot change the sum.  Anything outside the guards takes the true division.
__device__ __forceinline__ float div_by_rcp(float a, float b, float y) {
    const float q0 = a * y;
    const float r = __builtin_fmaf(-q0, b, a);
    return __builtin_fmaf(r, y, q0);
}
// RN(1/b) and RN(sqrt(x)) without the scaling / fix-up selects of hipcc's general expansions.  These
// lean forms rest on the hardware approximation instructions, so the engine PROVES them when a
// handle is created by enumerating every f32 of the range on the GPU against `1.0f / b` and
// __builtin_sqrtf (k_verify_unary: 3.4e8 + 6.7e8 inputs, well under a millisecond); without the
// proof the force pass keeps its true divisions.
__device__ __forceinline__ float rcp_rn_fast(float b) {           // 2^-20 <= b <= 2^20
    const float y = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
__device__ __forceinline__ float sqrt_rn_fast(float x) {          // 2^-40 <= x <= 2^40
    const float s = __builtin_amdgcn_sqrtf(x);
    const float hlf = 0.5f * __builtin_amdgcn_rsqf(x);
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, hlf, s);
}
#define FS_RCP_LO 0x1p-20f
#define FS_RCP_HI 0x1p20f
#define FS_SQRT_LO 0x1p-40f
#define FS_SQRT_HI 0x1p40f

// Range predicates as WAVE masks (one v_cmp into an SGPR pair each, combined on the scalar unit):
// the caller only needs "does any active lane fall outside", never a per-lane flag.
typedef unsigned long long wave_mask;
__device__ __forceinline__ wave_mask wm(bool c) { return __builtin_amdgcn_ballot_w64(c); }
__device__ __forceinline__ wave_mask rcp_num_lo_ok(float a) { return wm(fabsf(a) >= 0x1p-60f) | wm(a == 0.0f); }       // NaN: 0
__device__ __forceinline__ wave_mask rcp_num_ok(float a) { return wm(fabsf(a) <= 0x1p60f) & rcp_num_lo_ok(a); }

// ---- per-particle "safe operand" classification ---------------------------------------------------------------
// div_by_rcp's numerator guards used to be evaluated per PAIR (18 compares).  Most of them follow from properties of
// the two particles alone, so they are decided once per particle per step (k_reorder: coordinates and velocity;
// k_density: density and pressure) and travel as the SIGN of the stored reciprocal density {rho, +-RN(1/rho)}:
//   * a component c is "lo-safe" when c == 0 or |c| >= 2^-53.  The difference of two lo-safe floats is 0 or has
//     magnitude >= 2^-76 (equal -> 0; one zero -> the other; else >= ulp(2^-53) = 2^-76): the numerators
//     q - me (positions) and v_j - v_i are inside [2^-76, .] or 0 whenever both particles are safe.  2^-76 is what
//     div_by_rcp needs with 2^-20 <= b <= 2^20: q0 = RN(a y) >= 2^-96, the residual a - b q0 is a multiple of
//     2^(ea-47) >= 2^-123, the quotient >= 2^-96 — every intermediate normal, which is all the mantissa-pair proof
//     assumes.  (Round 1 used 2^-60 / 2^-36 here; in the first steps of a lattice scene the x velocities are rounding
//     noise of 1e-12..1e-9, which the old threshold sent to the true-division path by the thousand.)
//   * |v| <= 2^59 per component  ->  |v_j - v_i| <= 2^60;
//   * rho <= 2^20 (the reciprocal's proven range; rho >= 0.1 by construction) and |k (rho - rho0)| <= 2^39  ->
//     |dir * kern * (P_i + P_j)/2| <= (1 + 2^-22) * h*spiky * 2^39 <= 2^60 given h * spiky <= 2^19 (checked on the host
//     before the shared-reciprocal path is enabled).
// What remains per pair: r2 >= 2^-40, both particles safe (two sign tests), and the lower bound of the two pressure
// numerators (a product of three factors can be tiny without any factor being unusual).
__device__ __forceinline__ bool lo_safe(float c) { return c == 0.0f || fabsf(c) >= 0x1p-53f; }          // NaN: false
__device__ __forceinline__ bool kin_safe(float2 pred, float2 vel) {
    return lo_safe(pred.x) && lo_safe(pred.y) && lo_safe(vel.x) && lo_safe(vel.y) && fabsf(vel.x) <= 0x1p59f &&
           fabsf(vel.y) <= 0x1p59f;
}
#define FS_PRESSURE_HI 0x1p39f
#define FS_HSPIKY_HI 0x1p19f

// ---- XCD-aware block index ---------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one; MI355X guide, "Workgroup
// dispatch").  A strip of particles and the strips one grid row above/below it (~64 blocks away)
// read the same neighbour rows, so each XCD is given ONE contiguous eighth of the strips: its
// private 4 MB L2 then holds the few rows it is working on instead of every XCD streaming all of
// them.  `nblocks` = blocks that hold live particles (in slab mode the grid covers the whole
// capacity; mapping over that would park the dead tail on the last XCDs and idle them).  The
// grid must have at least ceil(nblocks/8)*8 blocks; returns false for blocks with no work.
// Chunked form (P.xcd_chunk_log2 = c): the strips are cut into chunks of 2^c blocks dealt round-robin to the XCDs, and an
// XCD walks its chunks in order.  One contiguous eighth per XCD (c = large) minimises shared-row traffic but binds an
// XCD to one region of the fluid: as the column compresses, the XCD that owns the bottom rows has several times the
// neighbour pairs of the one that owns the surface and the launch waits for it.  Chunks of a few grid rows keep the
// row sharing (a chunk re-reads one row above and one below) and spread every depth over all eight XCDs.
// Grid: 8 * ceil(ceil(nblocks / 2^c) / 8) * 2^c blocks (xcd_grid()).
__device__ __forceinline__ bool xcd_block(const StepParams& P, uint32_t nblocks, uint32_t* logical) {
    const uint32_t c = P.xcd_chunk_log2;
    const uint32_t slot = blockIdx.x >> 3, xcd = blockIdx.x & 7u;
    const uint32_t chunk = ((slot >> c) << 3) | xcd;          // the XCD's (slot >> c)-th chunk
    const uint32_t lb = (chunk << c) | (slot & ((1u << c) - 1u));
    *logical = lb;
    return lb < nblocks;
}

// ---- edge-first slab step, column-major ids: the blocks that hold the edge columns ------------------------------------
// With StepParams::transposed a rank's cell columns are contiguous index ranges of the sorted array, so the columns left of the
// interior [adv_lo, adv_hi) are the blocks [0, eL) and the ones right of it the blocks [eR, nb).  The edge columns' kernels are
// launched with a SMALL fixed grid that walks exactly these blocks (the counts live in the device-side cell table): a launch
// over all of a rank's ~8 000 blocks, most of which return at once, still has to find a wave slot for each of them — beside
// the interior columns' kernels, which keep every slot busy, that alone took ~20 us per launch.
// `extra`: columns of the interior to include as well (the density launch needs one).
struct EdgeBlocks { uint32_t eL, eR, nb; };
__device__ __forceinline__ EdgeBlocks edge_blocks(const StepParams& P, const uint32_t* __restrict__ cs, uint32_t n, uint32_t extra) {
    EdgeBlocks E;
    E.nb = (n + 255u) / 256u;
    E.eL = 0u; E.eR = E.nb;
    if (P.adv_lo > P.own_lo) {
        uint32_t c = (uint32_t)((int32_t)P.adv_lo - P.col_origin) + extra;       // first local column not needed
        if (c > P.grid_v) c = P.grid_v;
        const uint32_t e = (cs[c * P.grid_u] + 255u) / 256u;
        E.eL = e < E.nb ? e : E.nb;
    }
    if (P.adv_hi < P.own_hi) {
        const uint32_t c = (uint32_t)((int32_t)P.adv_hi - P.col_origin) - extra;  // first local column needed
        const uint32_t b = cs[c * P.grid_u] / 256u;
        E.eR = b < E.nb ? b : E.nb;
    }
    if (E.eR < E.eL) E.eR = E.eL;                    // the two ends meet: every block once
    return E;
}
__device__ __forceinline__ uint32_t edge_block_count(const EdgeBlocks& E) { return E.eL + (E.nb - E.eR); }
__device__ __forceinline__ uint32_t edge_block_at(const EdgeBlocks& E, uint32_t t) { return t < E.eL ? t : E.eR + (t - E.eL); }
__device__ __forceinline__ bool edge_block_has(const EdgeBlocks& E, uint32_t b) { return b < E.eL || (b >= E.eR && b < E.nb); }

// ---- block neighbour tiles (shared by the 2D and 3D density / force kernels) ------------------
// A 256-thread workgroup owns 256 consecutive sorted particles (a strip of cells in one grid
// row), so the candidates of ALL its lanes for one sweep row form one short contiguous index
// range.  block_tile_bounds() reduces the per-lane [lo, hi) of three sweep rows to block-wide
// ranges and says whether they fit an LDS tile of `tile` entries each.
struct RowRanges { uint32_t lo[3], hi[3]; };

// Block-wide [min lo, max hi) per sweep row; returns true when all three fit the tile.
// WAVES: waves per workgroup (4 for the 256-thread kernels; 1 for the 3D density / force kernels, whose workgroup is one
// wave so that no barrier couples waves with different neighbour counts).
template <int WAVES = 4>
__device__ __forceinline__ bool block_tile_bounds(const RowRanges& R, uint32_t* s_red /*[24]*/, uint32_t* blo,
                                                  uint32_t* bhi, uint32_t tile) {
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        // Lanes hold consecutive sorted particles, so their cell ids — and with them cs[id_lo] and
        // cs[id_hi] of one sweep row — are non-decreasing in lane order (quirk_lo_fix keeps that:
        // lo_fix <= cs[first cell + 1] <= any non-zero start).  Min lo / max hi over the lanes that
        // have candidates are therefore the FIRST such lane's lo and the LAST one's hi: a ballot
        // and two readlanes instead of twelve shuffle steps.
        const bool has = R.lo[r] < R.hi[r];
        const unsigned long long hm = __builtin_amdgcn_ballot_w64(has);
        uint32_t mn = 0xFFFFFFFFu, mx = 0u;
        if (hm) {
            mn = (uint32_t)__builtin_amdgcn_readlane((int)R.lo[r], __builtin_ctzll(hm));
            mx = (uint32_t)__builtin_amdgcn_readlane((int)R.hi[r], 63 - __builtin_clzll(hm));
        }
        if (lane == 0) { s_red[(r * 2) * 4 + w] = mn; s_red[(r * 2 + 1) * 4 + w] = mx; }
    }
    __syncthreads();
    bool fit = true;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        uint32_t mn = s_red[(r * 2) * 4], mx = s_red[(r * 2 + 1) * 4];
#pragma unroll
        for (int k = 1; k < WAVES; ++k) {
            const uint32_t a = s_red[(r * 2) * 4 + k], b = s_red[(r * 2 + 1) * 4 + k];
            mn = a < mn ? a : mn;
            mx = b > mx ? b : mx;
        }
        if (mx <= mn) { mn = 0; mx = 0; }
        blo[r] = mn; bhi[r] = mx;
        fit = fit && (mx - mn <= tile);
    }
    return fit;
}


// ---- dense cell-start table fill (shared by the 2D, slab and 3D reorder kernels) -------------
// cs[c] = index of the first sorted particle whose key is >= c.  Short gaps are written by the
// boundary lane; long gaps (empty regions of the domain) go to a worklist drained by k_fill_gaps.
struct GapEntry { uint32_t begin, end, value; };
#define FS_GAP_INLINE 16u
#define FS_GAP_CHUNK 16384u

__device__ __forceinline__ void fill_cells(uint32_t* __restrict__ cs, uint32_t begin, uint32_t end, uint32_t value,
                                           GapEntry* __restrict__ work, uint32_t* __restrict__ counter,
                                           uint32_t work_cap) {
    if (end <= begin) return;
    if (end - begin <= FS_GAP_INLINE) {
        for (uint32_t c = begin; c < end; ++c) cs[c] = value;
        return;
    }
    for (uint32_t b = begin; b < end; b += FS_GAP_CHUNK) {
        const uint32_t e = (end - b > FS_GAP_CHUNK) ? b + FS_GAP_CHUNK : end;
        const uint32_t slot = atomicAdd(counter, 1u);
        if (slot < work_cap) {
            work[slot] = GapEntry{b, e, value};
        } else {
            for (uint32_t c = b; c < e; ++c) cs[c] = value;   // never expected: capacity covers the worst case
        }
    }
}

}  // namespace fsd

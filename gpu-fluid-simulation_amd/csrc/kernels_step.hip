// kernels_step.hip — the non-sort passes of the SPH step as CDNA4 (gfx950) kernels.
//
// Device state is SoA (float2 pos / vel / pred, f32 density): coalesced 8-byte
// per-lane streams instead of the reference's 32-byte AoS records.  Pass map:
//   (predict_next_position + create_spatial_lookup, compute.wgsl:8-42, are fused into the first
//    kernel of the sort: kernels_sort.hip k_bitonic_local<INIT, KEYGEN> / kernels_csort.hip k_cs_hist)
//   k_reorder       = payload gather after the (key,index) sort + compute_start_indices
//                     (compute.wgsl:45-56) + dense cell-start table
//   k_density       = calculate_density (compute.wgsl:59-74, funcs.wgsl:157-203)
//   k_force         = move_particle + both force sweeps fused (compute.wgsl:79-299)
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "fs_device.h"
#include "fs_kernels.h"

namespace fsd {

#define FS_BLOCK 256

// --------------------------------------------------- dense cell-start table fill
// cs[c] = index of the first sorted particle whose key is >= c (c in [0, ncell]).
// Short gaps are written by the boundary lane; long gaps go to a worklist.
__global__ __launch_bounds__(FS_BLOCK) void k_fill_gaps(uint32_t* __restrict__ cs, const GapEntry* __restrict__ work,
                                                        const uint32_t* __restrict__ counter, uint32_t work_cap) {
    uint32_t count = *counter;
    if (count > work_cap) count = work_cap;
    for (uint32_t e = blockIdx.x; e < count; e += gridDim.x) {
        const GapEntry g = work[e];
        for (uint32_t c = g.begin + threadIdx.x; c < g.end; c += FS_BLOCK) cs[c] = g.value;
    }
}

// -------------------------------------------------------------------- reorder
// Gathers the payload into cell order (the reference swaps whole 32-byte records
// inside the sort, sort.wgsl:44-50; sorting (key,index) pairs and gathering once
// gives the identical arrangement because the network only looks at keys).
template <bool FILL>
__global__ __launch_bounds__(FS_BLOCK) void k_reorder(StepParams P, const u64* __restrict__ pairs,
                                                      const float2* __restrict__ pos_in,
                                                      const float2* __restrict__ vel_in, float2* __restrict__ pos_s,
                                                      float2* __restrict__ vel_s, float2* __restrict__ pred_s,
                                                      uint32_t* __restrict__ key_s, uint32_t* __restrict__ cs,
                                                      uint32_t* __restrict__ start_ref, GapEntry* __restrict__ work,
                                                      uint32_t* __restrict__ counter, uint32_t work_cap,
                                                      unsigned long long* __restrict__ safe, uint32_t* __restrict__ force_defer,
                                                      uint32_t* __restrict__ force_work_count) {
    const uint32_t i = blockIdx.x * FS_BLOCK + threadIdx.x;
    if (threadIdx.x == 0) {                      // the force pass's worklists of this step (same block size and count)
        force_defer[2u * blockIdx.x] = 0u;       // [2 blk] pre-registered by k_density, [2 blk + 1] found late by k_force
        force_defer[2u * blockIdx.x + 1u] = 0u;
        if (blockIdx.x == 0) { force_work_count[0] = 0u; force_work_count[1] = 0u; }
    }
    if (i >= P.n) return;
    const u64 pr = pairs[i];
    const uint32_t key = (uint32_t)(pr >> 32);
    const uint32_t src = (uint32_t)pr;
    const float2 p = pos_in[src];
    const float2 v = vel_in[src];
    if (pos_s) pos_s[i] = p;                  // uniform; nullptr: the force pass reads pos_in[src] itself (StepParams::pos_by_src)
    vel_s[i] = v;
    const float2 pd = predict_pos(P, p, v);   // same expression as the key generation in the sort -> same bits
    pred_s[i] = pd;
    if (key_s) key_s[i] = key;                // uniform; single-domain handles read the key back from `pairs` instead
    {   // fs_device.h "safe operand" classification (finished by k_density): one 64-bit word per wave
        const unsigned long long sb = __builtin_amdgcn_ballot_w64(kin_safe(pd, v));   // lanes that returned above: 0
        if ((threadIdx.x & 63u) == 0u) safe[i >> 6] = sb;
    }

    const uint32_t kc = key < P.ncell ? key : P.ncell;   // clamp for table writes only
    if (i == 0) {
        if (!P.ref_quirks && key < P.ncell) start_ref[key] = 0;   // compute.wgsl:50 skips index 0
        if (FILL) fill_cells(cs, 0u, kc + 1u, 0u, work, counter, work_cap);
    } else {
        const uint32_t prev = (uint32_t)(pairs[i - 1] >> 32);
        if (key != prev) {
            if (key < P.ncell) start_ref[key] = i;                // compute.wgsl:53-55
            const uint32_t pc = prev < P.ncell ? prev : P.ncell;
            if (FILL) fill_cells(cs, pc + 1u, kc + 1u, i, work, counter, work_cap);
        }
    }
    if (FILL && i == P.n - 1) fill_cells(cs, kc + 1u, P.ncell + 1u, P.n, work, counter, work_cap);
}

// ------------------------------------------------------------ neighbour ranges
// Cells (cx-1..cx+1, y) are consecutive ids, and particles are in id order, so a
// row of the 3x3 sweep is ONE contiguous index range [cs[id_lo], cs[id_lo+3]).
// Visiting it ascending is exactly the reference order (offset_x inner, index
// ascending: funcs.wgsl:161-199).
//
// Quirk: the cell of sorted index 0 never gets its start written
// (compute.wgsl:50), so the reference walks it from a stale start v.  Its
// particles are [0,cnt); the walk sees [min(v,cnt), cnt).  Any row range that
// begins at index 0 begins with that cell, so `lo == 0 -> lo = lo_fix`.
__device__ __forceinline__ uint32_t quirk_lo_fix(const StepParams& P, const u64* __restrict__ pairs,
                                                 const uint32_t* __restrict__ cs,
                                                 const uint32_t* __restrict__ start_ref) {
    if (!P.ref_quirks) return 0u;
    const uint32_t cmin = (uint32_t)(pairs[0] >> 32);
    if (cmin >= P.ncell) return 0u;
    const uint32_t v = start_ref[cmin];
    const uint32_t cnt = cs[cmin + 1];
    return v < cnt ? v : cnt;
}

// (cx, y) are the cell's (u, v) of the handle's id layout (fs_device.h StepParams::transposed; the reference layout: u = x, v = y).
__device__ __forceinline__ bool row_range(const StepParams& P, const uint32_t* __restrict__ cs, uint32_t cx,
                                          uint32_t y, uint32_t lo_fix, uint32_t* lo, uint32_t* hi) {
    if (y >= P.grid_v) return false;             // id >= ncell: OOB start_indices read -> nothing
    const uint32_t id_lo = y * P.grid_u + cx - 1u;
    if (id_lo >= P.ncell) return false;
    uint32_t id_hi = id_lo + 3u;
    if (id_hi > P.ncell) id_hi = P.ncell;
    uint32_t a = cs[id_lo];
    const uint32_t b = cs[id_hi];
    if (a == 0u) a = lo_fix;
    *lo = a;
    *hi = b;
    return a < b;
}

// ------------------------------------------------------------ block neighbour tiles
// A workgroup owns 256 consecutive sorted particles (a strip of cells in one grid row), so
// the candidates of ALL its lanes for sweep row r form one short contiguous index range
// [blo_r, bhi_r).  The three ranges are staged into LDS with coalesced loads once and the
// per-lane loops then read LDS instead of issuing one gather per candidate.  Strips that
// straddle a grid-row end (or very sparse ones) exceed the tile and take the global path.
#define NB_TILE 640          // staged candidates per sweep row
#ifndef NBF_TILE
#define NBF_TILE 544         // ... of the force pass (k_force).  384 (round 2) left the blocks of the fluid's free surface — half-empty
                             // cells: 256 particles span 128 cells and their full neighbour row holds 516 - 526 — to the general
                             // kernel's unstaged sweep: 32 blocks per step at 16 M even on the lattice, a ~15 us tail behind the
                             // lean kernel in every step (force 0.603 -> 0.595 ms at 16 M; more at 1 M and per slab rank)
#endif

// -------------------------------------------------------------------- density
__device__ __forceinline__ float density_cube_tol(float h2, float2 me, float2 q, float acc) {
    const float dx = q.x - me.x, dy = q.y - me.y;
    const float t = fmaxf(h2 - __builtin_fmaf(dx, dx, dy * dy), 0.0f);      // NaN candidate: contributes nothing
    return __builtin_fmaf(t * t, t, acc);
}

// MASS1: the tick's particle_mass is exactly 1.0f (the reference's default, src/renderer.rs:374-388): `mass * kern` IS kern then
// (x * 1.0f == x for every f32), and the multiplication — one of the ~14 instructions per candidate — is left out.
template <bool MASS1 = false>
__device__ __forceinline__ float density_term(const StepParams& P, float h2, float2 me, float2 q) {
    const float dx = q.x - me.x, dy = q.y - me.y;
    const float r2 = dx * dx + dy * dy;
    float kern = 0.0f;
    if (!(r2 > h2)) {
        const float diff = h2 - r2;
        kern = P.poly6_norm * diff * diff * diff;       // funcs.wgsl:77
    }
    return MASS1 ? kern : P.mass * kern * 1.0f;         // funcs.wgsl:192
}

// TOL (fs_options.math_mode = FS_MATH_TOLERANCE): r2 by one fma, max(h2 - r2, 0) instead of the compare/select, the
// constant factor mass * 4/(pi h^8) applied once to the sum; stores {pressure_i, 1/rho_i} for the merged force terms.
template <bool TOL, bool MASS1>
__device__ __forceinline__ void density_block(const StepParams& P, uint32_t blk, uint32_t n, const float2* __restrict__ pred,
                                              const uint32_t* __restrict__ cs, const uint32_t* __restrict__ start_ref,
                                              const u64* __restrict__ pairs, const unsigned long long* __restrict__ safe,
                                              float* __restrict__ rho_out, float2* __restrict__ rho2_out,
                                              uint32_t* __restrict__ force_defer, uint32_t* __restrict__ force_work,
                                              uint32_t* __restrict__ force_count, float2 (*s_pred)[NB_TILE], uint32_t* s_red) {
    const uint32_t i = blk * FS_BLOCK + threadIdx.x;
    const bool live = i < n;
    const uint32_t lo_fix = quirk_lo_fix(P, pairs, cs, start_ref);
    const float2 me = pred[live ? i : n - 1];
    uint32_t cx, cy;                // (u, v) of the cell-id layout: (x, y) unless the handle is a transposed slab rank
    int32_t cg;
    uv_local(P, me, &cx, &cy, &cg);
    const float h2 = P.h * P.h;     // funcs.wgsl:73
    RowRanges R;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        R.lo[r] = 0; R.hi[r] = 0;
        if (live) (void)row_range(P, cs, cx, cy + (uint32_t)(r - 1), lo_fix, &R.lo[r], &R.hi[r]);
        if (R.hi[r] < R.lo[r]) R.hi[r] = R.lo[r];
    }
    uint32_t blo[3], bhi[3];
    const bool fit = block_tile_bounds(R, s_red, blo, bhi, NB_TILE);
    if (P.block_bounds && threadIdx.x == 0) {       // the force pass reads these instead of reducing the same ranges again
        uint32_t* bb = P.block_bounds + 8u * blk;
        bb[0] = blo[0]; bb[1] = blo[1]; bb[2] = blo[2]; bb[3] = bhi[0]; bb[4] = bhi[1]; bb[5] = bhi[2];
    }
    {   // The force pass sweeps the same row ranges: a wave it could not finish on its lean path — a row longer than
        // 32 candidates, or a block whose rows do not fit ITS LDS stage — is named here already, so that the general
        // workgroups of the force launch can start on it at once, beside the lean ones (k_force).
        const bool unfit = bhi[0] - blo[0] > NBF_TILE || bhi[1] - blo[1] > NBF_TILE || bhi[2] - blo[2] > NBF_TILE;
        const bool long_row = R.hi[0] - R.lo[0] > 32u || R.hi[1] - R.lo[1] > 32u || R.hi[2] - R.lo[2] > 32u;
        if ((unfit || __any(long_row)) && __builtin_amdgcn_ballot_w64(live) != 0 && (threadIdx.x & 63u) == 0u) {
            const uint32_t old = atomicOr(&force_defer[2u * blk], 1u << (threadIdx.x >> 6));
            if (old == 0u) force_work[atomicAdd(&force_count[0], 1u)] = blk;
        }
    }
    float rho = 0.0f;
    if (fit) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
            for (uint32_t j = threadIdx.x; j < bhi[r] - blo[r]; j += FS_BLOCK) s_pred[r][j] = pred[blo[r] + j];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            // four candidates per trip (independent LDS reads and kernel evaluations give the wave
            // ILP), adds in index order; then a scalar tail
            const float2* sp = s_pred[r] - 0;
            const bool any = R.lo[r] < R.hi[r];
            const uint32_t hi = any ? R.hi[r] - blo[r] : 0u;
            uint32_t k = any ? R.lo[r] - blo[r] : 0u;
            if (TOL) {
                for (; k + 4u <= hi; k += 4u) {
                    const float2 q0 = sp[k], q1 = sp[k + 1u], q2 = sp[k + 2u], q3 = sp[k + 3u];
                    rho = density_cube_tol(h2, me, q0, rho); rho = density_cube_tol(h2, me, q1, rho);
                    rho = density_cube_tol(h2, me, q2, rho); rho = density_cube_tol(h2, me, q3, rho);
                }
                for (; k < hi; ++k) rho = density_cube_tol(h2, me, sp[k], rho);
                continue;
            }
            for (; k + 4u <= hi; k += 4u) {
                const float t0 = density_term<MASS1>(P, h2, me, sp[k]);
                const float t1 = density_term<MASS1>(P, h2, me, sp[k + 1u]);
                const float t2 = density_term<MASS1>(P, h2, me, sp[k + 2u]);
                const float t3 = density_term<MASS1>(P, h2, me, sp[k + 3u]);
                rho += t0; rho += t1; rho += t2; rho += t3;
            }
            for (; k < hi; ++k) rho += density_term<MASS1>(P, h2, me, sp[k]);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r)
            for (uint32_t k = R.lo[r]; k < R.hi[r]; ++k) {
                if (TOL) rho = density_cube_tol(h2, me, pred[k], rho);
                else rho += density_term<MASS1>(P, h2, me, pred[k]);
            }
    }
    if (!live) return;
    if (TOL) {
        rho = rho * (P.mass * P.poly6_norm);                    // sum of (h2 - r2)^3 -> density
        rho = fmaxf(fmaxf(rho, 1.19209290e-07f), 0.1f);
        rho_out[i] = rho;
        rho2_out[i] = make_float2(P.pressure_k * (rho - P.rest_density),
                                  (P.share_div && rho <= FS_RCP_HI) ? rcp_rn_fast(rho) : __fdiv_rn(1.0f, rho));   // same bits (proven range)
        return;
    }
    rho = fmaxf(rho, 1.19209290e-07f);                          // funcs.wgsl:202
    rho = fmaxf(rho, 0.1f);                                     // compute.wgsl:70
    if (rho_out) rho_out[i] = rho;                              // uniform; single-domain handles read it back from rho2.x
    // {rho, +-RN(1/rho)}: the force pass divides by neighbours' densities; the sign carries the particle's
    // "safe operand" classification (fs_device.h) — negative sends every pair it takes part in to true divisions
    const float press = P.pressure_k * (rho - P.rest_density);  // the expression the force pass evaluates
    const bool ok = ((safe[i >> 6] >> (i & 63u)) & 1ull) != 0ull && rho <= FS_RCP_HI && fabsf(press) <= FS_PRESSURE_HI;
    // rho >= 0.1; the lean reciprocal is proven correctly rounded on [2^-20, 2^20] (share_div implies that proof)
    const float y = (P.share_div && rho <= FS_RCP_HI) ? rcp_rn_fast(rho) : __fdiv_rn(1.0f, rho);
    rho2_out[i] = make_float2(rho, ok ? y : -y);
}

#define FS_DENSITY_ARGS                                                                                                   \
    StepParams P, const float2* __restrict__ pred, const uint32_t* __restrict__ cs, const uint32_t* __restrict__ start_ref, \
        const u64* __restrict__ pairs, const unsigned long long* __restrict__ safe, float* __restrict__ rho_out,         \
        float2* __restrict__ rho2_out, uint32_t* __restrict__ force_defer, uint32_t* __restrict__ force_work,            \
        uint32_t* __restrict__ force_count
template <bool TOL, bool MASS1>
__global__ __launch_bounds__(FS_BLOCK) void k_density(FS_DENSITY_ARGS) {
    __shared__ float2 s_pred[3][NB_TILE];
    __shared__ uint32_t s_red[24];
    const uint32_t n = P.n_live ? *P.n_live : P.n;
    uint32_t blk;
    if (!xcd_block(P, (n + FS_BLOCK - 1) / FS_BLOCK, &blk)) return;   // uniform: no live particle in this block
    density_block<TOL, MASS1>(P, blk, n, pred, cs, start_ref, pairs, safe, rho_out, rho2_out, force_defer, force_work, force_count, s_pred, s_red);
}
// Edge-first slab step, column-major ids (fs_device.h EdgeBlocks): the density of the columns the edge columns' force launch
// reads — the edge columns and one more towards the interior — ahead of the full launch, on the exchange stream.  (The full
// launch writes the same values again.)
template <bool TOL, bool MASS1>
__global__ __launch_bounds__(FS_BLOCK) void k_density_edge(FS_DENSITY_ARGS) {
    __shared__ float2 s_pred[3][NB_TILE];
    __shared__ uint32_t s_red[24];
    const uint32_t n = *P.n_live;
    const EdgeBlocks E = edge_blocks(P, cs, n, 1u);
    for (uint32_t t = blockIdx.x; t < edge_block_count(E); t += gridDim.x) {
        density_block<TOL, MASS1>(P, edge_block_at(E, t), n, pred, cs, start_ref, pairs, safe, rho_out, rho2_out, force_defer, force_work,
                           force_count, s_pred, s_red);
        __syncthreads();                             // the LDS stage is reused
    }
}

// ---------------------------------------------------------- force + integrate
// Two phases per lane so the expensive body (pressure + viscosity terms of one in-radius neighbour)
// runs with dense lanes:
//   scan  - test `k != i && !(r2 > sqr_radius)` (compute.wgsl:195,202) for the candidates of the three
//           row ranges and record the outcome as pass bits in registers (no branches, no lists);
//   heavy - every lane walks its set bits in the reference visiting order, so sums keep their association.
// force_sweep_masks handles the common case (all three rows of every lane of the wave <= 32 candidates:
// three masks, walked without idle lanes), force_sweep_chunks everything else.

struct ForceAcc { float fpx, fpy, fvx, fvy; uint32_t seed; };
struct ForceTerms { float px, py, vx, vy; };

// One in-radius neighbour: pressure (compute.wgsl:207-223) and viscosity (:283-288) terms.
// `seed` only advances on the coincident-particle path (dst == 0, compute.wgsl:211-212).
// FAST (fs_options.math_mode = FS_MATH_WGSL_ULP): `/` becomes n * v_rcp_f32(d) (<= ~1.5 ulp) and sqrt
// the native v_sqrt_f32 (1 ulp) — inside WGSL's own accuracy contract for the reference shaders (f32
// division 2.5 ULP, sqrt via inverseSqrt 2 ULP), but no longer bit-identical to the IEEE oracle.
template <bool FAST> __device__ __forceinline__ float fs_div(float n, float d) {
    return FAST ? n * __builtin_amdgcn_rcpf(d) : __fdiv_rn(n, d);
}
template <bool FAST> __device__ __forceinline__ float fs_sqrt(float x) {
    return FAST ? __builtin_amdgcn_sqrtf(x) : sqrt_rn(x);
}

template <bool FAST>
__device__ __forceinline__ ForceTerms force_terms(const StepParams& P, const float2 me, const float2 mv,
                                                  float pressure, const float2 q, const float2 nv, float nrho,
                                                  uint32_t& seed) {
    const float h = P.h;
    const float ox = q.x - me.x, oyv = q.y - me.y;
    const float r2 = ox * ox + oyv * oyv;
    const float dst = fs_sqrt<FAST>(r2);                                // compute.wgsl:207,283
    float dx, dy;
    if (dst == 0.0f) {                                                  // :211-212
        const float rx = rand_f32(&seed);
        const float ry = rand_f32(&seed);
        const float len = fs_sqrt<FAST>(rx * rx + ry * ry);
        dx = fs_div<FAST>(rx, len);
        dy = fs_div<FAST>(ry, len);
    } else {
        dx = fs_div<FAST>(ox, dst);
        dy = fs_div<FAST>(oyv, dst);
    }
    const float npress = P.pressure_k * (nrho - P.rest_density);
    const float kern = (dst <= h) ? (-(h - dst)) * P.spiky : 0.0f;      // funcs.wgsl:101-109
    const float shared = (pressure + npress) * 0.5f;
    ForceTerms T;
    T.px = fs_div<FAST>(dx * kern * shared, nrho);                         // compute.wgsl:223
    T.py = fs_div<FAST>(dy * kern * shared, nrho);
    float kv = 0.0f;                                                    // funcs.wgsl:112-123
    if (dst <= h) {
        if (dst == 0.0f) {
            kv = P.visc_k;
        } else {
            // the two constant denominators go through div_const (bit-identical to `/`, proven per
            // constant at create time); 2.0f*h*h*h and h*h are exactly P.div_2h3.c / P.div_h2.c
            // (proven for FS_CONSTDIV_MIN <= |x| <= c; dst >= 2^-20 puts dst^2 and dst^3 inside, and
            //  dst <= h keeps them <= h^2 and h^3 = c/2)
            const bool tiny = dst < 9.5367431640625e-07f;                                   // 2^-20: rare, true division
            float a, b;
            if (FAST) {
                a = fs_div<true>(-(dst * dst * dst), 2.0f * h * h * h);
                b = fs_div<true>(dst * dst, h * h);
            } else if (tiny) {
                a = __fdiv_rn(-(dst * dst * dst), P.div_2h3.c);
                b = __fdiv_rn(dst * dst, P.div_h2.c);
            } else {
                a = div_const(P.div_2h3, -(dst * dst * dst));
                b = div_const(P.div_h2, dst * dst);
            }
            kv = P.visc_k * (a + b + (fs_div<FAST>(h, 2.0f * dst)) - 1.0f);
        }
    }
    T.vx = fs_div<FAST>(nv.x - mv.x, nrho) * kv;                           // compute.wgsl:288
    T.vy = fs_div<FAST>(nv.y - mv.y, nrho) * kv;
    return T;
}

// ---- tolerance mode (MODE 2, fs_options.math_mode = FS_MATH_TOLERANCE) ------------------------------------------
// The pressure and viscosity terms of one in-radius neighbour merged algebraically (compute.wgsl:207-223, :283-288,
// funcs.wgsl:101-123): one v_rsq_f32, fused multiply-adds, pressure_j and 1/rho_j precomputed per particle by
// k_density<true> — 24 issue slots per pair instead of ~85.  Within rtol 1e-5 / atol 1e-4*h of the IEEE oracle per
// step (tests/test_parity_gpu.py::test_tolerance_mode_*); cell keys and start_indices stay bit-exact (they come
// from the sort and the reorder pass, which this mode does not touch).  Coincident particles (r == 0) keep the
// reference's xorshift direction.
struct TolConsts { float cP, c3, c2, hh; };
__device__ __forceinline__ TolConsts tol_consts(const StepParams& P) {
    TolConsts C;
    const float h = P.h;
    C.cP = -0.5f * P.spiky;                       // kern * 0.5 = -(h - dst) * spiky * 0.5
    C.c3 = -1.0f / (2.0f * h * h * h);
    C.c2 = 1.0f / (h * h);
    C.hh = 0.5f * h;
    return C;
}
__device__ __forceinline__ void force_accum_tol(const StepParams& P, const TolConsts& C, const float2 me, const float2 mv,
                                                float pressure, const float2 q, const float2 nv,
                                                const float2 nd /* {pressure_j, 1/rho_j} */, ForceAcc& A) {
    const float ox = q.x - me.x, oy = q.y - me.y;
    const float r2 = __builtin_fmaf(ox, ox, oy * oy);
    float dirx = ox, diry = oy, inv, dst;
    if (r2 == 0.0f) {                                                   // compute.wgsl:211-212 (rare)
        const float rx = rand_f32(&A.seed), ry = rand_f32(&A.seed);
        const float il = __builtin_amdgcn_rsqf(__builtin_fmaf(rx, rx, ry * ry));
        dirx = rx * il; diry = ry * il;
        dst = 0.0f; inv = 1.0f;                                         // dir is already normalised
    } else {
        inv = __builtin_amdgcn_rsqf(r2);
        dst = r2 * inv;
    }
    const float w = fmaxf(P.h - dst, 0.0f);                             // dst <= h for every admitted candidate
    const float coefP = (w * C.cP) * (pressure + nd.x) * nd.y * inv;
    float u = __builtin_fmaf(C.c3, dst, C.c2);
    u = __builtin_fmaf(u, r2, -1.0f);
    u = r2 == 0.0f ? 1.0f : __builtin_fmaf(C.hh, inv, u);               // funcs.wgsl:116: r == 0 -> the bare constant
    const float kvv = u * (P.visc_k * nd.y);
    A.fpx = __builtin_fmaf(dirx, coefP, A.fpx);
    A.fpy = __builtin_fmaf(diry, coefP, A.fpy);
    A.fvx = __builtin_fmaf(nv.x - mv.x, kvv, A.fvx);
    A.fvy = __builtin_fmaf(nv.y - mv.y, kvv, A.fvy);
}

// The same terms with ONE true division per denominator (1/dst, 1/nrho) and div_by_rcp() for the
// seven quotients — bit-identical to force_terms<false> whenever `good` comes back all-ones (operands
// inside the proven range, fs_device.h).  Straight-line: no PRNG path, no tiny-distance path;
// those (and any out-of-range operand) clear the lane's bit in `good`, and the caller re-evaluates
// the pair with the exact body for the whole wave when any active lane's bit is missing.
// Guards per pair: r2 >= 2^-40 (excludes r2 == 0 = the PRNG path, NaN and div_const's tiny range; r2 <= h*h
// because the scan admitted it, and the host only enables this path for h <= 2^19: the proven sqrt range), the
// neighbour's "safe operand" sign (fs_device.h; the lane's own is folded in by the caller), and the lower bound
// of the two pressure numerators.
__device__ __forceinline__ wave_mask num_lo_ok(float a) { return wm(fabsf(a) >= 0x1p-76f) | wm(a == 0.0f); }   // NaN: 0 (fs_device.h: why 2^-76)
__device__ __forceinline__ ForceTerms force_terms_shared(const StepParams& P, const float2 me, const float2 mv,
                                                         float pressure, const float2 q, const float2 nv,
                                                         const float2 nd /* {density, +-RN(1/density)} */, wave_mask& good) {
    const float h = P.h;
    const float nrho = nd.x, yrho = nd.y;
    const float ox = q.x - me.x, oyv = q.y - me.y;
    const float r2 = ox * ox + oyv * oyv;
    good = wm(r2 >= FS_SQRT_LO) & wm(yrho > 0.0f);
    const float dst = sqrt_rn_fast(r2);                                 // in [2^-20, ~h]
    const float ydst = rcp_rn_fast(dst);
    const float dx = div_by_rcp(ox, dst, ydst);
    const float dy = div_by_rcp(oyv, dst, ydst);
    const float npress = P.pressure_k * (nrho - P.rest_density);
    const bool inside = dst <= h;
    const float kern = inside ? (-(h - dst)) * P.spiky : 0.0f;
    const float shared = (pressure + npress) * 0.5f;
    const float apx = dx * kern * shared, apy = dy * kern * shared;
    const float dvx = nv.x - mv.x, dvy = nv.y - mv.y;
    good &= num_lo_ok(apx) & num_lo_ok(apy);
    ForceTerms T;
    T.px = div_by_rcp(apx, nrho, yrho);
    T.py = div_by_rcp(apy, nrho, yrho);
    // share_div implies both constant-division proofs succeeded (engine.hip)
    const float a = div_const_fast(-(dst * dst * dst), P.div_2h3.c, P.div_2h3.y);
    const float b = div_const_fast(dst * dst, P.div_h2.c, P.div_h2.y);
    const float hq = div_by_rcp(h, 2.0f * dst, 0.5f * ydst);           // RN(1/(2 dst)) == RN(1/dst)/2 exactly
    const float kv = inside ? P.visc_k * (a + b + hq - 1.0f) : 0.0f;
    T.vx = div_by_rcp(dvx, nrho, yrho) * kv;
    T.vy = div_by_rcp(dvy, nrho, yrho) * kv;
    return T;
}

// k_force stages the predicted positions of its three sweep rows in LDS: NBF_TILE candidates per row plus
// NBF_PAD of slack (the mask scans read up to 32 entries from a range start, whatever the range's length).
// Velocity and {density, 1/density} of the few in-radius neighbours are gathered in the heavy phase instead
// (staging them too cost occupancy and measured slower).
#define NBF_PAD 32u
#define NBF_ROW (NBF_TILE + NBF_PAD)     // LDS row pitch

__device__ __forceinline__ void shift_in_not_greater(uint32_t& mask, float r2, float lim) {
    // !(lim < r2) == !(r2 > lim), NaN included; this operand order lets `lim` stay in an SGPR
    asm("v_cmp_nlt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(r2), "s"(lim) : "vcc");
}

// ---- chunked sweep: the general case (a row range of the wave is longer than 32, or the rows do not
// fit the LDS tile: dense clusters).  Same machinery as the mask sweep below, one 32-candidate chunk of
// one row at a time: wave-uniform scan of the chunk into a register mask (v_cmp + v_addc_co per
// candidate), then every lane walks its set bits.  Rows and chunks are taken in order, so a lane still
// visits its neighbours in the reference order; lanes idle while others finish a chunk (dense regions
// only — the common case never comes here).  STAGED: candidates from the LDS tile, else from global
// memory (the pred array is allocated with FS_PRED_SLACK elements of slack for the read-ahead).
template <bool STAGED, int MODE>
__device__ __forceinline__ void force_sweep_chunks(const StepParams& P, const RowRanges& R, const uint32_t* blo,
                                                   uint32_t ii, const float2 me, const float2 mv, float pressure,
                                                   const float2* __restrict__ pred, const float2* __restrict__ vel_s,
                                                   const float2* __restrict__ rho2, const float2* s_flat, bool me_ok,
                                                   ForceAcc& A) {
    const float lim = P.sqr_radius;
    constexpr bool FAST = MODE == 1;
    const TolConsts TC = tol_consts(P);
    const wave_mask me_okm = wm(me_ok);      // the lane's own "safe operand" classification (all lanes active here)
    // plain registers: as arrays the row selects below become dynamic indexing, which the compiler
    // serves from scratch / promoted LDS
    uint32_t lo0 = R.lo[0], lo1 = R.lo[1], lo2 = R.lo[2], hi0 = R.hi[0], hi1 = R.hi[1], hi2 = R.hi[2];
    uint32_t b00 = blo[0], b01 = blo[1], b02 = blo[2];
    asm volatile("" : "+v"(lo0), "+v"(lo1), "+v"(lo2), "+v"(hi0), "+v"(hi1), "+v"(hi2), "+v"(b00), "+v"(b01), "+v"(b02));
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
        const uint32_t lo = r == 0 ? lo0 : r == 1 ? lo1 : lo2;
        const uint32_t hi = r == 0 ? hi0 : r == 1 ? hi1 : hi2;
        const uint32_t b0 = r == 0 ? b00 : r == 1 ? b01 : b02;
        const uint32_t len = hi - lo;
        // Round 3: FS_CHUNK_BATCH chunks of 32 candidates are scanned before the walk starts, and their masks are walked as
        // ONE shift register (cur <- n1 <- n2 <- n3; the chunks of a batch are consecutive in the row, so a refill only
        // advances the two bases by 32 candidates).  With one chunk per walk a lane waited for the wave's slowest lane after
        // every ~11 hits (lane utilisation 0.66 in the dense regime, profiles/r03_counters_2d_dense.md); over 128
        // candidates the hit counts of the lanes differ relatively less.
#ifndef FS_CHUNK_BATCH
#define FS_CHUNK_BATCH 4
#endif
#pragma unroll 1
        for (uint32_t c0 = 0; __any(c0 < len); c0 += 32u * FS_CHUNK_BATCH) {   // c0 is wave-uniform
            uint32_t mq[FS_CHUNK_BATCH];
            const uint32_t g0 = c0 < len ? lo + c0 : 0u;                  // global index of the batch's first candidate
            // byte offset of the batch's first candidate: into the LDS tile, or (32-bit, n <= 2^28) into pred
            const uint32_t boff0 = (STAGED ? (c0 < len ? (uint32_t)r * NBF_ROW + (g0 - b0) : 0u) : g0) << 3;
            const char* src = STAGED ? reinterpret_cast<const char*>(s_flat) : reinterpret_cast<const char*>(pred);
#define FS_CAND(off, k) (*reinterpret_cast<const float2*>(src + ((off) + ((k) << 3))))
#pragma unroll
            for (int q = 0; q < FS_CHUNK_BATCH; ++q) {
                const uint32_t cq = c0 + 32u * (uint32_t)q;
                const uint32_t clen = cq < len ? (len - cq < 32u ? len - cq : 32u) : 0u;
                const uint32_t boff = clen ? boff0 + 256u * (uint32_t)q : 0u;
                uint32_t mask = 0, t = 0;
                for (; __any(t < clen); t += 4u) {
                    const float2 q0 = FS_CAND(boff, t), q1 = FS_CAND(boff, t + 1u), q2 = FS_CAND(boff, t + 2u), q3 = FS_CAND(boff, t + 3u);
                    const float2 qq[4] = {q0, q1, q2, q3};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float ox = qq[u].x - me.x, oyv = qq[u].y - me.y;
                        shift_in_not_greater(mask, ox * ox + oyv * oyv, lim);
                    }
                }
                mask = t ? mask << (32u - t) : 0u;
                mask &= clen ? 0xFFFFFFFFu << (32u - clen) : 0u;
                const uint32_t g = g0 + 32u * (uint32_t)q;
                if (r == 1 && clen && ii - g < clen) mask &= ~(0x80000000u >> (ii - g));   // k != i
                mq[q] = mask;
            }
            // walk, software-pipelined by one neighbour; (boff, goff) are the bases of the chunk `cur` belongs to
            uint32_t cur = mq[0], n1 = FS_CHUNK_BATCH > 1 ? mq[1 % FS_CHUNK_BATCH] : 0u, n2 = FS_CHUNK_BATCH > 2 ? mq[2 % FS_CHUNK_BATCH] : 0u,
                     n3 = FS_CHUNK_BATCH > 3 ? mq[3 % FS_CHUNK_BATCH] : 0u;
            uint32_t boff = boff0, goff = g0 << 3;
            float2 qn = make_float2(0.0f, 0.0f), vn = qn, dn = qn;
            bool have = false, pending = false;
#define FS_FETCH_NEXT1()                                                                                             \
    do {                                                                                                             \
        if (cur == 0u) { cur = n1; n1 = n2; n2 = n3; n3 = 0u; boff += 256u; goff += 256u; }   /* next chunk of the batch */ \
        have = cur != 0u;                                                                                            \
        pending = (cur | n1 | n2 | n3) != 0u;            /* an empty chunk in the middle costs this lane one idle trip */ \
        if (have) {                                                                                                  \
            const uint32_t t8 = (uint32_t)__builtin_clz(cur) << 3;                                                   \
            cur ^= 0x80000000u >> (t8 >> 3);                                                                         \
            qn = FS_CAND(boff, t8 >> 3);                                                                             \
            const uint32_t off = goff + t8;                                                                          \
            vn = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(vel_s) + off);                       \
            dn = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(rho2) + off);                        \
        }                                                                                                            \
    } while (0)
            FS_FETCH_NEXT1();
            while (__any(pending)) {
                const bool cur_valid = have;
                const float2 q0 = qn, v0 = vn, d0 = dn;
                FS_FETCH_NEXT1();
                if (cur_valid && MODE == 2) {
                    force_accum_tol(P, TC, me, mv, pressure, q0, v0, d0, A);
                } else if (cur_valid) {
                    ForceTerms T0;
                    if (FAST) {
                        T0 = force_terms<true>(P, me, mv, pressure, q0, v0, d0.x, A.seed);
                    } else {
                        wave_mask good = 0;
                        if (P.share_div) { T0 = force_terms_shared(P, me, mv, pressure, q0, v0, d0, good); good &= me_okm; }
                        if (good != wm(true)) {
                            T0 = force_terms<false>(P, me, mv, pressure, q0, v0, d0.x, A.seed);
                        }
                    }
                    A.fpx += T0.px; A.fpy += T0.py; A.fvx += T0.vx; A.fvy += T0.vy;
                }
            }
#undef FS_FETCH_NEXT1
#undef FS_CAND
        }
    }
}

// ---- mask sweep: the normal case (staged tiles, no row range of the wave longer than 32) ----------
//   scan  — per sweep row one 32-bit pass mask in a register.  Per candidate: the LDS read, r2, and
//           v_cmp_ngt + v_addc_co, which shifts `!(r2 > sqr_radius)` (compute.wgsl:202; true for NaN
//           like the shader's test) into the mask — no branch, no LDS write.  Trip counts are
//           wave-uniform (longest range of the wave, in fours); a lane masks off what lies past its
//           own range afterwards, and the middle row clears the lane's own bit (`k != i`, :195).
//   heavy — every lane walks its set bits, row 0, 1, 2, ascending = the reference visiting order, so
//           the sums keep their association; all lanes stay busy until the longest list is done.
// GENERAL = false (the lean main kernel): a pair whose operands fall outside the proven ranges is not re-evaluated
// here — the wave remembers it (`bad`) and the caller hands the whole wave to the general kernel instead, so the
// exact true-division body never enters this kernel's register allocation.
template <int MODE, bool GENERAL>
__device__ __forceinline__ bool force_sweep_masks(const StepParams& P, const RowRanges& R, const uint32_t* blo,
                                                  uint32_t ii, const float2 me, const float2 mv, float pressure,
                                                  const float2* __restrict__ vel_s, const float2* __restrict__ rho2,
                                                  const float2* s_flat /* [3][NBF_ROW] */, bool me_ok, ForceAcc& A) {
    uint32_t m[3], la[3];                    // masks (bit 31-t <=> candidate lo+t), flat LDS index of lo
    const float lim = P.sqr_radius;
    constexpr bool FAST = MODE == 1;
    const TolConsts TC = tol_consts(P);
    bool bad = false;                        // wave-uniform
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint32_t len = R.hi[r] - R.lo[r];                           // <= 32 (caller)
        la[r] = (uint32_t)r * NBF_ROW + (len ? R.lo[r] - blo[r] : 0u);
        const float2* base = s_flat + la[r];
        uint32_t mask = 0, t = 0;
        for (; __any(t < len); t += 4u) {                                 // t is wave-uniform
            const float2 q0 = base[t], q1 = base[t + 1u], q2 = base[t + 2u], q3 = base[t + 3u];
            const float2 qq[4] = {q0, q1, q2, q3};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ox = qq[u].x - me.x, oyv = qq[u].y - me.y;
                shift_in_not_greater(mask, ox * ox + oyv * oyv, lim);
            }
        }
        // candidate t sits at bit (trips - 1 - t): left-align, keep the lane's own len candidates
        mask = t ? mask << (32u - t) : 0u;
        mask &= len ? 0xFFFFFFFFu << (32u - len) : 0u;
        if (r == 1 && ii - R.lo[1] < len) mask &= ~(0x80000000u >> (ii - R.lo[1]));
        m[r] = mask;
    }
    // The three masks are walked as a shift register (round 3): `cur` is the mask being consumed with its LDS / global
    // bases, (n1, n2) wait behind it.  Empty masks are squeezed out first, so "cur == 0 -> pull n1" is all a refill ever
    // needs, and the per-neighbour bit extraction touches ONE mask and ONE pair of bases instead of selecting among three
    // masks and six bases.  Row order 0, 1, 2 (= the reference visiting order) is kept.
    // Software-pipelined: the LDS read and the two gathers of a later neighbour are issued before the terms of
    // neighbour k are evaluated, so a lane's own arithmetic covers their latency.  FS_PIPE_DEPTH = 1: neighbour
    // k+1 (one slot, rotated by moves); 2: neighbours k+1 and k+2 (three slots A, B, C refilled in turn, the loop
    // unrolled by three so no value is moved).
    const wave_mask me_okm = wm(me_ok);      // the lane's own "safe operand" classification (all lanes active here)
    uint32_t cur = m[0], n1 = m[1], n2 = m[2];
    uint32_t lac = la[0] << 3, la_1 = la[1] << 3, la_2 = la[2] << 3, loc = R.lo[0] << 3, lo_1 = R.lo[1] << 3, lo_2 = R.lo[2] << 3;   // bytes
    if (n1 == 0u) { n1 = n2; la_1 = la_2; lo_1 = lo_2; n2 = 0u; }
    if (cur == 0u) { cur = n1; lac = la_1; loc = lo_1; n1 = n2; la_1 = la_2; lo_1 = lo_2; n2 = 0u; }
#define FS_FETCH(have, qn, vn, dn)                                                                                   \
    do {                                                                                                             \
        have = cur != 0u;                                                                                            \
        if (have) {                                                                                                  \
            const uint32_t t8 = (uint32_t)__builtin_clz(cur) << 3;                                                   \
            cur ^= 0x80000000u >> (t8 >> 3);                                                                         \
            qn = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(s_flat) + (lac + t8));               \
            /* both arrays hold 8-B elements: one 32-bit byte offset from the two SGPR bases (n <= 2^28) */          \
            const uint32_t off = loc + t8;                                                                           \
            vn = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(vel_s) + off);                       \
            dn = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(rho2) + off); /* {rho, 1/rho} */     \
            if (cur == 0u) { cur = n1; lac = la_1; loc = lo_1; n1 = n2; la_1 = la_2; lo_1 = lo_2; n2 = 0u; }         \
        }                                                                                                            \
    } while (0)
#define FS_PAIR(cur_valid, q0, v0, d0)                                                                               \
    do {                                                                                                             \
        if (cur_valid && MODE == 2) {                                                                                \
            force_accum_tol(P, TC, me, mv, pressure, q0, v0, d0, A);                                                 \
        } else if (cur_valid) {                                                                                      \
            ForceTerms T0;                                                                                           \
            if (FAST) {                                                                                              \
                T0 = force_terms<true>(P, me, mv, pressure, q0, v0, d0.x, A.seed);                                   \
            } else {                                                                                                 \
                wave_mask good = 0;                                                                                  \
                if (P.share_div) { T0 = force_terms_shared(P, me, mv, pressure, q0, v0, d0, good); good &= me_okm; } \
                if (good != wm(true)) {             /* rare, wave-uniform */                                         \
                    if (GENERAL) T0 = force_terms<false>(P, me, mv, pressure, q0, v0, d0.x, A.seed);                 \
                    else bad = true;                                                                                 \
                }                                                                                                    \
            }                                                                                                        \
            A.fpx += T0.px; A.fpy += T0.py; A.fvx += T0.vx; A.fvy += T0.vy;                                          \
        }                                                                                                            \
    } while (0)
    // measured at 16M: depth 2 is worth 2.3 % to the strict kernel (0.721 -> 0.705 ms) and COSTS the tolerance-mode
    // kernel 5 % (0.57 -> 0.60 ms: with 24 instructions per pair the extra selects and registers outweigh the cover)
    if constexpr (MODE == 2) {
    float2 qn = make_float2(0.0f, 0.0f), vn = qn, dn = qn;
    bool have = false;
    FS_FETCH(have, qn, vn, dn);
    while (__any(have)) {
        const bool cur_valid = have;
        const float2 q0 = qn, v0 = vn, d0 = dn;
        FS_FETCH(have, qn, vn, dn);
        FS_PAIR(cur_valid, q0, v0, d0);
    }
    } else {
    float2 qA = make_float2(0.0f, 0.0f), vA = qA, dA = qA, qB = qA, vB = qA, dB = qA, qC = qA, vC = qA, dC = qA;
    bool hA = false, hB = false, hC = false;
    FS_FETCH(hA, qA, vA, dA);
    FS_FETCH(hB, qB, vB, dB);
    FS_FETCH(hC, qC, vC, dC);
    for (;;) {       // a slot is refilled right after its neighbour's terms: two bodies later it is consumed
        if (!__any(hA)) break;
        { const bool cv = hA; const float2 q0 = qA, v0 = vA, d0 = dA; FS_PAIR(cv, q0, v0, d0); }
        FS_FETCH(hA, qA, vA, dA);
        if (!__any(hB)) break;
        { const bool cv = hB; const float2 q0 = qB, v0 = vB, d0 = dB; FS_PAIR(cv, q0, v0, d0); }
        FS_FETCH(hB, qB, vB, dB);
        if (!__any(hC)) break;
        { const bool cv = hC; const float2 q0 = qC, v0 = vC, d0 = dC; FS_PAIR(cv, q0, v0, d0); }
        FS_FETCH(hC, qC, vC, dC);
    }
    }
#undef FS_PAIR
#undef FS_FETCH
    // `bad` was set under the exec mask of the lanes that were evaluating the failing pair: make it the wave's
    return __any(bad);
}

// amdgpu_waves_per_eu(8, 8): with the chunked sweep inlined next to the mask sweep the allocator would take
// 83 VGPRs (5 waves/SIMD) and the common path loses 9 %; capped at 64 it spills in the rarely taken
// branches instead (measured: 0.77 vs 0.86 ms in the bench window, 2.67 vs 2.82 ms in the dense regime).
struct AosParticle { float2 position, predicted, velocity; float density; uint32_t grid; };   // ParticleInstance, 32 B

// Integration of one particle from its accumulated force sums (compute.wgsl:93-153, :298) and the stores of its new state.
template <int MODE, bool AOS>
__device__ __forceinline__ void integrate_store(const StepParams& P, uint32_t i, const float2 me, const float2 mv, const float2 mrec,
                                                float mrho, const float2 p_own, const ForceAcc& A, uint32_t cx, uint32_t cy,
                                                const float2* __restrict__ tex, float2* __restrict__ pos_out,
                                                float2* __restrict__ vel_out, AosParticle* __restrict__ aos_out,
                                                const float* __restrict__ rho_arr) {
    const float fvx = A.fvx * P.visc_coeff;                             // compute.wgsl:298
    const float fvy = A.fvy * P.visc_coeff;

    // integrate (compute.wgsl:93-153)
    float2 v = mv;
    float2 p = p_own;
    const float ax = A.fpx + fvx, ay = A.fpy + fvy;
    if (MODE == 2) {
        v.x = __builtin_fmaf(ax * mrec.y, P.dt, v.x);
        v.y = __builtin_fmaf(ay * mrec.y, P.dt, v.y);
    } else {
        v.x += __fdiv_rn(ax, mrho) * P.dt;
        v.y += __fdiv_rn(ay, mrho) * P.dt;
    }
    v.x += P.gx * P.dt;
    v.y += P.gy * P.dt;
    if (P.mouse_state != 0) {
        const float dx = P.mouse_x - me.x, dy = P.mouse_y - me.y;
        const float dist = sqrt_rn(dx * dx + dy * dy);
        if (dist <= P.mouse_radius) {
            const float dirx = __fdiv_rn(__fdiv_rn(dx, dist), dist);
            const float diry = __fdiv_rn(__fdiv_rn(dy, dist), dist);
            const float ratio = __fdiv_rn(dist, P.mouse_radius);
            v.x += dirx * P.mouse_power * (float)P.mouse_state * ratio;
            v.y += diry * P.mouse_power * (float)P.mouse_state * ratio;
        }
    }
    if (!(v.x == v.x && v.y == v.y)) { v.x = 0.0f; v.y = 0.0f; }
    if (MODE == 2) {
        const float s2 = __builtin_fmaf(v.x, v.x, v.y * v.y);
        if (s2 > 250000.0f) { const float k = 500.0f * __builtin_amdgcn_rsqf(s2); v.x *= k; v.y *= k; }
    } else {
        // compute.wgsl:118-122.  The square root (an IEEE sequence of ~15 instructions) is only needed near the clamp:
        // for s2 <= 249 000 it is at most 498.999 < 500 whatever the rounding, so nothing can change; NaN was reset
        // above and an infinite s2 takes the branch.
        const float s2 = v.x * v.x + v.y * v.y;
        if (s2 > 249000.0f) {
            const float speed = sqrt_rn(s2);
            if (speed > 500.0f) {
                v.x = __fdiv_rn(v.x, speed) * 500.0f;
                v.y = __fdiv_rn(v.y, speed) * 500.0f;
            }
        }
    }
    p.x += v.x * P.dt;
    p.y += v.y * P.dt;

    float2 force = make_float2(0.0f, 0.0f);
    if (!P.tex_zero) {   // uniform; an all-zero field (the default, and the benchmark's) changes nothing below
        const float uvx = (__fdiv_rn(me.x, P.bounds_x) * 1.0f) + 0.5f;      // compute.wgsl:127
        const float uvy = (__fdiv_rn(me.y, P.bounds_y) * 1.0f) + 0.5f;
        const uint32_t px = f32_to_u32_sat(uvx * P.tex_w);
        const uint32_t py = f32_to_u32_sat(uvy * P.tex_h);
        const uint32_t tix = py * P.tex_w_u + px;
        if (tix < P.tex_len) force = tex[tix];
    }
    if (force.x != 0.0f || force.y != 0.0f) {                           // compute.wgsl:131-140
        const float p2wx = __fdiv_rn(P.bounds_x * 2.0f, P.tex_w);
        const float p2wy = __fdiv_rn(P.bounds_y * 2.0f, P.tex_h);
        const float fwx = force.x * p2wx, fwy = force.y * p2wy;
        const float len = sqrt_rn(force.x * force.x + force.y * force.y);
        const float nx = __fdiv_rn(force.x, len), ny = __fdiv_rn(force.y, len);
        p.x += fwx;
        p.y += fwy;
        const float vn = v.x * nx + v.y * ny;
        v.x -= (1.0f - P.damping) * vn * nx;
        v.y -= (1.0f - P.damping) * vn * ny;
    }
    if (fabsf(p.x) > P.bs_x) { p.x = P.bs_x * sign_f32(p.x); v.x *= -1.0f * P.damping; }
    if (fabsf(p.y) > P.bs_y) { p.y = P.bs_y * sign_f32(p.y); v.y *= -1.0f * P.damping; }
    pos_out[i] = p;
    vel_out[i] = v;
    if (AOS) {       // compile-time (even unused, the store costs the plain kernel 5 %): a renderer hand-off is registered (fs_export_handle) — the 32-byte ParticleInstance the
                     // reference's fragment shader binds (src/simulation.rs:552-559) is written here, no export pass
        AosParticle a;
        a.position = p; a.predicted = me; a.velocity = v; a.density = MODE == 2 ? rho_arr[i] : mrho;
        a.grid = cy * P.grid_u + cx;      // == the sorted key: same expression as cell_of_point(pred) (single-domain handles only)
        aos_out[i] = a;
    }
}

#ifndef FS_FORCE_WAVES
#define FS_FORCE_WAVES 8
#endif
// One workgroup's 256 particles.  GENERAL = false is the lean main path: mask sweep with the shared-reciprocal terms
// only.  A wave it cannot finish that way — its tile does not fit the LDS stage, one of its sweep rows is longer than
// 32 candidates (dense clusters), or an operand fell outside the proven quotient ranges — is handed to the general
// kernel through a device worklist (`defer_bits[blk]` bit w, the block id pushed once) and writes nothing.
// GENERAL = true is the complete body (chunked sweeps, true-division fallback) for the waves named in `wave_bits`.
// The split keeps the rare paths out of the common kernel's register allocation: 39 VGPRs instead of 64 + 35
// spilled, force 0.72 -> 0.67 ms at 16M (profiles/r02_c_force_split.txt).
// defer_bits[2 blk] / worklist[0 .. nblk) / work_count[0]: waves named by k_density before the launch ("pre");
// defer_bits[2 blk + 1] / worklist[nblk ..) / work_count[1]: waves the lean path gives up on itself ("late").
template <int MODE, bool AOS, bool GENERAL>
__device__ __forceinline__ void force_block(const StepParams& P, uint32_t blk, uint32_t n, uint32_t wave_bits,
                                            const float2* __restrict__ pos_s, const float2* __restrict__ vel_s,
                                            const float2* __restrict__ pred, const float2* __restrict__ rho2,
                                            const uint32_t* __restrict__ cs, const uint32_t* __restrict__ start_ref,
                                            const u64* __restrict__ pairs, const float2* __restrict__ tex,
                                            float2* __restrict__ pos_out, float2* __restrict__ vel_out,
                                            AosParticle* __restrict__ aos_out, const float* __restrict__ rho_arr,
                                            uint32_t* __restrict__ defer_bits, uint32_t* __restrict__ worklist,
                                            uint32_t* __restrict__ work_count, float2 (*s_pred)[NBF_ROW],
                                            uint32_t* s_red) {
    const uint32_t tid = threadIdx.x;
    const uint32_t i = blk * FS_BLOCK + tid;
    bool live = i < n;
    if (GENERAL) live = live && ((wave_bits >> (tid >> 6)) & 1u);   // only the waves handed over to the general path
    // lean path: a wave k_density pre-registered is being finished by a general workgroup of this same launch
    const bool pre = !GENERAL && ((wave_bits >> (tid >> 6)) & 1u);
    if (pre) live = false;
    const uint32_t ii = i < n ? i : n - 1;           // dead lanes shadow the last particle, store nothing
    const uint32_t lo_fix = quirk_lo_fix(P, pairs, cs, start_ref);
    const float2 me = pred[ii];
    const float2 mv = vel_s[ii];
    const float2 mrec = rho2[ii];                   // {rho, +-1/rho}; MODE 2: {pressure, 1/rho}
    // own position at the start of the step: the sorted copy, or (pos_by_src) the previous state through the pair's source index
    const float2 p_own = pos_s[P.pos_by_src ? (uint32_t)pairs[ii] : ii];
    const float mrho = MODE == 2 ? 0.0f : mrec.x;
    const bool me_ok = mrec.y > 0.0f;               // this particle's "safe operand" classification (fs_device.h)
    const float pressure = MODE == 2 ? mrec.x : P.pressure_k * (mrho - P.rest_density);      // funcs.wgsl:152-154
    ForceAcc A;
    A.fpx = A.fpy = A.fvx = A.fvy = 0.0f;
    A.seed = ii * 12u + P.frame_time * 69u;                             // compute.wgsl:161
    uint32_t cx, cy;                // (u, v) of the cell-id layout: (x, y) unless the handle is a transposed slab rank
    int32_t cg;
    uv_local(P, me, &cx, &cy, &cg);
    if (P.n_live) {   // slab mode: ghosts (outside the owned columns) are not advanced, and an overlapped step splits the owned
                      // columns between two launches (fs_device.h slab_advances)
        if (!slab_advances(P, cg)) live = false;
    }
    RowRanges R;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        R.lo[r] = 0; R.hi[r] = 0;
        if (live) (void)row_range(P, cs, cx, cy + (uint32_t)(r - 1), lo_fix, &R.lo[r], &R.hi[r]);
        if (R.hi[r] < R.lo[r]) R.hi[r] = R.lo[r];
    }
    uint32_t blo[3], bhi[3];
    bool staged;
    if (P.block_bounds) {
        // the density pass of this step reduced the same ranges over the same 256 particles (a slab launch that advances only
        // some columns zeroes the other lanes' ranges: the stored bounds are then a superset — more is staged, nothing is missed)
        const uint32_t* bb = P.block_bounds + 8u * blk;
        blo[0] = bb[0]; blo[1] = bb[1]; blo[2] = bb[2]; bhi[0] = bb[3]; bhi[1] = bb[4]; bhi[2] = bb[5];
        staged = bhi[0] - blo[0] <= NBF_TILE && bhi[1] - blo[1] <= NBF_TILE && bhi[2] - blo[2] <= NBF_TILE;
    } else {
        staged = block_tile_bounds(R, s_red, blo, bhi, NBF_TILE);
    }
    bool defer = !staged;                            // lean path only; wave-uniform from here on
    if (staged) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
            for (uint32_t j = tid; j < bhi[r] - blo[r]; j += FS_BLOCK) {
                s_pred[r][j] = pred[blo[r] + j];
            }
        __syncthreads();
        const bool long_row = R.hi[0] - R.lo[0] > 32u || R.hi[1] - R.lo[1] > 32u || R.hi[2] - R.lo[2] > 32u;
        if (!__any(long_row))
            defer = force_sweep_masks<MODE, GENERAL>(P, R, blo, ii, me, mv, pressure, vel_s, rho2, &s_pred[0][0], me_ok, A);
        else if (GENERAL)
            force_sweep_chunks<true, MODE>(P, R, blo, ii, me, mv, pressure, pred, vel_s, rho2, &s_pred[0][0], me_ok, A);
        else
            defer = true;
    } else if (GENERAL) {
        force_sweep_chunks<false, MODE>(P, R, blo, ii, me, mv, pressure, pred, vel_s, rho2, &s_pred[0][0], me_ok, A);
    }
    defer = __any(defer);
    if (!GENERAL && defer) {                         // wave-uniform: hand this wave over (late list), write nothing
        if (__builtin_amdgcn_ballot_w64(live) != 0 && (tid & 63u) == 0u) {
            const uint32_t old = atomicOr(&defer_bits[2u * blk + 1u], 1u << (tid >> 6));
            if (old == 0u) worklist[P.n / FS_BLOCK + 8u + atomicAdd(&work_count[1], 1u)] = blk;   // first wave of the block
        }
        return;
    }
    if (!live) return;
    integrate_store<MODE, AOS>(P, i, me, mv, mrec, mrho, p_own, A, cx, cy, tex, pos_out, vel_out, aos_out, rho_arr);
}

// Slab ranks with column-major cell ids (StepParams::transposed): a block's 256 consecutive sorted particles span the cell
// columns [column of its first key, column of its last key].  A launch that advances only the edge columns (or only the interior,
// fs_device.h slab_advances) leaves every block whose span misses its columns after two scalar loads.
__device__ __forceinline__ bool block_may_advance(const StepParams& P, const u64* __restrict__ pairs, uint32_t blk, uint32_t n) {
    if (!P.n_live || !P.transposed) return true;
    const uint32_t i0 = blk * FS_BLOCK;
    const uint32_t i1 = (i0 + FS_BLOCK < n ? i0 + FS_BLOCK : n) - 1u;
    const int32_t cf = (int32_t)((uint32_t)(pairs[i0] >> 32) / P.grid_u) + P.col_origin;
    const int32_t cl = (int32_t)((uint32_t)(pairs[i1] >> 32) / P.grid_u) + P.col_origin;
    if (P.adv_outside)
        return (cf < (int32_t)P.adv_lo && cl >= (int32_t)P.own_lo) || (cl >= (int32_t)P.adv_hi && cf < (int32_t)P.own_hi);
    return cl >= (int32_t)P.adv_lo && cf < (int32_t)P.adv_hi;
}

#define FS_FORCE_ARGS                                                                                                  \
    StepParams P, const float2* __restrict__ pos_s, const float2* __restrict__ vel_s, const float2* __restrict__ pred,  \
        const float2* __restrict__ rho2, const uint32_t* __restrict__ cs, const uint32_t* __restrict__ start_ref,       \
        const u64* __restrict__ pairs, const float2* __restrict__ tex, float2* __restrict__ pos_out,                   \
        float2* __restrict__ vel_out, AosParticle* __restrict__ aos_out, const float* __restrict__ rho_arr,            \
        uint32_t* __restrict__ defer_bits, uint32_t* __restrict__ worklist, uint32_t* __restrict__ work_count

// Lean main kernel: every block once — mask sweep + shared reciprocals only, skipping the waves k_density
// pre-registered.  (Folding the general workgroups into this launch was tried: the kernel then carries the general
// body's spills and scratch set-up and the strict lean path ran 1.7x slower; two kernels on two streams instead.)
template <int MODE, bool AOS>
__global__ __launch_bounds__(FS_BLOCK) __attribute__((amdgpu_waves_per_eu(FS_FORCE_WAVES, FS_FORCE_WAVES))) void k_force(FS_FORCE_ARGS, uint32_t which) {
    __shared__ float2 s_pred[3][NBF_ROW];
    __shared__ uint32_t s_red[24];
    const uint32_t n = P.n_live ? *P.n_live : P.n;
    uint32_t blk;
    if (!xcd_block(P, (n + FS_BLOCK - 1) / FS_BLOCK, &blk)) return;   // uniform: no live particle in this block
    if (!block_may_advance(P, pairs, blk, n)) return;                 // uniform: none of its columns belongs to this launch
    force_block<MODE, AOS, false>(P, blk, n, defer_bits[2u * blk], pos_s, vel_s, pred, rho2, cs, start_ref, pairs, tex,
                                  pos_out, vel_out, aos_out, rho_arr, defer_bits, worklist, work_count, s_pred, s_red);
}

// Edge-first slab step, column-major ids: the lean kernel over the blocks that hold the edge columns only (fs_device.h
// EdgeBlocks), a small fixed grid walking them.
template <int MODE, bool AOS>
__global__ __launch_bounds__(FS_BLOCK) __attribute__((amdgpu_waves_per_eu(FS_FORCE_WAVES, FS_FORCE_WAVES))) void k_force_edge(FS_FORCE_ARGS, uint32_t which) {
    __shared__ float2 s_pred[3][NBF_ROW];
    __shared__ uint32_t s_red[24];
    const uint32_t n = *P.n_live;
    const EdgeBlocks E = edge_blocks(P, cs, n, 0u);
    for (uint32_t t = blockIdx.x; t < edge_block_count(E); t += gridDim.x) {
        const uint32_t blk = edge_block_at(E, t);
        if (block_may_advance(P, pairs, blk, n))         // uniform (the ghost columns' blocks at the very ends)
            force_block<MODE, AOS, false>(P, blk, n, defer_bits[2u * blk], pos_s, vel_s, pred, rho2, cs, start_ref, pairs, tex,
                                          pos_out, vel_out, aos_out, rho_arr, defer_bits, worklist, work_count, s_pred, s_red);
        __syncthreads();                                 // the LDS stage is reused
    }
}

// General kernel: a fixed grid walks one of the two worklists with the complete body.
//   which = 0: the waves k_density pre-registered (long rows / unstaged tiles: dense clusters) — launched on the
//              simulation's second stream so that it runs BESIDE the lean kernel: a general workgroup's latency (a wave
//              alone with 100+ neighbours per particle) hides under the lean work;
//   which = 1: the waves the lean kernel gave up on itself (an operand outside the proven quotient ranges) — after
//              both, on the main stream; usually empty.
// amdgpu_waves_per_eu(5, 5): 94 VGPRs, no spills, no scratch — at 8 waves (64 VGPRs, 52 spilled) even an EMPTY launch
// cost ~17 us for the scratch set-up of its 1024 workgroups (bench window force 0.707 -> 0.690 ms).
#ifndef FS_GENERAL_WAVES
#define FS_GENERAL_WAVES 5
#endif
template <int MODE, bool AOS>
__global__ __launch_bounds__(FS_BLOCK) __attribute__((amdgpu_waves_per_eu(FS_GENERAL_WAVES, FS_GENERAL_WAVES))) void k_force_general(FS_FORCE_ARGS, uint32_t which, uint32_t* __restrict__ hint) {
    __shared__ float2 s_pred[3][NBF_ROW];
    __shared__ uint32_t s_red[24];
    const uint32_t n = P.n_live ? *P.n_live : P.n;
    // how much work the lists held: the host sizes the next steps' grid of this kernel from it (a few steps late,
    // through pinned memory; an idle launch costs what its workgroups cost to come and go)
    if (hint && blockIdx.x == 0 && threadIdx.x == 0) *hint = which == 2u ? work_count[0] + work_count[1] : work_count[which];
    // Which entry a workgroup starts with.  A list shorter than the grid (a small scene, the first dense clusters) would
    // otherwise be worked off by the FIRST workgroups of the grid, neighbours in dispatch order, while most of the chip runs the
    // workgroups that find nothing: grids of 40 k workgroups (sort_policy.h) deal consecutive entries to the 8 XCDs and, inside
    // an XCD, to workgroups five apart (1 M particles: force pass -1.5 us).  Longer lists keep the plain order (consecutive
    // entries are neighbouring blocks: they share their candidates in one XCD's L2).
    const uint32_t J = gridDim.x >> 3, j = blockIdx.x >> 3;
    const bool spread_ok = (gridDim.x & 7u) == 0u && J % 5u == 0u;
    const uint32_t e_spread = spread_ok ? (((j % 5u) * (J / 5u) + j / 5u) << 3) | (blockIdx.x & 7u) : blockIdx.x;
    // which = 0 / 1: one list; which = 2: the pre-registered list, then the late one (the usual single follow-up launch)
    for (uint32_t w = (which == 2u ? 0u : which); w <= (which == 2u ? 1u : which); ++w) {
        const uint32_t count = work_count[w];        // written earlier in the stream (k_density / the lean kernel)
        const uint32_t* list = worklist + (w ? P.n / FS_BLOCK + 8u : 0u);
        for (uint32_t e = count < gridDim.x ? e_spread : blockIdx.x; e < count; e += gridDim.x) {
            const uint32_t blk = list[e];
            if (!block_may_advance(P, pairs, blk, n)) continue;          // uniform
            const uint32_t bits = defer_bits[2u * blk + w];
            force_block<MODE, AOS, true>(P, blk, n, bits, pos_s, vel_s, pred, rho2, cs, start_ref, pairs,
                                         tex, pos_out, vel_out, aos_out, rho_arr, defer_bits, worklist, work_count, s_pred,
                                         s_red);
            __syncthreads();                         // the LDS stage is reused by the next entry
        }
    }
}

// ---- k_force_quad: a SHORT pre-registered list, four lanes per particle (round 4) -------------------------
// The general kernel's time on a short list — a small scene, a slab rank, the first dense clusters of any scene — is the
// latency of ONE wave: 64 particles of 60 - 70 in-radius neighbours each, ~11 000 instructions at the single-wave issue rate
// (40 us per step at 1 M particles, profiles/r04_rejected.md), with most of the chip idle.  Here a deferred wave's 64
// particles are spread over a whole 256-thread workgroup, FOUR LANES PER PARTICLE: the quad scans the particle's candidates
// 64 at a time (16 each), ORs the hits into one 64-bit mask in candidate order, and walks it four neighbours per round — lane
// l evaluates the l-th set bit — after which the four terms are added to the (replicated) sums in lane order by quad
// broadcasts: neighbour order and association are exactly the reference's.  (A lane without a neighbour contributes +0.0f:
// the sums start at +0.0f and a round-to-nearest sum is never -0.0f unless both addends are, so x + 0.0f == x bit for bit.)
// Shared-reciprocal terms only (strict math), native forms (FS_MATH_WGSL_ULP); a pair outside the proven ranges — or a
// coincident pair, whose direction comes from the particle's serial random sequence — sends the wave to the late list, which
// k_force_general takes afterwards.  Candidates are read from global memory: no tile, any row length.
__device__ __forceinline__ uint32_t quad_or(uint32_t x) {
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1 /* quad_perm [1, 0, 3, 2] */, 0xf, 0xf, true);
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E /* quad_perm [2, 3, 0, 1] */, 0xf, 0xf, true);
    return x;
}
template <int K> __device__ __forceinline__ float quad_lane(float x) {      // the value lane K of the quad holds
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), K * 0x55, 0xf, 0xf, true));
}
__device__ __forceinline__ void mask64_clear_top(uint32_t& hi, uint32_t& lo) {
    const uint32_t t = hi ? hi : lo;
    const uint32_t bit = t ? 0x80000000u >> __builtin_clz(t) : 0u;
    if (hi) hi ^= bit; else lo ^= bit;
}
template <int MODE, bool AOS>
__global__ __launch_bounds__(FS_BLOCK) void k_force_quad(FS_FORCE_ARGS, uint32_t* __restrict__ hint) {
    constexpr bool FAST = MODE == 1;
    const uint32_t n = P.n_live ? *P.n_live : P.n;
    const uint32_t count = work_count[0];                // written by k_density earlier in the stream
    if (hint && blockIdx.x == 0 && threadIdx.x == 0) *hint = count + work_count[1];
    const uint32_t tid = threadIdx.x, l = tid & 3u;
    const float lim = P.sqr_radius;
    const uint32_t lbit = 0x80000000u >> l;             // candidate 4 t + l of a chunk: bit 31 - (4 t + l) of its half
    // consecutive work items to workgroups far apart in dispatch order (grids of 40 k workgroups; see k_force_general)
    const uint32_t J = gridDim.x >> 3, jq = blockIdx.x >> 3;
    const uint32_t item0 = ((gridDim.x & 7u) == 0u && J % 5u == 0u) ? ((((jq % 5u) * (J / 5u) + jq / 5u) << 3) | (blockIdx.x & 7u)) : blockIdx.x;
    for (uint32_t item = item0; item < 4u * count; item += gridDim.x) {
        const uint32_t blk = worklist[item >> 2], w = item & 3u;
        if (!((defer_bits[2u * blk] >> w) & 1u)) continue;           // uniform: this wave of the block was not deferred
        if (!block_may_advance(P, pairs, blk, n)) continue;          // uniform
        const uint32_t i = blk * FS_BLOCK + w * 64u + (tid >> 2);
        bool live = i < n;
        const uint32_t ii = live ? i : n - 1u;
        const uint32_t lo_fix = quirk_lo_fix(P, pairs, cs, start_ref);
        const float2 me = pred[ii];
        const float2 mv = vel_s[ii];
        const float2 mrec = rho2[ii];
        const float2 p_own = pos_s[P.pos_by_src ? (uint32_t)pairs[ii] : ii];
        const float mrho = mrec.x;
        const float pressure = P.pressure_k * (mrho - P.rest_density);
        ForceAcc A;
        A.fpx = A.fpy = A.fvx = A.fvy = 0.0f;
        A.seed = 0u;                                      // never drawn from here
        uint32_t cx, cy;
        int32_t cg;
        uv_local(P, me, &cx, &cy, &cg);
        if (P.n_live && !slab_advances(P, cg)) live = false;
        bool bad = live && !FAST && !(mrec.y > 0.0f);     // the particle's own operands are outside the proven ranges
        const wave_mask allm = wm(true);
#pragma unroll 1
        for (int r = 0; r < 3; ++r) {
            uint32_t lo = 0, hi = 0;
            if (live) (void)row_range(P, cs, cx, cy + (uint32_t)(r - 1), lo_fix, &lo, &hi);
            const uint32_t len = hi > lo ? hi - lo : 0u;
#pragma unroll 1
            for (uint32_t c0 = 0; __any(c0 < len); c0 += 64u) {
                // scan: this lane's sixteen candidates of the chunk, c0 + 4 t + l
                uint32_t Mh = 0, Ml = 0;
                {   // all sixteen loads in flight at once: in groups of four every group waited out a full round trip, and the
                    // round trips, not the arithmetic, were this kernel's time
                    float2 q[16];
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const uint32_t c = c0 + 4u * (uint32_t)t + l;
                        q[t] = pred[c < len ? lo + c : ii];
                    }
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const uint32_t c = c0 + 4u * (uint32_t)t + l;
                        const float ox = q[t].x - me.x, oyv = q[t].y - me.y;
                        const float r2 = ox * ox + oyv * oyv;
                        const bool hit = c < len && !(r2 > lim) && !(r == 1 && lo + c == ii);       // compute.wgsl:195,202
                        const uint32_t b = hit ? lbit >> (4 * (t & 7)) : 0u;
                        if (t < 8) Mh |= b; else Ml |= b;
                    }
                }
                Mh = quad_or(Mh); Ml = quad_or(Ml);       // the particle's hits in candidate order, in all four lanes
                const uint32_t jbase = lo + c0;
                // walk: four neighbours per round, lane l the l-th of them; two register sets, refilled in turn
#define FS_QSELECT(has, q, v, d)                                                                                     \
    do {                                                                                                             \
        uint32_t th = Mh, tl = Ml;                                                                                   \
        if (l > 0u) mask64_clear_top(th, tl);                                                                        \
        if (l > 1u) mask64_clear_top(th, tl);                                                                        \
        if (l > 2u) mask64_clear_top(th, tl);                                                                        \
        has = (th | tl) != 0u;                                                                                       \
        const uint32_t kk = th ? (uint32_t)__builtin_clz(th) : 32u + (uint32_t)__builtin_clz(tl | 1u);               \
        const uint32_t j = has ? jbase + kk : ii;                                                                    \
        q = pred[j]; v = vel_s[j]; d = rho2[j];                                                                      \
        mask64_clear_top(Mh, Ml); mask64_clear_top(Mh, Ml); mask64_clear_top(Mh, Ml); mask64_clear_top(Mh, Ml);      \
    } while (0)
#define FS_QROUND(has, q, v, d)                                                                                      \
    do {                                                                                                             \
        ForceTerms T0;                                                                                               \
        if (FAST) {                                                                                                  \
            const float ox = q.x - me.x, oyv = q.y - me.y;                                                           \
            if (has && ox * ox + oyv * oyv == 0.0f) bad = true;          /* coincident: the serial random direction */ \
            uint32_t seed = 0u;                                                                                      \
            T0 = force_terms<true>(P, me, mv, pressure, q, v, d.x, seed);                                            \
        } else {                                                                                                     \
            wave_mask good = 0;                                                                                      \
            T0 = force_terms_shared(P, me, mv, pressure, q, v, d, good);                                             \
            if ((good | ~wm(has)) != allm) bad = true;                   /* wave-uniform */                          \
        }                                                                                                            \
        if (!has) { T0.px = 0.0f; T0.py = 0.0f; T0.vx = 0.0f; T0.vy = 0.0f; }                                        \
        A.fpx += quad_lane<0>(T0.px); A.fpy += quad_lane<0>(T0.py); A.fvx += quad_lane<0>(T0.vx); A.fvy += quad_lane<0>(T0.vy); \
        A.fpx += quad_lane<1>(T0.px); A.fpy += quad_lane<1>(T0.py); A.fvx += quad_lane<1>(T0.vx); A.fvy += quad_lane<1>(T0.vy); \
        A.fpx += quad_lane<2>(T0.px); A.fpy += quad_lane<2>(T0.py); A.fvx += quad_lane<2>(T0.vx); A.fvy += quad_lane<2>(T0.vy); \
        A.fpx += quad_lane<3>(T0.px); A.fpy += quad_lane<3>(T0.py); A.fvx += quad_lane<3>(T0.vx); A.fvy += quad_lane<3>(T0.vy); \
    } while (0)
                bool hA = false, hB = false, hC = false;
                float2 qA, vA, dA, qB, vB, dB, qC, vC, dC;
                FS_QSELECT(hA, qA, vA, dA);
                FS_QSELECT(hB, qB, vB, dB);
                for (;;) {                                 // two rounds' gathers in flight behind the one being evaluated
                    if (!__any(hA)) break;
                    FS_QSELECT(hC, qC, vC, dC);
                    FS_QROUND(hA, qA, vA, dA);
                    if (!__any(hB)) break;
                    FS_QSELECT(hA, qA, vA, dA);
                    FS_QROUND(hB, qB, vB, dB);
                    if (!__any(hC)) break;
                    FS_QSELECT(hB, qB, vB, dB);
                    FS_QROUND(hC, qC, vC, dC);
                }
#undef FS_QROUND
#undef FS_QSELECT
            }
        }
        if (__syncthreads_or(bad ? 1 : 0)) {             // hand the whole wave to the late list, write nothing
            if (tid == 0) {
                const uint32_t old = atomicOr(&defer_bits[2u * blk + 1u], 1u << w);
                if (old == 0u) worklist[P.n / FS_BLOCK + 8u + atomicAdd(&work_count[1], 1u)] = blk;
            }
            continue;
        }
        if (live && l == 0u)
            integrate_store<MODE, AOS>(P, i, me, mv, mrec, mrho, p_own, A, cx, cy, tex, pos_out, vel_out, aos_out, rho_arr);
    }
}

// --------------------------------------------------------------- AoS <-> SoA

__global__ __launch_bounds__(FS_BLOCK) void k_export_aos(uint32_t n, const float2* __restrict__ pos,
                                                         const float2* __restrict__ pred,
                                                         const float2* __restrict__ vel,
                                                         const float* __restrict__ rho,
                                                         const uint32_t* __restrict__ key,
                                                         const u64* __restrict__ pairs,
                                                         const float2* __restrict__ rho2,
                                                         AosParticle* __restrict__ out) {
    const uint32_t i = blockIdx.x * FS_BLOCK + threadIdx.x;
    if (i >= n) return;
    AosParticle a;
    a.position = pos[i]; a.predicted = pred[i]; a.velocity = vel[i]; a.density = rho2 ? rho2[i].x : rho[i];
    a.grid = pairs ? (uint32_t)(pairs[i] >> 32) : key[i];     // after a step the sorted (key, source) pairs hold the keys
    out[i] = a;
}

__global__ __launch_bounds__(FS_BLOCK) void k_import_aos(uint32_t n, const AosParticle* __restrict__ in,
                                                         float2* __restrict__ pos, float2* __restrict__ pred,
                                                         float2* __restrict__ vel, float* __restrict__ rho,
                                                         uint32_t* __restrict__ key) {
    const uint32_t i = blockIdx.x * FS_BLOCK + threadIdx.x;
    if (i >= n) return;
    const AosParticle a = in[i];
    pos[i] = a.position; pred[i] = a.predicted; vel[i] = a.velocity; rho[i] = a.density; key[i] = a.grid;
}

// ------------------------------------------------------- proof kernel for div_const
// Enumerates EVERY f32 x with lo <= |x| <= hi (both signs; lo, hi > 0 given as bit patterns) and
// counts those for which div_const_fast(x, c, y) differs bitwise from the correctly rounded x / c.
__global__ __launch_bounds__(FS_BLOCK) void k_verify_constdiv(float c, float y, uint32_t lo_bits, uint32_t hi_bits,
                                                              uint32_t* __restrict__ mismatches) {
    const uint32_t tid = blockIdx.x * FS_BLOCK + threadIdx.x;
    const uint32_t total_threads = gridDim.x * FS_BLOCK;
    uint32_t bad = 0;
    for (uint64_t b = (uint64_t)lo_bits + tid; b <= (uint64_t)hi_bits; b += total_threads) {
        const float x = __uint_as_float((uint32_t)b);                 // positive floats are ordered like their bits
        bad += __float_as_uint(div_const_fast(x, c, y)) != __float_as_uint(__fdiv_rn(x, c)) ? 1u : 0u;
        bad += __float_as_uint(div_const_fast(-x, c, y)) != __float_as_uint(__fdiv_rn(-x, c)) ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, bad);
}

void launch_verify_constdiv(hipStream_t st, float c, float y, float lo, float hi, uint32_t* mismatches) {
    uint32_t lb, hb;
    memcpy(&lb, &lo, 4);
    memcpy(&hb, &hi, 4);
    hipLaunchKernelGGL(k_verify_constdiv, dim3(256 * 32), dim3(FS_BLOCK), 0, st, c, y, lb, hb, mismatches);
}

// ------------------------------------------------------- proof kernel for rcp_rn_fast / sqrt_rn_fast
// Enumerates EVERY f32 in [lo, hi] (bit patterns; positive) and counts inputs whose lean result
// differs bitwise from the correctly rounded 1.0f / x (which == 0) or __builtin_sqrtf(x) (which == 1).
__global__ __launch_bounds__(FS_BLOCK) void k_verify_unary(int which, uint32_t lo_bits, uint32_t hi_bits,
                                                           uint32_t* __restrict__ mismatches) {
    const uint32_t total_threads = gridDim.x * FS_BLOCK;
    uint32_t bad = 0;
    for (uint64_t b = (uint64_t)lo_bits + blockIdx.x * FS_BLOCK + threadIdx.x; b <= (uint64_t)hi_bits; b += total_threads) {
        const float x = __uint_as_float((uint32_t)b);
        const uint32_t got = __float_as_uint(which == 0 ? rcp_rn_fast(x) : sqrt_rn_fast(x));
        const uint32_t ref = __float_as_uint(which == 0 ? __fdiv_rn(1.0f, x) : sqrt_rn(x));
        bad += got != ref ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, bad);
}

void launch_verify_unary(hipStream_t st, int which, float lo, float hi, uint32_t* mismatches) {
    uint32_t lb, hb;
    memcpy(&lb, &lo, 4);
    memcpy(&hb, &hi, 4);
    hipLaunchKernelGGL(k_verify_unary, dim3(256 * 32), dim3(FS_BLOCK), 0, st, which, lb, hb, mismatches);
}

// ------------------------------------------------------- density-splat image (fluid_shader.wgsl:27-102)
__device__ __forceinline__ float smoothstep_f(float a, float b, float x) {
    float t = __fdiv_rn(x - a, b - a);
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

__global__ __launch_bounds__(FS_BLOCK) void k_render_density(StepParams P, float2 wmin, float2 wmax, uint32_t width,
                                                             uint32_t height, const float2* __restrict__ pred,
                                                             const float2* __restrict__ vel,
                                                             const uint32_t* __restrict__ cs,
                                                             const uint32_t* __restrict__ start_ref,
                                                             const u64* __restrict__ pairs, float4* __restrict__ out) {
    const uint32_t pix = blockIdx.x * FS_BLOCK + threadIdx.x;
    if (pix >= width * height) return;
    const uint32_t i = pix % width, j = pix / width;
    float2 pt;
    pt.x = wmin.x + __fdiv_rn((float)i + 0.5f, (float)width) * (wmax.x - wmin.x);
    pt.y = wmin.y + __fdiv_rn((float)j + 0.5f, (float)height) * (wmax.y - wmin.y);
    const uint32_t lo_fix = quirk_lo_fix(P, pairs, cs, start_ref);
    uint32_t cx, cy;
    xy_local(P, pt, &cx, &cy);
    float density = 0.0f, vfac = 0.0f;
    const float denom = P.sqr_radius / 2.0f;                            // fluid_shader.wgsl:66
    for (int oy = -2; oy < 3; ++oy) {                                   // :39-40 (5x5 cells)
        const uint32_t y = cy + (uint32_t)oy;
        if (y >= P.grid_h) continue;
        const int32_t xl = (int32_t)cx - 2, xh = (int32_t)cx + 3;
        const uint32_t xlo = xl < 0 ? 0u : (uint32_t)xl;
        const uint32_t xhi = xh > (int32_t)P.grid_w ? P.grid_w : (uint32_t)xh;
        if (xlo >= xhi) continue;
        uint32_t a = cs[y * P.grid_w + xlo];
        const uint32_t b = cs[y * P.grid_w + xhi];
        if (a == 0u) a = lo_fix;
        for (uint32_t k = a; k < b; ++k) {
            const float2 q = pred[k];
            const float2 v = vel[k];
            const float ox = q.x - pt.x, oyv = q.y - pt.y;
            const float r2 = ox * ox + oyv * oyv;
            const float contrib = expf(__fdiv_rn(-r2, denom));
            density += contrib;
            vfac += contrib * sqrt_rn(v.x * v.x + v.y * v.y);           // :68
        }
    }
    vfac = vfac * 0.01f;                                                // :79-83
    vfac = __fdiv_rn(logf(1.0f + 5.0f * vfac), logf(1.0f + 5.0f));
    vfac = fminf(fmaxf(vfac, 0.0f), 1.0f);
    const float interior = smoothstep_f(0.5f, 1.5f, density);           // :86
    float edge = smoothstep_f(0.7f, 1.0f, density) - smoothstep_f(1.0f, 1.5f, density);
    edge = edge * (1.0f + vfac * 2.0f);                                 // :89-90
    const float br = (0.0f * (1.0f - vfac) + 1.0f * vfac) * interior;   // mix(blue, red, vfac) * interior, :93
    const float bg = (0.5f * (1.0f - vfac) + 0.0f * vfac) * interior;
    const float bb = (1.0f * (1.0f - vfac) + 0.0f * vfac) * interior;
    out[pix] = make_float4(br + edge, bg + edge, bb + edge, fminf(fmaxf(interior, 0.0f), 1.0f));
}

void launch_render_density(hipStream_t st, const StepParams& P, float2 wmin, float2 wmax, uint32_t width,
                           uint32_t height, const float2* pred, const float2* vel, const uint32_t* cs,
                           const uint32_t* start_ref, const u64* pairs, float4* out) {
    const uint32_t npix = width * height;
    hipLaunchKernelGGL(k_render_density, dim3((npix + FS_BLOCK - 1) / FS_BLOCK), dim3(FS_BLOCK), 0, st, P, wmin, wmax,
                       width, height, pred, vel, cs, start_ref, pairs, out);
}

// ------------------------------------------------------------------ launchers
static inline uint32_t nblk(uint32_t n) { return (n + FS_BLOCK - 1) / FS_BLOCK; }
static inline uint32_t xcd_grid(uint32_t nb, uint32_t c) {      // blocks to launch for xcd_block() (fs_device.h)
    const uint32_t chunks = (nb + (1u << c) - 1u) >> c;
    return (((chunks + 7u) >> 3) << 3) << c;
}

void launch_reorder(hipStream_t st, const StepParams& P, const u64* pairs, const float2* pos_in, const float2* vel_in,
                    float2* pos_s, float2* vel_s, float2* pred_s, uint32_t* key_s, uint32_t* cs, uint32_t* start_ref,
                    void* work, uint32_t* counter, uint32_t work_cap, unsigned long long* safe, uint32_t* force_defer,
                    uint32_t* force_work_count, bool cs_ready) {
    if (cs_ready) {   // counting sort already produced the dense table
        hipLaunchKernelGGL(k_reorder<false>, dim3(nblk(P.n)), dim3(FS_BLOCK), 0, st, P, pairs, pos_in, vel_in, pos_s,
                           vel_s, pred_s, key_s, cs, start_ref, (GapEntry*)work, counter, work_cap, safe, force_defer, force_work_count);
        return;
    }
    hipLaunchKernelGGL(k_reorder<true>, dim3(nblk(P.n)), dim3(FS_BLOCK), 0, st, P, pairs, pos_in, vel_in, pos_s, vel_s,
                       pred_s, key_s, cs, start_ref, (GapEntry*)work, counter, work_cap, safe, force_defer, force_work_count);
    hipLaunchKernelGGL(k_fill_gaps, dim3(1024), dim3(FS_BLOCK), 0, st, cs, (const GapEntry*)work, counter, work_cap);
}

void launch_density(hipStream_t st, const StepParams& P, const float2* pred, const uint32_t* cs,
                    const uint32_t* start_ref, const u64* pairs, const unsigned long long* safe, float* rho, float2* rho2,
                    uint32_t* force_defer, uint32_t* force_work, uint32_t* force_count, uint32_t edge_grid) {
    static const bool no_mass1 = getenv("FS_NO_MASS1") != nullptr;          // A/B: always the general form
    const bool tol = P.fast_math == 2, mass1 = P.mass == 1.0f && !tol && !no_mass1;     // (the tolerance form applies the constant factor once anyway)
#define FS_LAUNCH_DENSITY(K, G)                                                                                        \
    do {                                                                                                               \
        if (tol) hipLaunchKernelGGL((K<true, false>), dim3(G), dim3(FS_BLOCK), 0, st, P, pred, cs, start_ref, pairs, safe, rho, rho2, force_defer, force_work, force_count); \
        else if (mass1) hipLaunchKernelGGL((K<false, true>), dim3(G), dim3(FS_BLOCK), 0, st, P, pred, cs, start_ref, pairs, safe, rho, rho2, force_defer, force_work, force_count); \
        else hipLaunchKernelGGL((K<false, false>), dim3(G), dim3(FS_BLOCK), 0, st, P, pred, cs, start_ref, pairs, safe, rho, rho2, force_defer, force_work, force_count); \
    } while (0)
    if (edge_grid) {   // edge-first slab step: the edge columns' blocks only (k_density_edge)
        FS_LAUNCH_DENSITY(k_density_edge, edge_grid);
        return;
    }
    const uint32_t nb = nblk(P.n), grid = xcd_grid(nb, P.xcd_chunk_log2);
    FS_LAUNCH_DENSITY(k_density, grid);
#undef FS_LAUNCH_DENSITY
}

void launch_force(hipStream_t st, const StepParams& P, const float2* pos_s, const float2* vel_s, const float2* pred,
                  const float2* rho2, const uint32_t* cs, const uint32_t* start_ref, const u64* pairs, const float2* tex,
                  float2* pos_out, float2* vel_out, const float* rho_arr, uint32_t* defer_bits, uint32_t* worklist,
                  uint32_t* work_count, void* aos_out, hipStream_t side, hipEvent_t ev_fork, hipEvent_t ev_join,
                  uint32_t general_grid, uint32_t* general_hint, uint32_t edge_grid, hipEvent_t done, uint32_t quad_entries) {
    const uint32_t nb = nblk(P.n), grid = xcd_grid(nb, P.xcd_chunk_log2);
    hipEvent_t stop_ev = nullptr;      // set for the pass's last launch only
#define FS_LAUNCH_FORCE(K, M, A, G, S, ...)                                                                         \
    hipExtLaunchKernelGGL((K<M, A>), dim3(G), dim3(FS_BLOCK), 0, S, nullptr, stop_ev, 0, P, pos_s, vel_s, pred, rho2, cs, start_ref, pairs, tex, \
                       pos_out, vel_out, (AosParticle*)aos_out, rho_arr, defer_bits, worklist, work_count, __VA_ARGS__)
#define FS_LAUNCH_FORCE_MODE(K, G, S, ...)                                                                          \
    do {                                                                                                            \
        if (P.fast_math == 2) { if (aos_out) FS_LAUNCH_FORCE(K, 2, true, G, S, __VA_ARGS__); else FS_LAUNCH_FORCE(K, 2, false, G, S, __VA_ARGS__); } \
        else if (P.fast_math == 1) { if (aos_out) FS_LAUNCH_FORCE(K, 1, true, G, S, __VA_ARGS__); else FS_LAUNCH_FORCE(K, 1, false, G, S, __VA_ARGS__); } \
        else { if (aos_out) FS_LAUNCH_FORCE(K, 0, true, G, S, __VA_ARGS__); else FS_LAUNCH_FORCE(K, 0, false, G, S, __VA_ARGS__); }      \
    } while (0)
#ifndef FS_GENERAL_GRID
#define FS_GENERAL_GRID 4080u   // 16M, steps 150-250: force 1.175 (1024) -> 1.126 (2048) -> 1.117 ms (4096); steps 10-110 unchanged
#endif
    uint32_t gg = general_grid ? general_grid : FS_GENERAL_GRID;      // the host's choice (engine.hip), else the full grid
    if (gg > nb) gg = nb;
    if (side) {   // fork: the pre-registered waves on the second stream, beside the lean kernel
        (void)hipEventRecord(ev_fork, st);
        (void)hipStreamWaitEvent(side, ev_fork, 0);
        FS_LAUNCH_FORCE_MODE(k_force_general, gg, side, 0u, (uint32_t*)nullptr);
        (void)hipEventRecord(ev_join, side);
    }
    if (edge_grid) FS_LAUNCH_FORCE_MODE(k_force_edge, edge_grid, st, 0u);      // edge-first slab step: the edge columns' blocks only
    else FS_LAUNCH_FORCE_MODE(k_force, grid, st, 0u);
    if (side) {
        (void)hipStreamWaitEvent(st, ev_join, 0);
        stop_ev = done;
        FS_LAUNCH_FORCE_MODE(k_force_general, (nb < 256u ? nb : 256u), st, 1u, (uint32_t*)nullptr);
    } else if (quad_entries != 0u && P.fast_math != 2 && (P.fast_math == 1 || P.share_div)) {
        // a short pre-registered list (the host's view of it, a few steps old): four lanes per particle (k_force_quad), then the
        // late list — what the lean kernel and the quad kernel gave up on — in the general kernel
        uint32_t qg = 8u * quad_entries;                 // 4 work items per entry, twice that for a list that has grown since
        qg = qg < 80u ? 80u : qg > 4080u ? 4080u : (qg + 39u) / 40u * 40u;       // a multiple of 40 (the kernel's item mapping)
        FS_LAUNCH_FORCE_MODE(k_force_quad, qg, st, general_hint);
        stop_ev = done;
        FS_LAUNCH_FORCE_MODE(k_force_general, 240u, st, 1u, (uint32_t*)nullptr);
    } else {
        stop_ev = done;
        FS_LAUNCH_FORCE_MODE(k_force_general, gg, st, 2u, general_hint);      // both lists in one follow-up launch
    }
#undef FS_LAUNCH_FORCE_MODE
#undef FS_LAUNCH_FORCE
}

void launch_export_aos(hipStream_t st, uint32_t n, const float2* pos, const float2* pred, const float2* vel,
                       const float* rho, const uint32_t* key, void* out, const u64* pairs, const float2* rho2) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_export_aos, dim3(nblk(n)), dim3(FS_BLOCK), 0, st, n, pos, pred, vel, rho, key, pairs, rho2,
                       (AosParticle*)out);
}

__global__ __launch_bounds__(FS_BLOCK) void k_keys_from_pairs(uint32_t n, const u64* __restrict__ pairs, uint32_t* __restrict__ key,
                                                              const float2* __restrict__ rho2, float* __restrict__ rho) {
    const uint32_t i = blockIdx.x * FS_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (pairs) key[i] = (uint32_t)(pairs[i] >> 32);
    if (rho2) rho[i] = rho2[i].x;
}
void launch_keys_from_pairs(hipStream_t st, uint32_t n, const u64* pairs, uint32_t* key, const float2* rho2, float* rho) {
    if (n) hipLaunchKernelGGL(k_keys_from_pairs, dim3(nblk(n)), dim3(FS_BLOCK), 0, st, n, pairs, key, rho2, rho);
}

void launch_import_aos(hipStream_t st, uint32_t n, const void* in, float2* pos, float2* pred, float2* vel, float* rho,
                       uint32_t* key) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_import_aos, dim3(nblk(n)), dim3(FS_BLOCK), 0, st, n, (const AosParticle*)in, pos, pred, vel,
                       rho, key);
}

size_t gap_entry_size() { return sizeof(GapEntry); }

void launch_fill_gaps(hipStream_t st, uint32_t* cs, const void* work, const uint32_t* counter, uint32_t work_cap) {
    hipLaunchKernelGGL(k_fill_gaps, dim3(1024), dim3(FS_BLOCK), 0, st, cs, (const GapEntry*)work, counter, work_cap);
}

}  // namespace fsd

// sim3d.hip — 3D extension of the step (27-cell neighbour path).  NOT in the reference (2D
// only); build-defined per SURVEY.md Appendix B.3, normative statement oracle/sph_oracle3d.cpp.
// Same pass chain as 2D: predict+key -> bitonic (key,index) sort -> reorder + dense cell
// starts -> density -> force+integrate.  SoA with 16-byte lanes: pos4 / vel4 / pred4 (xyz,
// pred.w carries the density for the force pass), so a neighbour is two 16-B loads.
// A row of the 27-cell sweep (fixed z,y; x-1..x+1) is ONE contiguous index range, visited
// z outer, y, x inner, index ascending = the oracle's order, so sums are bit-identical.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/fluidsim.h"
#include "fs_kernels.h"
#include "sort_policy.h"

static_assert(sizeof(fs3_particle) == 48, "fs3_particle is 48 bytes");

namespace fsd {

void set_last_error(const std::string& msg);   // engine.hip

struct Params3 {
    uint32_t n, gw, gh, gd, ncell;
    float dt, h, h2;
    float bx, by, bz;            // bounds * 0.5
    float mass, poly6, pressure_k, rest_density, damping, visc_coeff, spiky, visc_k;
    float gx, gy, gz;
    uint32_t frame;
    ConstDiv div_2h3, div_h2;    // exact constant divisions, proven at create (fs_device.h div_const)
    int32_t share_div;           // one reciprocal per denominator + div_by_rcp in the force pass (fs_device.h)
    int32_t handoff;             // k3_density stores its nine 64-bit pass masks per particle, k3_force walks them (no second scan)
    uint32_t xcd_chunk_log2;     // xcd_block3(): blocks per chunk dealt to one XCD
};

// Workgroup -> block of particles for the density / force kernels, XCD-aware as in 2D (fs_device.h xcd_block): the
// hardware deals consecutive workgroup ids round-robin to the 8 XCDs, and a block's nine sweep rows are the rows of the
// blocks 3 (next cell row) and ~310 (next z-plane) away — dealt block by block, EVERY XCD's L2 fetches every row.  Chunks
// of 2^c consecutive blocks per XCD keep the y-neighbour rows in one L2.  Grid: xcd_grid3() blocks.
__device__ __forceinline__ bool xcd_block3(const Params3& P, uint32_t nblocks, uint32_t* logical) {
    const uint32_t c = P.xcd_chunk_log2;
    const uint32_t slot = blockIdx.x >> 3, xcd = blockIdx.x & 7u;
    const uint32_t chunk = ((slot >> c) << 3) | xcd;
    const uint32_t lb = (chunk << c) | (slot & ((1u << c) - 1u));
    *logical = lb;
    return lb < nblocks;
}
static inline uint32_t xcd_grid3(uint32_t nb, uint32_t c) {
    const uint32_t chunks = (nb + (1u << c) - 1u) >> c;
    return (((chunks + 7u) >> 3) << 3) << c;
}

#define B3 256

__device__ __forceinline__ float4 predict3(const Params3& P, float4 p, float4 v) {
    float4 r;
    r.x = p.x + v.x * P.dt; r.y = p.y + v.y * P.dt; r.z = p.z + v.z * P.dt; r.w = 0.0f;
    if (fabsf(r.x) > P.bx) r.x = P.bx * sign_f32(r.x);
    if (fabsf(r.y) > P.by) r.y = P.by * sign_f32(r.y);
    if (fabsf(r.z) > P.bz) r.z = P.bz * sign_f32(r.z);
    return r;
}
__device__ __forceinline__ void cell3(const Params3& P, float4 pt, uint32_t* cx, uint32_t* cy, uint32_t* cz) {
    *cx = f32_to_u32_sat(floorf(__fdiv_rn(pt.x + P.bx, P.h))) + 1u;
    *cy = f32_to_u32_sat(floorf(__fdiv_rn(pt.y + P.by, P.h))) + 1u;
    *cz = f32_to_u32_sat(floorf(__fdiv_rn(pt.z + P.bz, P.h))) + 1u;
}

__global__ __launch_bounds__(B3) void k3_predict_key(Params3 P, const float4* __restrict__ pos,
                                                     const float4* __restrict__ vel, u64* __restrict__ pairs,
                                                     uint32_t* __restrict__ gap_counter) {
    const uint32_t i = blockIdx.x * B3 + threadIdx.x;
    if (i == 0) *gap_counter = 0;
    if (i >= P.n) return;
    uint32_t cx, cy, cz;
    cell3(P, predict3(P, pos[i], vel[i]), &cx, &cy, &cz);
    pairs[i] = ((u64)((cz * P.gh + cy) * P.gw + cx) << 32) | (u64)i;
}

__global__ __launch_bounds__(B3) void k3_reorder(Params3 P, const u64* __restrict__ pairs,
                                                 const float4* __restrict__ pos_in, const float4* __restrict__ vel_in,
                                                 float4* __restrict__ pos_s, float4* __restrict__ vel_s,
                                                 float4* __restrict__ pred, uint32_t* __restrict__ key_s,
                                                 uint32_t* __restrict__ cs, GapEntry* __restrict__ work,
                                                 uint32_t* __restrict__ counter, uint32_t work_cap) {
    const uint32_t i = blockIdx.x * B3 + threadIdx.x;
    if (i >= P.n) return;
    const u64 pr = pairs[i];
    const uint32_t key = (uint32_t)(pr >> 32), src = (uint32_t)pr;
    const float4 p = pos_in[src], v = vel_in[src];
    const float4 pd = predict3(P, p, v);
    // "safe operand" classification, kinematic part (fs_device.h): coordinates and velocity components 0 or >= 2^-53,
    // |v| <= 2^59 — carried as the sign of vel_s.w (the lane is free: a velocity has three components); k3_density
    // finishes it with the density / pressure bounds and leaves +-RN(1/rho) there for the force pass
    const bool ksafe = lo_safe(pd.x) && lo_safe(pd.y) && lo_safe(pd.z) && lo_safe(v.x) && lo_safe(v.y) && lo_safe(v.z) &&
                       fabsf(v.x) <= 0x1p59f && fabsf(v.y) <= 0x1p59f && fabsf(v.z) <= 0x1p59f;
    float4 vs = v;
    vs.w = ksafe ? 1.0f : -1.0f;
    // no sorted copy of the positions: k3_force takes its own particle's position from the previous state through the pair's
    // source index and writes the new state into the spare buffer (16 B / particle less in this HBM-bound pass)
    vel_s[i] = vs; pred[i] = pd; key_s[i] = key;
    const uint32_t kc = key < P.ncell ? key : P.ncell;
    if (i == 0) {
        fill_cells(cs, 0u, kc + 1u, 0u, work, counter, work_cap);
    } else {
        const uint32_t prev = (uint32_t)(pairs[i - 1] >> 32);
        if (key != prev) fill_cells(cs, (prev < P.ncell ? prev : P.ncell) + 1u, kc + 1u, i, work, counter, work_cap);
    }
    if (i == P.n - 1) fill_cells(cs, kc + 1u, P.ncell + 1u, P.n, work, counter, work_cap);
}

// row j in 0..8 -> (oz, oy) = (j/3 - 1, j%3 - 1); false when the row is outside the grid
__device__ __forceinline__ bool row3(const Params3& P, const uint32_t* __restrict__ cs, uint32_t cx, uint32_t cy,
                                     uint32_t cz, int j, uint32_t* lo, uint32_t* hi) {
    const uint32_t y = cy + (uint32_t)(j % 3 - 1), z = cz + (uint32_t)(j / 3 - 1);
    if (y >= P.gh || z >= P.gd) return false;
    const uint32_t xlo = cx - 1u;                         // cx >= 1 always
    uint32_t xhi = cx + 2u;                               // exclusive; cells past the row end do not exist
    if (xhi > P.gw) xhi = P.gw;
    const uint32_t base = (z * P.gh + y) * P.gw;
    *lo = cs[base + xlo];
    *hi = cs[base + xhi];
    return *lo < *hi;
}

// The same row from the particle's KEY (round 3): cells (cx-1 .. cx+1, cy+oy, cz+oz) are the ids key + (oz gh + oy) gw - 1 .. + 2,
// so the density and force passes need no cell coordinates (three IEEE divisions per particle in cell3) — only the stored
// key.  Equivalent to row3 for every reachable key: cell index 0 of every row / plane is padding and always empty
// (coordinates are floor(..) + 1 >= 1), so a row that wraps into the next row or plane reads an empty range exactly where
// row3 says "outside the grid", and ids past the table are cut off here.
__device__ __forceinline__ bool row3_key(const Params3& P, const uint32_t* __restrict__ cs, uint32_t key, int j,
                                         uint32_t* lo, uint32_t* hi) {
    const int32_t off = ((j / 3 - 1) * (int32_t)P.gh + (j % 3 - 1)) * (int32_t)P.gw - 1;      // scalar
    const uint32_t id_lo = key + (uint32_t)off;                       // wraps for a row below the grid: >= ncell
    if (id_lo >= P.ncell) return false;
    const uint32_t id_hi = id_lo + 3u > P.ncell ? P.ncell : id_lo + 3u;
    *lo = cs[id_lo];
    *hi = cs[id_hi];
    return *lo < *hi;
}

__device__ __forceinline__ float dens3(const Params3& P, float4 me, float4 q) {
    const float dx = q.x - me.x, dy = q.y - me.y, dz = q.z - me.z;
    const float r2 = dx * dx + dy * dy + dz * dz;
    float kern = 0.0f;
    if (!(r2 > P.h2)) { const float d = P.h2 - r2; kern = P.poly6 * d * d * d; }
    return P.mass * kern * 1.0f;
}

// Workgroup of the density / force kernels: 256 threads.  -DB3F=64 (one wave per workgroup: no barrier couples waves with
// different neighbour counts, each wave stages its own 10-cell rows) was measured and is SLOWER — density 0.93 -> 1.11 ms,
// force 1.94 -> 2.14 ms at 8 M (profiles/r03_rejected.md): the kernels' waiting is not barrier skew.
#ifndef B3F
#define B3F 256
#endif
#define W3F (B3F / 64)
#if B3F == 256
#ifndef TILE3
#define TILE3 400            // staged candidates per sweep row; one z-plane (3 rows) is staged at a time.  8 M, steps 10-110, strict / tolerance step: 352: 3.30 / 2.70, 384: 3.21 / 2.60, 400: 3.18 / 2.56, 408: 3.18 / 2.56 ms (408 is the most four workgroups per CU have room for)
#endif
#define TILE3_ROW TILE3      // rows 0 and 1 over-read into the next row's stage (masked off), only the last row needs the slack
#else
#define TILE3 96             // a wave's row at rest: 10 cells x 8 particles; longer rows take the unstaged chunked sweep
#define TILE3_ROW TILE3      // rows 0 and 1 over-read into the next row's stage, only the last row needs the slack below
#endif
#define TILE3_PAD 72u        // the wave-uniform scan reads up to the wave's longest row (<= 64) + 3 past a lane's own range
#define TILE3_LDS (3 * TILE3_ROW + TILE3_PAD)
// k3_force stages the neighbours' VELOCITY records {vx, vy, vz, +-1/rho} behind the positions, same row pitch: the walk's
// second fetch is then an LDS read at a constant offset from the first instead of a 16-byte gather per neighbour (with the
// masks handed over the kernel was bound by exactly those gathers: waves parked 65 - 79 %, profiles/r03_counters_3d*.md).
// 19.6 + 18.4 KB per workgroup: four workgroups (16 waves) per CU.
#ifndef FS3_STAGE_VEL
#define FS3_STAGE_VEL 1
#endif
#ifndef FS3_CHUNK_BATCH
#define FS3_CHUNK_BATCH 4    // 32-candidate chunks scanned per walk in the chunked sweep (2 .. 4)
#endif
#define TILE3_VEL_OFF (TILE3_LDS * 16u)          // bytes from a staged position to the same candidate's velocity
#define TILE3_FORCE_LDS (TILE3_LDS + (FS3_STAGE_VEL ? 3 * TILE3_ROW : 0))
typedef unsigned long long u64m;

// ---- pass masks of one staged z-plane -----------------------------------------------------------------------
// A 3D row of three cells holds ~24 candidates at rest (8 particles per cell) and passes 32 as soon as the column
// compresses, so the pass masks are 64 bits, filled as two 32-bit shift registers: v_cmp + one v_addc_co per candidate shift
// `!(r2 > h^2)` in (see kernels_step.hip force_sweep_masks for the 2D form).  Candidate t of a row ends up at bit 63 - t.
// Valid for waves whose three rows hold <= 64 candidates each; the rows are read from the LDS stage `s_flat`
// (TILE3_ROW entries per row).  Both the density and the force pass need exactly these masks: k3_density computes
// them, walks them for its own sum and (Params3::handoff) stores them — 72 B per particle — so that k3_force does not
// scan the 216 candidates a second time (~2 600 of its ~9 900 VALU instructions per wave).
__device__ __forceinline__ void shift_in_not_greater32(uint32_t& mask, float r2, float lim) {
    asm("v_cmp_nlt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(r2), "s"(lim) : "vcc");
}
__device__ __forceinline__ void scan3_plane(const Params3& P, const RowRanges& R, const uint32_t* blo, float4 me,
                                            const float4* s_flat, u64m m[3], uint32_t la[3]) {
    const float lim = P.h2;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint32_t len = R.hi[r] - R.lo[r];                           // <= 64 (caller)
        la[r] = (uint32_t)r * TILE3_ROW + (len ? R.lo[r] - blo[r] : 0u);
        const float4* base = s_flat + la[r];
        uint32_t mlo = 0, mhi = 0, t = 0;
        // Two 32-bit shift registers, one v_addc_co per candidate (the 64-bit form needs two): candidates 0 .. 31 go
        // through `mhi`, the rest through `mlo`; t is wave-uniform, so the switch is a scalar branch.
        for (; t < 32u && __any(t < len); t += 4u) {
            const float4 q0 = base[t], q1 = base[t + 1u], q2 = base[t + 2u], q3 = base[t + 3u];
            const float4 qq[4] = {q0, q1, q2, q3};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ox = qq[u].x - me.x, oy = qq[u].y - me.y, oz = qq[u].z - me.z;
                shift_in_not_greater32(mhi, ox * ox + oy * oy + oz * oz, lim);
            }
        }
        const uint32_t ta = t;                                            // <= 32: candidates that went through mhi
        for (; __any(t < len); t += 4u) {
            const float4 q0 = base[t], q1 = base[t + 1u], q2 = base[t + 2u], q3 = base[t + 3u];
            const float4 qq[4] = {q0, q1, q2, q3};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ox = qq[u].x - me.x, oy = qq[u].y - me.y, oz = qq[u].z - me.z;
                shift_in_not_greater32(mlo, ox * ox + oy * oy + oz * oz, lim);
            }
        }
        {   // candidate k sits at bit 63 - k: left-align each half, keep the lane's own len candidates
            const uint32_t hi32 = ta ? mhi << (32u - ta) : 0u;
            const uint32_t lo32 = t > ta ? mlo << (32u - (t - ta)) : 0u;
            u64m mask = ((u64m)hi32 << 32) | lo32;
            mask &= len ? ~0ull << (64u - len) : 0ull;
            m[r] = mask;
        }
    }
}
// Is the mask form available for this wave's plane?  k3_density and k3_force must agree, so both call this with the
// RowRanges / block bounds they derive from the same cell table.
__device__ __forceinline__ bool plane_masked(const RowRanges& R, bool fit) {
    const bool long_row = R.hi[0] - R.lo[0] > 64u || R.hi[1] - R.lo[1] > 64u || R.hi[2] - R.lo[2] > 64u;
    return fit && !__any(long_row);
}

// Rows of 65 .. 128 candidates (the compressing column: 5 % of the waves at step 60, 12 - 14 % from step 80 on,
// tools/rows3d_stats.py): the same hand-off with TWO 64-bit words per row — candidates 0 .. 63 in `hi` (stored in
// masks[0 .. 9n)), 64 .. 127 in `lo` (masks[9n .. 18n)).  plane_class(): 1 = every row of the wave <= 64 (plane_masked),
// 2 = every row <= 128, 0 = the chunked sweep.  k3_density and k3_force call it with the same ranges.
#ifndef FS3_MASK128
#define FS3_MASK128 1
#endif
__device__ __forceinline__ int plane_class(const RowRanges& R, bool fit) {
    const uint32_t l0 = R.hi[0] - R.lo[0], l1 = R.hi[1] - R.lo[1], l2 = R.hi[2] - R.lo[2];
    const uint32_t mx = l0 > l1 ? (l0 > l2 ? l0 : l2) : (l1 > l2 ? l1 : l2);
    if (!fit) return 0;
    if (!__any(mx > 64u)) return 1;
    return (FS3_MASK128 && !__any(mx > 128u)) ? 2 : 0;
}
// One row of up to 128 candidates into four 32-bit shift registers (t is wave-uniform: the switches are scalar branches).
__device__ __forceinline__ void scan3_row128(const Params3& P, const float4* base, uint32_t len, float4 me, u64m* hi, u64m* lo) {
    const float lim = P.h2;
    uint32_t w0 = 0u, w1 = 0u, w2 = 0u, w3 = 0u, t = 0u;
#define FS3_SCAN32(W, LIMIT)                                                                                          \
    for (; t < (LIMIT) && __any(t < len); t += 4u) {                                                                  \
        const float4 q0 = base[t], q1 = base[t + 1u], q2 = base[t + 2u], q3 = base[t + 3u];                           \
        const float4 qq[4] = {q0, q1, q2, q3};                                                                        \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                               \
            const float ox = qq[u].x - me.x, oy = qq[u].y - me.y, oz = qq[u].z - me.z;                                \
            shift_in_not_greater32(W, ox * ox + oy * oy + oz * oz, lim);                                              \
        }                                                                                                             \
    }
    FS3_SCAN32(w0, 32u) const uint32_t t0 = t;
    FS3_SCAN32(w1, 64u) const uint32_t t1 = t;
    FS3_SCAN32(w2, 96u) const uint32_t t2 = t;
    FS3_SCAN32(w3, 128u)
#undef FS3_SCAN32
    // candidate c of the row sits at bit 31 - (c & 31) of word c / 32: left-align each word by the candidates it took
    const uint32_t a0 = t0 ? w0 << (32u - t0) : 0u, a1 = t1 > t0 ? w1 << (32u - (t1 - t0)) : 0u;
    const uint32_t a2 = t2 > t1 ? w2 << (32u - (t2 - t1)) : 0u, a3 = t > t2 ? w3 << (32u - (t - t2)) : 0u;
    u64m h = ((u64m)a0 << 32) | a1, l = ((u64m)a2 << 32) | a3;
    h &= len >= 64u ? ~0ull : (len ? ~0ull << (64u - len) : 0ull);
    l &= len > 64u ? ~0ull << (128u - len) : 0ull;      // len <= 128
    *hi = h; *lo = l;
}

__device__ __forceinline__ float dens3_tol(const Params3& P, float4 me, float4 q, float acc) {
    const float dx = q.x - me.x, dy = q.y - me.y, dz = q.z - me.z;
    const float r2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
    const float t = fmaxf(P.h2 - r2, 0.0f);                               // NaN candidate: contributes nothing
    return __builtin_fmaf(t * t, t, acc);
}

// The 27-cell sweep runs plane by plane (z outer): per plane the workgroup's three row ranges are
// staged into LDS with coalesced loads (fs_device.h block_tile_bounds).  Waves whose rows fit the 64-bit masks
// scan the plane into masks and add the terms of the set bits (row 0, 1, 2, ascending: the oracle's order — the
// candidates outside the radius contribute +0 there, which changes no bit of a non-negative sum); other waves loop
// over their candidates directly.  MODE 2 (FS_MATH_TOLERANCE): FMA terms, the constant applied once.
#ifndef FS3_DENSITY_WAVES
#define FS3_DENSITY_WAVES 0   // > 0: pin the register budget (A/B: tools/ab_variant3d.py)
#endif
#if FS3_DENSITY_WAVES > 0
#define FS3_DENSITY_ATTR __attribute__((amdgpu_waves_per_eu(FS3_DENSITY_WAVES, FS3_DENSITY_WAVES)))
#else
#define FS3_DENSITY_ATTR
#endif
template <int MODE>
__global__ __launch_bounds__(B3F) FS3_DENSITY_ATTR void k3_density(Params3 P, float4* __restrict__ pred, const uint32_t* __restrict__ cs,
                                                 float4* __restrict__ vel_s, u64m* __restrict__ masks,
                                                 const uint32_t* __restrict__ key_s) {
    __shared__ float4 s_pred[TILE3_LDS + (FS3_MASK128 ? 64 : 0)];   // a 128-candidate scan reads up to 131 entries from a range start
    __shared__ uint32_t s_red[24];
    uint32_t blk;
    if (!xcd_block3(P, (P.n + B3F - 1) / B3F, &blk)) return;       // uniform
    const uint32_t i = blk * B3F + threadIdx.x;
    const bool live = i < P.n;
    const float4 me = pred[live ? i : P.n - 1];
    const uint32_t key = key_s[live ? i : P.n - 1];
    float rho = 0.0f;
    uint32_t lo9[9], hi9[9];        // all 18 cell-start lookups up front: independent loads, one latency
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        lo9[j] = 0; hi9[j] = 0;
        if (live && !row3_key(P, cs, key, j, &lo9[j], &hi9[j])) { lo9[j] = 0; hi9[j] = 0; }
    }
#pragma unroll 1
    for (int plane = 0; plane < 3; ++plane) {
        RowRanges R;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            R.lo[r] = plane == 0 ? lo9[r] : plane == 1 ? lo9[3 + r] : lo9[6 + r];
            R.hi[r] = plane == 0 ? hi9[r] : plane == 1 ? hi9[3 + r] : hi9[6 + r];
        }
        uint32_t blo[3], bhi[3];
        const bool fit = block_tile_bounds<W3F>(R, s_red, blo, bhi, TILE3);
        if (fit) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
                for (uint32_t j = threadIdx.x; j < bhi[r] - blo[r]; j += B3F) s_pred[r * TILE3_ROW + j] = pred[blo[r] + j];
            __syncthreads();
            const int pclass = plane_class(R, fit);
            if (pclass == 2) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const uint32_t len = R.hi[r] - R.lo[r];
                    const float4* base = s_pred + ((uint32_t)r * TILE3_ROW + (len ? R.lo[r] - blo[r] : 0u));
                    u64m mh, ml;
                    scan3_row128(P, base, len, me, &mh, &ml);
                    if (P.handoff && live) {
                        masks[(size_t)(plane * 3 + r) * P.n + i] = mh;
                        masks[(size_t)(9 + plane * 3 + r) * P.n + i] = ml;
                    }
                    while (mh) {
                        const uint32_t t = (uint32_t)__builtin_clzll(mh);
                        mh ^= 0x8000000000000000ull >> t;
                        if (MODE == 2) rho = dens3_tol(P, me, base[t], rho);
                        else rho += dens3(P, me, base[t]);
                    }
                    while (ml) {
                        const uint32_t t = (uint32_t)__builtin_clzll(ml);
                        ml ^= 0x8000000000000000ull >> t;
                        if (MODE == 2) rho = dens3_tol(P, me, base[64u + t], rho);
                        else rho += dens3(P, me, base[64u + t]);
                    }
                }
            } else if (pclass == 1) {
                u64m m[3];
                uint32_t la[3];
                scan3_plane(P, R, blo, me, s_pred, m, la);
                if (P.handoff && live) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) masks[(size_t)(plane * 3 + r) * P.n + i] = m[r];
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    u64m mm = m[r];
                    const float4* base = s_pred + la[r];
                    while (mm) {
                        const uint32_t t = (uint32_t)__builtin_clzll(mm);
                        mm ^= 0x8000000000000000ull >> t;
                        if (MODE == 2) rho = dens3_tol(P, me, base[t], rho);
                        else rho += dens3(P, me, base[t]);
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const bool any = R.lo[r] < R.hi[r];
                    const uint32_t hi = any ? R.hi[r] - blo[r] : 0u;
                    uint32_t k = any ? R.lo[r] - blo[r] : 0u;
                    const float4* sp = s_pred + r * TILE3_ROW;
                    if (MODE == 2) { for (; k < hi; ++k) rho = dens3_tol(P, me, sp[k], rho); continue; }
                    for (; k + 4u <= hi; k += 4u) {
                        const float t0 = dens3(P, me, sp[k]), t1 = dens3(P, me, sp[k + 1u]);
                        const float t2 = dens3(P, me, sp[k + 2u]), t3 = dens3(P, me, sp[k + 3u]);
                        rho += t0; rho += t1; rho += t2; rho += t3;
                    }
                    for (; k < hi; ++k) rho += dens3(P, me, sp[k]);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r)
                for (uint32_t k = R.lo[r]; k < R.hi[r]; ++k) {
                    if (MODE == 2) rho = dens3_tol(P, me, pred[k], rho);
                    else rho += dens3(P, me, pred[k]);
                }
        }
        __syncthreads();     // the next plane reuses s_pred / s_red
    }
    if (!live) return;
    if (MODE == 2) rho = rho * (P.mass * P.poly6);                 // sum of (h2 - r2)^3 -> density
    rho = fmaxf(rho, 1.19209290e-07f);
    rho = fmaxf(rho, 0.1f);
    reinterpret_cast<float*>(pred + i)[3] = rho;                   // pred.w <- density (other lanes read .xyz only)
    // vel_s.w <- +-RN(1/rho): what the force pass divides by, once per particle instead of once per pair; positive only
    // when every operand this particle brings to a pair is inside the proven quotient ranges (fs_device.h)
    float* yw = reinterpret_cast<float*>(vel_s + i) + 3;
    const float y = (P.share_div && rho <= FS_RCP_HI) ? rcp_rn_fast(rho) : __fdiv_rn(1.0f, rho);
    if (MODE == 2) { *yw = y; return; }                            // tolerance mode: no classification, the force pass has no exact quotients
    const bool ksafe = *yw > 0.0f;
    const float press = P.pressure_k * (rho - P.rest_density);     // the expression the force pass evaluates
    const bool ok = ksafe && rho <= FS_RCP_HI && fabsf(press) <= FS_PRESSURE_HI;
    *yw = ok ? y : -y;
}

struct Acc3 { float px, py, pz, vx, vy, vz; uint32_t seed; };
struct Terms3 { float px, py, pz, vx, vy, vz; };

__device__ __forceinline__ Terms3 terms3(const Params3& P, float4 me, float4 mv, float pressure, float4 q, float4 nv,
                                         uint32_t& seed) {
    const float h = P.h;
    const float ox = q.x - me.x, oy = q.y - me.y, oz = q.z - me.z;
    const float r2 = ox * ox + oy * oy + oz * oz;
    const float dst = sqrt_rn(r2);
    float dx, dy, dz;
    if (dst == 0.0f) {
        const float rx = rand_f32(&seed), ry = rand_f32(&seed), rz = rand_f32(&seed);
        const float len = sqrt_rn(rx * rx + ry * ry + rz * rz);
        dx = __fdiv_rn(rx, len); dy = __fdiv_rn(ry, len); dz = __fdiv_rn(rz, len);
    } else {
        dx = __fdiv_rn(ox, dst); dy = __fdiv_rn(oy, dst); dz = __fdiv_rn(oz, dst);
    }
    const float nrho = q.w;
    const float npress = P.pressure_k * (nrho - P.rest_density);
    const float kern = (dst <= h) ? (-(h - dst)) * P.spiky : 0.0f;
    const float shared = (pressure + npress) * 0.5f;
    float kv = 0.0f;
    if (dst <= h) {
        if (dst == 0.0f) {
            kv = P.visc_k;
        } else {
            float a, b;
            if (dst < 9.5367431640625e-07f) {            // 2^-20: outside the proven range, true division
                a = __fdiv_rn(-(dst * dst * dst), P.div_2h3.c);
                b = __fdiv_rn(dst * dst, P.div_h2.c);
            } else {
                a = div_const(P.div_2h3, -(dst * dst * dst));
                b = div_const(P.div_h2, dst * dst);
            }
            kv = P.visc_k * (a + b + (__fdiv_rn(h, 2.0f * dst)) - 1.0f);
        }
    }
    Terms3 T;
    T.px = __fdiv_rn(dx * kern * shared, nrho);
    T.py = __fdiv_rn(dy * kern * shared, nrho);
    T.pz = __fdiv_rn(dz * kern * shared, nrho);
    T.vx = __fdiv_rn(nv.x - mv.x, nrho) * kv;
    T.vy = __fdiv_rn(nv.y - mv.y, nrho) * kv;
    T.vz = __fdiv_rn(nv.z - mv.z, nrho) * kv;
    return T;
}

// The same terms with one reciprocal per denominator (dst, neighbour density) and div_by_rcp() for
// the ten quotients — bit-identical to terms3() for every lane whose bit stays set in `good`
// (fs_device.h: the proven ranges).  No PRNG / tiny-distance path: those lanes clear their bit and
// the caller re-evaluates the pair with terms3() for the whole wave.
__device__ __forceinline__ wave_mask num_lo_ok3(float a) { return wm(fabsf(a) >= 0x1p-76f) | wm(a == 0.0f); }   // NaN: 0
__device__ __forceinline__ Terms3 terms3_shared(const Params3& P, float4 me, float4 mv, float pressure, float4 q,
                                                float4 nv, wave_mask& good) {
    const float h = P.h;
    const float ox = q.x - me.x, oy = q.y - me.y, oz = q.z - me.z;
    const float r2 = ox * ox + oy * oy + oz * oz;
    const float nrho = q.w;                                            // >= 0.1 (k3_density)
    const float yrho = nv.w;                                           // +-RN(1/nrho): the sign is the neighbour's classification
    good = wm(r2 >= FS_SQRT_LO) & wm(yrho > 0.0f);
    const float dst = sqrt_rn_fast(r2);                                // r2 <= h*h: the scan admitted it
    const float ydst = rcp_rn_fast(dst);
    const float dx = div_by_rcp(ox, dst, ydst), dy = div_by_rcp(oy, dst, ydst), dz = div_by_rcp(oz, dst, ydst);
    const float npress = P.pressure_k * (nrho - P.rest_density);
    const bool inside = dst <= h;
    const float kern = inside ? (-(h - dst)) * P.spiky : 0.0f;
    const float shared = (pressure + npress) * 0.5f;
    const float apx = dx * kern * shared, apy = dy * kern * shared, apz = dz * kern * shared;
    const float dvx = nv.x - mv.x, dvy = nv.y - mv.y, dvz = nv.z - mv.z;
    // both particles safe => every numerator is 0 or in [2^-76, 2^60] except the lower bound of the three pressure
    // numerators (a product of three factors can be tiny without any factor being unusual)
    good &= num_lo_ok3(apx) & num_lo_ok3(apy) & num_lo_ok3(apz);
    const float a = div_const_fast(-(dst * dst * dst), P.div_2h3.c, P.div_2h3.y);   // share_div implies both proofs
    const float b = div_const_fast(dst * dst, P.div_h2.c, P.div_h2.y);
    const float hq = div_by_rcp(h, 2.0f * dst, 0.5f * ydst);
    const float kv = inside ? P.visc_k * (a + b + hq - 1.0f) : 0.0f;
    Terms3 T;
    T.px = div_by_rcp(apx, nrho, yrho); T.py = div_by_rcp(apy, nrho, yrho); T.pz = div_by_rcp(apz, nrho, yrho);
    T.vx = div_by_rcp(dvx, nrho, yrho) * kv; T.vy = div_by_rcp(dvy, nrho, yrho) * kv; T.vz = div_by_rcp(dvz, nrho, yrho) * kv;
    return T;
}

__device__ __forceinline__ void acc3_add(Acc3& A, const Terms3& T) {
    A.px += T.px; A.py += T.py; A.pz += T.pz; A.vx += T.vx; A.vy += T.vy; A.vz += T.vz;
}

__device__ __forceinline__ Terms3 pair3(const Params3& P, float4 me, float4 mv, float pressure, float4 q, float4 nv,
                                        Acc3& A) {
    wave_mask good = 0;
    Terms3 T;
    if (P.share_div) { T = terms3_shared(P, me, mv, pressure, q, nv, good); good &= wm(mv.w > 0.0f); }   // + the lane's own classification
    if (good != wm(true)) T = terms3(P, me, mv, pressure, q, nv, A.seed);      // rare, wave-uniform
    return T;
}

// (the general sweep, sweep3_chunks, follows the mask sweep below: it shares its helpers)

#ifndef FS3_FORCE_WAVES
#define FS3_FORCE_WAVES (FS3_STAGE_VEL ? 4 : 7)   // with the velocity stage the LDS allows 4 waves per SIMD: take their registers.  Without it: round 2 (own scan): 4: 2.265, 5: 2.232, 6: 2.215 ms.  Round 3 (masks handed over by k3_density), steps 10-50 / 50-110: 5: 1.42 / 2.48, 6: 1.45 / 2.45, 7: 1.38 / 2.38 ms
#endif

// ---- tolerance mode (fs3_create_ex math_mode = FS_MATH_TOLERANCE): the pressure and viscosity terms of one in-radius
// neighbour merged algebraically, as kernels_step.hip force_accum_tol does in 2D: one v_rsq_f32, fused multiply-adds,
// 1/rho_j from the density pass (vel_s.w), ~32 issue slots per pair instead of ~95.  Coincident particles keep the
// oracle's xorshift direction.
struct Tol3 { float cP, c3, c2, hh, kp0; };
__device__ __forceinline__ Tol3 tol3_consts(const Params3& P) {
    Tol3 C;
    const float h = P.h;
    C.cP = -0.5f * P.spiky;
    C.c3 = -1.0f / (2.0f * h * h * h);
    C.c2 = 1.0f / (h * h);
    C.hh = 0.5f * h;
    C.kp0 = -P.pressure_k * P.rest_density;           // pressure_j = fma(k, rho_j, kp0)
    return C;
}
__device__ __forceinline__ void accum3_tol(const Params3& P, const Tol3& C, float4 me, float4 mv, float pressure, float4 q,
                                           float4 nv, Acc3& A) {
    const float ox = q.x - me.x, oy = q.y - me.y, oz = q.z - me.z;
    const float r2 = __builtin_fmaf(ox, ox, __builtin_fmaf(oy, oy, oz * oz));
    float dx = ox, dy = oy, dz = oz, inv, dst;
    if (r2 == 0.0f) {                                                   // rare: the PRNG direction
        const float rx = rand_f32(&A.seed), ry = rand_f32(&A.seed), rz = rand_f32(&A.seed);
        const float il = __builtin_amdgcn_rsqf(__builtin_fmaf(rx, rx, __builtin_fmaf(ry, ry, rz * rz)));
        dx = rx * il; dy = ry * il; dz = rz * il;
        dst = 0.0f; inv = 1.0f;
    } else {
        inv = __builtin_amdgcn_rsqf(r2);
        dst = r2 * inv;
    }
    const float yrho = fabsf(nv.w);                                     // 1 / rho_j
    const float pj = __builtin_fmaf(P.pressure_k, q.w, C.kp0);
    const float w = fmaxf(P.h - dst, 0.0f);
    const float coefP = (w * C.cP) * (pressure + pj) * yrho * inv;
    float u = __builtin_fmaf(C.c3, dst, C.c2);
    u = __builtin_fmaf(u, r2, -1.0f);
    u = r2 == 0.0f ? 1.0f : __builtin_fmaf(C.hh, inv, u);
    const float kvv = u * (P.visc_k * yrho);
    A.px = __builtin_fmaf(dx, coefP, A.px); A.py = __builtin_fmaf(dy, coefP, A.py); A.pz = __builtin_fmaf(dz, coefP, A.pz);
    A.vx = __builtin_fmaf(nv.x - mv.x, kvv, A.vx); A.vy = __builtin_fmaf(nv.y - mv.y, kvv, A.vy); A.vz = __builtin_fmaf(nv.z - mv.z, kvv, A.vz);
}
template <int MODE>
__device__ __forceinline__ void pair3_accum(const Params3& P, const Tol3& C, float4 me, float4 mv, float pressure, float4 q,
                                            float4 nv, Acc3& A) {
    if (MODE == 2) accum3_tol(P, C, me, mv, pressure, q, nv, A);
    else acc3_add(A, pair3(P, me, mv, pressure, q, nv, A));
}

// Mask sweep of one staged z-plane (see kernels_step.hip force_sweep_masks): every lane walks the set bits of its three
// 64-bit pass masks, row 0, 1, 2, ascending — the oracle's visiting order.  The masks come from k3_density
// (Params3::handoff, `masks` != nullptr: three coalesced 8-byte loads) or from a scan of the staged plane.
// `self_plane`: the lane's own particle sits in row 1 of the middle plane and is skipped (k != i).
// The walk shared by the 64-bit and the 128-bit mask sweeps: three mask words with the LDS index (and, without the velocity
// stage, the global index) of their first candidate, consumed in order.
template <int MODE>
__device__ __forceinline__ void walk3(const Params3& P, const Tol3& C, const u64m* m, const uint32_t* la, const uint32_t* gl,
                                      float4 me, float4 mv, float pressure, const float4* __restrict__ vel_s,
                                      const float4* s_flat, Acc3& A) {
    // The three masks are walked as a shift register (round 3): `cur` is the mask being consumed with its LDS / global
    // bases, (n1, n2) wait behind it.  Empty masks are squeezed out first, so "cur == 0 -> pull n1" is all a refill ever
    // needs and the per-neighbour bit extraction touches ONE 64-bit mask and ONE pair of bases (the round-2 form selected
    // among three masks and six bases for every neighbour: ~40 instructions, now ~23).  Row order 0, 1, 2 is kept.
    u64m cur = m[0], n1 = m[1], n2 = m[2];
    uint32_t lac = la[0] << 4, la_1 = la[1] << 4, la_2 = la[2] << 4, loc = gl[0], lo_1 = gl[1], lo_2 = gl[2];   // la* in bytes
    if (n1 == 0ull) { n1 = n2; la_1 = la_2; lo_1 = lo_2; n2 = 0ull; }
    if (cur == 0ull) { cur = n1; lac = la_1; loc = lo_1; n1 = n2; la_1 = la_2; lo_1 = lo_2; n2 = 0ull; }
    // Software-pipelined (as in the 2D kernel): the LDS read and the velocity gather of later neighbours are issued
    // before the terms of neighbour k are evaluated — two neighbours ahead (k+1 and k+2: three slots refilled in turn, the
    // loop unrolled by three so no value is moved).  With the velocity stage the kernel runs at 4 waves per SIMD and has the
    // registers for it; the one-deep form it replaced, and the measurements at 7 waves, are in profiles/r03_rejected.md.
#define FS3_FETCH(have, qn, vn)                                                                                      \
    do {                                                                                                             \
        have = cur != 0ull;                                                                                          \
        if (have) {                                                                                                  \
            const uint32_t t = (uint32_t)__builtin_clzll(cur);                                                       \
            cur ^= 0x8000000000000000ull >> t;                                                                       \
            qn = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_flat) + (lac + (t << 4)));         \
            /* 32-bit byte offset from the SGPR base (n <= 2^28) instead of 64-bit address arithmetic */             \
            if (FS3_STAGE_VEL) vn = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_flat) + (lac + (t << 4)) + TILE3_VEL_OFF); \
            else vn = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(vel_s) + ((loc + t) << 4));     \
            if (cur == 0ull) { cur = n1; lac = la_1; loc = lo_1; n1 = n2; la_1 = la_2; lo_1 = lo_2; n2 = 0ull; }     \
        }                                                                                                            \
    } while (0)
    float4 qA = make_float4(0.0f, 0.0f, 0.0f, 0.0f), vA = qA, qB = qA, vB = qA, qC = qA, vC = qA;
    bool hA = false, hB = false, hC = false;
    FS3_FETCH(hA, qA, vA);
    FS3_FETCH(hB, qB, vB);
    FS3_FETCH(hC, qC, vC);
    for (;;) {       // a slot is refilled right after its neighbour's terms: two bodies later it is consumed
        if (!__any(hA)) break;
        { const bool cv = hA; const float4 q0 = qA, v0 = vA; if (cv) pair3_accum<MODE>(P, C, me, mv, pressure, q0, v0, A); }
        FS3_FETCH(hA, qA, vA);
        if (!__any(hB)) break;
        { const bool cv = hB; const float4 q0 = qB, v0 = vB; if (cv) pair3_accum<MODE>(P, C, me, mv, pressure, q0, v0, A); }
        FS3_FETCH(hB, qB, vB);
        if (!__any(hC)) break;
        { const bool cv = hC; const float4 q0 = qC, v0 = vC; if (cv) pair3_accum<MODE>(P, C, me, mv, pressure, q0, v0, A); }
        FS3_FETCH(hC, qC, vC);
    }
#undef FS3_FETCH
}

template <int MODE>
__device__ __forceinline__ void sweep3_masks(const Params3& P, const Tol3& C, const RowRanges& R, const uint32_t* blo, bool self_plane,
                                             uint32_t ii, float4 me, float4 mv, float pressure,
                                             const float4* __restrict__ vel_s, const float4* s_flat,
                                             const u64m* __restrict__ masks, Acc3& A) {
    u64m m[3];
    uint32_t la[3];
    if (masks) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const uint32_t len = R.hi[r] - R.lo[r];
            la[r] = (uint32_t)r * TILE3_ROW + (len ? R.lo[r] - blo[r] : 0u);
            m[r] = masks[(size_t)r * P.n + ii];                            // all-zero for lanes past the end (never written: masked below)
            m[r] &= len ? ~0ull << (64u - len) : 0ull;
        }
    } else {
        scan3_plane(P, R, blo, me, s_flat, m, la);
    }
    if (self_plane && ii - R.lo[1] < R.hi[1] - R.lo[1]) m[1] &= ~(0x8000000000000000ull >> (ii - R.lo[1]));
    walk3<MODE>(P, C, m, la, R.lo, me, mv, pressure, vel_s, s_flat, A);
}

// Rows of up to 128 candidates (plane_class() == 2): two words per row, walked as (r0.hi, r0.lo, r1.hi) then (r1.lo, r2.hi,
// r2.lo) — the same visiting order.  `masks` / `masks_lo`: the plane's words from k3_density, or nullptr (own scan).
template <int MODE>
__device__ __forceinline__ void sweep3_masks128(const Params3& P, const Tol3& C, const RowRanges& R, const uint32_t* blo, bool self_plane,
                                                uint32_t ii, float4 me, float4 mv, float pressure,
                                                const float4* __restrict__ vel_s, const float4* s_flat,
                                                const u64m* __restrict__ masks, const u64m* __restrict__ masks_lo, Acc3& A) {
    u64m mh[3], ml[3];
    uint32_t la[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint32_t len = R.hi[r] - R.lo[r];
        la[r] = (uint32_t)r * TILE3_ROW + (len ? R.lo[r] - blo[r] : 0u);
        if (masks) {
            mh[r] = masks[(size_t)r * P.n + ii] & (len >= 64u ? ~0ull : (len ? ~0ull << (64u - len) : 0ull));
            ml[r] = masks_lo[(size_t)r * P.n + ii] & (len > 64u ? ~0ull << (128u - len) : 0ull);
        } else {
            scan3_row128(P, s_flat + la[r], len, me, &mh[r], &ml[r]);
        }
    }
    if (self_plane && ii - R.lo[1] < R.hi[1] - R.lo[1]) {                  // k != i
        const uint32_t d = ii - R.lo[1];
        if (d < 64u) mh[1] &= ~(0x8000000000000000ull >> d);
        else ml[1] &= ~(0x8000000000000000ull >> (d - 64u));
    }
    {
        const u64m m[3] = {mh[0], ml[0], mh[1]};
        const uint32_t l[3] = {la[0], la[0] + 64u, la[1]}, g[3] = {R.lo[0], R.lo[0] + 64u, R.lo[1]};
        walk3<MODE>(P, C, m, l, g, me, mv, pressure, vel_s, s_flat, A);
    }
    {
        const u64m m[3] = {ml[1], mh[2], ml[2]};
        const uint32_t l[3] = {la[1] + 64u, la[2], la[2] + 64u}, g[3] = {R.lo[1] + 64u, R.lo[2], R.lo[2] + 64u};
        walk3<MODE>(P, C, m, l, g, me, mv, pressure, vel_s, s_flat, A);
    }
}

// General sweep of three rows (one z-plane) for waves that hold a row longer than 64 candidates, or whose
// plane does not fit the LDS tile: the same machinery one 32-candidate chunk of one row at a time (see
// kernels_step.hip force_sweep_chunks) — wave-uniform scan into a 32-bit mask, pipelined walk.  Rows and
// chunks in order = the oracle's visiting order.  STAGED: candidates from the LDS tile, else from global
// memory (pred is allocated with FS_PRED_SLACK elements of slack for the read-ahead).
template <bool STAGED, int MODE>
__device__ __forceinline__ void sweep3_chunks(const Params3& P, const Tol3& C, const RowRanges& R, const uint32_t* blo, bool self_plane,
                                              uint32_t ii, float4 me, float4 mv, float pressure,
                                              const float4* __restrict__ pred, const float4* __restrict__ vel_s,
                                              const float4* s_flat, Acc3& A) {
    const float lim = P.h2;
    uint32_t lo0 = R.lo[0], lo1 = R.lo[1], lo2 = R.lo[2], hi0 = R.hi[0], hi1 = R.hi[1], hi2 = R.hi[2];
    uint32_t b00 = blo[0], b01 = blo[1], b02 = blo[2];
    asm volatile("" : "+v"(lo0), "+v"(lo1), "+v"(lo2), "+v"(hi0), "+v"(hi1), "+v"(hi2), "+v"(b00), "+v"(b01), "+v"(b02));
#pragma unroll 1
    for (int r = 0; r < 3; ++r) {
        const uint32_t lo = r == 0 ? lo0 : r == 1 ? lo1 : lo2;
        const uint32_t hi = r == 0 ? hi0 : r == 1 ? hi1 : hi2;
        const uint32_t b0 = r == 0 ? b00 : r == 1 ? b01 : b02;
        const uint32_t len = hi - lo;
        // FS3_CHUNK_BATCH chunks of 32 candidates are scanned before the walk starts and their masks are walked as one shift
        // register (kernels_step.hip force_sweep_chunks: a lane then waits for the wave's slowest lane once per 128
        // candidates instead of once per 32); the chunks of a batch are consecutive in the row, a refill advances the bases
#pragma unroll 1
        for (uint32_t c0 = 0; __any(c0 < len); c0 += 32u * FS3_CHUNK_BATCH) {   // c0 is wave-uniform
            uint32_t mq[FS3_CHUNK_BATCH];
            const uint32_t g0 = c0 < len ? lo + c0 : 0u;                 // global index of the batch's first candidate
            const uint32_t boff0 = (STAGED ? (c0 < len ? (uint32_t)r * TILE3_ROW + (g0 - b0) : 0u) : g0) << 4;
            const char* src = STAGED ? reinterpret_cast<const char*>(s_flat) : reinterpret_cast<const char*>(pred);
#define FS3_CAND(off, k) (*reinterpret_cast<const float4*>(src + ((off) + ((k) << 4))))
#pragma unroll
            for (int q = 0; q < FS3_CHUNK_BATCH; ++q) {
                const uint32_t cq = c0 + 32u * (uint32_t)q;
                const uint32_t clen = cq < len ? (len - cq < 32u ? len - cq : 32u) : 0u;
                const uint32_t boff = clen ? boff0 + 512u * (uint32_t)q : 0u;
                uint32_t mask = 0, t = 0;
                for (; __any(t < clen); t += 4u) {
                    const float4 q0 = FS3_CAND(boff, t), q1 = FS3_CAND(boff, t + 1u), q2 = FS3_CAND(boff, t + 2u), q3 = FS3_CAND(boff, t + 3u);
                    const float4 qq[4] = {q0, q1, q2, q3};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float ox = qq[u].x - me.x, oy = qq[u].y - me.y, oz = qq[u].z - me.z;
                        shift_in_not_greater32(mask, ox * ox + oy * oy + oz * oz, lim);
                    }
                }
                mask = t ? mask << (32u - t) : 0u;
                mask &= clen ? 0xFFFFFFFFu << (32u - clen) : 0u;
                const uint32_t g = g0 + 32u * (uint32_t)q;
                if (r == 1 && self_plane && clen && ii - g < clen) mask &= ~(0x80000000u >> (ii - g));   // k != i
                mq[q] = mask;
            }
            uint32_t cur = mq[0], n1 = mq[1 % FS3_CHUNK_BATCH], n2 = FS3_CHUNK_BATCH > 2 ? mq[2 % FS3_CHUNK_BATCH] : 0u,
                     n3 = FS3_CHUNK_BATCH > 3 ? mq[3 % FS3_CHUNK_BATCH] : 0u;
            uint32_t boff = boff0, goff = g0 << 4;
            float4 qn = make_float4(0.0f, 0.0f, 0.0f, 0.0f), vn = qn;
            bool have = false, pending = false;
#define FS3_FETCH_NEXT1()                                                                                            \
    do {                                                                                                             \
        if (cur == 0u) { cur = n1; n1 = n2; n2 = n3; n3 = 0u; boff += 512u; goff += 512u; }   /* next chunk of the batch */ \
        have = cur != 0u;                                                                                            \
        pending = (cur | n1 | n2 | n3) != 0u;            /* an empty chunk in the middle costs this lane one idle trip */ \
        if (have) {                                                                                                  \
            const uint32_t tt = (uint32_t)__builtin_clz(cur);                                                        \
            cur ^= 0x80000000u >> tt;                                                                                \
            qn = FS3_CAND(boff, tt);                                                                                 \
            if (STAGED && FS3_STAGE_VEL) vn = *reinterpret_cast<const float4*>(src + (boff + (tt << 4)) + TILE3_VEL_OFF);  \
            else vn = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(vel_s) + (goff + (tt << 4)));   \
        }                                                                                                            \
    } while (0)
            FS3_FETCH_NEXT1();
            while (__any(pending)) {
                const bool cur_valid = have;
                const float4 q0 = qn, v0 = vn;
                FS3_FETCH_NEXT1();
                if (cur_valid) pair3_accum<MODE>(P, C, me, mv, pressure, q0, v0, A);
            }
#undef FS3_FETCH_NEXT1
#undef FS3_CAND
        }
    }
}

// The 27-cell sweep runs plane by plane (z outer).  Per plane the workgroup's three row ranges are staged
// into LDS (as in k3_density) and swept with register pass-masks (sweep3_masks); waves that hold a range
// longer than 64, and planes whose rows do not fit the tile, take the chunked sweep.
template <int MODE>
__device__ __forceinline__ void force3_body(const Params3& P, const float4* __restrict__ pos_s,
                                            const float4* __restrict__ vel_s, const float4* __restrict__ pred,
                                            const uint32_t* __restrict__ cs, float4* __restrict__ pos_out,
                                            float4* __restrict__ vel_out, const u64m* __restrict__ masks,
                                            const uint32_t* __restrict__ key_s, const u64* __restrict__ srcs, float4* s_buf,
                                            uint32_t* s_red) {
    const uint32_t tid = threadIdx.x;
    uint32_t blk;
    if (!xcd_block3(P, (P.n + B3F - 1) / B3F, &blk)) return;       // uniform
    const uint32_t i = blk * B3F + tid;
    const bool live = i < P.n;
    const uint32_t ii = live ? i : P.n - 1;
    const float4 me = pred[ii];
    const float4 mv = vel_s[ii];
    const float mrho = me.w;
    const float pressure = P.pressure_k * (mrho - P.rest_density);
    const Tol3 C = tol3_consts(P);                      // dead code unless MODE == 2
    Acc3 A;
    A.px = A.py = A.pz = A.vx = A.vy = A.vz = 0.0f;
    A.seed = ii * 12u + P.frame * 69u;
    const uint32_t key = key_s[ii];
    uint32_t lo9[9], hi9[9];        // all 18 cell-start lookups up front: independent loads, one latency
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        lo9[j] = 0; hi9[j] = 0;
        if (live && !row3_key(P, cs, key, j, &lo9[j], &hi9[j])) { lo9[j] = 0; hi9[j] = 0; }
    }
#pragma unroll 1
    for (int plane = 0; plane < 3; ++plane) {
        RowRanges R;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            R.lo[r] = plane == 0 ? lo9[r] : plane == 1 ? lo9[3 + r] : lo9[6 + r];
            R.hi[r] = plane == 0 ? hi9[r] : plane == 1 ? hi9[3 + r] : hi9[6 + r];
        }
        uint32_t blo[3], bhi[3];
        const bool fit = block_tile_bounds<W3F>(R, s_red, blo, bhi, TILE3);
        if (fit) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
                for (uint32_t j = tid; j < bhi[r] - blo[r]; j += B3F) {
                    s_buf[r * TILE3_ROW + j] = pred[blo[r] + j];
                    if (FS3_STAGE_VEL) s_buf[TILE3_LDS + r * TILE3_ROW + j] = vel_s[blo[r] + j];
                }
            __syncthreads();
            const int pclass = plane_class(R, fit);      // the same predicate as k3_density: its masks exist exactly for these planes
            if (pclass == 1)
                sweep3_masks<MODE>(P, C, R, blo, plane == 1, ii, me, mv, pressure, vel_s, s_buf,
                                   masks ? masks + (size_t)plane * 3u * P.n : nullptr, A);
            else if (pclass == 2)
                sweep3_masks128<MODE>(P, C, R, blo, plane == 1, ii, me, mv, pressure, vel_s, s_buf,
                                      masks ? masks + (size_t)plane * 3u * P.n : nullptr,
                                      masks ? masks + (size_t)(9 + plane * 3) * P.n : nullptr, A);
            else sweep3_chunks<true, MODE>(P, C, R, blo, plane == 1, ii, me, mv, pressure, pred, vel_s, s_buf, A);
        } else {
            sweep3_chunks<false, MODE>(P, C, R, blo, plane == 1, ii, me, mv, pressure, pred, vel_s, s_buf, A);
        }
        __syncthreads();     // the next plane reuses s_buf / s_red
    }
    if (!live) return;
    float4 v = mv, p = pos_s[(uint32_t)srcs[i]];        // pos_s: the PREVIOUS state, source order (see k3_reorder)
    const float ax = A.px + A.vx * P.visc_coeff, ay = A.py + A.vy * P.visc_coeff, az = A.pz + A.vz * P.visc_coeff;
    v.x += __fdiv_rn(ax, mrho) * P.dt; v.y += __fdiv_rn(ay, mrho) * P.dt; v.z += __fdiv_rn(az, mrho) * P.dt;
    v.x += P.gx * P.dt; v.y += P.gy * P.dt; v.z += P.gz * P.dt;
    if (!(v.x == v.x && v.y == v.y && v.z == v.z)) { v.x = 0.0f; v.y = 0.0f; v.z = 0.0f; }
    const float s2 = v.x * v.x + v.y * v.y + v.z * v.z;
    if (s2 > 249000.0f) {                           // below that the root is < 500 whatever the rounding: no clamp (kernels_step.hip)
        const float speed = sqrt_rn(s2);
        if (speed > 500.0f) {
            v.x = __fdiv_rn(v.x, speed) * 500.0f; v.y = __fdiv_rn(v.y, speed) * 500.0f; v.z = __fdiv_rn(v.z, speed) * 500.0f;
        }
    }
    p.x += v.x * P.dt; p.y += v.y * P.dt; p.z += v.z * P.dt;
    if (fabsf(p.x) > P.bx) { p.x = P.bx * sign_f32(p.x); v.x *= -1.0f * P.damping; }
    if (fabsf(p.y) > P.by) { p.y = P.by * sign_f32(p.y); v.y *= -1.0f * P.damping; }
    if (fabsf(p.z) > P.bz) { p.z = P.bz * sign_f32(p.z); v.z *= -1.0f * P.damping; }
    p.w = 0.0f; v.w = 0.0f;
    pos_out[i] = p;
    vel_out[i] = v;
}
// One kernel per math mode: the register budget that measured best differs (strict: 7 waves per SIMD, tolerance: 6).
#ifndef FS3_FORCE_WAVES_TOL
#define FS3_FORCE_WAVES_TOL (FS3_STAGE_VEL ? 4 : 6)
#endif
template <int MODE> __global__ void k3_force(Params3 P, const float4* __restrict__ pos_s, const float4* __restrict__ vel_s,
                                             const float4* __restrict__ pred, const uint32_t* __restrict__ cs,
                                             float4* __restrict__ pos_out, float4* __restrict__ vel_out,
                                             const u64m* __restrict__ masks, const uint32_t* __restrict__ key_s,
                                             const u64* __restrict__ srcs);
template <>
__global__ __launch_bounds__(B3F) __attribute__((amdgpu_waves_per_eu(FS3_FORCE_WAVES, FS3_FORCE_WAVES))) void k3_force<0>(
    Params3 P, const float4* __restrict__ pos_s, const float4* __restrict__ vel_s, const float4* __restrict__ pred,
    const uint32_t* __restrict__ cs, float4* __restrict__ pos_out, float4* __restrict__ vel_out, const u64m* __restrict__ masks,
    const uint32_t* __restrict__ key_s, const u64* __restrict__ srcs) {
    __shared__ float4 s_buf[TILE3_FORCE_LDS];     // the staged plane: positions, then velocities
    __shared__ uint32_t s_red[24];
    force3_body<0>(P, pos_s, vel_s, pred, cs, pos_out, vel_out, masks, key_s, srcs, s_buf, s_red);
}
template <>
__global__ __launch_bounds__(B3F) __attribute__((amdgpu_waves_per_eu(FS3_FORCE_WAVES_TOL, FS3_FORCE_WAVES_TOL))) void k3_force<2>(
    Params3 P, const float4* __restrict__ pos_s, const float4* __restrict__ vel_s, const float4* __restrict__ pred,
    const uint32_t* __restrict__ cs, float4* __restrict__ pos_out, float4* __restrict__ vel_out, const u64m* __restrict__ masks,
    const uint32_t* __restrict__ key_s, const u64* __restrict__ srcs) {
    __shared__ float4 s_buf[TILE3_FORCE_LDS];
    __shared__ uint32_t s_red[24];
    force3_body<2>(P, pos_s, vel_s, pred, cs, pos_out, vel_out, masks, key_s, srcs, s_buf, s_red);
}

__global__ __launch_bounds__(B3) void k3_export(uint32_t n, const float4* __restrict__ pos, const float4* __restrict__ pred,
                                                const float4* __restrict__ vel, const uint32_t* __restrict__ key,
                                                fs3_particle* __restrict__ out) {
    const uint32_t i = blockIdx.x * B3 + threadIdx.x;
    if (i >= n) return;
    const float4 p = pos[i], q = pred[i], v = vel[i];
    fs3_particle a;
    a.position = fs_vec3{p.x, p.y, p.z};
    a.predicted_position = fs_vec3{q.x, q.y, q.z};
    a.velocity = fs_vec3{v.x, v.y, v.z};
    a.density = q.w; a.grid = key[i]; a.pad = 0;
    out[i] = a;
}
__global__ __launch_bounds__(B3) void k3_import(uint32_t n, const fs3_particle* __restrict__ in, float4* __restrict__ pos,
                                                float4* __restrict__ pred, float4* __restrict__ vel,
                                                uint32_t* __restrict__ key) {
    const uint32_t i = blockIdx.x * B3 + threadIdx.x;
    if (i >= n) return;
    const fs3_particle a = in[i];
    pos[i] = make_float4(a.position.x, a.position.y, a.position.z, 0.0f);
    pred[i] = make_float4(a.predicted_position.x, a.predicted_position.y, a.predicted_position.z, a.density);
    vel[i] = make_float4(a.velocity.x, a.velocity.y, a.velocity.z, 0.0f);
    key[i] = a.grid;
}

}  // namespace fsd

// ------------------------------------------------------------------------------------ host
namespace {
fs_status fail3(fs_status st, const std::string& m) { fsd::set_last_error(m); return st; }
template <class T> struct Dev3 {
    T* p = nullptr; size_t n = 0;
    hipError_t alloc(size_t c) { n = c; return c ? hipMalloc((void**)&p, c * sizeof(T)) : hipSuccess; }
    void release() { if (p) (void)hipFree(p); p = nullptr; }
};
void lattice3(const fs3_settings& st, fs_vec3 off, fs3_particle* dst, size_t n) {
    const uint32_t side = (uint32_t)std::llround(std::cbrt((double)st.particle_count));
    const float half = (float)side * 0.5f, s = st.particle_spacing;
    for (uint32_t i = 0; i < st.particle_count && i < n; ++i) {
        const uint32_t ix = i % side, iy = (i / side) % side, iz = i / (side * side);
        fs3_particle q;
        std::memset(&q, 0, sizeof q);
        q.position.x = ((float)ix - half + 0.5f) * s + off.x;
        q.position.y = ((float)iy - half + 0.5f) * s + off.y;
        q.position.z = ((float)iz - half + 0.5f) * s + off.z;
        q.predicted_position = q.position;
        dst[i] = q;
    }
}
}  // namespace

#define H3(expr)                                                                                         \
    do {                                                                                                 \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess)                                                                           \
            return fail3(e__ == hipErrorOutOfMemory ? FS_ERR_OOM : FS_ERR_DEVICE,                        \
                         std::string(#expr) + ": " + hipGetErrorString(e__));                            \
    } while (0)

struct fs_sim3 {
    fs3_settings st{};
    uint32_t n = 0, gw = 0, gh = 0, gd = 0, ncell = 0, tick = 0, work_cap = 0;
    int device = 0;
    int math_mode = FS_MATH_IEEE;
    hipStream_t stream = nullptr;
    Dev3<float4> pos, vel, pos_s, vel_s, pred;
    Dev3<uint32_t> key, cs, counter, dirty;
    Dev3<fsd::u64> pairs;
    Dev3<fsd::u64> masks;        // 9 x n pass masks of the 27-cell sweep, k3_density -> k3_force (Params3::handoff)
    bool handoff = true;
    Dev3<unsigned char> work;
    Dev3<fs3_particle> aos;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool profile = false;
    std::vector<hipEvent_t> ev;
    uint32_t pending = 0;
    double ms[FS_PASS_COUNT] = {};
    uint64_t steps = 0;
    fsd::ConstDiv div_2h3{}, div_h2{};
    bool share_div = false;      // all create-time proofs of the shared-denominator path succeeded
    fsd::SortPolicy sortp;       // host side of the sort's late-stage plan (sort_policy.h)
    void release() {
        sortp.release();
        pos.release(); vel.release(); pos_s.release(); vel_s.release(); pred.release(); key.release(); cs.release();
        counter.release(); dirty.release(); pairs.release(); masks.release(); work.release(); aos.release();
        for (auto& e : ev) (void)hipEventDestroy(e);
        if (t0) (void)hipEventDestroy(t0);
        if (t1) (void)hipEventDestroy(t1);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

static const uint32_t RING3 = 256;

static fs_status drain3(fs_sim3* s) {
    if (!s->pending) return FS_OK;
    const size_t stride = FS_PASS_COUNT + 1;
    H3(hipEventSynchronize(s->ev[(size_t)(s->pending - 1) * stride + FS_PASS_COUNT]));
    for (uint32_t j = 0; j < s->pending; ++j)
        for (int k = 0; k < FS_PASS_COUNT; ++k) {
            float ms = 0;
            H3(hipEventElapsedTime(&ms, s->ev[j * stride + k], s->ev[j * stride + k + 1]));
            s->ms[k] += ms;
        }
    s->steps += s->pending;
    s->pending = 0;
    return FS_OK;
}

static fs_status enqueue3(fs_sim3* s, const fs3_tick_settings* t) {
    using namespace fsd;
    s->tick += 1;
    const float h = s->st.smoothing_radius;
    const float PI3 = 3.14159265359f;
    Params3 P;
    std::memset(&P, 0, sizeof P);
    P.n = s->n; P.gw = s->gw; P.gh = s->gh; P.gd = s->gd; P.ncell = s->ncell;
    P.dt = t->delta; P.h = h; P.h2 = h * h;
    P.bx = s->st.size.x * 0.5f; P.by = s->st.size.y * 0.5f; P.bz = s->st.size.z * 0.5f;
    P.mass = t->mass;
    P.poly6 = 315.0f / (64.0f * PI3 * std::pow(h, 9.0f));      // host libm, as in the oracle
    P.spiky = 15.0f / (PI3 * std::pow(h, 5.0f));
    P.visc_k = 15.0f / (2.0f * PI3 * (h * h * h));
    P.pressure_k = t->pressure_constant; P.rest_density = t->rest_density; P.damping = t->damping_factor;
    P.visc_coeff = t->viscosity_coefficient;
    P.gx = t->gravity.x; P.gy = t->gravity.y; P.gz = t->gravity.z;
    P.frame = s->tick;
    P.div_2h3 = s->div_2h3;
    P.div_h2 = s->div_h2;
    // the classification bounds the pressure numerators by (1 + 2^-22) h spiky 2^39 <= 2^60 (fs_device.h)
    P.share_div = (s->share_div && h * P.spiky <= FS_HSPIKY_HI) ? 1 : 0;
    P.handoff = s->handoff ? 1 : 0;
    {   // chunks of ~1/128 of the blocks, at most 2^7 (8 M: 31 250 blocks, a z-plane of the cube is ~310): FS3_XCD_CHUNK_LOG2 overrides
        static const int forced = getenv("FS3_XCD_CHUNK_LOG2") ? atoi(getenv("FS3_XCD_CHUNK_LOG2")) : -1;
        const uint32_t nb = (s->n + B3F - 1) / B3F;
        uint32_t c = 0;
        while (c < 7u && (128u << (c + 1u)) <= nb) ++c;
        // 8 M, steps 10-110, strict / tolerance step (ms): c = 0: 3.329 / 2.732, 3: 3.277 / 2.675, 5: 3.260 / 2.645,
        // 7: 3.255 / 2.636, 8: 3.281 / 2.650, 10: 3.403 / 2.738, 12: 3.425 / 2.761 (large chunks bind an XCD to one depth)
        P.xcd_chunk_log2 = forced >= 0 ? (uint32_t)(forced > 16 ? 16 : forced) : c;
    }
    const bool tol = s->math_mode == FS_MATH_TOLERANCE;
    hipStream_t st = s->stream;
    hipEvent_t* ev = nullptr;
    if (s->profile) {
        if (s->ev.empty()) { s->ev.resize((size_t)RING3 * (FS_PASS_COUNT + 1)); for (auto& e : s->ev) H3(hipEventCreate(&e)); }
        if (s->pending == RING3) { fs_status r = drain3(s); if (r != FS_OK) return r; }
        ev = &s->ev[(size_t)s->pending * (FS_PASS_COUNT + 1)];
    }
    const dim3 grid((s->n + B3 - 1) / B3), block(B3);
    H3(s->sortp.throttle());                           // at most SortPolicy::FLIGHT steps ahead of the device
    if (ev) H3(hipEventRecord(ev[0], st));
    // predict + key are fused into the first sort kernel (k_bitonic_local<true, 2, *>), as in 2D: no separate launch,
    // the unsorted pairs never touch HBM.  FS3_SEPARATE_KEYGEN=1 keeps the round-2 kernel (A/B measurements).
    static const bool separate_keygen = getenv("FS3_SEPARATE_KEYGEN") != nullptr;
    if (separate_keygen) hipLaunchKernelGGL(k3_predict_key, grid, block, 0, st, P, s->pos.p, s->vel.p, s->pairs.p, s->counter.p);
    if (ev) H3(hipEventRecord(ev[1], st));
    fsd::SortPlan plan;
    if (!s->sortp.plan(s->n, &plan)) return fail3(FS_ERR_DEVICE, "sort: the stand-by kernel's grid barrier timed out");
    if (separate_keygen) {
        launch_bitonic_sort(st, s->pairs.p, s->n, s->dirty.p, nullptr, nullptr, nullptr, nullptr, &plan);
    } else {
        const fsd::KeyGen3 kg{P.dt, P.h, P.bx, P.by, P.bz, P.gw, P.gh};
        launch_bitonic_sort(st, s->pairs.p, s->n, s->dirty.p, nullptr, nullptr, nullptr, s->counter.p, &plan, &kg, s->pos.p, s->vel.p);
    }
    if (ev) H3(hipEventRecord(ev[2], st));
    hipLaunchKernelGGL(k3_reorder, grid, block, 0, st, P, s->pairs.p, s->pos.p, s->vel.p, s->pos_s.p, s->vel_s.p,
                       s->pred.p, s->key.p, s->cs.p, (GapEntry*)s->work.p, s->counter.p, s->work_cap);
    launch_fill_gaps(st, s->cs.p, s->work.p, s->counter.p, s->work_cap);
    if (ev) H3(hipEventRecord(ev[3], st));
    const fsd::u64* fm = s->handoff ? s->masks.p : nullptr;
    const dim3 gridf(xcd_grid3((s->n + B3F - 1) / B3F, P.xcd_chunk_log2)), blockf(B3F);
    if (tol) hipLaunchKernelGGL(k3_density<2>, gridf, blockf, 0, st, P, s->pred.p, s->cs.p, s->vel_s.p, s->masks.p, s->key.p);
    else hipLaunchKernelGGL(k3_density<0>, gridf, blockf, 0, st, P, s->pred.p, s->cs.p, s->vel_s.p, s->masks.p, s->key.p);
    if (ev) H3(hipEventRecord(ev[4], st));
    // positions ping-pong: read the previous state (s->pos, source order) through the pairs, write the new one into s->pos_s
    // the step's completion event (sort_policy.h: the host stays at most four steps ahead) rides on the force kernel as its
    // completion signal — no marker packet behind it (engine.hip fs_step does the same); a profiled step records markers anyway
    hipEvent_t done = ev ? nullptr : s->sortp.flight_event();
    if (tol) hipExtLaunchKernelGGL(k3_force<2>, gridf, blockf, 0, st, nullptr, done, 0, P, s->pos.p, s->vel_s.p, s->pred.p, s->cs.p, s->pos_s.p, s->vel.p, fm, s->key.p, s->pairs.p);
    else hipExtLaunchKernelGGL(k3_force<0>, gridf, blockf, 0, st, nullptr, done, 0, P, s->pos.p, s->vel_s.p, s->pred.p, s->cs.p, s->pos_s.p, s->vel.p, fm, s->key.p, s->pairs.p);
    { float4* t = s->pos.p; s->pos.p = s->pos_s.p; s->pos_s.p = t; }
    if (ev) { H3(hipEventRecord(ev[5], st)); H3(hipEventRecord(ev[6], st)); /* FS_PASS_BOUNDARY: slab handles only */ s->pending += 1; }
    if (ev) H3(s->sortp.step_enqueued(st));
    else s->sortp.step_bound();
    H3(hipGetLastError());
    return FS_OK;
}

extern "C" {

fs_status fs3_reference_lattice(const fs3_settings* st, fs_vec3 off, fs3_particle* dst, size_t n) {
    if (!st || (!dst && n)) return fail3(FS_ERR_INVALID, "null argument");
    lattice3(*st, off, dst, n);
    return FS_OK;
}

fs_status fs3_create(const fs3_settings* st, int device, fs_vec3 off, fs_sim3** out) {
    return fs3_create_ex(st, device, off, FS_MATH_IEEE, out);
}

fs_status fs3_create_ex(const fs3_settings* st, int device, fs_vec3 off, int math_mode, fs_sim3** out) {
    if (!st || !out) return fail3(FS_ERR_INVALID, "null argument");
    *out = nullptr;
    if (math_mode != FS_MATH_IEEE && math_mode != FS_MATH_TOLERANCE)
        return fail3(FS_ERR_UNSUPPORTED, "3D math_mode must be FS_MATH_IEEE or FS_MATH_TOLERANCE");
    if (st->particle_count <= 1) return fail3(FS_ERR_INVALID, "particle_count <= 1");
    if (st->particle_count > (1u << 28)) return fail3(FS_ERR_INVALID, "particle_count > 2^28 (32-bit byte offsets)");
    if (!(st->smoothing_radius > 0.0f) || !(st->size.x > 0) || !(st->size.y > 0) || !(st->size.z > 0))
        return fail3(FS_ERR_INVALID, "bad settings");
    const uint32_t side = (uint32_t)std::llround(std::cbrt((double)st->particle_count));
    if ((uint64_t)side * side * side != st->particle_count) return fail3(FS_ERR_INVALID, "particle_count must be a cube");
    const double gw = std::ceil((double)st->size.x / st->smoothing_radius) + 2, gh = std::ceil((double)st->size.y / st->smoothing_radius) + 2,
                 gd = std::ceil((double)st->size.z / st->smoothing_radius) + 2;
    if (gw * gh * gd >= 4294967295.0) return fail3(FS_ERR_INVALID, "grid does not fit u32 cell ids");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail3(FS_ERR_DEVICE, "no HIP device: the engine has no CPU fallback");
    if (device < 0 || device >= ndev) return fail3(FS_ERR_INVALID, "device ordinal out of range");
    H3(hipSetDevice(device));
    fs_sim3* s = new (std::nothrow) fs_sim3();
    if (!s) return fail3(FS_ERR_OOM, "host allocation failed");
    s->st = *st; s->n = st->particle_count; s->device = device; s->math_mode = math_mode;
    s->gw = (uint32_t)((size_t)std::ceil(st->size.x / st->smoothing_radius) + 2);
    s->gh = (uint32_t)((size_t)std::ceil(st->size.y / st->smoothing_radius) + 2);
    s->gd = (uint32_t)((size_t)std::ceil(st->size.z / st->smoothing_radius) + 2);
    s->ncell = s->gw * s->gh * s->gd;
    s->work_cap = s->ncell / 16u + 1024u;
#define T3(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { s->release(); delete s; return fail3(FS_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); } } while (0)
    T3(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    const size_t n = s->n;
    T3(s->pos.alloc(n)); T3(s->vel.alloc(n)); T3(s->pos_s.alloc(n)); T3(s->vel_s.alloc(n)); T3(s->pred.alloc(n + FS_PRED_SLACK));
    T3(s->key.alloc(n)); T3(s->pairs.alloc(n)); T3(s->cs.alloc((size_t)s->ncell + 1)); T3(s->counter.alloc(4));
    T3(s->dirty.alloc(fsd::sort_tile_count((uint32_t)n))); T3(s->work.alloc((size_t)s->work_cap * fsd::gap_entry_size()));
    T3(s->aos.alloc(n));
    s->handoff = !(getenv("FS3_HANDOFF") && atoi(getenv("FS3_HANDOFF")) == 0);
    if (s->handoff) T3(s->masks.alloc((FS3_MASK128 ? 18 : 9) * (size_t)n));   // hi words, then the lo words of rows of 65 .. 128
    T3(hipEventCreate(&s->t0)); T3(hipEventCreate(&s->t1));
    T3(hipMemsetAsync(s->cs.p, 0, s->cs.n * 4, s->stream));
    T3(hipMemsetAsync(s->counter.p, 0, 16, s->stream));
    T3(hipMemsetAsync(s->dirty.p, 0, s->dirty.n * 4, s->stream));
    T3(s->sortp.init(5));         // a z-plane of the cube holds n^(2/3) particles: the moves are long, start at stage S - 5
    {
        std::vector<fs3_particle> host(n);
        lattice3(*st, off, host.data(), n);
        T3(hipMemcpyAsync(s->aos.p, host.data(), n * sizeof(fs3_particle), hipMemcpyHostToDevice, s->stream));
        hipLaunchKernelGGL(fsd::k3_import, dim3((s->n + B3 - 1) / B3), dim3(B3), 0, s->stream, s->n, s->aos.p, s->pos.p,
                           s->pred.p, s->vel.p, s->key.p);
        T3(hipStreamSynchronize(s->stream));
    }
    for (int k = 0; k < 2; ++k) {   // prove the two constant divisions for this h (see engine.hip prove_constdiv)
        fsd::ConstDiv& K = k == 0 ? s->div_2h3 : s->div_h2;
        const float hh = st->smoothing_radius;
        K.c = k == 0 ? 2.0f * hh * hh * hh : hh * hh;
        K.y = 1.0f / K.c;
        K.ok = 0;
        if (!(K.c > 4.0f * FS_CONSTDIV_MIN) || !std::isfinite(K.c) || !std::isfinite(K.y)) continue;
        uint32_t bad = 1;
        T3(hipMemsetAsync(s->counter.p + 1, 0, 4, s->stream));
        fsd::launch_verify_constdiv(s->stream, K.c, K.y, FS_CONSTDIV_MIN, K.c, s->counter.p + 1);
        T3(hipMemcpyAsync(&bad, s->counter.p + 1, 4, hipMemcpyDeviceToHost, s->stream));
        T3(hipStreamSynchronize(s->stream));
        K.ok = bad == 0 ? 1 : 0;
    }
    {   // lean reciprocal / square root of the shared-denominator path, over their whole ranges (engine.hip)
        uint32_t bad[2] = {1, 1};
        const float hh = st->smoothing_radius;
        if (!getenv("FS_NO_SHAREDIV")) {
            T3(hipMemsetAsync(s->counter.p + 1, 0, 8, s->stream));
            fsd::launch_verify_unary(s->stream, 0, FS_RCP_LO, FS_RCP_HI, s->counter.p + 1);
            fsd::launch_verify_unary(s->stream, 1, FS_SQRT_LO, FS_SQRT_HI, s->counter.p + 2);
            T3(hipMemcpyAsync(bad, s->counter.p + 1, 8, hipMemcpyDeviceToHost, s->stream));
            T3(hipStreamSynchronize(s->stream));
        }
        s->share_div = bad[0] == 0 && bad[1] == 0 && s->div_2h3.ok && s->div_h2.ok && hh >= 0x1p-19f && hh <= 0x1p19f;
    }
#undef T3
    *out = s;
    return FS_OK;
}

void fs3_destroy(fs_sim3* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    s->release();
    delete s;
}

fs_status fs3_step(fs_sim3* s, const fs3_tick_settings* t) {
    if (!s || !t) return fail3(FS_ERR_INVALID, "null argument");
    H3(hipSetDevice(s->device));
    return enqueue3(s, t);
}
// a barrier time-out of the sort's stand-by kernel leaves the particle order undefined: reported wherever state is handed over
static fs_status sort_health3(fs_sim3* s) {
    H3(s->sortp.check_timeout(s->dirty.p, s->n));
    if (s->sortp.dead) return fail3(FS_ERR_DEVICE, "sort: the stand-by kernel's grid barrier timed out: the particle order is undefined from that step on; destroy the handle");
    return FS_OK;
}
fs_status fs3_sync(fs_sim3* s) { if (!s) return fail3(FS_ERR_INVALID, "null"); H3(hipStreamSynchronize(s->stream)); return sort_health3(s); }
uint32_t fs3_tick_count(const fs_sim3* s) { return s ? s->tick : 0; }
uint32_t fs3_particle_count(const fs_sim3* s) { return s ? s->n : 0; }
fs_status fs3_grid_dims(const fs_sim3* s, uint32_t* w, uint32_t* h, uint32_t* d) {
    if (!s || !w || !h || !d) return fail3(FS_ERR_INVALID, "null argument");
    *w = s->gw; *h = s->gh; *d = s->gd;
    return FS_OK;
}
fs_status fs3_download_particles(fs_sim3* s, fs3_particle* dst, size_t n) {
    if (!s || (!dst && n)) return fail3(FS_ERR_INVALID, "null argument");
    if (n > s->n) n = s->n;
    H3(hipSetDevice(s->device));
    hipLaunchKernelGGL(fsd::k3_export, dim3((s->n + B3 - 1) / B3), dim3(B3), 0, s->stream, s->n, s->pos.p, s->pred.p,
                       s->vel.p, s->key.p, s->aos.p);
    if (n) H3(hipMemcpyAsync(dst, s->aos.p, n * sizeof(fs3_particle), hipMemcpyDeviceToHost, s->stream));
    H3(hipStreamSynchronize(s->stream));
    return sort_health3(s);
}
fs_status fs3_upload_particles(fs_sim3* s, const fs3_particle* src, size_t n) {
    if (!s || (!src && n)) return fail3(FS_ERR_INVALID, "null argument");
    if (n > s->n) n = s->n;
    H3(hipSetDevice(s->device));
    if (n) H3(hipMemcpyAsync(s->aos.p, src, n * sizeof(fs3_particle), hipMemcpyHostToDevice, s->stream));
    if (n) hipLaunchKernelGGL(fsd::k3_import, dim3(((uint32_t)n + B3 - 1) / B3), dim3(B3), 0, s->stream, (uint32_t)n,
                              s->aos.p, s->pos.p, s->pred.p, s->vel.p, s->key.p);
    H3(hipStreamSynchronize(s->stream));
    s->sortp.touched();
    return FS_OK;
}
fs_status fs3_timed_steps(fs_sim3* s, const fs3_tick_settings* t, uint32_t steps, double* ms_total) {
    if (!s || !t || !ms_total) return fail3(FS_ERR_INVALID, "null argument");
    H3(hipSetDevice(s->device));
    H3(hipEventRecord(s->t0, s->stream));
    for (uint32_t k = 0; k < steps; ++k) { fs_status r = enqueue3(s, t); if (r != FS_OK) return r; }
    H3(hipEventRecord(s->t1, s->stream));
    H3(hipEventSynchronize(s->t1));
    float ms = 0;
    H3(hipEventElapsedTime(&ms, s->t0, s->t1));
    *ms_total = ms;
    return sort_health3(s);
}
fs_status fs3_profile_enable(fs_sim3* s, int enable) { if (!s) return fail3(FS_ERR_INVALID, "null"); s->profile = enable != 0; return FS_OK; }
fs_status fs3_profile_read(fs_sim3* s, double ms[FS_PASS_COUNT], uint64_t* steps, int reset) {
    if (!s || !ms) return fail3(FS_ERR_INVALID, "null argument");
    fs_status r = drain3(s);
    if (r != FS_OK) return r;
    for (int k = 0; k < FS_PASS_COUNT; ++k) ms[k] = s->ms[k];
    if (steps) *steps = s->steps;
    if (reset) { for (auto& m : s->ms) m = 0; s->steps = 0; }
    return FS_OK;
}

}  // extern "C"

/*
 * sph_oracle.cpp — CPU restatement of the reference SPH step.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / reported CPU baseline — never as a
 * product path.  The product (libfluidsim_hip.so) does not link or call it.
 *
 * PARITY PINNING: the reference (Rust + WGSL through wgpu 25.0.2 / naga 25.0.1)
 * ships no tests, no golden vectors and cannot be built or run here (no
 * cargo/rustc/naga/Vulkan).  This oracle is therefore "parity unpinned" by
 * reference fixtures; it is pinned instead by known answers derived from the
 * reference source (kernel constants, interior lattice density
 * 101.4609, struct layouts 32/120 B, dispatch counts) — see tests/test_oracle.py.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference tree).  All arithmetic is IEEE f32, evaluated in the written
 * association; build with -ffp-contract=off (see oracle/Makefile).
 */
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/fluidsim.h"

namespace {

// funcs.wgsl:54-55
const float PI_F = 3.14159265359f;       // == std::f32::consts::PI as f32
const float EPSILON_F = 1.19209290e-07f;

// Rust f32::powi -> llvm.powi -> compiler-rt __powisf2 (square-and-multiply).
float powi_f32(float a, int b) {
    const bool recip = b < 0;
    float r = 1.0f;
    while (true) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0f / r : r;
}

// WGSL f32 -> u32 conversion saturates; NaN -> 0.
uint32_t f32_to_u32_sat(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

float sign_f32(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }  // WGSL sign()

struct OrcSim {
    fs_settings settings{};
    fs_vec2 offset{0, 0};
    int ref_quirks = 1;
    uint32_t tick = 0;
    uint32_t grid_w = 0, grid_h = 0;
    fs_uniform u{};
    float poly6_norm = 0.0f;
    std::vector<fs_particle> p;
    std::vector<fs_particle> snap;       // Jacobi snapshot for move_particle
    std::vector<uint32_t> start_indices; // persistent, never cleared (simulation.rs:204-209)
    std::vector<fs_vec2> texture;        // force field (simulation.rs:213-218)
};

// src/simulation.rs:140-141 (f32 divide, f32 ceil, +2 padding ring)
void grid_dims(const fs_settings& s, uint32_t* gw, uint32_t* gh) {
    *gw = (uint32_t)((size_t)std::ceil(s.size.x / s.smoothing_radius) + 2);
    *gh = (uint32_t)((size_t)std::ceil(s.size.y / s.smoothing_radius) + 2);
}

// src/simulation.rs:147-163 — initial lattice, quirks of SURVEY A.6d kept.
void lattice(const fs_settings& s, fs_vec2 off, fs_particle* dst, size_t n) {
    const uint32_t count = s.particle_count;
    const float spacing = s.particle_spacing;
    const float ppr = std::sqrt((float)count);               // :147
    const float ppc = ((float)count - 1.0f) / ppr + 1.0f;    // :148
    const size_t ppr_usize = (size_t)ppr;                    // `as usize` truncation, :152
    for (uint32_t i = 0; i < count && i < n; ++i) {
        const size_t xi = (size_t)i % ppr_usize;                                         // :152
        const float x = ((float)xi - ppr * 0.5f + 0.5f) * spacing;                       // :153
        const float y = (std::floor((float)i / ppr) - ppc * 0.5f + 0.5f) * spacing;      // :154
        fs_particle q;
        std::memset(&q, 0, sizeof q);
        q.position.x = x + off.x;   // offset is a build extension (0 reproduces the reference)
        q.position.y = y + off.y;
        q.predicted_position = q.position;                                               // :158
        dst[i] = q;
    }
}

// src/simulation.rs:470-497
void build_uniform(const fs_settings& s, const fs_tick_settings& t, uint32_t tick, fs_uniform* u) {
    const float h = s.smoothing_radius;
    uint32_t gw, gh;
    grid_dims(s, &gw, &gh);
    u->delta = t.delta;
    u->particle_count = s.particle_count;
    u->sqr_radius = h * h;                                        // :473
    u->frame_time = tick;                                         // :474
    u->gravity = t.gravity;
    u->bounds = s.size;
    u->mouse_pos = t.mouse_pos;
    u->smoothing_radius = h;
    u->particle_mass = t.mass;
    u->pressure_constant = t.pressure_constant;
    u->rest_density = t.rest_density;
    u->damping_factor = t.damping_factor;
    u->viscosity_coefficient = t.viscosity_coefficient;
    u->surface_tension_treshold = t.surface_tension_treshold;
    u->surface_tension_coefficient = t.surface_tension_coefficient;
    u->poly6_kernel_volume = 4.0f / (PI_F * powi_f32(h, 8));      // :486
    u->poly6_kernel_derivative = 24.0f / (PI_F * powi_f32(h, 8)); // :487
    u->poly6_kernel_laplacian = 8.0f / (PI_F * powi_f32(h, 8));   // :488
    u->spiky_kernel_derivative = 12.0f / (powi_f32(h, 4) * PI_F); // :489
    u->viscosity_kernel = 15.0f / (2.0f * PI_F * powi_f32(h, 3)); // :490
    u->mouse_state = t.mouse_state;
    u->mouse_force_radius = t.mouse_force_radius;
    u->mouse_force_power = t.mouse_force_power;
    u->grid_w = gw;
    u->grid_h = gh;
    u->texture_size.x = (float)s.texture_size.x;                  // :496 as_vec2
    u->texture_size.y = (float)s.texture_size.y;
}

// funcs.wgsl:212-214
inline void xy_of_point(const fs_uniform& u, fs_vec2 pt, uint32_t* cx, uint32_t* cy) {
    const float fx = std::floor((pt.x + u.bounds.x * 0.5f) / u.smoothing_radius);
    const float fy = std::floor((pt.y + u.bounds.y * 0.5f) / u.smoothing_radius);
    *cx = f32_to_u32_sat(fx) + 1u;
    *cy = f32_to_u32_sat(fy) + 1u;
}
// funcs.wgsl:216-218 (wrapping u32)
inline uint32_t grid_pos_to_id(const fs_uniform& u, uint32_t x, uint32_t y) { return y * u.grid_w + x; }

// compute.wgsl:8-30
void predict(OrcSim& s) {
    const fs_uniform& u = s.u;
    const float bsx = u.bounds.x * 0.5f, bsy = u.bounds.y * 0.5f;
#pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i < u.particle_count; ++i) {
        fs_particle& q = s.p[i];
        q.predicted_position.x = q.position.x + q.velocity.x * u.delta;   // :16
        q.predicted_position.y = q.position.y + q.velocity.y * u.delta;
        if (std::fabs(q.predicted_position.x) > bsx) q.predicted_position.x = bsx * sign_f32(q.predicted_position.x);
        if (std::fabs(q.predicted_position.y) > bsy) q.predicted_position.y = bsy * sign_f32(q.predicted_position.y);
    }
}

// compute.wgsl:33-42 + funcs.wgsl:206-218
void spatial_lookup(OrcSim& s) {
#pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i < s.u.particle_count; ++i) {
        uint32_t cx, cy;
        xy_of_point(s.u, s.p[i].predicted_position, &cx, &cy);
        s.p[i].grid = grid_pos_to_id(s.u, cx, cy);
    }
}

// src/simulation.rs:323-347 — schedule; sort.wgsl:27-51 — compare-exchange step.
size_t sort_schedule(uint32_t n, fs_sort_step* dst, size_t cap) {
    if (n <= 1) return 0;
    uint32_t p2 = 1;
    while (p2 < n) p2 <<= 1;
    const uint32_t num_pairs = p2 / 2;
    uint32_t stages = 0;
    while ((1u << stages) < num_pairs * 2) ++stages;   // ilog2(num_pairs*2)
    size_t k = 0;
    for (uint32_t stage = 0; stage < stages; ++stage) {
        for (uint32_t step = 0; step <= stage; ++step) {
            if (dst && k < cap) {
                const uint32_t gw = 1u << (stage - step);
                dst[k] = fs_sort_step{gw, 2 * gw - 1, step, n};
            }
            ++k;
        }
    }
    return k;
}

template <class T, class KeyOf>
void bitonic_network(T* v, uint32_t n, KeyOf key) {
    if (n <= 1) return;
    uint32_t p2 = 1;
    while (p2 < n) p2 <<= 1;
    const uint32_t num_pairs = p2 / 2;
    const uint32_t threads = ((num_pairs + 127) / 128) * 128;   // simulation.rs:325 dispatch × WG 128
    uint32_t stages = 0;
    while ((1u << stages) < p2) ++stages;
    for (uint32_t stage = 0; stage < stages; ++stage) {
        for (uint32_t step = 0; step <= stage; ++step) {
            const uint32_t gw = 1u << (stage - step), gh = 2 * gw - 1;
#pragma omp parallel for schedule(static) if (threads > 65536)      // pairs of one dispatch are disjoint
            for (uint32_t i = 0; i < threads; ++i) {                       // sort.wgsl:29-50
                const uint32_t hh = i & (gw - 1);
                const uint32_t lo = hh + (gh + 1) * (i / gw);
                const uint32_t hi = lo + (step == 0 ? gh - 2 * hh : (gh + 1) / 2);
                if (hi >= n) continue;
                if (key(v[lo]) > key(v[hi])) { T t = v[lo]; v[lo] = v[hi]; v[hi] = t; }
            }
        }
    }
}

// compute.wgsl:45-56 — never cleared, index 0 skipped.  With
// ref_quirks == 0 the table is rebuilt cleanly instead (build extension).
void cell_starts(OrcSim& s) {
    const uint32_t n = s.u.particle_count;
    if (!s.ref_quirks) {
        // clean semantics: every occupied cell gets its true start, incl. index 0
        if (n > 0 && s.p[0].grid < s.start_indices.size()) s.start_indices[s.p[0].grid] = 0;
    }
    for (uint32_t i = 1; i < n; ++i) {
        const uint32_t g = s.p[i].grid;
        if (g != s.p[i - 1].grid && g < s.start_indices.size()) s.start_indices[g] = i;
    }
}

// funcs.wgsl:72-78.  pow(h, 8.0) is evaluated once per tick with libm powf
// (WGSL pow accuracy is implementation-defined; see DESIGN.md "float contract").
inline float poly6(const OrcSim& s, float r2) {
    const float h = s.u.smoothing_radius;
    const float h2 = h * h;
    if (r2 > h2) return 0.0f;
    const float diff = h2 - r2;
    return s.poly6_norm * diff * diff * diff;
}

// Walk one cell the way the shaders do (funcs.wgsl:166-197, compute.wgsl:178-229).
template <class F>
inline void walk_cell(const OrcSim& s, const std::vector<fs_particle>& arr, uint32_t id, F&& body) {
    if (id >= s.start_indices.size()) return;    // OOB start_indices read -> no contribution
    uint32_t k = s.start_indices[id];
    const uint32_t n = s.u.particle_count;
    while (true) {
        if (k >= n) break;
        const fs_particle& nb = arr[k];
        if (nb.grid != id) break;
        const uint32_t i = k;
        k += 1;
        body(i, nb);
    }
}

// compute.wgsl:59-74 + funcs.wgsl:157-203.  `reach` = 3 reproduces the 7x7
// sweep as written; 1 is the bit-identical 3x3 sweep.
void density(OrcSim& s, int reach) {
    const fs_uniform& u = s.u;
#pragma omp parallel for schedule(dynamic, 1024)
    for (uint32_t pi = 0; pi < u.particle_count; ++pi) {
        const fs_vec2 point = s.p[pi].predicted_position;
        uint32_t cxu, cyu;
        xy_of_point(u, point, &cxu, &cyu);
        const int32_t cx = (int32_t)cxu, cy = (int32_t)cyu;
        float rho = 0.0f;
        for (int oy = -reach; oy <= reach; ++oy) {
            for (int ox = -reach; ox <= reach; ++ox) {
                const uint32_t x = (uint32_t)(cx + ox), y = (uint32_t)(cy + oy);
                const uint32_t id = grid_pos_to_id(u, x, y);
                walk_cell(s, s.p, id, [&](uint32_t, const fs_particle& nb) {
                    const float dx = nb.predicted_position.x - point.x;
                    const float dy = nb.predicted_position.y - point.y;
                    const float r2 = dx * dx + dy * dy;
                    const float kern = poly6(s, r2);
                    const float mul = 1.0f;
                    rho += u.particle_mass * kern * mul;        // funcs.wgsl:192
                });
            }
        }
        rho = std::fmax(rho, EPSILON_F);                        // funcs.wgsl:202
        s.p[pi].density = std::fmax(rho, 0.1f);                 // compute.wgsl:70
    }
}

// funcs.wgsl:129-149
inline uint32_t xorshift32(uint32_t* st) {
    uint32_t x = *st;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    *st = x;
    return x;
}
inline float rand_f32(uint32_t* st) { return (float)xorshift32(st) / 4294967296.0f; }

inline float calc_pressure(const fs_uniform& u, float rho) { return u.pressure_constant * (rho - u.rest_density); } // funcs.wgsl:152-154

// funcs.wgsl:101-109
inline float spiky_derivative(const fs_uniform& u, float h, float r) {
    if (r <= h) { const float v = h - r; return -v * u.spiky_kernel_derivative; }
    return 0.0f;
}
// funcs.wgsl:112-123
inline float viscosity_kernel(const fs_uniform& u, float h, float r) {
    if (r <= h) {
        const float c = u.viscosity_kernel;
        if (r == 0.0f) return c;
        return c * ((-(r * r * r) / (2.0f * h * h * h)) + ((r * r) / (h * h)) + (h / (2.0f * r)) - 1.0f);
    }
    return 0.0f;
}

// compute.wgsl:160-235 — reads the Jacobi snapshot `src`.
fs_vec2 pressure_force(const OrcSim& s, const std::vector<fs_particle>& src, uint32_t pid) {
    const fs_uniform& u = s.u;
    uint32_t seed = pid * 12u + u.frame_time * 69u;             // :161
    const fs_particle& me = src[pid];
    const float pressure = calc_pressure(u, me.density);
    const fs_vec2 pos = me.predicted_position;
    float fx = 0.0f, fy = 0.0f;
    uint32_t cxu, cyu;
    xy_of_point(u, pos, &cxu, &cyu);
    const int32_t cx = (int32_t)cxu, cy = (int32_t)cyu;
    for (int oy = -1; oy <= 1; ++oy) {
        for (int ox = -1; ox <= 1; ++ox) {
            const uint32_t id = grid_pos_to_id(u, (uint32_t)(cx + ox), (uint32_t)(cy + oy));
            walk_cell(s, src, id, [&](uint32_t i, const fs_particle& nb) {
                if (i == pid) return;                           // :195
                const float ox_ = nb.predicted_position.x - pos.x;
                const float oy_ = nb.predicted_position.y - pos.y;
                const float r2 = ox_ * ox_ + oy_ * oy_;
                if (r2 > u.sqr_radius) return;                  // :202
                const float dst = std::sqrt(r2);
                float dx, dy;
                if (dst == 0.0f) {                              // :211-212
                    const float rx = rand_f32(&seed);
                    const float ry = rand_f32(&seed);
                    const float len = std::sqrt(rx * rx + ry * ry);
                    dx = rx / len; dy = ry / len;
                } else {
                    dx = ox_ / dst; dy = oy_ / dst;
                }
                const float nrho = nb.density;
                const float npress = calc_pressure(u, nb.density);
                const float kern = spiky_derivative(u, u.smoothing_radius, dst);
                const float shared = (pressure + npress) * 0.5f;
                fx += dx * kern * shared / nrho;                // :223
                fy += dy * kern * shared / nrho;
            });
        }
    }
    return fs_vec2{fx, fy};
}

// compute.wgsl:238-299
fs_vec2 viscosity_force(const OrcSim& s, const std::vector<fs_particle>& src, uint32_t pid) {
    const fs_uniform& u = s.u;
    const fs_particle& me = src[pid];
    const fs_vec2 pos = me.predicted_position;
    float fx = 0.0f, fy = 0.0f;
    uint32_t cxu, cyu;
    xy_of_point(u, pos, &cxu, &cyu);
    const int32_t cx = (int32_t)cxu, cy = (int32_t)cyu;
    for (int oy = -1; oy <= 1; ++oy) {
        for (int ox = -1; ox <= 1; ++ox) {
            const uint32_t id = grid_pos_to_id(u, (uint32_t)(cx + ox), (uint32_t)(cy + oy));
            walk_cell(s, src, id, [&](uint32_t i, const fs_particle& nb) {
                if (i == pid) return;
                const float ox_ = nb.predicted_position.x - pos.x;
                const float oy_ = nb.predicted_position.y - pos.y;
                const float r2 = ox_ * ox_ + oy_ * oy_;
                if (r2 > u.sqr_radius) return;
                const float dst = std::sqrt(r2);
                const float nrho = nb.density;
                const float kern = viscosity_kernel(u, u.smoothing_radius, dst);
                fx += (nb.velocity.x - me.velocity.x) / nrho * kern;   // :288
                fy += (nb.velocity.y - me.velocity.y) / nrho * kern;
            });
        }
    }
    return fs_vec2{fx * u.viscosity_coefficient, fy * u.viscosity_coefficient};   // :298
}

// compute.wgsl:79-157 — Jacobi semantics: neighbours come from the pre-pass snapshot.
void move_particles(OrcSim& s) {
    const fs_uniform& u = s.u;
    s.snap = s.p;
    const std::vector<fs_particle>& src = s.snap;
    const uint32_t tex_w = f32_to_u32_sat(u.texture_size.x);
#pragma omp parallel for schedule(dynamic, 1024)
    for (uint32_t id = 0; id < u.particle_count; ++id) {
        fs_particle q = src[id];
        const fs_vec2 fp = pressure_force(s, src, id);
        const fs_vec2 fv = viscosity_force(s, src, id);
        const float ax = fp.x + fv.x, ay = fp.y + fv.y;                  // :93
        q.velocity.x += (ax / q.density) * u.delta;                     // :95
        q.velocity.y += (ay / q.density) * u.delta;
        q.velocity.x += u.gravity.x * u.delta;                          // :96
        q.velocity.y += u.gravity.y * u.delta;
        if (u.mouse_state != 0) {                                       // :99-108
            const float dx = u.mouse_pos.x - q.predicted_position.x;
            const float dy = u.mouse_pos.y - q.predicted_position.y;
            const float dist = std::sqrt(dx * dx + dy * dy);
            if (dist <= u.mouse_force_radius) {
                const float dirx = dx / dist / dist, diry = dy / dist / dist;
                const float ratio = dist / u.mouse_force_radius;
                q.velocity.x += dirx * u.mouse_force_power * (float)u.mouse_state * ratio;
                q.velocity.y += diry * u.mouse_force_power * (float)u.mouse_state * ratio;
            }
        }
        if (!(q.velocity.x == q.velocity.x && q.velocity.y == q.velocity.y)) {   // :113-116
            q.velocity.x = 0.0f; q.velocity.y = 0.0f;
        }
        const float max_speed = 500.0f;
        const float speed = std::sqrt(q.velocity.x * q.velocity.x + q.velocity.y * q.velocity.y);
        if (speed > max_speed) {                                        // :120-122
            q.velocity.x = (q.velocity.x / speed) * max_speed;
            q.velocity.y = (q.velocity.y / speed) * max_speed;
        }
        q.position.x += q.velocity.x * u.delta;                         // :125
        q.position.y += q.velocity.y * u.delta;

        const float uvx = (q.predicted_position.x / u.bounds.x * 1.0f) + 0.5f;   // :127
        const float uvy = (q.predicted_position.y / u.bounds.y * 1.0f) + 0.5f;
        const uint32_t px = f32_to_u32_sat(uvx * u.texture_size.x);     // :128
        const uint32_t py = f32_to_u32_sat(uvy * u.texture_size.y);
        const uint32_t tix = py * tex_w + px;                           // :129 (wrapping u32)
        fs_vec2 force{0.0f, 0.0f};
        if (tix < s.texture.size()) force = s.texture[tix];            // OOB -> zero (robust access)
        const float p2wx = (u.bounds.x * 2.0f) / u.texture_size.x;      // :131
        const float p2wy = (u.bounds.y * 2.0f) / u.texture_size.y;
        const float fwx = force.x * p2wx, fwy = force.y * p2wy;         // :132
        if (force.x != 0.0f || force.y != 0.0f) {                       // :134-140
            const float len = std::sqrt(force.x * force.x + force.y * force.y);
            const float nx = force.x / len, ny = force.y / len;
            q.position.x += fwx; q.position.y += fwy;
            const float vn = q.velocity.x * nx + q.velocity.y * ny;
            q.velocity.x -= (1.0f - u.damping_factor) * vn * nx;
            q.velocity.y -= (1.0f - u.damping_factor) * vn * ny;
        }
        const float bsx = u.bounds.x * 0.5f, bsy = u.bounds.y * 0.5f;   // :143-153
        if (std::fabs(q.position.x) > bsx) {
            q.position.x = bsx * sign_f32(q.position.x);
            q.velocity.x *= -1.0f * u.damping_factor;
        }
        if (std::fabs(q.position.y) > bsy) {
            q.position.y = bsy * sign_f32(q.position.y);
            q.velocity.y *= -1.0f * u.damping_factor;
        }
        s.p[id] = q;                                                    // :155
    }
}

}  // namespace

extern "C" {

typedef struct orc_sim orc_sim;

/* Threads used by the passes above (default 1 = the scalar port).  Results do not depend on it:
 * every pass is independent per particle / per disjoint pair. */
void orc_set_threads(int n) { omp_set_num_threads(n < 1 ? 1 : n); }
int orc_max_threads(void) { return omp_get_num_procs(); }

orc_sim* orc_create(const fs_settings* st, float off_x, float off_y, int ref_quirks) {
    if (!st || st->particle_count <= 1) return nullptr;   // simulation.rs:323-324 would panic
    OrcSim* s = new OrcSim();
    s->settings = *st;
    s->offset = fs_vec2{off_x, off_y};
    s->ref_quirks = ref_quirks;
    grid_dims(*st, &s->grid_w, &s->grid_h);
    s->p.resize(st->particle_count);
    lattice(*st, s->offset, s->p.data(), s->p.size());
    s->start_indices.assign((size_t)s->grid_w * s->grid_h, 0u);
    s->texture.assign((size_t)st->texture_size.x * st->texture_size.y, fs_vec2{0, 0});
    return (orc_sim*)s;
}
void orc_destroy(orc_sim* h) { delete (OrcSim*)h; }

/* Individual passes in reference dispatch order (simulation.rs:512-537). */
void orc_begin_tick(orc_sim* h, const fs_tick_settings* t) {
    OrcSim& s = *(OrcSim*)h;
    s.tick += 1;                                                        // :460
    build_uniform(s.settings, *t, s.tick, &s.u);
    s.poly6_norm = 4.0f / (PI_F * std::pow(s.u.smoothing_radius, 8.0f)); // funcs.wgsl:76
}
void orc_predict(orc_sim* h) { predict(*(OrcSim*)h); }
void orc_spatial_lookup(orc_sim* h) { spatial_lookup(*(OrcSim*)h); }
void orc_sort(orc_sim* h) {
    OrcSim& s = *(OrcSim*)h;
    bitonic_network(s.p.data(), s.u.particle_count, [](const fs_particle& q) { return q.grid; });
}
/* FS_SORT_COUNTING counterpart (NOT the reference's sort): stable sort by cell key. */
void orc_sort_stable(orc_sim* h) {
    OrcSim& s = *(OrcSim*)h;
    std::stable_sort(s.p.begin(), s.p.begin() + s.u.particle_count,
                     [](const fs_particle& a, const fs_particle& b) { return a.grid < b.grid; });
}
void orc_cell_starts(orc_sim* h) { cell_starts(*(OrcSim*)h); }
void orc_density(orc_sim* h, int reach) { density(*(OrcSim*)h, reach); }
void orc_move(orc_sim* h) { move_particles(*(OrcSim*)h); }

void orc_step_stable(orc_sim* h, const fs_tick_settings* t) {   /* step with the stable sort */
    orc_begin_tick(h, t);
    orc_predict(h);
    orc_spatial_lookup(h);
    orc_sort_stable(h);
    orc_cell_starts(h);
    orc_density(h, 1);
    orc_move(h);
}

void orc_step(orc_sim* h, const fs_tick_settings* t) {
    orc_begin_tick(h, t);
    orc_predict(h);
    orc_spatial_lookup(h);
    orc_sort(h);
    orc_cell_starts(h);
    orc_density(h, 1);      // 3x3 == 7x7 bit-for-bit (checked in tests/test_oracle.py)
    orc_move(h);
}

uint32_t orc_tick(const orc_sim* h) { return ((const OrcSim*)h)->tick; }
uint32_t orc_count(const orc_sim* h) { return ((const OrcSim*)h)->settings.particle_count; }
void orc_grid(const orc_sim* h, uint32_t* gw, uint32_t* gh) { *gw = ((const OrcSim*)h)->grid_w; *gh = ((const OrcSim*)h)->grid_h; }
fs_particle* orc_particles(orc_sim* h) { return ((OrcSim*)h)->p.data(); }
uint32_t* orc_start_indices(orc_sim* h) { return ((OrcSim*)h)->start_indices.data(); }
size_t orc_start_indices_len(const orc_sim* h) { return ((const OrcSim*)h)->start_indices.size(); }
fs_vec2* orc_texture(orc_sim* h) { return ((OrcSim*)h)->texture.data(); }
void orc_uniform(const orc_sim* h, fs_uniform* out) { *out = ((const OrcSim*)h)->u; }
float orc_poly6_norm(const orc_sim* h) { return ((const OrcSim*)h)->poly6_norm; }

/* fluid_shader.wgsl:27-102 — density-splat fragment shader on the oracle's state, walking
 * start_indices exactly like the shader.  View mapping: see include/fluidsim.h fs_view. */
void orc_render(orc_sim* h, float wminx, float wminy, float wmaxx, float wmaxy, uint32_t width, uint32_t height, float* rgba) {
    OrcSim& s = *(OrcSim*)h;
    const fs_uniform& u = s.u;
    auto smooth = [](float a, float b, float x) { float t = (x - a) / (b - a); t = std::fmin(std::fmax(t, 0.0f), 1.0f); return t * t * (3.0f - 2.0f * t); };
    for (uint32_t j = 0; j < height; ++j)
        for (uint32_t i = 0; i < width; ++i) {
            fs_vec2 pt;
            pt.x = wminx + (((float)i + 0.5f) / (float)width) * (wmaxx - wminx);
            pt.y = wminy + (((float)j + 0.5f) / (float)height) * (wmaxy - wminy);
            uint32_t cxu, cyu;
            xy_of_point(u, pt, &cxu, &cyu);
            const int32_t cx = (int32_t)cxu, cy = (int32_t)cyu;
            float density = 0.0f, vfac = 0.0f;
            for (int oy = -2; oy < 3; ++oy)
                for (int ox = -2; ox < 3; ++ox) {
                    const uint32_t x = (uint32_t)(cx + ox), y = (uint32_t)(cy + oy);
                    if (x >= u.grid_w || y >= u.grid_h) continue;      // wrapped ids alias empty cells / OOB: nothing
                    walk_cell(s, s.p, grid_pos_to_id(u, x, y), [&](uint32_t, const fs_particle& nb) {
                        const float dx = nb.predicted_position.x - pt.x, dy = nb.predicted_position.y - pt.y;
                        const float r2 = dx * dx + dy * dy;
                        const float contrib = std::exp(-r2 / (u.sqr_radius / 2.0f));
                        density += contrib;
                        vfac += contrib * std::sqrt(nb.velocity.x * nb.velocity.x + nb.velocity.y * nb.velocity.y);
                    });
                }
            vfac = vfac * 0.01f;
            vfac = std::log(1.0f + 5.0f * vfac) / std::log(1.0f + 5.0f);
            vfac = std::fmin(std::fmax(vfac, 0.0f), 1.0f);
            const float interior = smooth(0.5f, 1.5f, density);
            float edge = smooth(0.7f, 1.0f, density) - smooth(1.0f, 1.5f, density);
            edge = edge * (1.0f + vfac * 2.0f);
            float* o = rgba + 4 * ((size_t)j * width + i);
            o[0] = (0.0f * (1.0f - vfac) + 1.0f * vfac) * interior + edge;
            o[1] = (0.5f * (1.0f - vfac) + 0.0f * vfac) * interior + edge;
            o[2] = (1.0f * (1.0f - vfac) + 0.0f * vfac) * interior + edge;
            o[3] = std::fmin(std::fmax(interior, 0.0f), 1.0f);
        }
}

/* src/main.rs:403-515 — generate_smooth_gradient_field, restated line by line. */
void orc_gradient_field(const uint8_t* img, uint32_t width, uint32_t height, fs_vec2* out) {
    const size_t W = width, H = height;
    std::vector<float> dist(W * H, 3.40282347e+38f);                    // f32::MAX, :408
    std::vector<uint32_t> nx_(W * H, 0), ny_(W * H, 0);                 // nearest, :410
    bool has_white = false;
    for (size_t y = 0; y < H; ++y)                                      // :413-422
        for (size_t x = 0; x < W; ++x)
            if (img[y * W + x] > 128) { dist[y * W + x] = 0.0f; nx_[y * W + x] = (uint32_t)x; ny_[y * W + x] = (uint32_t)y; has_white = true; }
    if (!has_white)                                                     // :426-438
        for (size_t y = 0; y < H; ++y)
            for (size_t x = 0; x < W; ++x)
                if (y == H - 1 || y == 0 || x == W - 1 || x == 0) { dist[y * W + x] = 0.0f; nx_[y * W + x] = (uint32_t)x; ny_[y * W + x] = (uint32_t)y; }
    auto sq = [](size_t x1, size_t y1, size_t x2, size_t y2) { const float dx = (float)x1 - (float)x2, dy = (float)y1 - (float)y2; return dx * dx + dy * dy; };
    auto relax = [&](size_t x, size_t y, size_t nx, size_t ny) {
        if (nx < W && ny < H) {                                         // usize wrap-around makes -1 huge
            const uint32_t cx = nx_[ny * W + nx], cy = ny_[ny * W + nx];
            const float cd = sq(x, y, cx, cy);
            if (cd < dist[y * W + x]) { dist[y * W + x] = cd; nx_[y * W + x] = cx; ny_[y * W + x] = cy; }
        }
    };
    for (size_t y = 0; y < H; ++y)                                      // forward pass, :447-468
        for (size_t x = 0; x < W; ++x) { relax(x, y, x - 1, y); relax(x, y, x - 1, y - 1); relax(x, y, x, y - 1); relax(x, y, x + 1, y - 1); }
    for (size_t y = H; y-- > 0;)                                        // backward pass, :470-491
        for (size_t x = W; x-- > 0;) { relax(x, y, x + 1, y); relax(x, y, x + 1, y + 1); relax(x, y, x, y + 1); relax(x, y, x - 1, y + 1); }
    for (size_t y = 0; y < H; ++y)                                      // :496-511
        for (size_t x = 0; x < W; ++x) {
            const float dx = (float)x - (float)nx_[y * W + x], dy = (float)y - (float)ny_[y * W + x];
            const float len = std::sqrt(dx * dx + dy * dy);
            out[y * W + x] = fs_vec2{-(len > 1e-6f ? dx : 0.0f), -(len > 1e-6f ? dy : 0.0f)};
        }
}

/* Pure helpers. */
void orc_lattice(const fs_settings* st, float off_x, float off_y, fs_particle* dst, size_t n) {
    lattice(*st, fs_vec2{off_x, off_y}, dst, n);
}
size_t orc_sort_schedule(uint32_t n, fs_sort_step* dst, size_t cap) { return sort_schedule(n, dst, cap); }
void orc_build_uniform(const fs_settings* st, const fs_tick_settings* t, uint32_t tick, fs_uniform* out) {
    build_uniform(*st, *t, tick, out);
}
void orc_grid_dims(const fs_settings* st, uint32_t* gw, uint32_t* gh) { grid_dims(*st, gw, gh); }
/* Bitonic network over bare keys; perm_out receives the source index of each output slot. */
void orc_bitonic_keys(uint32_t* keys, uint32_t* perm_out, uint32_t n) {
    struct KV { uint32_t k, i; };
    std::vector<KV> v(n);
    for (uint32_t i = 0; i < n; ++i) v[i] = KV{keys[i], i};
    bitonic_network(v.data(), n, [](const KV& a) { return a.k; });
    for (uint32_t i = 0; i < n; ++i) { keys[i] = v[i].k; if (perm_out) perm_out[i] = v[i].i; }
}
float orc_poly6_value(float h, float r2) {
    OrcSim s; s.u.smoothing_radius = h; s.poly6_norm = 4.0f / (PI_F * std::pow(h, 8.0f));
    return poly6(s, r2);
}

}  // extern "C"

/*
 * sph_oracle3d.cpp — CPU statement of the 3D extension of the step.  TEST INFRASTRUCTURE ONLY
 * (same rules as sph_oracle.cpp).
 *
 * The reference is 2D only: there is NO reference counterpart for anything in this file
 * (SURVEY.md §8c last row, Appendix B.3).  It keeps the reference's pass structure and
 * kernel *shapes* (compute.wgsl:8-299, funcs.wgsl:72-218) with a third coordinate:
 *   key        = (cz*grid_h + cy)*grid_w + cx, c = floor((pred + bounds/2)/h) + 1
 *   sweep      = 27 cells, z outer, y, x inner, index ascending
 *   poly6      = 315/(64 pi h^9) (h^2 - r^2)^3
 *   pressure   = -(h - r) * 15/(pi h^5)      (derivative shape of funcs.wgsl:101-109)
 *   viscosity  = 15/(2 pi h^3) * (-(r^3)/(2h^3) + r^2/h^2 + h/(2r) - 1)   (funcs.wgsl:112-123)
 *   guards     = density floor 0.1, NaN reset, |v| <= 500, damped wall bounce (compute.wgsl:113-153)
 * Cell starts are rebuilt cleanly every step (no stale-start quirk: that is a property of the
 * 2D reference only); no mouse force and no obstacle field in 3D.
 * IEEE f32, written association, build with -ffp-contract=off.
 * OpenMP over particles / over the disjoint pairs of one sort dispatch (orc_set_threads of sph_oracle.cpp
 * sets the count): every iteration writes its own record only, so results do not depend on the thread count
 * (the full-size parity tests run it on all host cores).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/fluidsim.h"

namespace {

const float PI3 = 3.14159265359f;
const float EPS3 = 1.19209290e-07f;

uint32_t u32sat(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
float sgn(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

struct Sim3 {
    fs3_settings st{};
    fs3_tick_settings tk{};
    uint32_t tick = 0;
    uint32_t gw = 0, gh = 0, gd = 0;
    float poly6 = 0, spiky = 0, visc = 0;
    std::vector<fs3_particle> p, snap;
    std::vector<uint32_t> starts;   // first index of each cell, 0xFFFFFFFF when empty (clean rebuild)
};

void dims(const fs3_settings& s, uint32_t* w, uint32_t* h, uint32_t* d) {
    *w = (uint32_t)((size_t)std::ceil(s.size.x / s.smoothing_radius) + 2);   // as src/simulation.rs:140-141
    *h = (uint32_t)((size_t)std::ceil(s.size.y / s.smoothing_radius) + 2);
    *d = (uint32_t)((size_t)std::ceil(s.size.z / s.smoothing_radius) + 2);
}

void cell_xyz(const Sim3& s, const float* pt, uint32_t* c) {
    c[0] = u32sat(std::floor((pt[0] + s.st.size.x * 0.5f) / s.st.smoothing_radius)) + 1u;
    c[1] = u32sat(std::floor((pt[1] + s.st.size.y * 0.5f) / s.st.smoothing_radius)) + 1u;
    c[2] = u32sat(std::floor((pt[2] + s.st.size.z * 0.5f) / s.st.smoothing_radius)) + 1u;
}
uint32_t cell_id(const Sim3& s, uint32_t x, uint32_t y, uint32_t z) { return (z * s.gh + y) * s.gw + x; }

template <class T, class K>
void bitonic(T* v, uint32_t n, K key) {   // the reference network (sort.wgsl:27-51) — same as 2D
    if (n <= 1) return;
    uint32_t p2 = 1, stages = 0;
    while (p2 < n) { p2 <<= 1; ++stages; }
    const uint32_t threads = ((p2 / 2 + 127) / 128) * 128;
    for (uint32_t stage = 0; stage < stages; ++stage)
        for (uint32_t step = 0; step <= stage; ++step) {
            const uint32_t gw = 1u << (stage - step), gh = 2 * gw - 1;
#pragma omp parallel for schedule(static) if (threads > 65536)      // pairs of one dispatch are disjoint
            for (uint32_t i = 0; i < threads; ++i) {
                const uint32_t hh = i & (gw - 1), lo = hh + (gh + 1) * (i / gw);
                const uint32_t hi = lo + (step == 0 ? gh - 2 * hh : (gh + 1) / 2);
                if (hi >= n) continue;
                if (key(v[lo]) > key(v[hi])) std::swap(v[lo], v[hi]);
            }
        }
}

template <class F>
void walk(const Sim3& s, const std::vector<fs3_particle>& arr, uint32_t id, F&& f) {
    if (id >= s.starts.size()) return;
    uint32_t k = s.starts[id];
    const uint32_t n = s.st.particle_count;
    while (k < n && arr[k].grid == id) { f(k, arr[k]); ++k; }
}

void step3(Sim3& s) {
    const uint32_t n = s.st.particle_count;
    const float dt = s.tk.delta, h = s.st.smoothing_radius;
    const float bs[3] = {s.st.size.x * 0.5f, s.st.size.y * 0.5f, s.st.size.z * 0.5f};
    // predict (compute.wgsl:16-26) + key (compute.wgsl:33-42)
#pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i < n; ++i) {
        fs3_particle& q = s.p[i];
        float* pr = &q.predicted_position.x;
        const float* po = &q.position.x;
        const float* ve = &q.velocity.x;
        for (int a = 0; a < 3; ++a) {
            pr[a] = po[a] + ve[a] * dt;
            if (std::fabs(pr[a]) > bs[a]) pr[a] = bs[a] * sgn(pr[a]);
        }
        uint32_t c[3];
        cell_xyz(s, pr, c);
        q.grid = cell_id(s, c[0], c[1], c[2]);
    }
    bitonic(s.p.data(), n, [](const fs3_particle& q) { return q.grid; });
    std::fill(s.starts.begin(), s.starts.end(), 0xFFFFFFFFu);
    for (uint32_t i = 0; i < n; ++i)
        if ((i == 0 || s.p[i].grid != s.p[i - 1].grid) && s.p[i].grid < s.starts.size()) s.starts[s.p[i].grid] = i;
    // density (compute.wgsl:59-74 shape)
    const float h2 = h * h;
#pragma omp parallel for schedule(dynamic, 1024)
    for (uint32_t i = 0; i < n; ++i) {
        const float* me = &s.p[i].predicted_position.x;
        uint32_t c[3];
        cell_xyz(s, me, c);
        float rho = 0.0f;
        for (int oz = -1; oz <= 1; ++oz)
            for (int oy = -1; oy <= 1; ++oy)
                for (int ox = -1; ox <= 1; ++ox) {
                    const uint32_t x = c[0] + ox, y = c[1] + oy, z = c[2] + oz;
                    if (x >= s.gw || y >= s.gh || z >= s.gd) continue;
                    walk(s, s.p, cell_id(s, x, y, z), [&](uint32_t, const fs3_particle& nb) {
                        const float dx = nb.predicted_position.x - me[0], dy = nb.predicted_position.y - me[1],
                                    dz = nb.predicted_position.z - me[2];
                        const float r2 = dx * dx + dy * dy + dz * dz;
                        float kern = 0.0f;
                        if (!(r2 > h2)) { const float d = h2 - r2; kern = s.poly6 * d * d * d; }
                        rho += s.tk.mass * kern * 1.0f;
                    });
                }
        rho = std::fmax(rho, EPS3);
        s.p[i].density = std::fmax(rho, 0.1f);
    }
    // force + integrate (compute.wgsl:79-157 shape), Jacobi snapshot
    s.snap = s.p;
    const std::vector<fs3_particle>& src = s.snap;
#pragma omp parallel for schedule(dynamic, 1024)
    for (uint32_t i = 0; i < n; ++i) {
        fs3_particle q = src[i];
        const float* me = &q.predicted_position.x;
        const float pressure = s.tk.pressure_constant * (q.density - s.tk.rest_density);
        uint32_t seed = i * 12u + s.tick * 69u;
        float fp[3] = {0, 0, 0}, fv[3] = {0, 0, 0};
        uint32_t c[3];
        cell_xyz(s, me, c);
        for (int oz = -1; oz <= 1; ++oz)
            for (int oy = -1; oy <= 1; ++oy)
                for (int ox = -1; ox <= 1; ++ox) {
                    const uint32_t x = c[0] + ox, y = c[1] + oy, z = c[2] + oz;
                    if (x >= s.gw || y >= s.gh || z >= s.gd) continue;
                    walk(s, src, cell_id(s, x, y, z), [&](uint32_t k, const fs3_particle& nb) {
                        if (k == i) return;
                        const float o[3] = {nb.predicted_position.x - me[0], nb.predicted_position.y - me[1],
                                            nb.predicted_position.z - me[2]};
                        const float r2 = o[0] * o[0] + o[1] * o[1] + o[2] * o[2];
                        if (r2 > h2) return;
                        const float dst = std::sqrt(r2);
                        float dir[3];
                        if (dst == 0.0f) {
                            float r[3];
                            for (int a = 0; a < 3; ++a) {
                                seed ^= seed << 13; seed ^= seed >> 17; seed ^= seed << 5;
                                r[a] = (float)seed / 4294967296.0f;
                            }
                            const float len = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
                            for (int a = 0; a < 3; ++a) dir[a] = r[a] / len;
                        } else {
                            for (int a = 0; a < 3; ++a) dir[a] = o[a] / dst;
                        }
                        const float nrho = nb.density;
                        const float npress = s.tk.pressure_constant * (nrho - s.tk.rest_density);
                        const float kern = (dst <= h) ? (-(h - dst)) * s.spiky : 0.0f;
                        const float shared = (pressure + npress) * 0.5f;
                        float kv = 0.0f;
                        if (dst <= h)
                            kv = (dst == 0.0f) ? s.visc
                                               : s.visc * ((-(dst * dst * dst) / (2.0f * h * h * h)) + ((dst * dst) / (h * h)) +
                                                           (h / (2.0f * dst)) - 1.0f);
                        const float* nv = &nb.velocity.x;
                        const float* mv = &q.velocity.x;
                        for (int a = 0; a < 3; ++a) {
                            fp[a] += dir[a] * kern * shared / nrho;
                            fv[a] += (nv[a] - mv[a]) / nrho * kv;
                        }
                    });
                }
        float* v = &q.velocity.x;
        float* x = &q.position.x;
        const float g[3] = {s.tk.gravity.x, s.tk.gravity.y, s.tk.gravity.z};
        for (int a = 0; a < 3; ++a) {
            const float acc = fp[a] + fv[a] * s.tk.viscosity_coefficient;
            v[a] += (acc / q.density) * dt;
            v[a] += g[a] * dt;
        }
        if (!(v[0] == v[0] && v[1] == v[1] && v[2] == v[2])) v[0] = v[1] = v[2] = 0.0f;
        const float speed = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (speed > 500.0f) for (int a = 0; a < 3; ++a) v[a] = (v[a] / speed) * 500.0f;
        for (int a = 0; a < 3; ++a) x[a] += v[a] * dt;
        for (int a = 0; a < 3; ++a)
            if (std::fabs(x[a]) > bs[a]) { x[a] = bs[a] * sgn(x[a]); v[a] *= -1.0f * s.tk.damping_factor; }
        s.p[i] = q;
    }
}

}  // namespace

extern "C" {

void orc3_lattice(const fs3_settings* st, float ox, float oy, float oz, fs3_particle* dst, size_t n) {
    // build-defined cube lattice: side = round(cbrt(N)); x fastest, then y, then z; centred, then offset
    const uint32_t side = (uint32_t)std::llround(std::cbrt((double)st->particle_count));
    const float half = (float)side * 0.5f, s = st->particle_spacing;
    for (uint32_t i = 0; i < st->particle_count && i < n; ++i) {
        const uint32_t ix = i % side, iy = (i / side) % side, iz = i / (side * side);
        fs3_particle q;
        std::memset(&q, 0, sizeof q);
        q.position.x = ((float)ix - half + 0.5f) * s + ox;
        q.position.y = ((float)iy - half + 0.5f) * s + oy;
        q.position.z = ((float)iz - half + 0.5f) * s + oz;
        q.predicted_position = q.position;
        dst[i] = q;
    }
}

void* orc3_create(const fs3_settings* st, float ox, float oy, float oz) {
    if (!st || st->particle_count <= 1) return nullptr;
    Sim3* s = new Sim3();
    s->st = *st;
    dims(*st, &s->gw, &s->gh, &s->gd);
    s->p.resize(st->particle_count);
    orc3_lattice(st, ox, oy, oz, s->p.data(), s->p.size());
    s->starts.assign((size_t)s->gw * s->gh * s->gd, 0xFFFFFFFFu);
    return s;
}
void orc3_destroy(void* h) { delete (Sim3*)h; }
void orc3_step(void* hh, const fs3_tick_settings* t) {
    Sim3& s = *(Sim3*)hh;
    s.tick += 1;
    s.tk = *t;
    const float h = s.st.smoothing_radius;
    s.poly6 = 315.0f / (64.0f * PI3 * std::pow(h, 9.0f));
    s.spiky = 15.0f / (PI3 * std::pow(h, 5.0f));
    s.visc = 15.0f / (2.0f * PI3 * (h * h * h));
    step3(s);
}
fs3_particle* orc3_particles(void* h) { return ((Sim3*)h)->p.data(); }
uint32_t orc3_count(void* h) { return ((Sim3*)h)->st.particle_count; }
void orc3_grid(void* h, uint32_t* w, uint32_t* hh, uint32_t* d) { Sim3* s = (Sim3*)h; *w = s->gw; *hh = s->gh; *d = s->gd; }
void orc3_constants(void* h, float* out3) { Sim3* s = (Sim3*)h; out3[0] = s->poly6; out3[1] = s->spiky; out3[2] = s->visc; }

}  // extern "C"

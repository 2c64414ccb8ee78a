// engine.hip — host side of the C ABI (include/fluidsim.h): simulation handle,
// SoA device state, pass chain on one HIP stream, AoS import/export.
//
// Mirrors FluidSimulation::{new,tick,accessors} (src/simulation.rs:139-564).
// There is no CPU fallback: every entry point that computes requires a HIP device.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/fluidsim.h"
#include "fs_kernels.h"
#include "sort_policy.h"

static_assert(sizeof(fs_particle) == 32, "ParticleInstance is 32 bytes (src/simulation.rs:126-135)");
static_assert(offsetof(fs_particle, predicted_position) == 8 && offsetof(fs_particle, velocity) == 16 &&
                  offsetof(fs_particle, density) == 24 && offsetof(fs_particle, grid) == 28,
              "ParticleInstance offsets");
static_assert(sizeof(fs_uniform) == 120, "SimulationUniform is 120 bytes (src/simulation.rs:53-90)");
static_assert(offsetof(fs_uniform, gravity) == 16 && offsetof(fs_uniform, smoothing_radius) == 40 &&
                  offsetof(fs_uniform, poly6_kernel_volume) == 72 && offsetof(fs_uniform, mouse_state) == 92 &&
                  offsetof(fs_uniform, grid_w) == 104 && offsetof(fs_uniform, texture_size) == 112,
              "SimulationUniform offsets");

namespace {

thread_local std::string g_err;

fs_status fail(fs_status st, const std::string& msg) {
    g_err = msg;
    return st;
}
}  // namespace
namespace fsd { void set_last_error(const std::string& msg) { g_err = msg; } }   // used by sim3d.hip
namespace {

#define FS_HIP(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess) {                                                                         \
            return fail(e__ == hipErrorOutOfMemory ? FS_ERR_OOM : FS_ERR_DEVICE,                         \
                        std::string(#expr) + ": " + hipGetErrorString(e__));                             \
        }                                                                                                \
    } while (0)

const float PI_F = 3.14159265359f;   // funcs.wgsl:54 == std::f32::consts::PI in f32

// Rust f32::powi (llvm.powi -> compiler-rt __powisf2): square-and-multiply.
float powi_f32(float a, int b) {
    float r = 1.0f;
    const bool recip = b < 0;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0f / r : r;
}

// src/simulation.rs:140-141
void grid_dims(const fs_settings& s, uint32_t* gw, uint32_t* gh) {
    *gw = (uint32_t)((size_t)std::ceil(s.size.x / s.smoothing_radius) + 2);
    *gh = (uint32_t)((size_t)std::ceil(s.size.y / s.smoothing_radius) + 2);
}

bool settings_valid(const fs_settings& s, std::string* why) {
    if (s.particle_count <= 1) { *why = "particle_count <= 1 (reference panics in ilog2, src/simulation.rs:323-324)"; return false; }
    if (s.particle_count > (1u << 28)) { *why = "particle_count > 2^28 (the kernels use 32-bit byte offsets into 8-byte arrays)"; return false; }
    if (!(s.smoothing_radius > 0.0f) || !std::isfinite(s.smoothing_radius)) { *why = "smoothing_radius must be finite and > 0"; return false; }
    if (!(s.size.x > 0.0f) || !(s.size.y > 0.0f) || !std::isfinite(s.size.x) || !std::isfinite(s.size.y)) { *why = "size must be finite and > 0"; return false; }
    if (!std::isfinite(s.particle_spacing)) { *why = "particle_spacing must be finite"; return false; }
    const double gw = std::ceil((double)s.size.x / s.smoothing_radius) + 2, gh = std::ceil((double)s.size.y / s.smoothing_radius) + 2;
    if (gw * gh >= 4294967295.0) { *why = "grid_w*grid_h does not fit u32 cell ids"; return false; }
    if ((double)s.texture_size.x * s.texture_size.y >= 4294967295.0) { *why = "texture too large"; return false; }
    return true;
}

template <class T>
struct DevArray {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        n = count;
        if (count == 0) { p = nullptr; return hipSuccess; }
        return hipMalloc((void**)&p, count * sizeof(T));
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

struct fs_sim {
    fs_settings settings{};
    fs_options opts{};
    uint32_t n = 0, capacity = 0;
    uint32_t grid_w = 0, grid_h = 0, ncell = 0;
    uint32_t tick = 0;
    fs_uniform uniform{};
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;     // second stream of the force pass (general workgroups beside the lean kernel)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int device = 0;

    // SoA state.  pos/vel: current state (cell order of the last step).  *_s: the
    // cell-sorted snapshot the density/force passes read (Jacobi semantics).
    DevArray<float2> pos, vel, pos_s, vel_s, pred;
    DevArray<float> rho;
    DevArray<float2> rho2;          // {density, RN(1/density)}: what the force pass gathers per neighbour
    DevArray<uint32_t> key;         // keys of an uploaded / initial state; after a step they live in `pairs` (key_in_pairs)
    bool key_in_pairs = false;
    bool rho_in_rho2 = false;       // likewise the densities: rho2.x after a strict / ulp step, `rho` after an upload or a tolerance step
    DevArray<uint32_t> fdefer, fwork;   // force pass: per-block deferred-wave bits and the worklist (counter[3] = its length)
    DevArray<uint32_t> bbounds;         // 8 words per 256-particle block: the density pass's block-wide sweep ranges, read by the force pass
    DevArray<unsigned long long> safe;   // one bit per sorted particle: coordinates / velocity inside the exact-quotient ranges (fs_device.h)
    DevArray<fsd::u64> pairs;
    DevArray<uint32_t> sort_dirty;  // per-tile flags of the bitonic sort
    DevArray<uint32_t> csort;       // scratch of the counting sort (FS_SORT_COUNTING)
    DevArray<uint32_t> cs;          // dense cell-start table, ncell+1
    DevArray<uint32_t> start_ref;   // reference start_indices (persistent, never cleared)
    DevArray<float2> tex;           // force field
    bool tex_zero = true;           // the host knows every entry is +-0 (zero-initialised, or an all-zero upload)
    DevArray<unsigned char> work;   // gap worklist
    DevArray<uint32_t> counter;
    uint32_t work_cap = 0;
    DevArray<fs_particle> aos;      // lazily allocated 32-byte view
    bool aos_live = false;          // a hand-off is registered: the force pass writes the AoS records itself
    uint32_t aos_tick = 0xFFFFFFFFu;   // tick whose state the AoS view holds (only meaningful with aos_live)

    fsd::ConstDiv div_2h3{}, div_h2{};   // exact constant divisions of the force pass, proven at create
    fsd::ConstDiv div_h{};               // ... and of the cell coordinates (x / h), over the numerators clamped positions give
    bool rcp_ok = false, sqrt_ok = false; // rcp_rn_fast / sqrt_rn_fast proven on this device at create

    fsd::SortPolicy sortp;          // host side of the sort's late-stage plan (sort_policy.h)

    // slab (multi-GPU) mode
    bool slab = false;
    fs_slab_config slab_cfg{};
    uint32_t slab_main = 0;         // capacity - 2 * recv_capacity
    DevArray<unsigned char> owned;
    DevArray<uint2> blockcnt;           // per 256-slot block: records for the left / right message (k_slab_pack)
    DevArray<uint32_t> stage;           // ... and the slots themselves, 2 x 256 entries per block
    DevArray<fsd::u64> msg_state;       // k_slab_msg's look-back words
    DevArray<uint32_t> slab_counters;   // [0] n_live, [2] lost, [3] overflow, [4] far_halo
    DevArray<uint32_t> hist;
    bool slab_packed = false;
    bool slab_prof = false;                // profiling state latched by fs_slab_pack for the matching fs_slab_step
    uint32_t state_lo = 0, state_hi = 0;   // the owned window the current keys / cell starts were built with

    // Overlapped slab step (counting sort only): fs_slab_pack enqueues the pack AND the whole step of the
    // interior columns; the halo exchange runs beside it on `comm`; fs_slab_step finishes the columns within `boundary_cols`
    // of a slab edge on the strip arrays (kernels_slab.hip "boundary strips").
    bool transposed = false;               // cell ids column-major (fs_device.h StepParams::transposed): ranks with neighbours, not the strip step
    bool overlap = false;                  // FS_SLAB_STRIPS: ghosts stay out of the main array, boundary strips after the exchange
    bool edge_first = false;               // the default: the next step's messages are built right after the edge columns' force launch
    bool prepacked = false;                // edge-first: the send buffers hold the messages of tick + 1 ...
    void *pp_left = nullptr, *pp_right = nullptr;   // ... these buffers (the ones the last fs_slab_pack was given)
    float pp_delta = 0.0f;                 // ... built with this delta
    uint32_t pp_lo = 0, pp_hi = 0;         // ... and this owned window
    uint32_t msg_epoch = 0;                // one number per k_slab_msg launch (its look-back state is never cleared)
    fs_tick_settings last_tick{};
    hipStream_t comm = nullptr;            // the exchange's stream (fs_slab_exchange, or the caller's transport between comm_begin / comm_end)
    hipEvent_t ev_packed = nullptr;        // main stream: both outgoing messages are complete
    hipEvent_t ev_exch = nullptr;          // comm stream: both incoming messages have arrived
    hipEvent_t ev_fork2 = nullptr;         // edge-first: the density pass is done, the edge columns' chain may start on `comm`
    bool exch_pending = false;             // fs_slab_step must wait for ev_exch
    bool join_pending = false;             // edge-first: the simulation's stream has not yet waited for the edge columns' chain of the last step (slab_join)
    bool edge_classified = false;          // edge-first: that chain also classified its particles' slots for the next pack
    uint32_t boundary_cols = 4;            // owned columns per neighboured edge left to the strips (>= 3)
    uint32_t pending_shift = 0;            // columns a window edge moved since the last pack: that step's migrants land deeper
    uint32_t adv_lo = 0, adv_hi = 0;       // interior columns of the step being enqueued
    uint32_t strip_win[4] = {0, 0, 0, 0};
    bool strip_active = false;
    struct Strip {
        DevArray<float2> pos, vel, pos_s, vel_s, pred, rho2, pos_out, vel_out;
        DevArray<float> rho;
        DevArray<fsd::u64> pairs;
        DevArray<uint32_t> csort, cs, start_ref, fdefer, fwork, counter, back, rowbase, counters, bbounds;
        DevArray<unsigned long long> safe;
        DevArray<unsigned char> owned;
        uint32_t cap = 0;
        void release() {
            pos.release(); vel.release(); pos_s.release(); vel_s.release(); pred.release(); rho2.release(); pos_out.release();
            vel_out.release(); rho.release(); pairs.release(); csort.release(); cs.release(); start_ref.release(); fdefer.release();
            fwork.release(); counter.release(); back.release(); rowbase.release(); counters.release(); safe.release(); owned.release(); bbounds.release();
        }
    } strip;

    // Per-pass timing: a ring of event sets recorded on the stream; drained (synchronised
    // and accumulated) only when read or when the ring is full, never per step.
    static const uint32_t PROF_RING = 256;
    bool profile = false;
    std::vector<hipEvent_t> ev;     // PROF_RING * (FS_PASS_COUNT + 1)
    uint32_t prof_pending = 0;
    double prof_ms[FS_PASS_COUNT] = {};
    uint64_t prof_steps = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;

    void release() {
        pos.release(); vel.release(); pos_s.release(); vel_s.release(); pred.release(); rho.release(); rho2.release();
        key.release(); safe.release(); fdefer.release(); fwork.release(); bbounds.release(); pairs.release(); sort_dirty.release(); csort.release(); cs.release(); start_ref.release(); tex.release(); work.release();
        counter.release(); aos.release();
        owned.release(); blockcnt.release(); stage.release(); msg_state.release(); slab_counters.release();
        hist.release(); strip.release();
        if (ev_packed) (void)hipEventDestroy(ev_packed);
        if (ev_exch) (void)hipEventDestroy(ev_exch);
        if (ev_fork2) (void)hipEventDestroy(ev_fork2);
        if (comm) (void)hipStreamDestroy(comm);
        comm = nullptr; ev_packed = ev_exch = ev_fork2 = nullptr;
        for (auto& e : ev) (void)hipEventDestroy(e);
        ev.clear();
        sortp.release();
        if (t0) (void)hipEventDestroy(t0);
        if (t1) (void)hipEventDestroy(t1);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (side) (void)hipStreamDestroy(side);
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr; side = nullptr; ev_fork = ev_join = nullptr;
    }
};

namespace {

void host_lattice(const fs_settings& s, fs_vec2 off, fs_particle* dst, size_t n) {
    // src/simulation.rs:147-163 — f32 arithmetic exactly as written.
    const uint32_t count = s.particle_count;
    const float per_row = std::sqrt((float)count);
    const float per_col = ((float)count - 1.0f) / per_row + 1.0f;
    const size_t per_row_trunc = (size_t)per_row;
    for (uint32_t i = 0; i < count && (size_t)i < n; ++i) {
        const size_t col = (size_t)i % per_row_trunc;
        fs_particle q;
        std::memset(&q, 0, sizeof q);
        q.position.x = ((float)col - per_row * 0.5f + 0.5f) * s.particle_spacing + off.x;
        q.position.y = (std::floor((float)i / per_row) - per_col * 0.5f + 0.5f) * s.particle_spacing + off.y;
        q.predicted_position = q.position;
        dst[i] = q;
    }
}

void host_uniform(const fs_settings& s, const fs_tick_settings& t, uint32_t tick, fs_uniform* u) {
    // src/simulation.rs:470-497
    const float h = s.smoothing_radius;
    std::memset(u, 0, sizeof *u);
    u->delta = t.delta;
    u->particle_count = s.particle_count;
    u->sqr_radius = h * h;
    u->frame_time = tick;
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
    const float h8 = powi_f32(h, 8);
    u->poly6_kernel_volume = 4.0f / (PI_F * h8);
    u->poly6_kernel_derivative = 24.0f / (PI_F * h8);
    u->poly6_kernel_laplacian = 8.0f / (PI_F * h8);
    u->spiky_kernel_derivative = 12.0f / (powi_f32(h, 4) * PI_F);
    u->viscosity_kernel = 15.0f / (2.0f * PI_F * powi_f32(h, 3));
    u->mouse_state = t.mouse_state;
    u->mouse_force_radius = t.mouse_force_radius;
    u->mouse_force_power = t.mouse_force_power;
    grid_dims(s, &u->grid_w, &u->grid_h);
    u->texture_size.x = (float)s.texture_size.x;
    u->texture_size.y = (float)s.texture_size.y;
}

fsd::StepParams make_params(const fs_sim& s) {
    const fs_uniform& u = s.uniform;
    fsd::StepParams P;
    std::memset(&P, 0, sizeof P);
    P.n = s.n;
    P.grid_w = s.grid_w; P.grid_h = s.grid_h; P.ncell = s.ncell;
    P.dt = u.delta;
    P.h = u.smoothing_radius;
    P.sqr_radius = u.sqr_radius;
    P.bounds_x = u.bounds.x; P.bounds_y = u.bounds.y;
    P.bs_x = u.bounds.x * 0.5f; P.bs_y = u.bounds.y * 0.5f;
    P.mass = u.particle_mass;
    // funcs.wgsl:76 — `4.0 / (PI * pow(h, 8.0))`: loop-invariant, evaluated once per tick on
    // the host with libm powf (the device pow is not correctly rounded).
    P.poly6_norm = 4.0f / (PI_F * std::pow(u.smoothing_radius, 8.0f));
    P.pressure_k = u.pressure_constant; P.rest_density = u.rest_density;
    P.damping = u.damping_factor; P.visc_coeff = u.viscosity_coefficient;
    P.spiky = u.spiky_kernel_derivative; P.visc_k = u.viscosity_kernel;
    P.gx = u.gravity.x; P.gy = u.gravity.y;
    P.mouse_x = u.mouse_pos.x; P.mouse_y = u.mouse_pos.y;
    P.mouse_radius = u.mouse_force_radius; P.mouse_power = u.mouse_force_power;
    P.mouse_state = u.mouse_state;
    P.frame_time = u.frame_time;
    P.tex_w = u.texture_size.x; P.tex_h = u.texture_size.y;
    P.tex_w_u = s.settings.texture_size.x;   // u32(u.texture_size.x), compute.wgsl:129
    P.tex_len = (uint32_t)s.tex.n;
    P.tex_zero = s.tex_zero ? 1 : 0;
    {   // xcd_block(): chunks of ~1/64 of the strips, at most 256 blocks (8 grid rows of the 16M scene), at least 1
        static const int forced = getenv("FS_XCD_CHUNK_LOG2") ? atoi(getenv("FS_XCD_CHUNK_LOG2")) : -1;
        const uint32_t nb = (s.capacity + 255u) / 256u;
        uint32_t c = 0;
        while (c < 8u && (64u << (c + 1u)) <= nb) ++c;
        P.xcd_chunk_log2 = forced >= 0 ? (uint32_t)forced : c;
    }
    P.ref_quirks = s.opts.ref_quirks;
    P.fast_math = s.opts.math_mode == FS_MATH_WGSL_ULP ? 1 : s.opts.math_mode == FS_MATH_TOLERANCE ? 2 : 0;
    P.div_2h3 = s.div_2h3;
    P.div_h2 = s.div_h2;
    P.div_h = s.div_h;
    // div_by_rcp's guards assume dst <= ~h <= 2^19 (fs_device.h); FS_NO_SHAREDIV=1 keeps every `/` a true division
    static const bool no_sharediv = getenv("FS_NO_SHAREDIV") != nullptr;
    // ... and the per-particle "safe operand" classification bounds the pressure numerators only if h * spiky <= 2^19
    const float hspiky = std::fabs(P.h * P.spiky);
    P.share_div = (!no_sharediv && s.div_2h3.ok && s.div_h2.ok && s.rcp_ok && s.sqrt_ok && P.h >= 0x1p-19f && P.h <= 0x1p19f &&
                   hspiky <= FS_HSPIKY_HI) ? 1 : 0;
    P.col_origin = 0;
    P.own_lo = 0; P.own_hi = s.grid_w;
    P.grid_w_global = s.grid_w;
    P.n_live = nullptr;
    if (s.slab) {
        // local window: 1 padding + 2 ghost columns each side of [own_lo, own_hi)
        P.grid_w = s.slab_cfg.own_hi - s.slab_cfg.own_lo + 6u;
        P.ncell = P.grid_w * s.grid_h;
        P.col_origin = (int32_t)s.slab_cfg.own_lo - 3;
        P.own_lo = s.slab_cfg.own_lo; P.own_hi = s.slab_cfg.own_hi;
        P.n = s.capacity;
        P.n_live = s.slab_counters.p;
        P.ref_quirks = 0;   // the global stale-start quirk cannot exist per rank (§8e)
        // the serial step advances every owned column in its one force launch; the overlapped step narrows this per launch
        P.adv_lo = P.own_lo; P.adv_hi = P.own_hi; P.adv_outside = 0;
        P.transposed = s.transposed ? 1 : 0;
    }
    P.grid_u = P.transposed ? P.grid_h : P.grid_w;
    P.grid_v = P.transposed ? P.grid_w : P.grid_h;
    static const bool no_bb = getenv("FS_NO_BLOCK_BOUNDS") != nullptr;      // A/B: every pass reduces its block bounds itself
    P.block_bounds = no_bb ? nullptr : s.bbounds.p;
    return P;
}

// Prove (exhaustively, on the device) that x / c == div_const_fast(x, c, RN(1/c)) for every f32 x.
// Stored keys, cell starts and the column origin belong to the window of the LAST step (or import); a
// window set since then (fs_slab_set_window) only takes effect at the next pack.  Anything that reads
// the stored state back in global coordinates must use this.
fsd::StepParams make_params_of_state(fs_sim& s) {
    const fs_slab_config cfg = s.slab_cfg;
    if (s.slab && s.state_hi > s.state_lo) { s.slab_cfg.own_lo = s.state_lo; s.slab_cfg.own_hi = s.state_hi; }
    const fsd::StepParams P = make_params(s);
    s.slab_cfg = cfg;
    return P;
}

fs_status prove_constdiv(hipStream_t st, uint32_t* scratch_word, float c, fsd::ConstDiv* out, float hi = 0.0f) {
    out->c = c;
    out->y = 1.0f / c;
    out->ok = 0;
    if (!(c > 4.0f * FS_CONSTDIV_MIN) || !std::isfinite(c) || !std::isfinite(out->y) || getenv("FS_NO_CONSTDIV")) return FS_OK;
    if (!(hi > 0.0f)) hi = c;                          // the force kernel's numerators: 2^-60 <= |x| <= c
    if (!(hi < 0x1p40f) || !(hi >= FS_CONSTDIV_MIN)) return FS_OK;
    uint32_t bad = 1;
    FS_HIP(hipMemsetAsync(scratch_word, 0, sizeof(uint32_t), st));
    fsd::launch_verify_constdiv(st, c, out->y, FS_CONSTDIV_MIN, hi, scratch_word);
    FS_HIP(hipMemcpyAsync(&bad, scratch_word, sizeof bad, hipMemcpyDeviceToHost, st));
    FS_HIP(hipStreamSynchronize(st));
    out->ok = bad == 0 ? 1 : 0;
    return FS_OK;
}

fs_status prove_force_constants(fs_sim* s) {
    const float h = s->settings.smoothing_radius;
    fs_status r = prove_constdiv(s->stream, s->counter.p + 1, 2.0f * h * h * h, &s->div_2h3);   // funcs.wgsl:119
    if (r != FS_OK) return r;
    r = prove_constdiv(s->stream, s->counter.p + 1, h * h, &s->div_h2);
    if (r != FS_OK) return r;
    {   // cell coordinates: (clamped position + half the bounds) / h, numerators 0 .. 2 bs (host_uniform: bs = size / 2 - ...)
        const float reach = 4.0f * fmaxf(fabsf(s->settings.size.x), fabsf(s->settings.size.y));
        r = prove_constdiv(s->stream, s->counter.p + 1, h, &s->div_h, reach);
        if (r != FS_OK) return r;
    }
    // the lean reciprocal / square root of the shared-denominator path, over their whole ranges
    s->rcp_ok = s->sqrt_ok = false;
    if (getenv("FS_NO_SHAREDIV")) return FS_OK;
    uint32_t bad[2] = {1, 1};
    FS_HIP(hipMemsetAsync(s->counter.p + 1, 0, 2 * sizeof(uint32_t), s->stream));
    fsd::launch_verify_unary(s->stream, 0, FS_RCP_LO, FS_RCP_HI, s->counter.p + 1);
    fsd::launch_verify_unary(s->stream, 1, FS_SQRT_LO, FS_SQRT_HI, s->counter.p + 2);
    FS_HIP(hipMemcpyAsync(bad, s->counter.p + 1, sizeof bad, hipMemcpyDeviceToHost, s->stream));
    FS_HIP(hipStreamSynchronize(s->stream));
    s->rcp_ok = bad[0] == 0;
    s->sqrt_ok = bad[1] == 0;
    return FS_OK;
}

fs_status ensure_events(fs_sim* s) {
    if (s->ev.empty()) {
        s->ev.resize((size_t)fs_sim::PROF_RING * (FS_PASS_COUNT + 1));
        for (auto& e : s->ev) FS_HIP(hipEventCreate(&e));
    }
    return FS_OK;
}

fs_status drain_profile(fs_sim* s) {
    if (s->prof_pending == 0) return FS_OK;
    const size_t stride = FS_PASS_COUNT + 1;
    FS_HIP(hipEventSynchronize(s->ev[(size_t)(s->prof_pending - 1) * stride + FS_PASS_COUNT]));
    for (uint32_t j = 0; j < s->prof_pending; ++j) {
        for (int k = 0; k < FS_PASS_COUNT; ++k) {
            float ms = 0.0f;
            FS_HIP(hipEventElapsedTime(&ms, s->ev[j * stride + k], s->ev[j * stride + k + 1]));
            s->prof_ms[k] += ms;
        }
    }
    s->prof_steps += s->prof_pending;
    s->prof_pending = 0;
    return FS_OK;
}

fs_status enqueue_step(fs_sim* s, const fs_tick_settings* t) {
    s->tick += 1;                                              // src/simulation.rs:460
    host_uniform(s->settings, *t, s->tick, &s->uniform);
    s->uniform.particle_count = s->n;
    fsd::StepParams P = make_params(*s);
    // reference-sort path: no sorted copy of the positions — the force pass takes its own particle's position from the
    // previous state through the pair's source index and writes the new state into the spare buffer (swapped below)
    static const bool pos_by_src_env = [] { const char* e = getenv("FS_POS_BY_SRC"); return e ? atoi(e) != 0 : true; }();
    const bool pos_by_src = pos_by_src_env && s->opts.sort_mode != FS_SORT_COUNTING;
    P.pos_by_src = pos_by_src ? 1 : 0;
    hipStream_t st = s->stream;
    const bool prof = s->profile;
    hipEvent_t* ev = nullptr;
    if (prof) {
        fs_status r = ensure_events(s);
        if (r != FS_OK) return r;
        if (s->prof_pending == fs_sim::PROF_RING) { r = drain_profile(s); if (r != FS_OK) return r; }
        ev = &s->ev[(size_t)s->prof_pending * (FS_PASS_COUNT + 1)];
    }
    if (s->n == 0) return FS_OK;
    FS_HIP(s->sortp.throttle());                       // at most SortPolicy::FLIGHT steps ahead of the device

    if (prof) FS_HIP(hipEventRecord(ev[0], st));
    // predict + key are fused into the first sort kernel of either mode (no separate launch)
    const bool counting = s->opts.sort_mode == FS_SORT_COUNTING;
    if (prof) FS_HIP(hipEventRecord(ev[1], st));
    if (counting) {
        fsd::launch_counting_sort(st, P, s->pos.p, s->vel.p, s->cs.p, s->csort.p, s->counter.p, s->safe.p, s->tick);
    } else {
        fsd::SortPlan plan;
        if (!s->sortp.plan(s->n, &plan)) return fail(FS_ERR_DEVICE, "sort: the stand-by kernel's grid barrier timed out");
        fsd::launch_bitonic_sort(st, s->pairs.p, s->n, s->sort_dirty.p, &P, s->pos.p, s->vel.p, s->counter.p, &plan);
    }
    if (prof) FS_HIP(hipEventRecord(ev[2], st));
    s->key_in_pairs = true;        // the 4 B / particle of a second copy of the keys stay unwritten
    if (counting)      // rank fix-up of the counting sort fused with the reorder pass (kernels_csort.hip)
        fsd::launch_counting_reorder(st, P, s->csort.p, s->pairs.p, s->cs.p, s->pos.p, s->vel.p, s->pos_s.p, s->vel_s.p, s->pred.p,
                                     (uint32_t*)nullptr, s->start_ref.p, s->safe.p, s->fdefer.p, s->counter.p + 4);
    else
        fsd::launch_reorder(st, P, s->pairs.p, s->pos.p, s->vel.p, pos_by_src ? (float2*)nullptr : s->pos_s.p, s->vel_s.p, s->pred.p,
                            (uint32_t*)nullptr, s->cs.p, s->start_ref.p, s->work.p, s->counter.p, s->work_cap, s->safe.p, s->fdefer.p,
                            s->counter.p + 4);
    if (prof) FS_HIP(hipEventRecord(ev[3], st));
    // strict / ulp modes: rho2.x IS the density; the separate 4-byte copy is only written in tolerance mode (rho2 = {P, 1/rho})
    s->rho_in_rho2 = P.fast_math != 2;
    fsd::launch_density(st, P, s->pred.p, s->cs.p, s->start_ref.p, s->pairs.p, s->safe.p, s->rho_in_rho2 ? (float*)nullptr : s->rho.p, s->rho2.p, s->fdefer.p, s->fwork.p, s->counter.p + 4);
    if (prof) FS_HIP(hipEventRecord(ev[4], st));
    fsd::launch_force(st, P, pos_by_src ? s->pos.p : s->pos_s.p, s->vel_s.p, s->pred.p, s->rho2.p, s->cs.p, s->start_ref.p, s->pairs.p,
                      s->tex.p, pos_by_src ? s->pos_s.p : s->pos.p, s->vel.p, s->rho.p, s->fdefer.p, s->fwork.p, s->counter.p + 4,
                      s->aos_live ? (void*)s->aos.p : nullptr, s->side, s->ev_fork, s->ev_join,
                      s->sortp.general_grid(), s->sortp.general_hint(), 0u, prof ? nullptr : s->sortp.flight_event(), s->sortp.quad_entries());
    if (pos_by_src) { float2* t = s->pos.p; s->pos.p = s->pos_s.p; s->pos_s.p = t; }   // the spare buffer now holds the state
    if (s->aos_live) s->aos_tick = s->tick;
    if (prof) {
        FS_HIP(hipEventRecord(ev[5], st));
        FS_HIP(hipEventRecord(ev[6], st));             // FS_PASS_BOUNDARY: slab handles only
        s->prof_pending += 1;
    }
    if (prof) FS_HIP(s->sortp.step_enqueued(st));      // (the profile's own events are markers anyway)
    else s->sortp.step_bound();                        // the general force launch carried the step's completion event
    FS_HIP(hipGetLastError());
    return FS_OK;
}

// workgroups of the edge columns' launches in an edge-first step with column-major ids (each walks the device-side block ranges)
#define FS_EDGE_GRID 1024u

// ---- overlapped slab step --------------------------------------------------------------------------
// Step parameters of the two force launches of an overlapped step: the interior launch (main array) and the strip launch.
fsd::StepParams overlap_params(const fs_sim& s, bool strip) {
    fsd::StepParams P = make_params(s);
    P.adv_lo = s.adv_lo; P.adv_hi = s.adv_hi;
    P.adv_outside = strip ? 1 : 0;
    if (strip) { P.n = s.strip.cap; P.n_live = s.strip.counters.p; if (P.block_bounds) P.block_bounds = s.strip.bbounds.p; }
    return P;
}

// Interior columns and strip windows of the step being packed.  Local columns: 0 and W-1 are padding, 1..2 and W-3..W-2 the
// ghost columns, own_lo is local column 3 (make_params).
void plan_overlap(fs_sim* s) {
    const fs_slab_config& c = s->slab_cfg;
    const uint32_t z = s->boundary_cols + s->pending_shift;
    s->pending_shift = 0;
    const uint32_t width = c.own_hi - c.own_lo, W = width + 6u;
    const uint32_t zl = c.has_left ? z : 0u, zr = c.has_right ? z : 0u;
    s->strip_active = c.has_left || c.has_right;
    s->strip_win[0] = s->strip_win[1] = s->strip_win[2] = s->strip_win[3] = 0u;
    if (zl + zr >= width) {                        // no interior: the strip holds the whole window
        s->adv_lo = s->adv_hi = c.own_lo;
        if (s->strip_active) { s->strip_win[0] = 1u; s->strip_win[1] = W - 1u; }
        return;
    }
    s->adv_lo = c.own_lo + zl; s->adv_hi = c.own_hi - zr;
    // a window = the two ghost columns + the boundary columns + two columns of interior context
    const uint32_t lhi = 3u + zl + 2u, rlo = (W - 3u) - zr - 2u;
    if (c.has_left && c.has_right && lhi >= rlo) { s->strip_win[0] = 1u; s->strip_win[1] = W - 1u; return; }
    if (c.has_left) { s->strip_win[0] = 1u; s->strip_win[1] = lhi < W - 1u ? lhi : W - 1u; }
    if (c.has_right) { s->strip_win[2] = rlo > 1u ? rlo : 1u; s->strip_win[3] = W - 1u; }
}

// Second half of fs_slab_pack: everything that does not need the incoming messages.
fs_status slab_interior(fs_sim* s) {
    hipStream_t st = s->stream;
    FS_HIP(hipEventRecord(s->ev_packed, st));      // the outgoing messages are complete: the exchange may start
    plan_overlap(s);
    const fsd::StepParams P = overlap_params(*s, false);
    hipEvent_t* ev = s->slab_prof ? &s->ev[(size_t)s->prof_pending * (FS_PASS_COUNT + 1)] : nullptr;
    if (ev) FS_HIP(hipEventRecord(ev[1], st));
    fsd::launch_counting_sort_pairs(st, s->capacity, P.ncell, s->ncell, s->cs.p, s->csort.p, s->slab_counters.p, s->tick, nullptr, s->safe.p);
    if (ev) FS_HIP(hipEventRecord(ev[2], st));
    fsd::launch_counting_reorder_slab(st, P, s->capacity, s->ncell, s->csort.p, s->pairs.p, s->cs.p, s->pos.p, s->vel.p, s->pos_s.p,
                                      s->vel_s.p, s->pred.p, s->key.p, s->owned.p, s->start_ref.p, s->safe.p, s->fdefer.p,
                                      s->counter.p + 4);
    if (s->strip_active) {                          // the main array's share of the strips: also independent of the messages
        fs_sim::Strip& T = s->strip;
        fsd::launch_strip_gather(st, P, s->strip_win, s->slab_cfg.recv_capacity, T.cap, s->cs.p, T.rowbase.p, s->pairs.p, s->pos_s.p,
                                 s->vel_s.p, T.pos.p, T.vel.p, fsd::counting_sort_kt(T.csort.p, T.cap, s->ncell),
                                 fsd::counting_sort_hist(T.csort.p), T.back.p, T.safe.p, T.counters.p, s->slab_counters.p);
    }
    if (ev) FS_HIP(hipEventRecord(ev[3], st));
    fsd::launch_density(st, P, s->pred.p, s->cs.p, s->start_ref.p, s->pairs.p, s->safe.p, s->rho.p, s->rho2.p, s->fdefer.p, s->fwork.p, s->counter.p + 4);
    if (ev) FS_HIP(hipEventRecord(ev[4], st));
    fsd::launch_force(st, P, s->pos_s.p, s->vel_s.p, s->pred.p, s->rho2.p, s->cs.p, s->start_ref.p, s->pairs.p,
                      s->tex.p, s->pos.p, s->vel.p, s->rho.p, s->fdefer.p, s->fwork.p, s->counter.p + 4, nullptr, s->side,
                      s->ev_fork, s->ev_join, s->sortp.general_grid(), s->sortp.general_hint());
    if (ev) FS_HIP(hipEventRecord(ev[5], st));
    FS_HIP(hipGetLastError());
    return FS_OK;
}

// fs_slab_step of an overlapped handle: the boundary strips, after the incoming messages.
fs_status slab_boundary(fs_sim* s, const void* recv_left, const void* recv_right) {
    hipStream_t st = s->stream;
    if (s->exch_pending) { FS_HIP(hipStreamWaitEvent(st, s->ev_exch, 0)); s->exch_pending = false; }
    hipEvent_t* ev = s->slab_prof ? &s->ev[(size_t)s->prof_pending * (FS_PASS_COUNT + 1)] : nullptr;
    if (s->strip_active) {
        fs_sim::Strip& T = s->strip;
        const fsd::StepParams P = overlap_params(*s, false), PS = overlap_params(*s, true);
        fsd::u64* kt = fsd::counting_sort_kt(T.csort.p, T.cap, s->ncell);
        fsd::launch_strip_unpack(st, P, s->slab_main, s->slab_cfg.recv_capacity, T.cap, s->slab_cfg.has_left ? recv_left : nullptr,
                                 s->slab_cfg.has_right ? recv_right : nullptr, T.pos.p, T.vel.p, kt, fsd::counting_sort_hist(T.csort.p),
                                 T.back.p, T.counters.p, s->slab_counters.p);
        fsd::launch_counting_sort_pairs(st, T.cap, PS.ncell, s->ncell, T.cs.p, T.csort.p, T.counters.p, s->tick, T.counters.p + 2);
        fsd::launch_counting_reorder_slab(st, PS, T.cap, s->ncell, T.csort.p, T.pairs.p, T.cs.p, T.pos.p, T.vel.p, T.pos_s.p, T.vel_s.p,
                                          T.pred.p, (uint32_t*)nullptr, T.owned.p, T.start_ref.p, T.safe.p, T.fdefer.p, T.counter.p + 4,
                                          T.counters.p + 2);
        fsd::launch_density(st, PS, T.pred.p, T.cs.p, T.start_ref.p, T.pairs.p, T.safe.p, T.rho.p, T.rho2.p, T.fdefer.p, T.fwork.p, T.counter.p + 4);
        fsd::launch_force(st, PS, T.pos_s.p, T.vel_s.p, T.pred.p, T.rho2.p, T.cs.p, T.start_ref.p, T.pairs.p, s->tex.p, T.pos_out.p,
                          T.vel_out.p, T.rho.p, T.fdefer.p, T.fwork.p, T.counter.p + 4, nullptr, nullptr, nullptr, nullptr, 256u, nullptr);
        fsd::launch_strip_writeback(st, PS, s->slab_main, T.cap, T.pairs.p, T.back.p, T.pos_out.p, T.vel_out.p, T.pred.p, T.rho.p,
                                    s->pos.p, s->vel.p, s->pred.p, s->rho.p, s->key.p, s->owned.p, s->slab_counters.p);
    }
    if (ev) { FS_HIP(hipEventRecord(ev[6], st)); s->prof_pending += 1; }
    FS_HIP(hipGetLastError());
    s->slab_packed = false;
    return FS_OK;
}

}  // namespace

// Edge-first slab step: the edge columns' chain of the last step (exchange stream) has advanced its particles and classified their
// slots; the simulation's stream only waits for it when something other than the next pack -> exchange -> step cycle needs that
// state (the cycle itself is ordered by the exchange's event).  Every entry point that reads or replaces the state calls this.
static fs_status slab_join(fs_sim* s) {
    if (s && s->slab && s->join_pending) {
        FS_HIP(hipStreamWaitEvent(s->stream, s->ev_packed, 0));
        s->join_pending = false;
    }
    return FS_OK;
}
#define FS_JOIN(s) do { fs_status jr_ = slab_join(s); if (jr_ != FS_OK) return jr_; } while (0)

extern "C" {

int fs_abi_version(void) { return FS_ABI_VERSION; }
const char* fs_last_error(void) { return g_err.c_str(); }

void fs_options_default(fs_options* o) {
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->device = 0;
    o->sort_mode = FS_SORT_BITONIC;
    o->ref_quirks = 1;
}

fs_status fs_create(const fs_settings* settings, int device, fs_sim** out) {
    fs_options o;
    fs_options_default(&o);
    o.device = device;
    return fs_create_ex(settings, &o, out);
}

fs_status fs_create_ex(const fs_settings* settings, const fs_options* opts, fs_sim** out) {
    if (!settings || !opts || !out) return fail(FS_ERR_INVALID, "null argument");
    *out = nullptr;
    std::string why;
    if (!settings_valid(*settings, &why)) return fail(FS_ERR_INVALID, why);
    if (opts->sort_mode != FS_SORT_BITONIC && opts->sort_mode != FS_SORT_COUNTING)
        return fail(FS_ERR_INVALID, "unknown sort_mode");
    if (opts->math_mode != FS_MATH_IEEE && opts->math_mode != FS_MATH_WGSL_ULP && opts->math_mode != FS_MATH_TOLERANCE)
        return fail(FS_ERR_INVALID, "unknown math_mode");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FS_ERR_DEVICE, "no HIP device: the engine has no CPU fallback");
    if (opts->device < 0 || opts->device >= ndev) return fail(FS_ERR_INVALID, "device ordinal out of range");
    FS_HIP(hipSetDevice(opts->device));

    fs_sim* s = new (std::nothrow) fs_sim();
    if (!s) return fail(FS_ERR_OOM, "host allocation failed");
    s->settings = *settings;
    s->opts = *opts;
    s->device = opts->device;
    s->n = settings->particle_count;
    s->capacity = opts->capacity > s->n ? opts->capacity : s->n;
    if (s->capacity > (1u << 28)) { delete s; return fail(FS_ERR_INVALID, "capacity > 2^28"); }
    grid_dims(*settings, &s->grid_w, &s->grid_h);
    s->ncell = s->grid_w * s->grid_h;
    s->work_cap = s->ncell / 16u + 1024u;

    auto bail = [&](fs_status st) { s->release(); delete s; return st; };
#define FS_TRY(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e__ = (expr);                                                                              \
        if (e__ != hipSuccess)                                                                                \
            return bail(fail(e__ == hipErrorOutOfMemory ? FS_ERR_OOM : FS_ERR_DEVICE,                         \
                             std::string(#expr) + ": " + hipGetErrorString(e__)));                            \
    } while (0)

    FS_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    // The pre-registered general work can run beside the lean force kernel on a second stream (FS_SIDE_STREAM=1).
    // Measured at 16M: force 0.725 -> 0.711 ms in the bench window and 1.10 -> 0.99 ms in the dense regime, but every
    // launch of the following sort then pays ~1 us more behind the cross-stream join (sort 0.70 -> 0.73 ms): a net
    // loss in the bench window, so it is off by default.
    if (getenv("FS_SIDE_STREAM")) {
        FS_TRY(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
        FS_TRY(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        FS_TRY(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
    }
    const size_t cap = s->capacity;
    FS_TRY(s->pos.alloc(cap)); FS_TRY(s->vel.alloc(cap)); FS_TRY(s->pos_s.alloc(cap)); FS_TRY(s->vel_s.alloc(cap));
    FS_TRY(s->pred.alloc(cap + FS_PRED_SLACK)); FS_TRY(s->rho.alloc(cap)); FS_TRY(s->rho2.alloc(cap)); FS_TRY(s->key.alloc(cap)); FS_TRY(s->safe.alloc((cap + 63) / 64 + 1)); FS_TRY(s->fdefer.alloc(2 * ((cap + 255) / 256 + 8))); FS_TRY(s->fwork.alloc(2 * (cap / 256 + 8) + 16)); FS_TRY(s->bbounds.alloc(8 * ((cap + 255) / 256 + 8))); FS_TRY(s->pairs.alloc(cap));
    FS_TRY(s->sort_dirty.alloc(fsd::sort_tile_count((uint32_t)cap)));
    FS_TRY(hipMemsetAsync(s->sort_dirty.p, 0, s->sort_dirty.n * sizeof(uint32_t), s->stream));
    FS_TRY(s->cs.alloc((size_t)s->ncell + 1));
    FS_TRY(s->start_ref.alloc(s->ncell));
    FS_TRY(s->tex.alloc((size_t)settings->texture_size.x * settings->texture_size.y));
    FS_TRY(s->work.alloc((size_t)s->work_cap * fsd::gap_entry_size()));
    FS_TRY(s->counter.alloc(8));
    if (opts->sort_mode == FS_SORT_COUNTING) {
        FS_TRY(s->csort.alloc(fsd::counting_sort_scratch_words((uint32_t)cap, s->ncell)));
        FS_TRY(hipMemsetAsync(s->csort.p, 0, s->csort.n * sizeof(uint32_t), s->stream));   // histogram / tickets: zero between steps
    }
    FS_TRY(hipEventCreate(&s->t0));
    FS_TRY(hipEventCreate(&s->t1));
    FS_TRY(s->sortp.init(8));
    // wgpu zero-initialises buffers: start_indices (simulation.rs:204-209), force field (:213-218)
    FS_TRY(hipMemsetAsync(s->start_ref.p, 0, s->start_ref.n * sizeof(uint32_t), s->stream));
    FS_TRY(hipMemsetAsync(s->cs.p, 0, s->cs.n * sizeof(uint32_t), s->stream));
    if (s->tex.n) FS_TRY(hipMemsetAsync(s->tex.p, 0, s->tex.n * sizeof(float2), s->stream));
    FS_TRY(hipMemsetAsync(s->counter.p, 0, 8 * sizeof(uint32_t), s->stream));
    FS_TRY(hipMemsetAsync(s->rho.p, 0, cap * sizeof(float), s->stream));
    FS_TRY(hipMemsetAsync(s->key.p, 0, cap * sizeof(uint32_t), s->stream));

    // initial lattice (simulation.rs:147-163) -> AoS staging -> SoA
    {
        std::vector<fs_particle> host(s->n);
        host_lattice(*settings, opts->initial_offset, host.data(), host.size());
        FS_TRY(s->aos.alloc(cap));
        FS_TRY(hipMemcpyAsync(s->aos.p, host.data(), host.size() * sizeof(fs_particle), hipMemcpyHostToDevice, s->stream));
        fsd::launch_import_aos(s->stream, s->n, s->aos.p, s->pos.p, s->pred.p, s->vel.p, s->rho.p, s->key.p);
        FS_TRY(hipStreamSynchronize(s->stream));
    }
#undef FS_TRY
    { fs_status r = prove_force_constants(s); if (r != FS_OK) { s->release(); delete s; return r; } }
    fs_tick_settings t0;
    std::memset(&t0, 0, sizeof t0);
    host_uniform(*settings, t0, 0, &s->uniform);
    *out = s;
    return FS_OK;
}

void fs_destroy(fs_sim* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->comm) (void)hipStreamSynchronize(s->comm);        // an exchange / edge chain still in flight reads this handle's buffers
    s->release();
    delete s;
}

fs_status fs_step(fs_sim* s, const fs_tick_settings* t) {
    if (!s || !t) return fail(FS_ERR_INVALID, "null argument");
    if (s->slab) return fail(FS_ERR_INVALID, "slab handle: use fs_slab_pack / fs_slab_step");
    FS_HIP(hipSetDevice(s->device));
    return enqueue_step(s, t);
}

// After a synchronisation of the stream: a barrier time-out of the sort's stand-by kernel makes the state undefined from
// that step on — whoever is about to receive state (or a "finished" signal) must hear about it.
static fs_status sort_health(fs_sim* s) {
    if (s->slab || s->opts.sort_mode != FS_SORT_BITONIC) return FS_OK;
    FS_HIP(s->sortp.check_timeout(s->sort_dirty.p, s->n));
    if (s->sortp.dead) return fail(FS_ERR_DEVICE, "sort: the stand-by kernel's grid barrier timed out: the particle order is undefined from that step on; destroy the handle");
    return FS_OK;
}

fs_status fs_sync(fs_sim* s) {
    if (!s) return fail(FS_ERR_INVALID, "null argument");
    FS_HIP(hipStreamSynchronize(s->stream));
    // an exchange issued but not yet consumed by fs_slab_step / the edge columns' chain of the last edge-first step
    if (s->comm && (s->exch_pending || s->join_pending)) { FS_HIP(hipStreamSynchronize(s->comm)); s->join_pending = false; }
    return sort_health(s);
}

uint32_t fs_tick_count(const fs_sim* s) { return s ? s->tick : 0; }
uint32_t fs_particle_count(const fs_sim* s) { return s ? s->n : 0; }
void* fs_stream(const fs_sim* s) { return s ? (void*)s->stream : nullptr; }

fs_status fs_grid_dims(const fs_sim* s, uint32_t* gw, uint32_t* gh) {
    if (!s || !gw || !gh) return fail(FS_ERR_INVALID, "null argument");
    *gw = s->grid_w; *gh = s->grid_h;
    return FS_OK;
}

fs_status fs_particles_device(fs_sim* s, const fs_particle** out) {
    if (!s || !out) return fail(FS_ERR_INVALID, "null argument");
    FS_JOIN(s);
    FS_HIP(hipSetDevice(s->device));
    if (!s->aos.p) FS_HIP(s->aos.alloc(s->capacity));
    if (!(s->aos_live && s->aos_tick == s->tick && s->tick != 0)) {   // live view: the force pass already wrote it
        fsd::launch_export_aos(s->stream, s->n, s->pos.p, s->pred.p, s->vel.p, s->rho.p, s->key.p, s->aos.p,
                               s->key_in_pairs ? s->pairs.p : nullptr, s->rho_in_rho2 ? s->rho2.p : nullptr);
        FS_HIP(hipGetLastError());
        if (s->aos_live) s->aos_tick = s->tick;
    }
    *out = s->aos.p;
    return FS_OK;
}

fs_status fs_start_indices_device(fs_sim* s, const uint32_t** out, size_t* count) {
    if (!s || !out) return fail(FS_ERR_INVALID, "null argument");
    FS_JOIN(s);
    *out = s->start_ref.p;
    if (count) *count = s->start_ref.n;
    return FS_OK;
}

fs_status fs_get_uniform(const fs_sim* s, fs_uniform* out) {
    if (!s || !out) return fail(FS_ERR_INVALID, "null argument");
    *out = s->uniform;
    return FS_OK;
}

fs_status fs_upload_force_field(fs_sim* s, const fs_vec2* field, uint32_t w, uint32_t h) {
    if (!s || !field) return fail(FS_ERR_INVALID, "null argument");
    FS_JOIN(s);
    if (w != s->settings.texture_size.x || h != s->settings.texture_size.y)
        return fail(FS_ERR_INVALID, "force field dimensions differ from settings.texture_size");
    FS_HIP(hipSetDevice(s->device));
    FS_HIP(hipMemcpyAsync(s->tex.p, field, (size_t)w * h * sizeof(fs_vec2), hipMemcpyHostToDevice, s->stream));
    bool zero = true;               // while the copy runs: does the field push anything at all? (`!= 0`: -0 is zero, NaN is not)
    for (size_t k = 0, m = (size_t)w * h; k < m && zero; ++k) zero = !(field[k].x != 0.0f) && !(field[k].y != 0.0f);
    FS_HIP(hipStreamSynchronize(s->stream));
    s->tex_zero = zero;
    return FS_OK;
}

fs_status fs_download_particles(fs_sim* s, fs_particle* dst, size_t n) {
    if (!s || (!dst && n)) return fail(FS_ERR_INVALID, "null argument");
    FS_JOIN(s);
    if (n > s->n) n = s->n;
    const fs_particle* dev = nullptr;
    fs_status r = fs_particles_device(s, &dev);
    if (r != FS_OK) return r;
    if (n) FS_HIP(hipMemcpyAsync(dst, dev, n * sizeof(fs_particle), hipMemcpyDeviceToHost, s->stream));
    FS_HIP(hipStreamSynchronize(s->stream));
    return sort_health(s);        // the records are in `dst` either way (diagnosis); FS_ERR_DEVICE says they are not to be trusted
}

fs_status fs_upload_particles(fs_sim* s, const fs_particle* src, size_t n) {
    if (!s || (!src && n)) return fail(FS_ERR_INVALID, "null argument");
    FS_JOIN(s);
    if (n > s->n) n = s->n;   // ResizableBuffer::write trims oversize data (src/buffer.rs:71-75)
    FS_HIP(hipSetDevice(s->device));
    if (!s->aos.p) FS_HIP(s->aos.alloc(s->capacity));
    if (n) FS_HIP(hipMemcpyAsync(s->aos.p, src, n * sizeof(fs_particle), hipMemcpyHostToDevice, s->stream));
    if (s->key_in_pairs || s->rho_in_rho2) {   // a partial upload keeps the other particles' keys / densities: bring them home first
        fsd::launch_keys_from_pairs(s->stream, s->n, s->key_in_pairs ? s->pairs.p : nullptr, s->key.p,
                                    s->rho_in_rho2 ? s->rho2.p : nullptr, s->rho.p);
        s->key_in_pairs = false; s->rho_in_rho2 = false;
    }
    fsd::launch_import_aos(s->stream, (uint32_t)n, s->aos.p, s->pos.p, s->pred.p, s->vel.p, s->rho.p, s->key.p);
    FS_HIP(hipStreamSynchronize(s->stream));
    s->aos_tick = 0xFFFFFFFFu;      // the live view (if any) no longer matches the state: re-materialise on demand
    s->sortp.touched();             // an arbitrary order: the per-stage launches stand by until reports pass again
    return FS_OK;
}

fs_status fs_download_start_indices(fs_sim* s, uint32_t* dst, size_t n) {
    if (!s || (!dst && n)) return fail(FS_ERR_INVALID, "null argument");
    FS_JOIN(s);
    if (n > s->start_ref.n) n = s->start_ref.n;
    FS_HIP(hipSetDevice(s->device));
    if (n) FS_HIP(hipMemcpyAsync(dst, s->start_ref.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    FS_HIP(hipStreamSynchronize(s->stream));
    return FS_OK;
}

fs_status fs_upload_start_indices(fs_sim* s, const uint32_t* src, size_t n) {
    if (!s || (!src && n)) return fail(FS_ERR_INVALID, "null argument");
    FS_JOIN(s);
    if (n > s->start_ref.n) n = s->start_ref.n;
    FS_HIP(hipSetDevice(s->device));
    if (n) FS_HIP(hipMemcpyAsync(s->start_ref.p, src, n * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    FS_HIP(hipStreamSynchronize(s->stream));
    return FS_OK;
}

fs_status fs_reference_lattice(const fs_settings* settings, fs_vec2 offset, fs_particle* dst, size_t n) {
    if (!settings || (!dst && n)) return fail(FS_ERR_INVALID, "null argument");
    if (settings->particle_count == 0) return FS_OK;
    host_lattice(*settings, offset, dst, n);
    return FS_OK;
}

size_t fs_sort_schedule(uint32_t particle_count, fs_sort_step* dst, size_t cap) {
    // src/simulation.rs:323-347
    if (particle_count <= 1) return 0;
    uint32_t p2 = 1, stages = 0;
    while (p2 < particle_count) { p2 <<= 1; ++stages; }
    size_t k = 0;
    for (uint32_t stage = 0; stage < stages; ++stage)
        for (uint32_t step = 0; step <= stage; ++step, ++k)
            if (dst && k < cap) {
                const uint32_t gw = 1u << (stage - step);
                dst[k] = fs_sort_step{gw, 2 * gw - 1, step, particle_count};
            }
    return k;
}

fs_status fs_build_uniform(const fs_settings* settings, const fs_tick_settings* tick, uint32_t tick_count,
                           fs_uniform* out) {
    if (!settings || !tick || !out) return fail(FS_ERR_INVALID, "null argument");
    host_uniform(*settings, *tick, tick_count, out);
    return FS_OK;
}

fs_status fs_generate_force_field(fs_sim* s, int device, const uint8_t* image, uint32_t w, uint32_t h,
                                  fs_vec2* field_host) {
    if (!image || w == 0 || h == 0) return fail(FS_ERR_INVALID, "null/empty image");
    if (h > 1024 || w >= 65536) return fail(FS_ERR_UNSUPPORTED, "image larger than 65535 x 1024");
    if (s) {
        if (s->slab) return fail(FS_ERR_UNSUPPORTED, "force field on a slab handle");
        if (w != s->settings.texture_size.x || h != s->settings.texture_size.y)
            return fail(FS_ERR_INVALID, "image dimensions differ from settings.texture_size");
        device = s->device;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
        return fail(FS_ERR_DEVICE, "no HIP device: the engine has no CPU fallback");
    FS_HIP(hipSetDevice(device));
    const size_t npix = (size_t)w * h;
    unsigned char* dimg = nullptr; float* ddist = nullptr; uint32_t* dnear = nullptr; float2* dfield = nullptr;
    hipStream_t st = s ? s->stream : nullptr;
    hipError_t e = hipMalloc((void**)&dimg, npix);
    if (e == hipSuccess) e = hipMalloc((void**)&ddist, npix * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&dnear, npix * sizeof(uint32_t));
    if (e == hipSuccess && !s) e = hipMalloc((void**)&dfield, npix * sizeof(float2));
    float2* out = s ? s->tex.p : dfield;
    if (s) s->tex_zero = false;     // produced on the device: contents unknown to the host
    if (e == hipSuccess) e = hipMemcpyAsync(dimg, image, npix, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) { fsd::launch_gradient_field(st, dimg, w, h, ddist, dnear, out); e = hipGetLastError(); }
    if (e == hipSuccess && field_host) e = hipMemcpyAsync(field_host, out, npix * sizeof(float2), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dimg); (void)hipFree(ddist); (void)hipFree(dnear); (void)hipFree(dfield);
    if (e != hipSuccess) return fail(FS_ERR_DEVICE, hipGetErrorString(e));
    return FS_OK;
}

fs_status fs_render_density(fs_sim* s, const fs_view* view, float* rgba_host) {
    if (!s || !view || !rgba_host) return fail(FS_ERR_INVALID, "null argument");
    if (s->slab) return fail(FS_ERR_UNSUPPORTED, "render on a slab handle");
    FS_JOIN(s);
    if (view->width == 0 || view->height == 0 || (uint64_t)view->width * view->height > (1ull << 28))
        return fail(FS_ERR_INVALID, "bad image size");
    if (s->tick == 0) return fail(FS_ERR_INVALID, "render needs at least one fs_step (cell table not built yet)");
    FS_HIP(hipSetDevice(s->device));
    const size_t npix = (size_t)view->width * view->height;
    float4* dimg = nullptr;
    FS_HIP(hipMalloc((void**)&dimg, npix * sizeof(float4)));
    const fsd::StepParams P = make_params(*s);
    // after a step: `pred` = predicted positions of this step, `vel` = updated velocities (what the
    // reference's fragment shader sees in in_particles at draw time)
    fsd::launch_render_density(s->stream, P, make_float2(view->world_min.x, view->world_min.y),
                               make_float2(view->world_max.x, view->world_max.y), view->width, view->height, s->pred.p,
                               s->vel.p, s->cs.p, s->start_ref.p, s->pairs.p, dimg);
    hipError_t e = hipMemcpyAsync(rgba_host, dimg, npix * sizeof(float4), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    (void)hipFree(dimg);
    if (e != hipSuccess) return fail(FS_ERR_DEVICE, hipGetErrorString(e));
    return FS_OK;
}

fs_status fs_profile_enable(fs_sim* s, int enable) {
    if (!s) return fail(FS_ERR_INVALID, "null argument");
    s->profile = enable != 0;
    return FS_OK;
}

fs_status fs_profile_read(fs_sim* s, double ms[FS_PASS_COUNT], uint64_t* steps, int reset) {
    if (!s || !ms) return fail(FS_ERR_INVALID, "null argument");
    FS_JOIN(s);
    fs_status r = drain_profile(s);
    if (r != FS_OK) return r;
    for (int k = 0; k < FS_PASS_COUNT; ++k) ms[k] = s->prof_ms[k];
    if (steps) *steps = s->prof_steps;
    if (reset) { for (auto& m : s->prof_ms) m = 0.0; s->prof_steps = 0; }
    return FS_OK;
}

fs_status fs_timed_steps(fs_sim* s, const fs_tick_settings* t, uint32_t steps, double* ms_total) {
    if (!s || !t || !ms_total) return fail(FS_ERR_INVALID, "null argument");
    if (s->slab) return fail(FS_ERR_INVALID, "slab handle: use fs_slab_pack / fs_slab_step");
    FS_HIP(hipSetDevice(s->device));
    FS_HIP(hipEventRecord(s->t0, s->stream));
    for (uint32_t k = 0; k < steps; ++k) {
        fs_status r = enqueue_step(s, t);
        if (r != FS_OK) return r;
    }
    FS_HIP(hipEventRecord(s->t1, s->stream));
    FS_HIP(hipEventSynchronize(s->t1));
    float ms = 0.0f;
    FS_HIP(hipEventElapsedTime(&ms, s->t0, s->t1));
    *ms_total = ms;
    return sort_health(s);
}

/* Exhaustive proof used by the force pass: number of f32 x with lo <= |x| <= hi for which the 3-op
 * constant division (x*y, fma, fma with the given reciprocal y) differs from the IEEE x / c.  Blocking. */
fs_status fs_selftest_constdiv(int device, float c, float y, float lo, float hi, uint32_t* mismatches) {
    if (!mismatches || !(lo > 0.0f) || !(hi >= lo) || !std::isfinite(hi)) return fail(FS_ERR_INVALID, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(FS_ERR_DEVICE, "no HIP device");
    FS_HIP(hipSetDevice(device));
    uint32_t* dm = nullptr;
    FS_HIP(hipMalloc((void**)&dm, sizeof(uint32_t)));
    FS_HIP(hipMemset(dm, 0, sizeof(uint32_t)));
    fsd::launch_verify_constdiv(nullptr, c, y, lo, hi, dm);
    hipError_t e = hipMemcpy(mismatches, dm, sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipFree(dm);
    if (e != hipSuccess) return fail(FS_ERR_DEVICE, hipGetErrorString(e));
    return FS_OK;
}

/* The sort on caller-supplied pairs (tests of the late-stage plans on adversarial inputs).  Blocking. */
fs_status fs_selftest_sort(int device, uint64_t* pairs, uint32_t n, int fuse_stage, uint32_t plan[2]) {
    if (!pairs || n == 0 || n > (1u << 28)) return fail(FS_ERR_INVALID, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(FS_ERR_DEVICE, "no HIP device");
    FS_HIP(hipSetDevice(device));
    unsigned long long* dp = nullptr;
    uint32_t* dd = nullptr;
    const size_t words = fsd::sort_tile_count(n);
    FS_HIP(hipMalloc((void**)&dp, (size_t)n * 8));
    hipError_t e = hipMalloc((void**)&dd, words * 4);
    if (e == hipSuccess) e = hipMemset(dd, 0, words * 4);
    if (e == hipSuccess) e = hipMemcpy(dp, pairs, (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        fsd::SortPlan sp;
        sp.fuse_stage = fuse_stage < 0 ? -1 : (fuse_stage & 0xFF);
        sp.fallback = fuse_stage >= 0 && (fuse_stage & 0x100) ? 1 : 0;
        fsd::launch_bitonic_sort(nullptr, dp, n, dd, nullptr, nullptr, nullptr, nullptr, &sp);
        e = hipMemcpy(pairs, dp, (size_t)n * 8, hipMemcpyDeviceToHost);
    }
    if (e == hipSuccess && plan) e = hipMemcpy(plan, dd + fsd::sort_plan_word(n) + 1, 8, hipMemcpyDeviceToHost);
    uint32_t timeouts = 0;
    if (e == hipSuccess) e = hipMemcpy(&timeouts, dd + fsd::sort_plan_word(n) + 4, 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && timeouts) { (void)hipFree(dp); (void)hipFree(dd); return fail(FS_ERR_DEVICE, "sort fallback: grid barrier timed out"); }
    (void)hipFree(dp);
    (void)hipFree(dd);
    if (e != hipSuccess) return fail(FS_ERR_DEVICE, hipGetErrorString(e));
    return FS_OK;
}

/* The plan policy replayed against a model of the flow (include/fluidsim.h); host only. */
fs_status fs_selftest_sort_policy(uint32_t S, int start_back, uint32_t lag, const uint32_t* required, size_t steps,
                                  uint32_t* stage_out, uint32_t* single_out) {
    if (!required || !stage_out || !single_out || S < 15 || S > 28) return fail(FS_ERR_INVALID, "bad argument");
    fsd::SortPolicy p;
    p.start_back = start_back;
    std::vector<uint32_t> used(steps);
    for (size_t i = 0; i < steps; ++i) {
        if (i >= lag && i - lag < steps) {             // the report of step i - lag arrives before step i is planned
            const size_t j = i - lag;
            const int st = (int)used[j];
            const bool passed = st >= (int)required[j];
            const int cls = passed ? (st - (int)required[j] > 3 ? 3 : st - (int)required[j]) : 0;
            p.observe((uint32_t)j + 1u, st, passed, cls, S);
        }
        used[i] = (uint32_t)(p.stage ? p.stage : p.first_stage(S));
        p.seq = (uint32_t)i + 1u;                      // what plan() does: this step's sequence number
        stage_out[i] = used[i];
        single_out[i] = p.single_standby() ? 1u : 0u;
    }
    return FS_OK;
}

fs_status fs_sort_plan_read(fs_sim* s, fs_sort_plan_info* out) {
    if (!s || !out) return fail(FS_ERR_INVALID, "null argument");
    FS_HIP(hipSetDevice(s->device));
    FS_HIP(hipStreamSynchronize(s->stream));
    uint32_t w[8] = {};
    const uint32_t count = s->slab ? s->capacity : s->n;
    if (count > 1) FS_HIP(hipMemcpy(w, s->sort_dirty.p + fsd::sort_plan_word(count), sizeof w, hipMemcpyDeviceToHost));
    out->shifted = w[1]; out->per_stage = w[2]; out->standby_runs = w[6]; out->timeouts = w[4]; out->wide_tiles = w[7];
    out->stage = (uint32_t)s->sortp.stage;
    out->standby_single = (s->sortp.force_single || (s->sortp.stage && s->sortp.trusted >= 2)) ? 1u : 0u;
    return FS_OK;
}

/* Did the create-time proofs succeed for this handle's constants (2h^3, h^2)?  Bits 0 / 1. */
int fs_constdiv_status(const fs_sim* s) {
    return s ? (s->div_2h3.ok ? 1 : 0) | (s->div_h2.ok ? 2 : 0) | (s->rcp_ok ? 4 : 0) | (s->sqrt_ok ? 8 : 0) | (s->div_h.ok ? 16 : 0) : 0;
}

/* ------------------------------------------------- renderer hand-off without a host round trip */
fs_status fs_export_handle(fs_sim* s, int which, fs_mem_handle* out) {
    if (!s || !out) return fail(FS_ERR_INVALID, "null argument");
    if (s->slab) return fail(FS_ERR_UNSUPPORTED, "export on a slab handle");
    if (which != FS_EXPORT_PARTICLES && which != FS_EXPORT_START_INDICES) return fail(FS_ERR_INVALID, "unknown export");
    FS_JOIN(s);
    FS_HIP(hipSetDevice(s->device));
    std::memset(out, 0, sizeof *out);
    void* base = nullptr;
    if (which == FS_EXPORT_PARTICLES) {
        if (!s->aos.p) FS_HIP(s->aos.alloc(s->capacity));
        if (!s->aos_live) {
            s->aos_live = true;                    // from now on k_force writes the records itself
            s->aos_tick = 0xFFFFFFFFu;
        }
        const fs_particle* dev = nullptr;          // make the view current for the state as it is now
        fs_status r = fs_particles_device(s, &dev);
        if (r != FS_OK) return r;
        base = s->aos.p;
        out->bytes = (uint64_t)s->n * sizeof(fs_particle);
    } else {
        base = s->start_ref.p;
        out->bytes = (uint64_t)s->start_ref.n * sizeof(uint32_t);
    }
    static_assert(sizeof(hipIpcMemHandle_t) <= sizeof(out->ipc), "fs_mem_handle.ipc too small");
    hipIpcMemHandle_t h;
    FS_HIP(hipIpcGetMemHandle(&h, base));
    std::memcpy(out->ipc, &h, sizeof h);
    out->device = s->device;
    out->dmabuf_fd = -1;
    // a dma-buf file descriptor of the same range, for consumers outside HIP (Vulkan / wgpu external memory);
    // optional: older runtimes lack the call, the IPC handle above is the portable path between HIP processes
    int fd = -1;
    if (hipMemGetHandleForAddressRange(&fd, base, (size_t)((out->bytes + 4095u) & ~(uint64_t)4095u), hipMemRangeHandleTypeDmaBufFd, 0) == hipSuccess)
        out->dmabuf_fd = fd;
    else
        (void)hipGetLastError();
    FS_HIP(hipStreamSynchronize(s->stream));
    return FS_OK;
}

fs_status fs_import_open(const fs_mem_handle* h, int device, void** ptr) {
    if (!h || !ptr) return fail(FS_ERR_INVALID, "null argument");
    *ptr = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(FS_ERR_DEVICE, "no HIP device");
    FS_HIP(hipSetDevice(device));
    hipIpcMemHandle_t ih;
    std::memcpy(&ih, h->ipc, sizeof ih);
    FS_HIP(hipIpcOpenMemHandle(ptr, ih, hipIpcMemLazyEnablePeerAccess));
    return FS_OK;
}

fs_status fs_import_read(const void* dev_ptr, size_t offset, void* dst, size_t bytes) {
    if (!dev_ptr || (!dst && bytes)) return fail(FS_ERR_INVALID, "null argument");
    FS_HIP(hipMemcpy(dst, (const char*)dev_ptr + offset, bytes, hipMemcpyDeviceToHost));
    return FS_OK;
}

fs_status fs_import_close(void* ptr) {
    if (!ptr) return FS_OK;
    FS_HIP(hipIpcCloseMemHandle(ptr));
    return FS_OK;
}

/* ------------------------------------------------------------ slab mode */
fs_status fs_slab_create(const fs_settings* settings, int device, const fs_slab_config* cfg, fs_sim** out) {
    if (!settings || !cfg || !out) return fail(FS_ERR_INVALID, "null argument");
    *out = nullptr;
    std::string why;
    if (!settings_valid(*settings, &why)) return fail(FS_ERR_INVALID, why);
    uint32_t gw, gh;
    grid_dims(*settings, &gw, &gh);
    if (cfg->own_lo >= cfg->own_hi || cfg->own_hi > gw) return fail(FS_ERR_INVALID, "bad owned window");
    if (cfg->own_hi - cfg->own_lo < 4) return fail(FS_ERR_INVALID, "slab narrower than 4 columns");
    if (cfg->max_cols < cfg->own_hi - cfg->own_lo) return fail(FS_ERR_INVALID, "max_cols < window");
    if (cfg->capacity <= 2 * cfg->recv_capacity || cfg->recv_capacity == 0)
        return fail(FS_ERR_INVALID, "capacity must exceed 2*recv_capacity");
    if (cfg->capacity > (1u << 28)) return fail(FS_ERR_INVALID, "capacity > 2^28");
    if (cfg->recv_capacity >= (1u << 20) - 2u) return fail(FS_ERR_INVALID, "recv_capacity >= 2^20 - 2 (message counters are 20-bit fields)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(FS_ERR_DEVICE, "no HIP device: the engine has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(FS_ERR_INVALID, "device ordinal out of range");
    FS_HIP(hipSetDevice(device));

    fs_sim* s = new (std::nothrow) fs_sim();
    if (!s) return fail(FS_ERR_OOM, "host allocation failed");
    s->settings = *settings;
    fs_options_default(&s->opts);
    s->opts.device = device;
    s->opts.ref_quirks = 0;
    // per-rank sorts can only be tolerance-parity with a single-domain run, so slabs
    // default to the O(N) counting sort; cfg->sort_mode = 1 + FS_SORT_BITONIC selects the network
    s->opts.sort_mode = (cfg->sort_mode & 0xFFu) == 1 + FS_SORT_BITONIC ? FS_SORT_BITONIC : FS_SORT_COUNTING;
    {   // the overlapped step needs the counting sort (ghosts out of the main array); FS_SLAB_SERIAL / FS_SLAB_OVERLAP=0: the serial step
        // FS_SLAB_MODE=serial|edge|strips overrides the configuration (A/B runs)
        const char* e = getenv("FS_SLAB_MODE");
        uint32_t m = (cfg->sort_mode & FS_SLAB_SERIAL) ? 0u : (cfg->sort_mode & FS_SLAB_STRIPS) ? 2u : 1u;
        if (e) m = !strcmp(e, "serial") ? 0u : !strcmp(e, "strips") ? 2u : !strcmp(e, "edge") ? 1u : m;
        if (s->opts.sort_mode != FS_SORT_COUNTING) m = 0u;     // the network's slab mode stays the serial step (bit-identity with the plain engine)
        s->overlap = m == 2u;
        s->edge_first = m == 1u;
        // column-major cell ids wherever a slab edge has a neighbour (the edge columns are then whole blocks at the two ends of the
        // sorted array); a slab without neighbours keeps the reference layout and stays bit-identical to the plain engine in
        // FS_SORT_COUNTING mode.  FS_SLAB_TRANSPOSE=0/1 overrides (A/B runs); the strip step's gather is written for rows.
        const char* te = getenv("FS_SLAB_TRANSPOSE");
        s->transposed = s->opts.sort_mode == FS_SORT_COUNTING && !s->overlap && (cfg->has_left || cfg->has_right) && !(cfg->sort_mode & FS_SLAB_ROWMAJOR);
        if (te && s->opts.sort_mode == FS_SORT_COUNTING && !s->overlap) s->transposed = atoi(te) != 0;
    }
    s->device = device;
    s->slab = true;
    s->slab_cfg = *cfg;
    s->capacity = cfg->capacity;
    s->n = 0;
    s->slab_main = cfg->capacity - 2 * cfg->recv_capacity;
    s->grid_w = gw; s->grid_h = gh;
    const uint32_t wmax = cfg->max_cols + 6u;
    s->ncell = wmax * gh;                       // allocation size of the local grid
    s->work_cap = s->ncell / 16u + 1024u;
    auto bail = [&](fs_status st) { s->release(); delete s; return st; };
#define FS_TRY(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e__ = (expr);                                                                              \
        if (e__ != hipSuccess)                                                                                \
            return bail(fail(e__ == hipErrorOutOfMemory ? FS_ERR_OOM : FS_ERR_DEVICE,                         \
                             std::string(#expr) + ": " + hipGetErrorString(e__)));                            \
    } while (0)
    FS_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    // The pre-registered general work can run beside the lean force kernel on a second stream (FS_SIDE_STREAM=1).
    // Measured at 16M: force 0.725 -> 0.711 ms in the bench window and 1.10 -> 0.99 ms in the dense regime, but every
    // launch of the following sort then pays ~1 us more behind the cross-stream join (sort 0.70 -> 0.73 ms): a net
    // loss in the bench window, so it is off by default.
    if (getenv("FS_SIDE_STREAM")) {
        FS_TRY(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
        FS_TRY(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        FS_TRY(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
    }
    FS_TRY(s->sortp.init(8));   // slab handles use the pinned words for the force pass's work report only (sort: per-stage plan)
    const size_t cap = s->capacity;
    FS_TRY(s->pos.alloc(cap)); FS_TRY(s->vel.alloc(cap)); FS_TRY(s->pos_s.alloc(cap)); FS_TRY(s->vel_s.alloc(cap));
    FS_TRY(s->pred.alloc(cap + FS_PRED_SLACK)); FS_TRY(s->rho.alloc(cap)); FS_TRY(s->rho2.alloc(cap)); FS_TRY(s->key.alloc(cap)); FS_TRY(s->safe.alloc((cap + 63) / 64 + 1)); FS_TRY(s->fdefer.alloc(2 * ((cap + 255) / 256 + 8))); FS_TRY(s->fwork.alloc(2 * (cap / 256 + 8) + 16)); FS_TRY(s->bbounds.alloc(8 * ((cap + 255) / 256 + 8))); FS_TRY(s->pairs.alloc(cap));
    FS_TRY(s->sort_dirty.alloc(fsd::sort_tile_count((uint32_t)cap)));
    FS_TRY(hipMemsetAsync(s->sort_dirty.p, 0, s->sort_dirty.n * sizeof(uint32_t), s->stream));
    FS_TRY(s->owned.alloc(cap));
    const size_t nblocks = (cap + 255) / 256;
    FS_TRY(s->blockcnt.alloc(2 * (nblocks + 1)));      // per 256-slot block: message counts, then message offsets (k_slab_msg)
    FS_TRY(s->stage.alloc(fsd::slab_stage_words((uint32_t)cap)));
    FS_TRY(s->msg_state.alloc(fsd::slab_msg_groups((uint32_t)cap) + 1));
    FS_TRY(hipMemsetAsync(s->msg_state.p, 0, s->msg_state.n * sizeof(fsd::u64), s->stream));
    FS_TRY(s->slab_counters.alloc(16));
    FS_TRY(s->hist.alloc(gw));
    FS_TRY(s->csort.alloc(fsd::counting_sort_scratch_words((uint32_t)cap, s->ncell)));
    FS_TRY(hipMemsetAsync(s->csort.p, 0, s->csort.n * sizeof(uint32_t), s->stream));       // histogram / tickets: zero between steps
    FS_TRY(s->cs.alloc((size_t)s->ncell + 1)); FS_TRY(s->start_ref.alloc(s->ncell));
    FS_TRY(s->tex.alloc((size_t)settings->texture_size.x * settings->texture_size.y));
    FS_TRY(s->work.alloc((size_t)s->work_cap * fsd::gap_entry_size()));
    FS_TRY(s->counter.alloc(8));
    FS_TRY(s->aos.alloc(cap));
    FS_TRY(hipEventCreate(&s->t0)); FS_TRY(hipEventCreate(&s->t1));
    if (s->overlap || s->edge_first) {
        int lo_prio = 0, hi_prio = 0;          // the exchange's kernel should not queue behind the interior columns' workgroups
        (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
        FS_TRY(hipStreamCreateWithPriority(&s->comm, hipStreamNonBlocking, hi_prio));
        FS_TRY(hipEventCreateWithFlags(&s->ev_packed, hipEventDisableTiming));
        FS_TRY(hipEventCreateWithFlags(&s->ev_exch, hipEventDisableTiming));
        // waited for by the exchange stream of this same device only: no system-scope fence (a write-back of every L2 behind the
        // reorder kernel, which the simulation's own stream would sit out)
        FS_TRY(hipEventCreateWithFlags(&s->ev_fork2, hipEventDisableTiming | hipEventDisableSystemFence));
        if (const char* e = getenv("FS_SLAB_BOUNDARY_COLS")) s->boundary_cols = (uint32_t)atoi(e) < 3u ? 3u : (uint32_t)atoi(e);
    }
    if (s->overlap) {
        // The strip could hold every particle of a narrow slab (all columns within the boundary zone) plus both messages:
        // same capacity as the main array (memory is not the constraint: ~100 B per slot); its kernels cover the slots in use only.
        fs_sim::Strip& T = s->strip;
        T.cap = (uint32_t)cap;
        FS_TRY(T.pos.alloc(cap)); FS_TRY(T.vel.alloc(cap)); FS_TRY(T.pos_s.alloc(cap)); FS_TRY(T.vel_s.alloc(cap));
        FS_TRY(T.pred.alloc(cap + FS_PRED_SLACK)); FS_TRY(T.rho2.alloc(cap)); FS_TRY(T.pos_out.alloc(cap)); FS_TRY(T.vel_out.alloc(cap));
        FS_TRY(T.rho.alloc(cap)); FS_TRY(T.pairs.alloc(cap)); FS_TRY(T.safe.alloc((cap + 63) / 64 + 1)); FS_TRY(T.owned.alloc(cap));
        FS_TRY(T.fdefer.alloc(2 * ((cap + 255) / 256 + 8))); FS_TRY(T.fwork.alloc(2 * (cap / 256 + 8) + 16));
        FS_TRY(T.bbounds.alloc(8 * ((cap + 255) / 256 + 8)));
        FS_TRY(T.csort.alloc(fsd::counting_sort_scratch_words((uint32_t)cap, s->ncell)));
        FS_TRY(hipMemsetAsync(T.csort.p, 0, T.csort.n * sizeof(uint32_t), s->stream));
        FS_TRY(T.cs.alloc((size_t)s->ncell + 1)); FS_TRY(T.start_ref.alloc(s->ncell));
        FS_TRY(T.counter.alloc(8)); FS_TRY(T.counters.alloc(8)); FS_TRY(T.back.alloc(cap)); FS_TRY(T.rowbase.alloc(2 * (size_t)gh + 2));
        FS_TRY(hipMemsetAsync(T.cs.p, 0, T.cs.n * sizeof(uint32_t), s->stream));
        FS_TRY(hipMemsetAsync(T.counter.p, 0, 8 * sizeof(uint32_t), s->stream));
        FS_TRY(hipMemsetAsync(T.counters.p, 0, 8 * sizeof(uint32_t), s->stream));
        FS_TRY(hipMemsetAsync(T.pred.p, 0, (cap + FS_PRED_SLACK) * sizeof(float2), s->stream));
        FS_TRY(hipMemsetAsync(T.pairs.p, 0xFF, cap * sizeof(fsd::u64), s->stream));
    }
    FS_TRY(hipMemsetAsync(s->start_ref.p, 0, s->start_ref.n * sizeof(uint32_t), s->stream));
    FS_TRY(hipMemsetAsync(s->cs.p, 0, s->cs.n * sizeof(uint32_t), s->stream));
    if (s->tex.n) FS_TRY(hipMemsetAsync(s->tex.p, 0, s->tex.n * sizeof(float2), s->stream));
    FS_TRY(hipMemsetAsync(s->counter.p, 0, 8 * sizeof(uint32_t), s->stream));
    FS_TRY(hipMemsetAsync(s->slab_counters.p, 0, 16 * sizeof(uint32_t), s->stream));
    FS_TRY(hipMemsetAsync(s->owned.p, 0, cap, s->stream));
    FS_TRY(hipMemsetAsync(s->rho.p, 0, cap * sizeof(float), s->stream));
    FS_TRY(hipMemsetAsync(s->pos.p, 0, cap * sizeof(float2), s->stream));
    FS_TRY(hipMemsetAsync(s->vel.p, 0, cap * sizeof(float2), s->stream));
    FS_TRY(hipMemsetAsync(s->pred.p, 0, cap * sizeof(float2), s->stream));
    FS_TRY(hipMemsetAsync(s->key.p, 0xFF, cap * sizeof(uint32_t), s->stream));
    FS_TRY(hipStreamSynchronize(s->stream));
#undef FS_TRY
    { fs_status r = prove_force_constants(s); if (r != FS_OK) { s->release(); delete s; return r; } }
    fs_tick_settings t0;
    std::memset(&t0, 0, sizeof t0);
    host_uniform(*settings, t0, 0, &s->uniform);
    *out = s;
    return FS_OK;
}

fs_status fs_slab_upload_owned(fs_sim* s, const fs_particle* src, size_t n) {
    if (!s || !s->slab || (!src && n)) return fail(FS_ERR_INVALID, "bad argument");
    if (n > s->slab_main) return fail(FS_ERR_INVALID, "more owned particles than main slots");
    FS_JOIN(s);
    s->prepacked = false;                  // the state is replaced: messages built from the old one are void
    FS_HIP(hipSetDevice(s->device));
    if (n) FS_HIP(hipMemcpyAsync(s->aos.p, src, n * sizeof(fs_particle), hipMemcpyHostToDevice, s->stream));
    const fsd::StepParams P = make_params(*s);
    fsd::launch_slab_import(s->stream, P, (uint32_t)n, s->capacity, s->aos.p, s->pos.p, s->pred.p, s->vel.p, s->rho.p,
                            s->key.p, s->owned.p);
    const uint32_t nl = (uint32_t)n;
    FS_HIP(hipMemcpyAsync(s->slab_counters.p, &nl, sizeof nl, hipMemcpyHostToDevice, s->stream));
    FS_HIP(hipStreamSynchronize(s->stream));
    s->state_lo = s->slab_cfg.own_lo; s->state_hi = s->slab_cfg.own_hi;
    return FS_OK;
}

fs_status fs_slab_set_window(fs_sim* s, uint32_t own_lo, uint32_t own_hi) {
    if (!s || !s->slab) return fail(FS_ERR_INVALID, "not a slab handle");
    FS_JOIN(s);
    if (own_lo >= own_hi || own_hi > s->grid_w || own_hi - own_lo < 4 || own_hi - own_lo > s->slab_cfg.max_cols)
        return fail(FS_ERR_INVALID, "bad owned window");
    if (s->slab_packed) return fail(FS_ERR_INVALID, "window change between pack and step");
    {   // the particles of a column that changes hands arrive at the new owner as migrants, that many columns deeper than usual:
        // the next (overlapped) step widens its boundary zone by the shift
        const uint32_t dl = s->slab_cfg.has_left ? (own_lo > s->slab_cfg.own_lo ? own_lo - s->slab_cfg.own_lo : s->slab_cfg.own_lo - own_lo) : 0u;
        const uint32_t dr = s->slab_cfg.has_right ? (own_hi > s->slab_cfg.own_hi ? own_hi - s->slab_cfg.own_hi : s->slab_cfg.own_hi - own_hi) : 0u;
        const uint32_t d = dl > dr ? dl : dr;
        if (d > s->pending_shift) s->pending_shift = d;
    }
    s->slab_cfg.own_lo = own_lo;
    s->slab_cfg.own_hi = own_hi;
    return FS_OK;
}

/* Overlapped step: owned columns per neighboured slab edge that are left to the boundary strips (computed AFTER the halo
 * exchange; everything farther inside runs while the messages are in flight).  A migrant must land at least 3 columns short of
 * the interior — cols >= 3 + the columns the fastest particle crosses in one step; violations are counted in far_halo. */
fs_status fs_slab_set_boundary_cols(fs_sim* s, uint32_t cols) {
    if (!s || !s->slab) return fail(FS_ERR_INVALID, "not a slab handle");
    if (s->slab_packed) return fail(FS_ERR_INVALID, "boundary change between pack and step");
    s->boundary_cols = cols < 3u ? 3u : cols;
    return FS_OK;
}
uint32_t fs_slab_boundary_cols(const fs_sim* s) { return (s && s->slab && (s->overlap || s->edge_first)) ? s->boundary_cols : 0u; }
int fs_slab_overlapped(const fs_sim* s) { return (s && s->slab) ? (s->edge_first ? 1 : s->overlap ? 2 : 0) : 0; }
void* fs_slab_comm_stream(const fs_sim* s) { return (s && s->slab) ? (void*)s->comm : nullptr; }

/* Transport hooks of the overlapped step (a no-op on a serial handle, whose exchange is ordered by the simulation's stream):
 * fs_slab_comm_begin makes the exchange stream wait for the packed messages, the caller then issues its send/recv ON
 * fs_slab_comm_stream(), fs_slab_comm_end records their completion for fs_slab_step to wait on.  fs_slab_exchange does all
 * three itself.  fs_slab_wait_packed blocks the HOST until the outgoing messages are complete (host-staged transports). */
fs_status fs_slab_comm_begin(fs_sim* s) {
    if (!s || !s->slab) return fail(FS_ERR_INVALID, "not a slab handle");
    if (!s->comm) return FS_OK;
    FS_HIP(hipSetDevice(s->device));
    if (!s->slab_packed) FS_HIP(hipEventRecord(s->ev_packed, s->stream));   // outside a step: behind whatever the simulation's stream holds
    FS_HIP(hipStreamWaitEvent(s->comm, s->ev_packed, 0));
    return FS_OK;
}
fs_status fs_slab_comm_end(fs_sim* s) {
    if (!s || !s->slab) return fail(FS_ERR_INVALID, "not a slab handle");
    if (!s->comm) return FS_OK;
    FS_HIP(hipSetDevice(s->device));
    FS_HIP(hipEventRecord(s->ev_exch, s->comm));
    s->exch_pending = true;
    return FS_OK;
}
fs_status fs_slab_wait_packed(fs_sim* s) {
    if (!s || !s->slab) return fail(FS_ERR_INVALID, "not a slab handle");
    FS_HIP(hipSetDevice(s->device));
    if (s->comm && s->slab_packed) FS_HIP(hipEventSynchronize(s->ev_packed));
    else FS_HIP(hipStreamSynchronize(s->stream));
    return FS_OK;
}

size_t fs_slab_message_bytes(const fs_sim* s) {
    return (s && s->slab) ? fsd::slab_message_bytes(s->slab_cfg.recv_capacity) : 0;
}

fs_status fs_slab_pack(fs_sim* s, const fs_tick_settings* t, void* send_left, void* send_right) {
    if (!s || !s->slab || !t) return fail(FS_ERR_INVALID, "bad argument");
    if (s->slab_packed) return fail(FS_ERR_INVALID, "fs_slab_pack called twice without fs_slab_step");
    if ((s->slab_cfg.has_left && !send_left) || (s->slab_cfg.has_right && !send_right))
        return fail(FS_ERR_INVALID, "missing outgoing message buffer");
    FS_HIP(hipSetDevice(s->device));
    s->tick += 1;
    host_uniform(s->settings, *t, s->tick, &s->uniform);
    const fsd::StepParams P = make_params(*s);
    s->slab_prof = s->profile;             // a toggle between pack and step must not leave ev[0] unrecorded
    if (s->slab_prof) {
        fs_status r = ensure_events(s);
        if (r != FS_OK) return r;
        if (s->prof_pending == fs_sim::PROF_RING) { r = drain_profile(s); if (r != FS_OK) return r; }
        FS_HIP(hipEventRecord(s->ev[(size_t)s->prof_pending * (FS_PASS_COUNT + 1)], s->stream));
    }
    const bool counting = s->opts.sort_mode == FS_SORT_COUNTING;
    // edge-first step: are the messages of this tick already in the send buffers (built by the last fs_slab_step)?  Only if
    // nothing they depend on has changed since: buffers, owned window, delta.
    const bool pre = s->edge_first && s->prepacked && s->pp_left == (s->slab_cfg.has_left ? send_left : nullptr) &&
                     s->pp_right == (s->slab_cfg.has_right ? send_right : nullptr) && s->pp_delta == t->delta &&
                     s->pp_lo == s->slab_cfg.own_lo && s->pp_hi == s->slab_cfg.own_hi;
    s->prepacked = false;
    // pre: this launch needs nothing of the edge columns' chain (their slots are classified already: skip_edge) — no join; the
    // chain is waited for through the exchange's event in fs_slab_step.  Otherwise: join, and take back the histogram counts
    // that chain added with the parameters it expected (the scan has left the table zero everywhere else)
    const bool skip_edge = pre && s->edge_classified;
    if (!pre) {
        FS_JOIN(s);
        if (s->edge_classified && counting) FS_HIP(hipMemsetAsync(fsd::counting_sort_hist(s->csort.p), 0, (size_t)s->ncell * sizeof(uint32_t), s->stream));
    }
    s->edge_classified = false;
    s->pp_left = s->slab_cfg.has_left ? send_left : nullptr;
    s->pp_right = s->slab_cfg.has_right ? send_right : nullptr;
    s->last_tick = *t;
    fsd::launch_slab_pack(s->stream, P, s->slab_main, s->slab_cfg.recv_capacity, (int)s->slab_cfg.has_left,
                          (int)s->slab_cfg.has_right, s->pos.p, s->vel.p, s->owned.p,
                          counting ? fsd::counting_sort_kt(s->csort.p, s->capacity, s->ncell) : s->pairs.p,
                          fsd::counting_sort_hist(s->csort.p), s->blockcnt.p, s->stage.p, s->msg_state.p, ++s->msg_epoch,
                          s->pp_left, s->pp_right, s->slab_counters.p, s->counter.p, s->safe.p, counting,
                          s->overlap, !pre, s->key.p, s->adv_lo, s->adv_hi, skip_edge);
    FS_HIP(hipGetLastError());
    s->slab_packed = true;
    s->state_lo = s->slab_cfg.own_lo; s->state_hi = s->slab_cfg.own_hi;
    if (s->overlap) return slab_interior(s);
    // edge-first: a pre-built message set was recorded complete (ev_packed) when it was built; a fresh one is complete now
    if (s->edge_first && !pre) FS_HIP(hipEventRecord(s->ev_packed, s->stream));
    return FS_OK;
}

fs_status fs_slab_step(fs_sim* s, const void* recv_left, const void* recv_right) {
    if (!s || !s->slab) return fail(FS_ERR_INVALID, "not a slab handle");
    if (!s->slab_packed) return fail(FS_ERR_INVALID, "fs_slab_step without fs_slab_pack");
    if ((s->slab_cfg.has_left && !recv_left) || (s->slab_cfg.has_right && !recv_right))
        return fail(FS_ERR_INVALID, "missing incoming message buffer");
    FS_HIP(hipSetDevice(s->device));
    if (s->overlap) return slab_boundary(s, recv_left, recv_right);
    const fsd::StepParams P = make_params(*s);
    hipStream_t st = s->stream;
    hipEvent_t* ev = s->slab_prof ? &s->ev[(size_t)s->prof_pending * (FS_PASS_COUNT + 1)] : nullptr;
    const bool counting = s->opts.sort_mode == FS_SORT_COUNTING;
    // the exchange was enqueued on the exchange stream behind the edge columns' chain: its event stands for the join as well
    if (s->exch_pending) { FS_HIP(hipStreamWaitEvent(st, s->ev_exch, 0)); s->exch_pending = false; s->join_pending = false; }
    else FS_JOIN(s);
    fsd::launch_slab_unpack(st, P, s->slab_main, s->slab_cfg.recv_capacity, s->slab_cfg.has_left ? recv_left : nullptr,
                            s->slab_cfg.has_right ? recv_right : nullptr, s->pos.p, s->vel.p,
                            counting ? fsd::counting_sort_kt(s->csort.p, s->capacity, s->ncell) : s->pairs.p,
                            fsd::counting_sort_hist(s->csort.p), s->slab_counters.p, counting);
    if (ev) FS_HIP(hipEventRecord(ev[1], st));
    if (counting) {
        fsd::launch_counting_sort_pairs(st, s->capacity, P.ncell, s->ncell, s->cs.p, s->csort.p, s->slab_counters.p, s->tick, nullptr, s->safe.p);
    } else {
        fsd::SortPlan per_stage;               // ghosts arrive at the end of the array every step: they travel far, no shifted merge
        per_stage.fuse_stage = 0;
        fsd::launch_bitonic_sort(st, s->pairs.p, s->capacity, s->sort_dirty.p, nullptr, nullptr, nullptr, nullptr, &per_stage);
    }
    if (ev) FS_HIP(hipEventRecord(ev[2], st));
    const bool edge_step = s->edge_first && (s->slab_cfg.has_left || s->slab_cfg.has_right);
    bool forked = false;
    if (edge_step) {
        plan_overlap(s);
        forked = s->transposed && s->adv_lo < s->adv_hi;      // the edge columns' chain forks off behind the reorder pass (below)
    }
    if (counting)
        fsd::launch_counting_reorder_slab(st, P, s->capacity, s->ncell, s->csort.p, s->pairs.p, s->cs.p, s->pos.p, s->vel.p, s->pos_s.p,
                                          s->vel_s.p, s->pred.p, s->key.p, s->owned.p, s->start_ref.p, s->safe.p, s->fdefer.p,
                                          s->counter.p + 4, nullptr, forked ? s->ev_fork2 : nullptr);
    else
        fsd::launch_slab_reorder(st, P, s->capacity, s->pairs.p, s->pos.p, s->vel.p, s->pos_s.p, s->vel_s.p, s->pred.p,
                                 s->key.p, s->owned.p, s->cs.p, s->start_ref.p, s->work.p, s->counter.p, s->work_cap,
                                 s->slab_counters.p, s->safe.p, s->fdefer.p, s->counter.p + 4);
    if (ev) FS_HIP(hipEventRecord(ev[3], st));
    if (edge_step) {
        if (forked) {
            // column-major ids: the edge columns' chain forks off BEFORE the density pass — their own density launch (the few
            // hundred blocks that hold the edge columns and one column more on either side; the full launch below computes the
            // same values again) runs on the exchange stream, so the chain is done, and the exchange under way, early in the
            // interior columns' force pass
            fsd::StepParams PD = P;
            PD.adv_lo = s->adv_lo; PD.adv_hi = s->adv_hi;
            FS_HIP(hipStreamWaitEvent(s->comm, s->ev_fork2, 0));     // signalled by the reorder kernel itself
            fsd::launch_density(s->comm, PD, s->pred.p, s->cs.p, s->start_ref.p, s->pairs.p, s->safe.p, s->rho.p, s->rho2.p, s->fdefer.p,
                                s->fwork.p, s->counter.p + 4, FS_EDGE_GRID);
        }
    }
    fsd::launch_density(st, P, s->pred.p, s->cs.p, s->start_ref.p, s->pairs.p, s->safe.p, s->rho.p, s->rho2.p, s->fdefer.p, s->fwork.p, s->counter.p + 4);
    if (ev) FS_HIP(hipEventRecord(ev[4], st));
    if (edge_step) {
        // Edge-first step.  Behind the density pass the stream forks: the handle's exchange stream (high priority) advances
        // the owned columns within boundary_cols of a neighboured edge — a few hundred blocks, latency-bound — then builds the
        // NEXT step's messages from their new state (k_slab_prepack .. k_slab_gather) and carries the exchange of those
        // messages (fs_slab_exchange / the caller's transport between fs_slab_comm_begin / _end); the simulation's stream
        // runs the force pass of the interior columns beside all that, and joins before anything reads the new state.
        fsd::StepParams PE = P, PI = P;
        PE.adv_lo = PI.adv_lo = s->adv_lo; PE.adv_hi = PI.adv_hi = s->adv_hi;
        PE.adv_outside = 1; PI.adv_outside = 0;
        // column-major ids: the edge columns are a few hundred consecutive blocks at the two ends of the sorted array, walked by
        // small fixed grids (fs_device.h EdgeBlocks)
        const uint32_t eg = s->transposed ? FS_EDGE_GRID : 0u;
        hipStream_t es = s->comm;
        if (!forked) FS_HIP(hipEventRecord(s->ev_fork2, st));
        // the simulation's stream first (its force launch is the long one: the host must not leave that stream empty while it
        // enqueues the six launches of the edge chain — seen under the profiler, where a launch costs 10 us), then the chain
        if (s->adv_lo < s->adv_hi)
            fsd::launch_force(st, PI, s->pos_s.p, s->vel_s.p, s->pred.p, s->rho2.p, s->cs.p, s->start_ref.p, s->pairs.p,
                              s->tex.p, s->pos.p, s->vel.p, s->rho.p, s->fdefer.p, s->fwork.p, s->counter.p + 4, nullptr, s->side,
                              s->ev_fork, s->ev_join, s->sortp.general_grid(), s->sortp.general_hint(), 0u, nullptr, s->sortp.quad_entries());
        if (ev) FS_HIP(hipEventRecord(ev[5], st));      // FS_PASS_FORCE: the interior launch
        if (!forked) FS_HIP(hipStreamWaitEvent(es, s->ev_fork2, 0));
        fsd::launch_force(es, PE, s->pos_s.p, s->vel_s.p, s->pred.p, s->rho2.p, s->cs.p, s->start_ref.p, s->pairs.p,
                          s->tex.p, s->pos.p, s->vel.p, s->rho.p, s->fdefer.p, s->fwork.p, s->counter.p + 4, nullptr, nullptr,
                          nullptr, nullptr, 256u, nullptr, eg);
        {   // what fs_slab_pack will see at tick + 1, if nothing changes in between (it checks)
            fs_uniform un;
            host_uniform(s->settings, s->last_tick, s->tick + 1, &un);
            const fs_uniform keep = s->uniform;
            s->uniform = un;
            fsd::StepParams PN = make_params(*s);
            s->uniform = keep;
            PN.adv_lo = s->adv_lo; PN.adv_hi = s->adv_hi; PN.adv_outside = 1;
            fsd::launch_slab_prepack(es, PN, s->capacity, s->slab_cfg.recv_capacity, (int)s->slab_cfg.has_left, (int)s->slab_cfg.has_right,
                                     s->pos.p, s->vel.p, s->owned.p, s->key.p, s->blockcnt.p, s->stage.p, s->msg_state.p, ++s->msg_epoch,
                                     s->pp_left, s->pp_right, s->slab_counters.p, s->cs.p, eg, true, counting, s->slab_main,
                                     counting ? fsd::counting_sort_kt(s->csort.p, s->capacity, s->ncell) : s->pairs.p,
                                     fsd::counting_sort_hist(s->csort.p));
            s->edge_classified = true;
            FS_HIP(hipEventRecord(s->ev_packed, es));       // the next step's messages are complete (and the edge columns advanced)
            s->prepacked = true;
            s->pp_delta = s->last_tick.delta; s->pp_lo = s->slab_cfg.own_lo; s->pp_hi = s->slab_cfg.own_hi;
        }
        // no join here: the next fs_slab_pack leaves the edge columns' slots alone, and fs_slab_step waits for the exchange that
        // follows their chain on the exchange stream; anything else that touches the state joins first (slab_join)
        s->join_pending = true;
        if (ev) { FS_HIP(hipEventRecord(ev[6], st)); s->prof_pending += 1; }     // FS_PASS_BOUNDARY: nothing left on this stream
    } else {
        fsd::launch_force(st, P, s->pos_s.p, s->vel_s.p, s->pred.p, s->rho2.p, s->cs.p, s->start_ref.p, s->pairs.p,
                          s->tex.p, s->pos.p, s->vel.p, s->rho.p, s->fdefer.p, s->fwork.p, s->counter.p + 4, nullptr, s->side,
                          s->ev_fork, s->ev_join, s->sortp.general_grid(), s->sortp.general_hint(), 0u, nullptr, s->sortp.quad_entries());
        if (ev) { FS_HIP(hipEventRecord(ev[5], st)); FS_HIP(hipEventRecord(ev[6], st)); s->prof_pending += 1; }
    }
    FS_HIP(hipGetLastError());
    s->slab_packed = false;
    return FS_OK;
}

fs_status fs_slab_counters_read(fs_sim* s, fs_slab_counters* out) {
    if (!s || !s->slab || !out) return fail(FS_ERR_INVALID, "bad argument");
    FS_JOIN(s);
    FS_HIP(hipSetDevice(s->device));
    uint32_t c[8];
    FS_HIP(hipMemcpyAsync(c, s->slab_counters.p, sizeof c, hipMemcpyDeviceToHost, s->stream));
    FS_HIP(hipStreamSynchronize(s->stream));
    out->n_live = c[0]; out->lost = c[2]; out->overflow = c[3]; out->far_halo = c[4];
    return FS_OK;
}

fs_status fs_slab_max_speed(fs_sim* s, float* out) {
    if (!s || !s->slab || !out) return fail(FS_ERR_INVALID, "bad argument");
    if (s->slab_packed) return fail(FS_ERR_INVALID, "fs_slab_max_speed between pack and step");
    FS_JOIN(s);
    FS_HIP(hipSetDevice(s->device));
    uint32_t bits = 0;
    FS_HIP(hipMemsetAsync(s->slab_counters.p + 5, 0, sizeof(uint32_t), s->stream));
    fsd::launch_slab_maxspeed(s->stream, s->slab_counters.p, s->vel.p, s->owned.p, s->slab_counters.p + 5,
                              s->slab_main, s->overlap ? 2u * s->slab_cfg.recv_capacity : 0u);
    FS_HIP(hipMemcpyAsync(&bits, s->slab_counters.p + 5, sizeof bits, hipMemcpyDeviceToHost, s->stream));
    FS_HIP(hipStreamSynchronize(s->stream));
    std::memcpy(out, &bits, sizeof bits);
    return FS_OK;
}

/* Re-balancing inputs left ON THE DEVICE, on the simulation's stream, nothing read back: `hist_dev[grid_w_global]` =
 * particles per global column (zero outside the owned window), `stats_dev[4]` = {lost, overflow, far_halo, bits of the
 * largest owned |velocity|} — all four reduce with MAX as u32 (non-negative floats order like their bits).  The caller
 * all-reduces both buffers (fs_comm_allreduce, or any collective ordered after this stream) and reads them once. */
fs_status fs_slab_rebalance_stats(fs_sim* s, uint32_t* stats_dev, uint32_t* hist_dev, size_t grid_w_global) {
    if (!s || !s->slab || !stats_dev || !hist_dev || grid_w_global < s->grid_w) return fail(FS_ERR_INVALID, "bad argument");
    if (s->slab_packed) return fail(FS_ERR_INVALID, "fs_slab_rebalance_stats between pack and step");
    FS_JOIN(s);
    FS_HIP(hipSetDevice(s->device));
    const fsd::StepParams P = make_params_of_state(*s);
    FS_HIP(hipMemsetAsync(hist_dev, 0, grid_w_global * sizeof(uint32_t), s->stream));
    const uint32_t migr = s->overlap ? 2u * s->slab_cfg.recv_capacity : 0u;     // overlapped step: last step's migrants sit past the main slots
    fsd::launch_slab_colhist(s->stream, P, s->cs.p, hist_dev, s->slab_main, migr, s->owned.p, s->key.p);
    FS_HIP(hipMemsetAsync(s->slab_counters.p + 5, 0, sizeof(uint32_t), s->stream));
    fsd::launch_slab_maxspeed(s->stream, s->slab_counters.p, s->vel.p, s->owned.p, s->slab_counters.p + 5, s->slab_main, migr);
    // counters [2] lost, [3] overflow, [4] far_halo, [5] max-speed bits are adjacent
    FS_HIP(hipMemcpyAsync(stats_dev, s->slab_counters.p + 2, 4 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s->stream));
    FS_HIP(hipGetLastError());
    return FS_OK;
}

fs_status fs_slab_download(fs_sim* s, fs_particle* dst, uint8_t* owned, size_t cap, uint32_t* n_live) {
    if (!s || !s->slab || !dst || !owned || !n_live) return fail(FS_ERR_INVALID, "bad argument");
    FS_JOIN(s);
    FS_HIP(hipSetDevice(s->device));
    const fsd::StepParams P = make_params_of_state(*s);
    uint32_t nl = 0;
    FS_HIP(hipMemcpyAsync(&nl, s->slab_counters.p, sizeof nl, hipMemcpyDeviceToHost, s->stream));
    FS_HIP(hipStreamSynchronize(s->stream));
    if (nl > s->capacity) nl = s->capacity;
    // overlapped step: the sorted prefix [0, nl) (owned + this rank's near-leavers) and, past the main slots, the 2R slots
    // that mirror the incoming messages — the migrants among them carry the owned flag; returned back to back
    const size_t migr = s->overlap ? 2u * (size_t)s->slab_cfg.recv_capacity : 0u;
    if (s->overlap && nl > s->slab_main) nl = s->slab_main;
    const size_t n = nl < cap ? nl : cap;
    const size_t m = cap - n < migr ? cap - n : migr;
    *n_live = (uint32_t)(n + m);
    if (n + m == 0) return FS_OK;
    // before the first step the state lives in pos/vel (import); afterwards pos/vel hold the advanced state
    fsd::launch_slab_export(s->stream, P, s->capacity, s->pos.p, s->pred.p, s->vel.p, s->rho.p, s->key.p, s->aos.p);
    if (n) FS_HIP(hipMemcpyAsync(dst, s->aos.p, n * sizeof(fs_particle), hipMemcpyDeviceToHost, s->stream));
    if (n) FS_HIP(hipMemcpyAsync(owned, s->owned.p, n, hipMemcpyDeviceToHost, s->stream));
    if (m) FS_HIP(hipMemcpyAsync(dst + n, s->aos.p + s->slab_main, m * sizeof(fs_particle), hipMemcpyDeviceToHost, s->stream));
    if (m) FS_HIP(hipMemcpyAsync(owned + n, s->owned.p + s->slab_main, m, hipMemcpyDeviceToHost, s->stream));
    FS_HIP(hipStreamSynchronize(s->stream));
    return FS_OK;
}

fs_status fs_slab_column_histogram(fs_sim* s, uint32_t* hist, size_t grid_w_global) {
    if (!s || !s->slab || !hist || grid_w_global < s->grid_w) return fail(FS_ERR_INVALID, "bad argument");
    FS_JOIN(s);
    FS_HIP(hipSetDevice(s->device));
    const fsd::StepParams P = make_params_of_state(*s);
    FS_HIP(hipMemsetAsync(s->hist.p, 0, s->hist.n * sizeof(uint32_t), s->stream));
    fsd::launch_slab_colhist(s->stream, P, s->cs.p, s->hist.p, s->slab_main, s->overlap ? 2u * s->slab_cfg.recv_capacity : 0u,
                             s->owned.p, s->key.p);
    std::vector<uint32_t> tmp(s->grid_w);
    FS_HIP(hipMemcpyAsync(tmp.data(), s->hist.p, tmp.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
    FS_HIP(hipStreamSynchronize(s->stream));
    for (uint32_t c = P.own_lo; c < P.own_hi; ++c) hist[c] = tmp[c];
    return FS_OK;
}

}  // extern "C"

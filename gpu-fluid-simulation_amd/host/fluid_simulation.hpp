// fluid_simulation.hpp — C++ host-side mirror of the reference's Rust interface for the hot
// path, layered on the C ABI (include/fluidsim.h).  Header-only; links libfluidsim_hip.so.
//
// The reference host code is Rust (src/simulation.rs, src/buffer.rs); this image has no Rust
// toolchain, so the host layer above the C ABI is C++ with the same names, argument meaning
// and error behaviour.  (The Rust twin is shipped as source in ../rust/.)
//
//   FluidSimulation::new_(settings)      <- FluidSimulation::new      src/simulation.rs:139
//   FluidSimulation::tick(tick_settings) <- FluidSimulation::tick     src/simulation.rs:459
//   FluidSimulation::tick_count()        <- pub tick: u32             src/simulation.rs:12
//   particles()/start_indices()/uniform()<- simulation_bg / simulation_settings_bg  :542-559
//   force_field_texture_write()          <- force_field_texture() + queue.write_buffer  :562, renderer.rs:497-502
//   ResizableBuffer<T>, SSBO<T>          <- src/buffer.rs:9-173
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fluidsim.h"

namespace fluidsim {

// The reference panics (unwrap / ilog2(0)); across a C ABI that becomes a status, and in this
// C++ mirror an exception carrying it.
struct Error : std::runtime_error {
    fs_status status;
    Error(fs_status s, const char* msg) : std::runtime_error(std::string("fluidsim: ") + msg), status(s) {}
};
inline void check(fs_status s) { if (s != FS_OK) throw Error(s, fs_last_error()); }

using SimulationSettings = fs_settings;   // src/simulation.rs:95-104
using TickSettings = fs_tick_settings;    // src/simulation.rs:107-122
using ParticleInstance = fs_particle;     // src/simulation.rs:126-135
using SimulationUniform = fs_uniform;     // src/simulation.rs:53-90

// Defaults: src/main.rs:48-54, src/renderer.rs:16.
inline SimulationSettings default_settings() {
    return SimulationSettings{100000u, 0.1f, 0.2f, fs_vec2{53.0f, 53.0f}, fs_uvec2{1024u, 1024u}};
}
// Defaults: src/renderer.rs:374-388.
inline TickSettings default_tick_settings() {
    TickSettings t{};
    t.delta = 1.0f / 120.0f; t.gravity = fs_vec2{0.0f, 0.0f}; t.mass = 1.0f; t.pressure_constant = 50.0f;
    t.rest_density = 0.0f; t.damping_factor = 0.1f; t.viscosity_coefficient = 25.0f;
    t.surface_tension_treshold = 0.1f; t.surface_tension_coefficient = 35.0f; t.mouse_force_radius = 5.0f;
    t.mouse_force_power = 150.0f; t.mouse_pos = fs_vec2{0.0f, 0.0f}; t.mouse_state = 0;
    return t;
}

class FluidSimulation {
public:
    // FluidSimulation::new(&device, settings): `device` is a HIP ordinal here.
    static FluidSimulation new_(int device, const SimulationSettings& settings) {
        FluidSimulation s;
        check(fs_create(&settings, device, &s.h_));
        return s;
    }
    static FluidSimulation with_options(const SimulationSettings& settings, const fs_options& opts) {
        FluidSimulation s;
        check(fs_create_ex(&settings, &opts, &s.h_));
        return s;
    }
    FluidSimulation(FluidSimulation&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    FluidSimulation& operator=(FluidSimulation&& o) noexcept { if (this != &o) { reset(); h_ = o.h_; o.h_ = nullptr; } return *this; }
    FluidSimulation(const FluidSimulation&) = delete;
    FluidSimulation& operator=(const FluidSimulation&) = delete;
    ~FluidSimulation() { reset(); }

    // tick(&mut self, queue, encoder, settings): enqueues on the simulation's HIP stream and
    // returns immediately (the reference only records; submit is src/main.rs:226).
    void tick(const TickSettings& t) { check(fs_step(h_, &t)); }
    void wait() { check(fs_sync(h_)); }                       // device.poll(Wait), src/main.rs:79
    uint32_t tick_count() const { return fs_tick_count(h_); }
    uint32_t particle_count() const { return fs_particle_count(h_); }

    // Device pointers for a renderer (the bind-group accessors of the reference).
    const ParticleInstance* particles() { const fs_particle* p = nullptr; check(fs_particles_device(h_, &p)); return p; }
    const uint32_t* start_indices(size_t* count = nullptr) { const uint32_t* p = nullptr; check(fs_start_indices_device(h_, &p, count)); return p; }
    SimulationUniform uniform() const { SimulationUniform u; check(fs_get_uniform(h_, &u)); return u; }
    void force_field_texture_write(const fs_vec2* field, uint32_t w, uint32_t h) { check(fs_upload_force_field(h_, field, w, h)); }

    std::vector<ParticleInstance> download() {
        std::vector<ParticleInstance> v(particle_count());
        check(fs_download_particles(h_, v.data(), v.size()));
        return v;
    }
    void upload(const std::vector<ParticleInstance>& v) { check(fs_upload_particles(h_, v.data(), v.size())); }
    fs_sim* handle() { return h_; }

private:
    FluidSimulation() = default;
    void reset() { if (h_) fs_destroy(h_); h_ = nullptr; }
    fs_sim* h_ = nullptr;
};

// ResizableBuffer<T> — src/buffer.rs:17-88.
template <class T>
class ResizableBuffer {
public:
    ResizableBuffer(const char* name, int device, size_t len) { check(fs_buffer_create(device, sizeof(T), len, name, &b_)); }
    ~ResizableBuffer() { if (b_) fs_buffer_destroy(b_); }
    ResizableBuffer(const ResizableBuffer&) = delete;
    ResizableBuffer& operator=(const ResizableBuffer&) = delete;
    bool resize(size_t new_cap) { int r = 0; check(fs_buffer_resize(b_, new_cap, &r)); return r != 0; }   // :46-67
    void write(size_t offset, const T* data, size_t count) { check(fs_buffer_write(b_, offset, data, count)); }  // :70-87
    size_t len() const { return fs_buffer_len(b_); }
    T* buffer() const { return static_cast<T*>(fs_buffer_device_ptr(b_)); }
protected:
    fs_buffer* b_ = nullptr;
};

// SSBO<T> — src/buffer.rs:9-14,91-173: a ResizableBuffer plus its binding; with HIP the
// "bind group" is just the device pointer.
template <class T>
class SSBO : public ResizableBuffer<T> {
public:
    using ResizableBuffer<T>::ResizableBuffer;
    void resize(size_t new_cap) { if (new_cap > this->len()) (void)ResizableBuffer<T>::resize(new_cap); }  // :127-150
    void update(const T* data, size_t count) { this->write(0, data, count); }                              // :153-155
    T* bind_group() const { return this->buffer(); }                                                       // :162-164
};

}  // namespace fluidsim

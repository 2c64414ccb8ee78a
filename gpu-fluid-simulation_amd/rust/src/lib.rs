//! Rust binding a maintainer of rookieCookies/gpu-fluid-simulation would add to run the
//! per-tick SPH step on an MI355X through `libfluidsim_hip.so` (C ABI: include/fluidsim.h).
//! Same type names, fields and method names as src/simulation.rs and src/buffer.rs so that
//! `renderer.rs` keeps compiling: `FluidSimulation::new(device, settings)`, `.tick(settings)`,
//! `.tick` (as `tick_count()`), `SimulationUniform`, `ResizableBuffer<T>`, `SSBO<T>`.
//! NOT BUILT in this repository's image (no Rust toolchain) — source only.  It cannot drift
//! silently: tests/test_rust_shim.py parses the `extern "C"` block below and checks every symbol,
//! its arity and its pointer / scalar shape against include/fluidsim.h and the built library.
#![allow(non_camel_case_types)]
use std::ffi::{c_char, c_int, c_void, CStr, CString};
use std::marker::PhantomData;

#[repr(C)] #[derive(Clone, Copy, Default, Debug, PartialEq)] pub struct Vec2 { pub x: f32, pub y: f32 }
#[repr(C)] #[derive(Clone, Copy, Default, Debug, PartialEq)] pub struct Vec3 { pub x: f32, pub y: f32, pub z: f32 }
#[repr(C)] #[derive(Clone, Copy, Default, Debug, PartialEq)] pub struct UVec2 { pub x: u32, pub y: u32 }

/// src/simulation.rs:95-104
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct SimulationSettings {
    pub particle_count: u32, pub particle_spacing: f32, pub smoothing_radius: f32,
    pub size: Vec2, pub texture_size: UVec2,
}
/// src/simulation.rs:107-122
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct TickSettings {
    pub delta: f32, pub gravity: Vec2, pub mass: f32, pub pressure_constant: f32, pub rest_density: f32,
    pub damping_factor: f32, pub viscosity_coefficient: f32, pub surface_tension_treshold: f32,
    pub surface_tension_coefficient: f32, pub mouse_force_radius: f32, pub mouse_force_power: f32,
    pub mouse_pos: Vec2, pub mouse_state: i32,
}
/// src/simulation.rs:126-135 (32 bytes)
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct ParticleInstance {
    pub position: Vec2, pub predicted_position: Vec2, pub velocity: Vec2, pub density: f32, pub grid: u32,
}
const _: () = assert!(std::mem::size_of::<ParticleInstance>() == 32);

/// src/simulation.rs:53-90 — the 120-byte uniform the renderer binds as `simulation_settings_bg`
/// (src/simulation.rs:552-554, src/renderer.rs:171,457; WGSL `Uniforms`, funcs.wgsl:17-51).
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct SimulationUniform {
    pub delta: f32, pub particle_count: u32, pub sqr_radius: f32, pub frame_time: u32,
    pub gravity: Vec2, pub bounds: Vec2, pub mouse_pos: Vec2,
    pub smoothing_radius: f32, pub particle_mass: f32, pub pressure_constant: f32, pub rest_density: f32,
    pub damping_factor: f32, pub viscosity_coefficient: f32,
    pub surface_tension_treshold: f32, pub surface_tension_coefficient: f32,
    pub poly6_kernel_volume: f32, pub poly6_kernel_derivative: f32, pub poly6_kernel_laplacian: f32,
    pub spiky_kernel_derivative: f32, pub viscosity_kernel: f32,
    pub mouse_state: i32, pub mouse_force_radius: f32, pub mouse_force_power: f32,
    pub grid_w: u32, pub grid_h: u32, pub texture_size: Vec2,
}
const _: () = assert!(std::mem::size_of::<SimulationUniform>() == 120);

/// SortUniform payload, src/simulation.rs:40-50 (without the 240-byte pad).
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct SortStep { pub group_width: u32, pub group_height: u32, pub step_index: u32, pub num_values: u32 }

/// fs_options (build-defined; the defaults reproduce the reference).
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct Options {
    pub device: i32, pub sort_mode: i32, pub ref_quirks: i32, pub math_mode: i32,
    pub initial_offset: Vec2, pub capacity: u32, pub reserved1: u32,
}
pub const SORT_BITONIC: i32 = 0;
pub const SORT_COUNTING: i32 = 1;
pub const MATH_IEEE: i32 = 0;
pub const MATH_WGSL_ULP: i32 = 1;
pub const MATH_TOLERANCE: i32 = 2;

/// fs_view: the renderer's orthographic full-domain camera (src/renderer.rs:558-561) made explicit.
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct View { pub world_min: Vec2, pub world_max: Vec2, pub width: u32, pub height: u32 }

/// fs_mem_handle: interprocess / external-memory handle of a device buffer of the simulation (80 bytes).
#[repr(C)] #[derive(Clone, Copy)]
pub struct MemHandle { pub ipc: [u8; 64], pub bytes: u64, pub device: i32, pub dmabuf_fd: i32 }
const _: () = assert!(std::mem::size_of::<MemHandle>() == 80);
pub const EXPORT_PARTICLES: c_int = 0;
pub const EXPORT_START_INDICES: c_int = 1;

/// fs_sort_plan_info: diagnostics of the sort's late-stage plan.
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct SortPlanInfo { pub shifted: u32, pub per_stage: u32, pub standby_runs: u32, pub stage: u32, pub standby_single: u32, pub timeouts: u32, pub wide_tiles: u32 }

/// fs_slab_config / fs_slab_counters (multi-GPU slabs; not in the reference).
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct SlabConfig { pub own_lo: u32, pub own_hi: u32, pub has_left: u32, pub has_right: u32, pub capacity: u32, pub recv_capacity: u32, pub max_cols: u32, pub sort_mode: u32 }
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct SlabCounters { pub n_live: u32, pub lost: u32, pub overflow: u32, pub far_halo: u32 }

/// 3D extension PODs (fs3_*; not in the reference).
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct Settings3 { pub particle_count: u32, pub particle_spacing: f32, pub smoothing_radius: f32, pub size: Vec3 }
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct TickSettings3 { pub delta: f32, pub gravity: Vec3, pub mass: f32, pub pressure_constant: f32, pub rest_density: f32, pub damping_factor: f32, pub viscosity_coefficient: f32 }
#[repr(C)] #[derive(Clone, Copy, Default, Debug)]
pub struct Particle3 { pub position: Vec3, pub predicted_position: Vec3, pub velocity: Vec3, pub density: f32, pub grid: u32, pub pad: u32 }
const _: () = assert!(std::mem::size_of::<Particle3>() == 48);

#[repr(C)] pub struct fs_sim { _p: [u8; 0] }
#[repr(C)] pub struct fs_sim3 { _p: [u8; 0] }
#[repr(C)] pub struct fs_buffer { _p: [u8; 0] }
#[repr(C)] pub struct fs_comm { _p: [u8; 0] }

pub const PASS_COUNT: usize = 6;   // FS_PASS_COUNT (the sixth, FS_PASS_BOUNDARY, is non-zero on slab handles only)
/// fs_slab_config.sort_mode | SLAB_SERIAL: the serial slab step instead of the overlapped one.
pub const SLAB_SERIAL: u32 = 0x100;
pub const COMM_ID_BYTES: usize = 128;

// Every entry point of include/fluidsim.h, in the header's order.
extern "C" {
    // lifecycle — FluidSimulation::new / Drop (src/simulation.rs:139, src/renderer.rs:31)
    fn fs_create(settings: *const SimulationSettings, device: c_int, out: *mut *mut fs_sim) -> c_int;
    fn fs_create_ex(settings: *const SimulationSettings, opts: *const Options, out: *mut *mut fs_sim) -> c_int;
    fn fs_options_default(opts: *mut Options);
    fn fs_destroy(sim: *mut fs_sim);
    // stepping — FluidSimulation::tick (src/simulation.rs:459-539)
    fn fs_step(sim: *mut fs_sim, tick: *const TickSettings) -> c_int;
    fn fs_sync(sim: *mut fs_sim) -> c_int;
    fn fs_tick_count(sim: *const fs_sim) -> u32;
    fn fs_particle_count(sim: *const fs_sim) -> u32;
    fn fs_grid_dims(sim: *const fs_sim, grid_w: *mut u32, grid_h: *mut u32) -> c_int;
    fn fs_stream(sim: *const fs_sim) -> *mut c_void;
    // data the renderer consumes (src/simulation.rs:542-564)
    fn fs_particles_device(sim: *mut fs_sim, out: *mut *const ParticleInstance) -> c_int;
    fn fs_start_indices_device(sim: *mut fs_sim, out: *mut *const u32, count: *mut usize) -> c_int;
    fn fs_get_uniform(sim: *const fs_sim, out: *mut SimulationUniform) -> c_int;
    fn fs_upload_force_field(sim: *mut fs_sim, field: *const Vec2, w: u32, h: u32) -> c_int;
    fn fs_export_handle(sim: *mut fs_sim, which: c_int, out: *mut MemHandle) -> c_int;
    fn fs_import_open(handle: *const MemHandle, device: c_int, dev_ptr: *mut *mut c_void) -> c_int;
    fn fs_import_read(dev_ptr: *const c_void, offset: usize, dst: *mut c_void, bytes: usize) -> c_int;
    fn fs_import_close(dev_ptr: *mut c_void) -> c_int;
    fn fs_download_particles(sim: *mut fs_sim, dst: *mut ParticleInstance, n: usize) -> c_int;
    fn fs_upload_particles(sim: *mut fs_sim, src: *const ParticleInstance, n: usize) -> c_int;
    fn fs_download_start_indices(sim: *mut fs_sim, dst: *mut u32, n: usize) -> c_int;
    fn fs_upload_start_indices(sim: *mut fs_sim, src: *const u32, n: usize) -> c_int;
    // host-side mirrors (pure)
    fn fs_reference_lattice(settings: *const SimulationSettings, offset: Vec2, dst: *mut ParticleInstance, n: usize) -> c_int;
    fn fs_sort_schedule(particle_count: u32, dst: *mut SortStep, cap: usize) -> usize;
    fn fs_build_uniform(settings: *const SimulationSettings, tick: *const TickSettings, tick_count: u32, out: *mut SimulationUniform) -> c_int;
    // obstacle field producer, headless density splat
    fn fs_generate_force_field(sim: *mut fs_sim, device: c_int, image: *const u8, w: u32, h: u32, field_host: *mut Vec2) -> c_int;
    fn fs_render_density(sim: *mut fs_sim, view: *const View, rgba_host: *mut f32) -> c_int;
    // profiling
    fn fs_profile_enable(sim: *mut fs_sim, enable: c_int) -> c_int;
    fn fs_profile_read(sim: *mut fs_sim, ms: *mut f64, steps: *mut u64, reset: c_int) -> c_int;
    fn fs_timed_steps(sim: *mut fs_sim, tick: *const TickSettings, steps: u32, ms_total: *mut f64) -> c_int;
    // multi-GPU slab mode
    fn fs_slab_create(global_settings: *const SimulationSettings, device: c_int, cfg: *const SlabConfig, out: *mut *mut fs_sim) -> c_int;
    fn fs_slab_upload_owned(sim: *mut fs_sim, src: *const ParticleInstance, n: usize) -> c_int;
    fn fs_slab_set_window(sim: *mut fs_sim, own_lo: u32, own_hi: u32) -> c_int;
    fn fs_slab_message_bytes(sim: *const fs_sim) -> usize;
    fn fs_slab_pack(sim: *mut fs_sim, tick: *const TickSettings, send_left: *mut c_void, send_right: *mut c_void) -> c_int;
    fn fs_slab_step(sim: *mut fs_sim, recv_left: *const c_void, recv_right: *const c_void) -> c_int;
    fn fs_slab_overlapped(sim: *const fs_sim) -> c_int;
    fn fs_slab_set_boundary_cols(sim: *mut fs_sim, cols: u32) -> c_int;
    fn fs_slab_boundary_cols(sim: *const fs_sim) -> u32;
    fn fs_slab_comm_stream(sim: *const fs_sim) -> *mut c_void;
    fn fs_slab_comm_begin(sim: *mut fs_sim) -> c_int;
    fn fs_slab_comm_end(sim: *mut fs_sim) -> c_int;
    fn fs_slab_wait_packed(sim: *mut fs_sim) -> c_int;
    fn fs_slab_counters_read(sim: *mut fs_sim, out: *mut SlabCounters) -> c_int;
    fn fs_slab_download(sim: *mut fs_sim, dst: *mut ParticleInstance, owned: *mut u8, cap: usize, n_live: *mut u32) -> c_int;
    fn fs_slab_max_speed(sim: *mut fs_sim, out: *mut f32) -> c_int;
    fn fs_slab_column_histogram(sim: *mut fs_sim, hist: *mut u32, grid_w_global: usize) -> c_int;
    fn fs_slab_rebalance_stats(sim: *mut fs_sim, stats: *mut u32, hist: *mut u32, grid_w_global: usize) -> c_int;
    // native RCCL transport
    fn fs_comm_unique_id(id: *mut u8) -> c_int;
    fn fs_comm_init(device: c_int, rank: c_int, world: c_int, id: *const u8, out: *mut *mut fs_comm) -> c_int;
    fn fs_comm_destroy(comm: *mut fs_comm);
    fn fs_slab_exchange(sim: *mut fs_sim, comm: *mut fs_comm, left_rank: c_int, right_rank: c_int,
                        send_left: *const c_void, send_right: *const c_void,
                        recv_left: *mut c_void, recv_right: *mut c_void) -> c_int;
    fn fs_comm_allreduce(sim: *mut fs_sim, comm: *mut fs_comm, device_buf: *mut c_void, count: usize, dtype: c_int, op: c_int) -> c_int;
    // 3D extension
    fn fs3_create(settings: *const Settings3, device: c_int, initial_offset: Vec3, out: *mut *mut fs_sim3) -> c_int;
    fn fs3_create_ex(settings: *const Settings3, device: c_int, initial_offset: Vec3, math_mode: c_int, out: *mut *mut fs_sim3) -> c_int;
    fn fs3_destroy(sim: *mut fs_sim3);
    fn fs3_step(sim: *mut fs_sim3, tick: *const TickSettings3) -> c_int;
    fn fs3_sync(sim: *mut fs_sim3) -> c_int;
    fn fs3_tick_count(sim: *const fs_sim3) -> u32;
    fn fs3_particle_count(sim: *const fs_sim3) -> u32;
    fn fs3_grid_dims(sim: *const fs_sim3, w: *mut u32, h: *mut u32, d: *mut u32) -> c_int;
    fn fs3_download_particles(sim: *mut fs_sim3, dst: *mut Particle3, n: usize) -> c_int;
    fn fs3_upload_particles(sim: *mut fs_sim3, src: *const Particle3, n: usize) -> c_int;
    fn fs3_reference_lattice(settings: *const Settings3, offset: Vec3, dst: *mut Particle3, n: usize) -> c_int;
    fn fs3_timed_steps(sim: *mut fs_sim3, tick: *const TickSettings3, steps: u32, ms_total: *mut f64) -> c_int;
    fn fs3_profile_enable(sim: *mut fs_sim3, enable: c_int) -> c_int;
    fn fs3_profile_read(sim: *mut fs_sim3, ms: *mut f64, steps: *mut u64, reset: c_int) -> c_int;
    // ResizableBuffer<T> (src/buffer.rs)
    fn fs_buffer_create(device: c_int, elem_size: usize, len: usize, name: *const c_char, out: *mut *mut fs_buffer) -> c_int;
    fn fs_buffer_resize(buf: *mut fs_buffer, new_cap: usize, resized: *mut c_int) -> c_int;
    fn fs_buffer_write(buf: *mut fs_buffer, offset: usize, data: *const c_void, count: usize) -> c_int;
    fn fs_buffer_read(buf: *mut fs_buffer, offset: usize, dst: *mut c_void, count: usize) -> c_int;
    fn fs_buffer_len(buf: *const fs_buffer) -> usize;
    fn fs_buffer_device_ptr(buf: *const fs_buffer) -> *mut c_void;
    fn fs_buffer_destroy(buf: *mut fs_buffer);
    // self-tests and diagnostics
    fn fs_selftest_constdiv(device: c_int, c: f32, y: f32, lo: f32, hi: f32, mismatches: *mut u32) -> c_int;
    fn fs_constdiv_status(sim: *const fs_sim) -> c_int;
    fn fs_selftest_sort(device: c_int, pairs: *mut u64, n: u32, fuse_stage: c_int, plan: *mut u32) -> c_int;
    fn fs_selftest_sort_policy(log2_count: u32, start_back: c_int, lag: u32, required: *const u32, steps: usize,
                               stage_out: *mut u32, single_out: *mut u32) -> c_int;
    fn fs_sort_plan_read(sim: *mut fs_sim, out: *mut SortPlanInfo) -> c_int;
    // errors
    fn fs_last_error() -> *const c_char;
    fn fs_abi_version() -> c_int;
}

fn check(status: c_int) {
    if status != 0 {
        // the reference unwrap()s everywhere; keep the panic-on-error behaviour on the Rust side
        let msg = unsafe { CStr::from_ptr(fs_last_error()) }.to_string_lossy().into_owned();
        panic!("fluidsim status {status}: {msg}");
    }
}

pub fn abi_version() -> i32 { unsafe { fs_abi_version() } }

pub struct FluidSimulation { raw: *mut fs_sim, settings: SimulationSettings }

impl FluidSimulation {
    /// `FluidSimulation::new(&wgpu::Device, SimulationSettings)` — src/simulation.rs:139.
    pub fn new(hip_device: i32, settings: SimulationSettings) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe { fs_create(&settings, hip_device, &mut raw) });
        Self { raw, settings }
    }
    /// The same with the build-defined options (sort mode, math mode, lattice offset, capacity).
    pub fn with_options(settings: SimulationSettings, opts: Options) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe { fs_create_ex(&settings, &opts, &mut raw) });
        Self { raw, settings }
    }
    pub fn default_options() -> Options { let mut o = Options::default(); unsafe { fs_options_default(&mut o) }; o }
    /// `tick(&mut self, &Queue, &mut CommandEncoder, TickSettings)` — src/simulation.rs:459.  Enqueues; does not wait.
    pub fn tick(&mut self, settings: TickSettings) { check(unsafe { fs_step(self.raw, &settings) }); }
    /// `device.poll(Wait)` — src/main.rs:79.
    pub fn wait(&mut self) { check(unsafe { fs_sync(self.raw) }); }
    /// `pub tick: u32` — src/simulation.rs:12.
    pub fn tick_count(&self) -> u32 { unsafe { fs_tick_count(self.raw) } }
    pub fn particle_count(&self) -> u32 { unsafe { fs_particle_count(self.raw) } }
    pub fn settings(&self) -> SimulationSettings { self.settings }
    /// `grid_w`, `grid_h` — src/simulation.rs:140-141.
    pub fn grid_dims(&self) -> (u32, u32) {
        let (mut w, mut h) = (0u32, 0u32); check(unsafe { fs_grid_dims(self.raw, &mut w, &mut h) }); (w, h)
    }
    /// `simulation_settings_bg()` — src/simulation.rs:552-554: the uniform of the last tick, by value.
    pub fn simulation_uniform(&self) -> SimulationUniform {
        let mut u = SimulationUniform::default(); check(unsafe { fs_get_uniform(self.raw, &mut u) }); u
    }
    /// The uniform `tick` would build (src/simulation.rs:470-497) without stepping.
    pub fn build_uniform(settings: &SimulationSettings, tick: &TickSettings, tick_count: u32) -> SimulationUniform {
        let mut u = SimulationUniform::default(); check(unsafe { fs_build_uniform(settings, tick, tick_count, &mut u) }); u
    }
    /// `simulation_bg()` binding 0 — src/simulation.rs:556-559: device pointer to the cell-sorted 32-byte records.
    pub fn particles_device(&mut self) -> *const ParticleInstance {
        let mut p = std::ptr::null(); check(unsafe { fs_particles_device(self.raw, &mut p) }); p
    }
    /// `simulation_bg()` binding 1: `start_indices`.
    pub fn start_indices_device(&mut self) -> (*const u32, usize) {
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { fs_start_indices_device(self.raw, &mut p, &mut n) }); (p, n)
    }
    /// `force_field_texture()` + `queue.write_buffer` — src/simulation.rs:562, src/renderer.rs:497-502.
    pub fn write_force_field(&mut self, field: &[Vec2], w: u32, h: u32) {
        assert_eq!(field.len(), (w * h) as usize);
        check(unsafe { fs_upload_force_field(self.raw, field.as_ptr(), w, h) });
    }
    /// `generate_smooth_gradient_field` (src/main.rs:403-515) on the GPU, straight into the force field.
    pub fn set_obstacle_image(&mut self, image: &[u8], w: u32, h: u32) {
        assert_eq!(image.len(), (w * h) as usize);
        check(unsafe { fs_generate_force_field(self.raw, 0, image.as_ptr(), w, h, std::ptr::null_mut()) });
    }
    /// One-copy hand-off for a wgpu renderer: `queue.write_buffer(&particles, 0, cast_slice(&v))`.
    pub fn download_particles(&mut self) -> Vec<ParticleInstance> {
        let mut v = vec![ParticleInstance::default(); self.particle_count() as usize];
        check(unsafe { fs_download_particles(self.raw, v.as_mut_ptr(), v.len()) }); v
    }
    pub fn upload_particles(&mut self, src: &[ParticleInstance]) { check(unsafe { fs_upload_particles(self.raw, src.as_ptr(), src.len()) }); }
    /// `start_indices` to the host (the renderer's second storage binding; u32[grid_w * grid_h]).
    pub fn download_start_indices(&mut self) -> Vec<u32> {
        let (w, h) = self.grid_dims();
        let mut v = vec![0u32; (w as usize) * (h as usize)];
        check(unsafe { fs_download_start_indices(self.raw, v.as_mut_ptr(), v.len()) }); v
    }
    pub fn upload_start_indices(&mut self, src: &[u32]) { check(unsafe { fs_upload_start_indices(self.raw, src.as_ptr(), src.len()) }); }
    /// Zero-copy hand-off: export the particle (or start_indices) allocation.  `dmabuf_fd` imports into Vulkan / wgpu
    /// as external memory (the buffers `simulation_bg` binds, src/simulation.rs:552-559); after this call the force
    /// pass writes the 32-byte records itself, so each frame costs no export pass and no PCIe copy.
    pub fn export(&mut self, which: c_int) -> MemHandle {
        let mut h = MemHandle { ipc: [0; 64], bytes: 0, device: 0, dmabuf_fd: -1 };
        check(unsafe { fs_export_handle(self.raw, which, &mut h) }); h
    }
    /// Headless `fluid_shader.wgsl:27-102`: width*height RGBA f32.
    pub fn render_density(&mut self, view: &View) -> Vec<f32> {
        let mut v = vec![0f32; 4 * view.width as usize * view.height as usize];
        check(unsafe { fs_render_density(self.raw, view, v.as_mut_ptr()) }); v
    }
    pub fn profile(&mut self, enable: bool) { check(unsafe { fs_profile_enable(self.raw, enable as c_int) }); }
    pub fn profile_read(&mut self, reset: bool) -> ([f64; PASS_COUNT], u64) {
        let (mut ms, mut steps) = ([0f64; PASS_COUNT], 0u64);
        check(unsafe { fs_profile_read(self.raw, ms.as_mut_ptr(), &mut steps, reset as c_int) }); (ms, steps)
    }
    pub fn timed_steps(&mut self, settings: TickSettings, steps: u32) -> f64 {
        let mut ms = 0f64; check(unsafe { fs_timed_steps(self.raw, &settings, steps, &mut ms) }); ms
    }
    pub fn sort_plan(&mut self) -> SortPlanInfo { let mut i = SortPlanInfo::default(); check(unsafe { fs_sort_plan_read(self.raw, &mut i) }); i }
    pub fn constdiv_status(&self) -> i32 { unsafe { fs_constdiv_status(self.raw) } }
    pub fn stream(&self) -> *mut c_void { unsafe { fs_stream(self.raw) } }
    pub fn raw(&mut self) -> *mut fs_sim { self.raw }
}
impl Drop for FluidSimulation { fn drop(&mut self) { unsafe { fs_destroy(self.raw) } } }

/// The reference lattice (src/simulation.rs:147-163) and sort schedule (:323-347) without a device.
pub fn reference_lattice(settings: &SimulationSettings, offset: Vec2) -> Vec<ParticleInstance> {
    let mut v = vec![ParticleInstance::default(); settings.particle_count as usize];
    check(unsafe { fs_reference_lattice(settings, offset, v.as_mut_ptr(), v.len()) }); v
}
pub fn sort_schedule(particle_count: u32) -> Vec<SortStep> {
    let n = unsafe { fs_sort_schedule(particle_count, std::ptr::null_mut(), 0) };
    let mut v = vec![SortStep::default(); n];
    unsafe { fs_sort_schedule(particle_count, v.as_mut_ptr(), n) }; v
}

/// Consumer side of `FluidSimulation::export` in another HIP process.
pub struct Imported { ptr: *mut c_void }
impl Imported {
    pub fn open(handle: &MemHandle, hip_device: i32) -> Self { let mut p = std::ptr::null_mut(); check(unsafe { fs_import_open(handle, hip_device, &mut p) }); Self { ptr: p } }
    pub fn read(&self, offset: usize, dst: &mut [u8]) { check(unsafe { fs_import_read(self.ptr, offset, dst.as_mut_ptr() as *mut c_void, dst.len()) }); }
    pub fn device_ptr(&self) -> *mut c_void { self.ptr }
}
impl Drop for Imported { fn drop(&mut self) { unsafe { fs_import_close(self.ptr); } } }

/// `ResizableBuffer<T>` — src/buffer.rs:17-88, over HIP device memory.  The wgpu arguments
/// (`device`, `usage`, `belt`, `encoder`) have no counterpart: `hip_device` selects the GPU at
/// `new`, copies are ordered on the buffer's own stream.
pub struct ResizableBuffer<T: Copy> { raw: *mut fs_buffer, pub len: usize, marker: PhantomData<T> }
impl<T: Copy> ResizableBuffer<T> {
    /// src/buffer.rs:27-43
    pub fn new(name: &'static str, hip_device: i32, len: usize) -> Self {
        let cname = CString::new(name).unwrap();
        let mut raw = std::ptr::null_mut();
        check(unsafe { fs_buffer_create(hip_device, std::mem::size_of::<T>(), len, cname.as_ptr(), &mut raw) });
        Self { raw, len, marker: PhantomData }
    }
    /// src/buffer.rs:46-67: grow-only, keeps the contents, clamps to the device maximum (warning); false when `new_cap < len`.
    pub fn resize(&mut self, new_cap: usize) -> bool {
        let mut r: c_int = 0;
        check(unsafe { fs_buffer_resize(self.raw, new_cap, &mut r) });
        self.len = unsafe { fs_buffer_len(self.raw) };
        r != 0
    }
    /// src/buffer.rs:70-87: data longer than the buffer is trimmed (and logged).
    pub fn write(&self, offset: usize, data: &[T]) {
        check(unsafe { fs_buffer_write(self.raw, offset, data.as_ptr() as *const c_void, data.len()) });
    }
    pub fn read(&self, offset: usize, count: usize) -> Vec<T> {
        let mut v: Vec<T> = Vec::with_capacity(count);
        check(unsafe { fs_buffer_read(self.raw, offset, v.as_mut_ptr() as *mut c_void, count) });
        unsafe { v.set_len(count) }; v
    }
    /// `.buffer` (the wgpu::Buffer): here the device pointer.
    pub fn device_ptr(&self) -> *mut c_void { unsafe { fs_buffer_device_ptr(self.raw) } }
}
impl<T: Copy> Drop for ResizableBuffer<T> { fn drop(&mut self) { unsafe { fs_buffer_destroy(self.raw) } } }

/// `SSBO<T>` — src/buffer.rs:9-14,91-173: a `ResizableBuffer<T>` plus its binding.  HIP kernels take
/// pointers, so `bind_group()` is the device pointer and `layout()` the element size.
pub struct SSBO<T: Copy> { pub buffer: ResizableBuffer<T> }
impl<T: Copy> SSBO<T> {
    pub fn new(name: &'static str, hip_device: i32, data_len: usize) -> Self { Self { buffer: ResizableBuffer::new(name, hip_device, data_len) } }
    /// src/buffer.rs:129-152: no-op unless `new_cap > len`.
    pub fn resize(&mut self, new_cap: usize) { if new_cap > self.buffer.len { self.buffer.resize(new_cap); } }
    /// src/buffer.rs:155-157: `write(0, data)`.
    pub fn update(&self, data: &[T]) { self.write(0, data) }
    pub fn write(&self, offset: usize, data: &[T]) { self.buffer.write(offset, data) }
    pub fn bind_group(&self) -> *mut c_void { self.buffer.device_ptr() }
    pub fn layout(&self) -> usize { std::mem::size_of::<T>() }
    pub fn len(&self) -> usize { self.buffer.len }
}

/// One rank of the multi-GPU slab decomposition (fs_slab_*): pack -> Comm::exchange -> step.
pub struct SlabSimulation { raw: *mut fs_sim }
impl SlabSimulation {
    pub fn new(global: &SimulationSettings, hip_device: i32, cfg: &SlabConfig) -> Self {
        let mut raw = std::ptr::null_mut(); check(unsafe { fs_slab_create(global, hip_device, cfg, &mut raw) }); Self { raw }
    }
    pub fn upload_owned(&mut self, src: &[ParticleInstance]) { check(unsafe { fs_slab_upload_owned(self.raw, src.as_ptr(), src.len()) }); }
    pub fn set_window(&mut self, own_lo: u32, own_hi: u32) { check(unsafe { fs_slab_set_window(self.raw, own_lo, own_hi) }); }
    pub fn message_bytes(&self) -> usize { unsafe { fs_slab_message_bytes(self.raw) } }
    pub unsafe fn pack(&mut self, tick: &TickSettings, send_left: *mut c_void, send_right: *mut c_void) { check(fs_slab_pack(self.raw, tick, send_left, send_right)); }
    pub unsafe fn step(&mut self, recv_left: *const c_void, recv_right: *const c_void) { check(fs_slab_step(self.raw, recv_left, recv_right)); }
    /// Overlapped step (the default with the counting sort): `pack` also enqueues the interior columns' whole step, `step` the
    /// boundary strips; the exchange between them runs on `comm_stream()` (Comm::exchange does the hand-over itself).
    pub fn overlapped(&self) -> bool { unsafe { fs_slab_overlapped(self.raw) != 0 } }
    pub fn set_boundary_cols(&mut self, cols: u32) { check(unsafe { fs_slab_set_boundary_cols(self.raw, cols) }); }
    pub fn boundary_cols(&self) -> u32 { unsafe { fs_slab_boundary_cols(self.raw) } }
    pub fn comm_stream(&self) -> *mut c_void { unsafe { fs_slab_comm_stream(self.raw) } }
    pub fn comm_begin(&mut self) { check(unsafe { fs_slab_comm_begin(self.raw) }); }
    pub fn comm_end(&mut self) { check(unsafe { fs_slab_comm_end(self.raw) }); }
    pub fn wait_packed(&mut self) { check(unsafe { fs_slab_wait_packed(self.raw) }); }
    pub fn counters(&mut self) -> SlabCounters { let mut c = SlabCounters::default(); check(unsafe { fs_slab_counters_read(self.raw, &mut c) }); c }
    pub fn max_speed(&mut self) -> f32 { let mut v = 0f32; check(unsafe { fs_slab_max_speed(self.raw, &mut v) }); v }
    pub fn column_histogram(&mut self, hist: &mut [u32]) { check(unsafe { fs_slab_column_histogram(self.raw, hist.as_mut_ptr(), hist.len()) }); }
    /// Device-side re-balancing inputs for `Comm::allreduce`: `stats` = 4 u32 {lost, overflow, far_halo, max-speed bits}, `hist` = grid_w u32.
    pub unsafe fn rebalance_stats(&mut self, stats: *mut u32, hist: *mut u32, grid_w_global: usize) { check(fs_slab_rebalance_stats(self.raw, stats, hist, grid_w_global)); }
    pub fn download(&mut self, cap: usize) -> (Vec<ParticleInstance>, Vec<u8>) {
        let (mut rec, mut owned, mut n) = (vec![ParticleInstance::default(); cap], vec![0u8; cap], 0u32);
        check(unsafe { fs_slab_download(self.raw, rec.as_mut_ptr(), owned.as_mut_ptr(), cap, &mut n) });
        rec.truncate(n as usize); owned.truncate(n as usize); (rec, owned)
    }
    pub fn wait(&mut self) { check(unsafe { fs_sync(self.raw) }); }
    pub fn raw(&mut self) -> *mut fs_sim { self.raw }
}
impl Drop for SlabSimulation { fn drop(&mut self) { unsafe { fs_destroy(self.raw) } } }

/// One RCCL communicator per process / GPU (multi-GPU slab runs: fs_slab_pack -> Comm::exchange -> fs_slab_step).
pub struct Comm { raw: *mut fs_comm }
pub const COMM_U32: c_int = 0; pub const COMM_U64: c_int = 1; pub const COMM_F32: c_int = 2;
pub const COMM_SUM: c_int = 0; pub const COMM_MAX: c_int = 1;
impl Comm {
    pub fn unique_id() -> [u8; COMM_ID_BYTES] { let mut id = [0u8; COMM_ID_BYTES]; check(unsafe { fs_comm_unique_id(id.as_mut_ptr()) }); id }
    pub fn init(hip_device: i32, rank: i32, world: i32, id: &[u8; COMM_ID_BYTES]) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe { fs_comm_init(hip_device, rank, world, id.as_ptr(), &mut raw) }); Self { raw }
    }
    /// Grouped ncclSend/ncclRecv of the two fixed-size slab messages on the simulation's own stream.
    pub unsafe fn exchange(&self, sim: *mut fs_sim, left: i32, right: i32, send_left: *const c_void,
                           send_right: *const c_void, recv_left: *mut c_void, recv_right: *mut c_void) {
        check(fs_slab_exchange(sim, self.raw, left, right, send_left, send_right, recv_left, recv_right));
    }
    /// In-place all-reduce on the simulation's stream (re-balancing histogram / violation counters).
    pub unsafe fn allreduce(&self, sim: *mut fs_sim, device_buf: *mut c_void, count: usize, dtype: c_int, op: c_int) {
        check(fs_comm_allreduce(sim, self.raw, device_buf, count, dtype, op));
    }
}
impl Drop for Comm { fn drop(&mut self) { unsafe { fs_comm_destroy(self.raw) } } }

/// 3D extension (fs3_*; no reference counterpart).
pub struct FluidSimulation3D { raw: *mut fs_sim3 }
impl FluidSimulation3D {
    pub fn new(hip_device: i32, settings: Settings3, initial_offset: Vec3) -> Self {
        let mut raw = std::ptr::null_mut(); check(unsafe { fs3_create(&settings, hip_device, initial_offset, &mut raw) }); Self { raw }
    }
    pub fn with_math_mode(hip_device: i32, settings: Settings3, initial_offset: Vec3, math_mode: i32) -> Self {
        let mut raw = std::ptr::null_mut(); check(unsafe { fs3_create_ex(&settings, hip_device, initial_offset, math_mode, &mut raw) }); Self { raw }
    }
    pub fn tick(&mut self, t: TickSettings3) { check(unsafe { fs3_step(self.raw, &t) }); }
    pub fn wait(&mut self) { check(unsafe { fs3_sync(self.raw) }); }
    pub fn tick_count(&self) -> u32 { unsafe { fs3_tick_count(self.raw) } }
    pub fn particle_count(&self) -> u32 { unsafe { fs3_particle_count(self.raw) } }
    pub fn grid_dims(&self) -> (u32, u32, u32) { let (mut w, mut h, mut d) = (0, 0, 0); check(unsafe { fs3_grid_dims(self.raw, &mut w, &mut h, &mut d) }); (w, h, d) }
    pub fn download_particles(&mut self) -> Vec<Particle3> {
        let mut v = vec![Particle3::default(); self.particle_count() as usize];
        check(unsafe { fs3_download_particles(self.raw, v.as_mut_ptr(), v.len()) }); v
    }
    pub fn upload_particles(&mut self, src: &[Particle3]) { check(unsafe { fs3_upload_particles(self.raw, src.as_ptr(), src.len()) }); }
    pub fn reference_lattice(settings: &Settings3, offset: Vec3) -> Vec<Particle3> {
        let mut v = vec![Particle3::default(); settings.particle_count as usize];
        check(unsafe { fs3_reference_lattice(settings, offset, v.as_mut_ptr(), v.len()) }); v
    }
    pub fn timed_steps(&mut self, t: TickSettings3, steps: u32) -> f64 { let mut ms = 0f64; check(unsafe { fs3_timed_steps(self.raw, &t, steps, &mut ms) }); ms }
    pub fn profile(&mut self, enable: bool) { check(unsafe { fs3_profile_enable(self.raw, enable as c_int) }); }
    pub fn profile_read(&mut self, reset: bool) -> ([f64; PASS_COUNT], u64) {
        let (mut ms, mut steps) = ([0f64; PASS_COUNT], 0u64);
        check(unsafe { fs3_profile_read(self.raw, ms.as_mut_ptr(), &mut steps, reset as c_int) }); (ms, steps)
    }
}
impl Drop for FluidSimulation3D { fn drop(&mut self) { unsafe { fs3_destroy(self.raw) } } }

/// Create-time proofs and the sort self-tests, for a Rust-side test suite.
pub mod selftest {
    use super::*;
    pub fn constdiv(hip_device: i32, c: f32, y: f32, lo: f32, hi: f32) -> u32 { let mut m = 0u32; check(unsafe { fs_selftest_constdiv(hip_device, c, y, lo, hi, &mut m) }); m }
    pub fn sort(hip_device: i32, pairs: &mut [u64], fuse_stage: i32) -> [u32; 2] {
        let mut plan = [0u32; 2];
        check(unsafe { fs_selftest_sort(hip_device, pairs.as_mut_ptr(), pairs.len() as u32, fuse_stage, plan.as_mut_ptr()) }); plan
    }
    pub fn sort_policy(log2_count: u32, start_back: i32, lag: u32, required: &[u32]) -> (Vec<u32>, Vec<u32>) {
        let (mut stage, mut single) = (vec![0u32; required.len()], vec![0u32; required.len()]);
        check(unsafe { fs_selftest_sort_policy(log2_count, start_back, lag, required.as_ptr(), required.len(), stage.as_mut_ptr(), single.as_mut_ptr()) });
        (stage, single)
    }
}

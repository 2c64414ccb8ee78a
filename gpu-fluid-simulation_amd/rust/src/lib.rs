//! Rust binding a maintainer of rookieCookies/gpu-fluid-simulation would add to run the
//! per-tick SPH step on an MI355X through `libfluidsim_hip.so` (C ABI: include/fluidsim.h).
//! Same type names and fields as src/simulation.rs so `renderer.rs` keeps compiling:
//! `FluidSimulation::new(device, settings)`, `.tick(settings)`, `.tick` (as `tick_count()`).
//! NOT BUILT in this repository's image (no Rust toolchain) — source only.
#![allow(non_camel_case_types)]
use std::ffi::{c_char, c_int, c_void, CStr};

#[repr(C)] #[derive(Clone, Copy, Default, Debug)] pub struct Vec2 { pub x: f32, pub y: f32 }
#[repr(C)] #[derive(Clone, Copy, Default, Debug)] pub struct UVec2 { pub x: u32, pub y: u32 }

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

#[repr(C)] pub struct fs_sim { _p: [u8; 0] }

extern "C" {
    fn fs_create(settings: *const SimulationSettings, device: c_int, out: *mut *mut fs_sim) -> c_int;
    fn fs_destroy(sim: *mut fs_sim);
    fn fs_step(sim: *mut fs_sim, tick: *const TickSettings) -> c_int;
    fn fs_sync(sim: *mut fs_sim) -> c_int;
    fn fs_tick_count(sim: *const fs_sim) -> u32;
    fn fs_particles_device(sim: *mut fs_sim, out: *mut *const ParticleInstance) -> c_int;
    fn fs_start_indices_device(sim: *mut fs_sim, out: *mut *const u32, count: *mut usize) -> c_int;
    fn fs_upload_force_field(sim: *mut fs_sim, field: *const Vec2, w: u32, h: u32) -> c_int;
    fn fs_download_particles(sim: *mut fs_sim, dst: *mut ParticleInstance, n: usize) -> c_int;
    fn fs_last_error() -> *const c_char;
    /// diagnostics of the sort's late-stage plan (include/fluidsim.h fs_sort_plan_info): six u32 counters
    fn fs_sort_plan_read(sim: *mut fs_sim, out: *mut [u32; 6]) -> c_int;
    // hand-off without a host round trip (include/fluidsim.h: fs_export_handle)
    fn fs_export_handle(sim: *mut fs_sim, which: c_int, out: *mut MemHandle) -> c_int;
    // native RCCL transport for a multi-GPU host (include/fluidsim.h: fs_comm_*, fs_slab_exchange)
    fn fs_comm_unique_id(id: *mut u8) -> c_int;
    fn fs_comm_init(device: c_int, rank: c_int, world: c_int, id: *const u8, out: *mut *mut c_void) -> c_int;
    fn fs_comm_destroy(comm: *mut c_void);
    fn fs_slab_exchange(sim: *mut fs_sim, comm: *mut c_void, left_rank: c_int, right_rank: c_int,
                        send_left: *const c_void, send_right: *const c_void,
                        recv_left: *mut c_void, recv_right: *mut c_void) -> c_int;
}

/// fs_mem_handle: interprocess / external-memory handle of a device buffer of the simulation (80 bytes).
#[repr(C)] #[derive(Clone, Copy)]
pub struct MemHandle { pub ipc: [u8; 64], pub bytes: u64, pub device: i32, pub dmabuf_fd: i32 }
const _: () = assert!(std::mem::size_of::<MemHandle>() == 80);
pub const EXPORT_PARTICLES: c_int = 0;
pub const EXPORT_START_INDICES: c_int = 1;

fn check(status: c_int) {
    if status != 0 {
        // the reference unwrap()s everywhere; keep the panic-on-error behaviour on the Rust side
        let msg = unsafe { CStr::from_ptr(fs_last_error()) }.to_string_lossy().into_owned();
        panic!("fluidsim status {status}: {msg}");
    }
}

pub struct FluidSimulation { raw: *mut fs_sim, settings: SimulationSettings }

impl FluidSimulation {
    /// `FluidSimulation::new(&wgpu::Device, SimulationSettings)` — src/simulation.rs:139.
    pub fn new(hip_device: i32, settings: SimulationSettings) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe { fs_create(&settings, hip_device, &mut raw) });
        Self { raw, settings }
    }
    /// `tick(&mut self, &Queue, &mut CommandEncoder, TickSettings)` — src/simulation.rs:459.
    pub fn tick(&mut self, settings: TickSettings) { check(unsafe { fs_step(self.raw, &settings) }); }
    pub fn wait(&mut self) { check(unsafe { fs_sync(self.raw) }); }
    /// `pub tick: u32` — src/simulation.rs:12.
    pub fn tick_count(&self) -> u32 { unsafe { fs_tick_count(self.raw) } }
    pub fn settings(&self) -> SimulationSettings { self.settings }
    /// Device pointer to the cell-sorted 32-byte records (simulation_bg binding 0).
    pub fn particles_device(&mut self) -> *const ParticleInstance {
        let mut p = std::ptr::null(); check(unsafe { fs_particles_device(self.raw, &mut p) }); p
    }
    pub fn start_indices_device(&mut self) -> (*const u32, usize) {
        let (mut p, mut n) = (std::ptr::null(), 0usize);
        check(unsafe { fs_start_indices_device(self.raw, &mut p, &mut n) }); (p, n)
    }
    /// `force_field_texture()` + `queue.write_buffer` — src/simulation.rs:562, src/renderer.rs:497-502.
    pub fn write_force_field(&mut self, field: &[Vec2], w: u32, h: u32) {
        assert_eq!(field.len(), (w * h) as usize);
        check(unsafe { fs_upload_force_field(self.raw, field.as_ptr(), w, h) });
    }
    /// One-copy hand-off for a wgpu renderer: `queue.write_buffer(&particles, 0, cast_slice(&v))`.
    pub fn download_particles(&mut self) -> Vec<ParticleInstance> {
        let mut v = vec![ParticleInstance::default(); self.settings.particle_count as usize];
        check(unsafe { fs_download_particles(self.raw, v.as_mut_ptr(), v.len()) }); v
    }
    /// Zero-copy hand-off: export the particle (or start_indices) allocation.  `dmabuf_fd` imports into Vulkan / wgpu
    /// as external memory (the buffers `simulation_bg` binds, src/simulation.rs:552-559); after this call the force
    /// pass writes the 32-byte records itself, so each frame costs no export pass and no PCIe copy.
    pub fn export(&mut self, which: c_int) -> MemHandle {
        let mut h = MemHandle { ipc: [0; 64], bytes: 0, device: 0, dmabuf_fd: -1 };
        check(unsafe { fs_export_handle(self.raw, which, &mut h) }); h
    }
    pub fn raw(&mut self) -> *mut c_void { self.raw as *mut c_void }
}

/// One RCCL communicator per process / GPU (multi-GPU slab runs: fs_slab_pack -> Comm::exchange -> fs_slab_step).
pub struct Comm { raw: *mut c_void }
impl Comm {
    pub fn unique_id() -> [u8; 128] { let mut id = [0u8; 128]; check(unsafe { fs_comm_unique_id(id.as_mut_ptr()) }); id }
    pub fn init(hip_device: i32, rank: i32, world: i32, id: &[u8; 128]) -> Self {
        let mut raw = std::ptr::null_mut();
        check(unsafe { fs_comm_init(hip_device, rank, world, id.as_ptr(), &mut raw) }); Self { raw }
    }
    /// Grouped ncclSend/ncclRecv of the two fixed-size slab messages on the simulation's own stream.
    pub unsafe fn exchange(&self, sim: *mut c_void, left: i32, right: i32, send_left: *const c_void,
                           send_right: *const c_void, recv_left: *mut c_void, recv_right: *mut c_void) {
        check(fs_slab_exchange(sim as *mut fs_sim, self.raw, left, right, send_left, send_right, recv_left, recv_right));
    }
}
impl Drop for Comm { fn drop(&mut self) { unsafe { fs_comm_destroy(self.raw) } } }
impl Drop for FluidSimulation { fn drop(&mut self) { unsafe { fs_destroy(self.raw) } } }

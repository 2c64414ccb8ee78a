/*
 * fluidsim.h — C ABI of the MI355X-native SPH fluid-step engine.
 *
 * This is the drop-in boundary for the reference's per-tick hot path
 * (rookieCookies/gpu-fluid-simulation).  Every entry point cites the reference
 * interface it replaces as file:line relative to the reference tree.  Plain
 * pointers and sizes only; no C++/torch types cross this boundary; nothing here
 * throws or aborts — every call returns an fs_status.
 *
 * Reference surface mirrored here:
 *   FluidSimulation::new      src/simulation.rs:139-455
 *   FluidSimulation::tick     src/simulation.rs:459-539
 *   accessors                 src/simulation.rs:542-564
 *   ParticleInstance (32 B)   src/simulation.rs:126-135, funcs.wgsl:1-8
 *   SimulationUniform (120 B) src/simulation.rs:53-90,   funcs.wgsl:17-51
 *   SimulationSettings        src/simulation.rs:95-104
 *   TickSettings              src/simulation.rs:107-122
 *   ResizableBuffer<T>/SSBO<T> src/buffer.rs:9-173
 */
#ifndef FLUIDSIM_H
#define FLUIDSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS_ABI_VERSION 2

/* ------------------------------------------------------------------ status */
typedef enum fs_status {
    FS_OK = 0,
    FS_ERR_INVALID = 1,      /* bad argument; N <= 1 (reference panics: simulation.rs:323-324) */
    FS_ERR_DEVICE = 2,       /* HIP runtime error (message in fs_last_error) */
    FS_ERR_OOM = 3,          /* device or host allocation failed */
    FS_ERR_UNSUPPORTED = 4,  /* option combination not built */
    FS_ERR_COMM = 5          /* RCCL error, or librccl could not be loaded (fs_comm_*, fs_slab_exchange) */
} fs_status;

/* ------------------------------------------------------------------- PODs */
typedef struct fs_vec2 { float x, y; } fs_vec2;
typedef struct fs_vec3 { float x, y, z; } fs_vec3;
typedef struct fs_uvec2 { uint32_t x, y; } fs_uvec2;

/* SimulationSettings — src/simulation.rs:95-104 (same fields, same order). */
typedef struct fs_settings {
    uint32_t particle_count;
    float particle_spacing;
    float smoothing_radius;
    fs_vec2 size;
    fs_uvec2 texture_size;
} fs_settings;

/* TickSettings — src/simulation.rs:107-122 (same fields, same order). */
typedef struct fs_tick_settings {
    float delta;
    fs_vec2 gravity;
    float mass;
    float pressure_constant;
    float rest_density;
    float damping_factor;
    float viscosity_coefficient;
    float surface_tension_treshold;     /* sic — spelling follows the reference */
    float surface_tension_coefficient;
    float mouse_force_radius;
    float mouse_force_power;
    fs_vec2 mouse_pos;
    int32_t mouse_state;
} fs_tick_settings;

/* ParticleInstance — src/simulation.rs:126-135; 32 bytes, offsets 0/8/16/24/28. */
typedef struct fs_particle {
    fs_vec2 position;
    fs_vec2 predicted_position;
    fs_vec2 velocity;
    float density;
    uint32_t grid;
} fs_particle;

/* SimulationUniform — src/simulation.rs:53-90; 120 bytes. */
typedef struct fs_uniform {
    float delta;
    uint32_t particle_count;
    float sqr_radius;
    uint32_t frame_time;
    fs_vec2 gravity;
    fs_vec2 bounds;
    fs_vec2 mouse_pos;
    float smoothing_radius;
    float particle_mass;
    float pressure_constant;
    float rest_density;
    float damping_factor;
    float viscosity_coefficient;
    float surface_tension_treshold;
    float surface_tension_coefficient;
    float poly6_kernel_volume;
    float poly6_kernel_derivative;
    float poly6_kernel_laplacian;
    float spiky_kernel_derivative;
    float viscosity_kernel;
    int32_t mouse_state;
    float mouse_force_radius;
    float mouse_force_power;
    uint32_t grid_w;
    uint32_t grid_h;
    fs_vec2 texture_size;
} fs_uniform;

/* SortUniform payload — src/simulation.rs:40-50 (without the 240-byte pad). */
typedef struct fs_sort_step {
    uint32_t group_width;
    uint32_t group_height;
    uint32_t step_index;
    uint32_t num_values;
} fs_sort_step;

/* Build-defined options (NOT in the reference; defaults reproduce the reference). */
typedef enum fs_sort_mode {
    FS_SORT_BITONIC = 0,   /* reference network (sort.wgsl:27-51): bit-exact permutation */
    FS_SORT_COUNTING = 1   /* O(N) cell counting sort; stable within a cell */
} fs_sort_mode;

typedef enum fs_math_mode {
    FS_MATH_IEEE = 0,      /* correctly rounded / and sqrt, no contraction: bit-identical to the CPU oracle */
    FS_MATH_WGSL_ULP = 1,  /* native rcp / sqrt in the force pass (<= ~1.5 ulp): within WGSL's own accuracy
                              contract for the reference shaders (division 2.5 ULP, sqrt 2 ULP), not bit-exact */
    FS_MATH_TOLERANCE = 2  /* density and force terms re-associated for speed (FMA, one rsqrt per pair, pressure and
                              1/density precomputed per particle): positions / velocities / densities within
                              rtol 1e-5, atol 1e-4*h of the IEEE oracle per step (north_star's float contract);
                              cell keys and start_indices remain bit-exact.  Never the default. */
} fs_math_mode;

typedef struct fs_options {
    int32_t device;            /* HIP device ordinal */
    int32_t sort_mode;         /* fs_sort_mode */
    int32_t ref_quirks;        /* 1 = reproduce compute.wgsl:49-55 stale cell-start behaviour */
    int32_t math_mode;         /* fs_math_mode (default FS_MATH_IEEE) */
    fs_vec2 initial_offset;    /* translation added to the reference lattice (dam-break scene) */
    uint32_t capacity;         /* particle slots to allocate (0 = particle_count); multi-GPU slabs */
    uint32_t reserved1;
} fs_options;

typedef struct fs_sim fs_sim;       /* opaque: one simulation, one HIP stream */
typedef struct fs_buffer fs_buffer; /* opaque: ResizableBuffer<T> */

/* ------------------------------------------------------------- lifecycle */
/* FluidSimulation::new (src/simulation.rs:139): builds the reference lattice
 * (:147-163), zeroed start_indices (:204-209) and force field (:213-218).
 * device = HIP ordinal.  N <= 1 -> FS_ERR_INVALID (reference: ilog2(0) panic);
 * N (or options.capacity) > 2^28 -> FS_ERR_INVALID (32-bit byte offsets in the kernels). */
fs_status fs_create(const fs_settings* settings, int device, fs_sim** out);
fs_status fs_create_ex(const fs_settings* settings, const fs_options* opts, fs_sim** out);
void fs_options_default(fs_options* opts);
/* Drop of FluidSimulation (Renderer owns it: src/renderer.rs:31). */
void fs_destroy(fs_sim* sim);

/* --------------------------------------------------------------- stepping */
/* FluidSimulation::tick (src/simulation.rs:459-539).  Pre-increments `tick`,
 * builds the 120-byte uniform (:470-497) and enqueues the whole pass chain on
 * the simulation's stream.  Non-blocking, like the reference (which only
 * records into a CommandEncoder; submit happens at src/main.rs:226).
 * FS_ERR_DEVICE from fs_step / fs_timed_steps is terminal for the handle: besides HIP runtime errors it reports that the
 * sort's stand-by kernel (csrc/kernels_sort.hip k_late_fallback, a persistent launch with a bounded-spin grid barrier)
 * timed out on a barrier in an EARLIER step — the host learns of it from the device's report a few steps later, and the
 * particle order of every step since is undefined.  Destroy the handle and re-create it (or re-upload a checkpoint into
 * a new one); no further call on the old handle is meaningful.  fs_sort_plan_info.timeouts counts such events
 * (0 on a healthy device: the grid is <= 128 workgroups on 256 CUs, all co-resident). */
fs_status fs_step(fs_sim* sim, const fs_tick_settings* tick);
/* device.poll(Wait) equivalent (src/main.rs:79). */
fs_status fs_sync(fs_sim* sim);
/* pub tick: u32 (src/simulation.rs:12). */
uint32_t fs_tick_count(const fs_sim* sim);
uint32_t fs_particle_count(const fs_sim* sim);
/* grid_w / grid_h (src/simulation.rs:140-141). */
fs_status fs_grid_dims(const fs_sim* sim, uint32_t* grid_w, uint32_t* grid_h);
/* The HIP stream the pass chain runs on (hipStream_t as void*). */
void* fs_stream(const fs_sim* sim);

/* ------------------------------------------- data the renderer consumes */
/* simulation_bg binding 0 (src/simulation.rs:552-559; fluid_shader.wgsl:38-75):
 * cell-sorted 32-byte AoS records, device pointer.  The engine keeps SoA state;
 * the AoS view is materialised on the stream by this call. */
fs_status fs_particles_device(fs_sim* sim, const fs_particle** out);
/* simulation_bg binding 1: start_indices u32[grid_w*grid_h] (persistent, never cleared). */
fs_status fs_start_indices_device(fs_sim* sim, const uint32_t** out, size_t* count);
/* simulation_settings_bg (src/simulation.rs:552-554): last uniform written by fs_step. */
fs_status fs_get_uniform(const fs_sim* sim, fs_uniform* out);
/* force_field_texture() (src/simulation.rs:562-564) + queue.write_buffer (src/renderer.rs:497-502). */
fs_status fs_upload_force_field(fs_sim* sim, const fs_vec2* field, uint32_t w, uint32_t h);

/* ---- hand-off without a host round trip --------------------------------------------
 * The reference's renderer binds the simulation's particle and start_indices buffers directly
 * (simulation_bg: src/simulation.rs:552-559, used at src/renderer.rs:457-458; fluid_shader.wgsl:38-75).
 * fs_export_handle gives a consumer in ANOTHER process (or another API) the same thing: an interprocess
 * handle of the device allocation holding the cell-sorted 32-byte records / the start_indices table.
 * Registering the particle export switches the engine to a LIVE AoS view: from the next fs_step on the force pass
 * writes the ParticleInstance records itself (no export pass, no copy); the consumer reads them after fs_sync
 * (or after its own wait on the step).  `ipc` is a hipIpcMemHandle_t (HIP consumers: fs_import_open);
 * `dmabuf_fd` is a dma-buf file descriptor of the same range for external-memory import by Vulkan / wgpu
 * (-1 when the runtime cannot export one); the caller owns and closes it. */
enum { FS_EXPORT_PARTICLES = 0, FS_EXPORT_START_INDICES = 1 };
typedef struct fs_mem_handle {
    uint8_t ipc[64];
    uint64_t bytes;       /* size of the exported range: particle_count * 32, or grid_w * grid_h * 4 */
    int32_t device;       /* HIP ordinal the allocation lives on */
    int32_t dmabuf_fd;
} fs_mem_handle;
fs_status fs_export_handle(fs_sim* sim, int which, fs_mem_handle* out);
/* Consumer side (any process with a HIP device): map an exported range, read from it, unmap. */
fs_status fs_import_open(const fs_mem_handle* handle, int device, void** dev_ptr);
fs_status fs_import_read(const void* dev_ptr, size_t offset, void* dst, size_t bytes);   /* blocking copy to host */
fs_status fs_import_close(void* dev_ptr);

/* Host copies (checkpoint / tests).  Blocking.  n = number of records. */
fs_status fs_download_particles(fs_sim* sim, fs_particle* dst, size_t n);
fs_status fs_upload_particles(fs_sim* sim, const fs_particle* src, size_t n);
fs_status fs_download_start_indices(fs_sim* sim, uint32_t* dst, size_t n);
fs_status fs_upload_start_indices(fs_sim* sim, const uint32_t* src, size_t n);

/* ------------------------------------------------- host-side mirrors (pure) */
/* Initial lattice, src/simulation.rs:147-163, f32 arithmetic as written. */
fs_status fs_reference_lattice(const fs_settings* settings, fs_vec2 offset, fs_particle* dst, size_t n);
/* Sort schedule, src/simulation.rs:323-347.  Returns the number of steps
 * S(S+1)/2; fills up to `cap` entries when dst != NULL. */
size_t fs_sort_schedule(uint32_t particle_count, fs_sort_step* dst, size_t cap);
/* Uniform construction, src/simulation.rs:470-497. */
fs_status fs_build_uniform(const fs_settings* settings, const fs_tick_settings* tick,
                           uint32_t tick_count, fs_uniform* out);

/* ------------------------------------------- obstacle field producer */
/* generate_smooth_gradient_field (src/main.rs:403-515): u8 mask (> 128 = obstacle source; none ->
 * the image border) -> per-pixel vector to the nearest source, by the reference's two-pass raster
 * propagation, reproduced exactly.  `field_host` may be NULL; with `sim` the result is also
 * written straight into the simulation's force field (the renderer's write_buffer,
 * src/renderer.rs:497-502), in which case (w, h) must equal settings.texture_size.  sim may be
 * NULL (then `device` selects the GPU).  Limits: h <= 1024, w < 65536.  The sweep is inherently sequential along
 * x + 2y (a pixel depends on its left / upper neighbours' RESULTS), so it runs as ONE workgroup — one thread per image
 * row, ~3 w + 2 h barrier steps — on one CU: ~1 ms at 1024^2, off the per-tick path (the reference runs it on a CPU
 * thread per rendered frame). */
fs_status fs_generate_force_field(fs_sim* sim, int device, const uint8_t* image, uint32_t w, uint32_t h,
                                  fs_vec2* field_host);

/* ------------------------------------------------- renderer hand-off */
/* Headless version of the reference's density-splat fragment shader (fluid_shader.wgsl:27-102):
 * per pixel a 5x5-cell walk over the cell-sorted particles, Gaussian splat of density and
 * speed, colour ramp.  The reference draws a full-screen quad through an orthographic
 * projection over the whole domain with +y down (src/renderer.rs:558-561); `fs_view` is that
 * mapping made explicit: pixel (i, j) samples world_min + ((i+0.5)/width, (j+0.5)/height) *
 * (world_max - world_min).  Output: width*height RGBA f32 (straight alpha), host memory. */
typedef struct fs_view {
    fs_vec2 world_min, world_max;
    uint32_t width, height;
} fs_view;
fs_status fs_render_density(fs_sim* sim, const fs_view* view, float* rgba_host);

/* -------------------------------------------------------------- profiling */
/* Per-pass device time measured with hipEvents on the simulation's stream. */
/* FS_PASS_BOUNDARY: slab handles only — the part of a step that runs after the halo exchange (overlapped step: waiting for
 * the incoming messages + the boundary strips); 0 on every other handle. */
enum { FS_PASS_PREDICT_KEY = 0, FS_PASS_SORT = 1, FS_PASS_REORDER = 2, FS_PASS_DENSITY = 3,
       FS_PASS_FORCE = 4, FS_PASS_BOUNDARY = 5, FS_PASS_COUNT = 6 };
fs_status fs_profile_enable(fs_sim* sim, int enable);
/* Accumulated milliseconds per pass since the last reset and number of steps. */
fs_status fs_profile_read(fs_sim* sim, double ms[FS_PASS_COUNT], uint64_t* steps, int reset);
/* Time `steps` consecutive fs_step calls with one hipEvent pair on the stream. */
fs_status fs_timed_steps(fs_sim* sim, const fs_tick_settings* tick, uint32_t steps, double* ms_total);

/* ------------------------------------------------ multi-GPU slab mode (build extension) */
/* NOT in the reference (single wgpu device, src/renderer.rs:108-133).  SURVEY.md §8e: a rank
 * owns the global cell columns [own_lo, own_hi) of the grid (src/simulation.rs:140-141) and
 * exchanges one fixed-size message per neighbour per step (migrants + 2-column ghost halo).
 * Messages are device buffers of fs_slab_message_bytes(); the caller moves them between
 * ranks (RCCL send/recv).  A step = fs_slab_pack -> exchange -> fs_slab_step, all
 * asynchronous on the simulation's stream; counts stay on the device. */
typedef struct fs_slab_config {
    uint32_t own_lo, own_hi;      /* owned window in global cell columns */
    uint32_t has_left, has_right; /* neighbours present */
    uint32_t capacity;            /* particle slots of the local array (incl. 2*recv_capacity) */
    uint32_t recv_capacity;       /* records per incoming message */
    uint32_t max_cols;            /* widest owned window this handle must support (re-balancing) */
    uint32_t sort_mode;           /* low byte: 0 = default (counting sort), 1 + fs_sort_mode selects explicitly; | FS_SLAB_SERIAL:
                                     the serial step (pack -> exchange -> everything) instead of the overlapped one */
} fs_slab_config;
#define FS_SLAB_SERIAL 0x100u
#define FS_SLAB_STRIPS 0x200u
#define FS_SLAB_ROWMAJOR 0x400u

typedef struct fs_slab_counters {
    uint32_t n_live;        /* live slots after the last step (owned + ghosts; overlapped step: the sorted prefix, i.e. the
                               carried-over particles — ghosts and this step's migrants live outside it) */
    uint32_t lost;          /* particles that left slab + halo in one step (must stay 0) */
    uint32_t overflow;      /* message capacity exceeded, or owned particles stranded past the main slots
                               (n_live > capacity - 2*recv_capacity at the next pack); must stay 0 */
    uint32_t far_halo;      /* migrants that landed in the receiver's far halo zone, or (overlapped step) within 2 columns
                               of its interior: the boundary zone was too narrow for their speed (must stay 0) */
} fs_slab_counters;

fs_status fs_slab_create(const fs_settings* global_settings, int device, const fs_slab_config* cfg, fs_sim** out);
/* Initial owned particles (host AoS records, any order). */
fs_status fs_slab_upload_owned(fs_sim* sim, const fs_particle* src, size_t n);
/* Move the owned window (re-balancing); takes effect at the next fs_slab_pack.  Until then fs_slab_download and
 * fs_slab_column_histogram still describe the stored state in the window it was built with. */
fs_status fs_slab_set_window(fs_sim* sim, uint32_t own_lo, uint32_t own_hi);
size_t fs_slab_message_bytes(const fs_sim* sim);
/* Begin a step: predict, classify, fill the two outgoing device messages (NULL = no neighbour).  Three step modes
 * (fs_slab_overlapped(): 1 edge-first — the default with the counting sort —, 2 strips, 0 serial; fs_slab_config.sort_mode
 * flags, FS_SLAB_MODE in the environment), one call pattern: fs_slab_pack -> exchange -> fs_slab_step.
 *   EDGE-FIRST (1): fs_slab_step forks after its reorder pass — the handle's exchange stream (fs_slab_comm_stream) advances
 *     the owned columns within the boundary zone of a neighboured edge first and builds the NEXT tick's two messages from
 *     their new state right away, into the buffers the last fs_slab_pack was given; the interior columns' density / force
 *     launches run beside that on the simulation's stream.  The next fs_slab_pack finds the messages built (same buffers,
 *     same delta, same window: otherwise it builds them itself) and only classifies the interior columns' slots for the sort.
 *     Keep the send buffers of a step untouched until the next fs_slab_pack returns, and exchange on the exchange stream
 *     (fs_slab_exchange does; other transports: fs_slab_comm_begin / _end).
 *   STRIPS (2): ghosts never enter the rank's sorted array; fs_slab_pack also enqueues sort, reorder, density and the interior
 *     columns' force launch; fs_slab_step finishes the boundary columns on a small second array after the exchange.
 *   SERIAL (0): pack, exchange and the whole step one after the other on the simulation's stream. */
fs_status fs_slab_pack(fs_sim* sim, const fs_tick_settings* tick, void* send_left, void* send_right);
/* Finish the step with the two incoming device messages (NULL = no neighbour). */
fs_status fs_slab_step(fs_sim* sim, const void* recv_left, const void* recv_right);
/* Step mode (see fs_slab_pack); owned columns per neighboured edge in the boundary zone (default 4, at least 3; set it to
 * 3 + the columns the fastest particle can cross in one step — a migrant that lands closer than 3 columns to the interior
 * is counted in fs_slab_counters.far_halo; a window edge moved by fs_slab_set_window widens the next step's zone by itself). */
int fs_slab_overlapped(const fs_sim* sim);
fs_status fs_slab_set_boundary_cols(fs_sim* sim, uint32_t cols);
uint32_t fs_slab_boundary_cols(const fs_sim* sim);
/* Transport hooks of the overlapped step (no-ops on a serial handle, where the simulation's own stream orders the exchange):
 * fs_slab_comm_begin makes fs_slab_comm_stream() wait for the packed messages; the caller issues its send / recv on THAT
 * stream; fs_slab_comm_end records their completion, which the next fs_slab_step waits for.  fs_slab_exchange does all three.
 * fs_slab_wait_packed blocks the host until the outgoing messages are complete (transports that stage through the host). */
void* fs_slab_comm_stream(const fs_sim* sim);
fs_status fs_slab_comm_begin(fs_sim* sim);
fs_status fs_slab_comm_end(fs_sim* sim);
fs_status fs_slab_wait_packed(fs_sim* sim);
fs_status fs_slab_counters_read(fs_sim* sim, fs_slab_counters* out);   /* blocking */
/* Live records (global cell keys) and their owned flags; blocking.  Returns n_live. */
fs_status fs_slab_download(fs_sim* sim, fs_particle* dst, uint8_t* owned, size_t cap, uint32_t* n_live);
/* Largest |velocity| among the owned particles (sizes the outer-edge margin between two re-balancing
 * steps: a wall-side slab must contain everything that can move before the next one); blocking. */
fs_status fs_slab_max_speed(fs_sim* sim, float* out);
/* Per-global-column particle counts of the owned columns (others untouched); blocking. */
fs_status fs_slab_column_histogram(fs_sim* sim, uint32_t* hist, size_t grid_w_global);

/* The three re-balancing inputs above without reading anything back: enqueued on the simulation's stream, results stay in
 * DEVICE buffers of the caller — hist_dev[grid_w_global] (zero outside the owned window) and stats_dev[4] = {lost,
 * overflow, far_halo, bits of the largest owned |velocity|}, all four MAX-reducible as u32.  All-reduce them on the
 * same stream (fs_comm_allreduce: SUM for the histogram, MAX for the stats) and read both once. */
fs_status fs_slab_rebalance_stats(fs_sim* sim, uint32_t* stats_dev, uint32_t* hist_dev, size_t grid_w_global);

/* ---- native RCCL transport (csrc/comm.hip): one process per GPU, any host language --------------------------
 * fs_comm_unique_id on rank 0 -> ship the 128 bytes to every rank -> fs_comm_init everywhere; then per step
 * fs_slab_pack -> fs_slab_exchange -> fs_slab_step.  The exchange is one grouped ncclSend/ncclRecv set on the
 * simulation's stream (left_rank / right_rank < 0 = no neighbour on that side); with left_rank == right_rank ==
 * own rank it is a self-exchange (recv_right receives send_right, recv_left receives send_left).  RCCL failures
 * return FS_ERR_COMM with the RCCL message in fs_last_error().  fs_comm_allreduce (in place, on the simulation's
 * stream) serves the re-balancing histogram / violation counters. */
#define FS_COMM_ID_BYTES 128
typedef struct fs_comm fs_comm;
enum { FS_COMM_U32 = 0, FS_COMM_U64 = 1, FS_COMM_F32 = 2 };
enum { FS_COMM_SUM = 0, FS_COMM_MAX = 1 };
fs_status fs_comm_unique_id(uint8_t id[FS_COMM_ID_BYTES]);
fs_status fs_comm_init(int device, int rank, int world, const uint8_t id[FS_COMM_ID_BYTES], fs_comm** out);
void fs_comm_destroy(fs_comm* comm);
fs_status fs_slab_exchange(fs_sim* sim, fs_comm* comm, int left_rank, int right_rank, const void* send_left,
                           const void* send_right, void* recv_left, void* recv_right);
fs_status fs_comm_allreduce(fs_sim* sim, fs_comm* comm, void* device_buf, size_t count, int dtype, int op);

/* ------------------------------------------------------------ 3D extension */
/* NOT in the reference (2D only).  Build-defined per SURVEY.md Appendix B.3: same pass
 * chain and kernel shapes with a third coordinate, 27-cell sweep, clean cell starts, no
 * mouse force / obstacle field.  Normative statement: oracle/sph_oracle3d.cpp. */
typedef struct fs3_settings {
    uint32_t particle_count;      /* must be a cube (side^3) for the built-in lattice */
    float particle_spacing;
    float smoothing_radius;
    fs_vec3 size;
} fs3_settings;

typedef struct fs3_tick_settings {
    float delta;
    fs_vec3 gravity;
    float mass;
    float pressure_constant;
    float rest_density;
    float damping_factor;
    float viscosity_coefficient;
} fs3_tick_settings;

typedef struct fs3_particle {      /* 48 bytes */
    fs_vec3 position;
    fs_vec3 predicted_position;
    fs_vec3 velocity;
    float density;
    uint32_t grid;
    uint32_t pad;
} fs3_particle;

typedef struct fs_sim3 fs_sim3;

fs_status fs3_create(const fs3_settings* settings, int device, fs_vec3 initial_offset, fs_sim3** out);
/* math_mode: FS_MATH_IEEE (default of fs3_create: bit-identical to oracle/sph_oracle3d.cpp) or FS_MATH_TOLERANCE (density
 * and force terms re-associated as in 2D: rtol 1e-5 per step against the 3D oracle, cell keys bit-exact).  The 3D
 * statement has no reference counterpart, so "exact" here means exact against this repository's own 3D oracle. */
fs_status fs3_create_ex(const fs3_settings* settings, int device, fs_vec3 initial_offset, int math_mode, fs_sim3** out);
void fs3_destroy(fs_sim3* sim);
fs_status fs3_step(fs_sim3* sim, const fs3_tick_settings* tick);
fs_status fs3_sync(fs_sim3* sim);
uint32_t fs3_tick_count(const fs_sim3* sim);
uint32_t fs3_particle_count(const fs_sim3* sim);
fs_status fs3_grid_dims(const fs_sim3* sim, uint32_t* w, uint32_t* h, uint32_t* d);
fs_status fs3_download_particles(fs_sim3* sim, fs3_particle* dst, size_t n);
fs_status fs3_upload_particles(fs_sim3* sim, const fs3_particle* src, size_t n);
fs_status fs3_reference_lattice(const fs3_settings* settings, fs_vec3 offset, fs3_particle* dst, size_t n);
fs_status fs3_timed_steps(fs_sim3* sim, const fs3_tick_settings* tick, uint32_t steps, double* ms_total);
fs_status fs3_profile_enable(fs_sim3* sim, int enable);
fs_status fs3_profile_read(fs_sim3* sim, double ms[FS_PASS_COUNT], uint64_t* steps, int reset);

/* ------------------------------------------------------- ResizableBuffer */
/* ResizableBuffer<T>::new (src/buffer.rs:27-43). */
fs_status fs_buffer_create(int device, size_t elem_size, size_t len, const char* name, fs_buffer** out);
/* ::resize (src/buffer.rs:46-67): grow-only, copies old contents, clamps to the
 * device maximum with a warning.  *resized = 0 when new_cap < len. */
fs_status fs_buffer_resize(fs_buffer* buf, size_t new_cap, int* resized);
/* ::write (src/buffer.rs:70-87): data longer than the buffer is trimmed (and
 * logged); offset-aware superset of the reference check. */
fs_status fs_buffer_write(fs_buffer* buf, size_t offset, const void* data, size_t count);
fs_status fs_buffer_read(fs_buffer* buf, size_t offset, void* dst, size_t count);
size_t fs_buffer_len(const fs_buffer* buf);      /* SSBO::len (src/buffer.rs:170-172) */
void* fs_buffer_device_ptr(const fs_buffer* buf); /* bind_group() equivalent (src/buffer.rs:162-164) */
void fs_buffer_destroy(fs_buffer* buf);

/* ------------------------------------------------------------- self-tests */
/* Bit-exact parity makes the force pass divide-bound; three identities let it keep IEEE results
 * with far fewer instructions, and each is PROVEN by enumeration on the GPU before it is used:
 *  - division by the two loop-invariant constants (2h^3, h^2; funcs.wgsl:119) as a 3-instruction
 *    form: every f32 numerator the kernel can feed it (2^-60 <= |x| <= c, both signs) is checked
 *    against `/` for the handle's constants when the handle is created;
 *  - RN(1/b) on [2^-20, 2^20] as v_rcp_f32 + one Newton step and RN(sqrt(x)) on [2^-40, 2^40] as
 *    v_sqrt_f32 + one Newton/Markstein correction: every f32 of both ranges is checked at create;
 *  - a/b from y = RN(1/b) as q0 = a*y, q = fma(fma(-q0, b, a), y, q0): holds for all 2^46 mantissa
 *    pairs (tools/div_markstein.hip, profiles/r01_f_div_by_rcp_exhaustive.txt), used inside range
 *    guards; operands outside the guards take the true division.
 * fs_selftest_constdiv exposes the first enumeration (mismatch count for constant c, reciprocal y,
 * range [lo, hi]); fs_constdiv_status returns bit 0 / 1 = proof succeeded for 2h^3 / h^2, bit 2 / 3
 * = for the lean reciprocal / square root, bit 4 = for the cell size h of the cell-coordinate quotients (funcs.wgsl:212-214,
 * numerators 2^-60 .. 4 x the larger bound) (31 = everything in use). */
fs_status fs_selftest_constdiv(int device, float c, float y, float lo, float hi, uint32_t* mismatches);
int fs_constdiv_status(const fs_sim* sim);
/* The engine's sort (the reference network of sort.wgsl:27-51 on (key << 32 | index) pairs) run on `n` caller-supplied
 * pairs, host memory, in place.  `fuse_stage` < 0: the engine's default late-stage plan; 0: per-stage launches only;
 * k: the shifted merge from stage k (kernels_sort.hip); k | 0x100: the same with the single stand-by launch instead of
 * the per-stage ones.  plan[0] / plan[1] (may be NULL): how often the device-side
 * certificate chose the shifted merge / the per-stage plan for this call (0, 0 when no plan was in play).  Blocking. */
fs_status fs_selftest_sort(int device, uint64_t* pairs, uint32_t n, int fuse_stage, uint32_t plan[2]);
/* The host policy that picks the sort's late-stage plan (csrc/sort_policy.h), replayed on the CPU against a model of the
 * flow: `required[i]` is the lowest stage whose window the moves of step i fit (a stage more doubles the window); the
 * certificate of step i, run at the stage the policy chose, passes iff stage >= required[i], fit class min(3, stage -
 * required[i]); its report reaches the policy `lag` steps later (the engine keeps at most 4 steps in flight).
 * Outputs per step: the stage chosen, 1 where the single stand-by launch was in the stream.  No device is touched. */
fs_status fs_selftest_sort_policy(uint32_t log2_count, int start_back, uint32_t lag, const uint32_t* required, size_t steps,
                                  uint32_t* stage_out, uint32_t* single_out);

/* Diagnostics of the sort's late-stage plan (csrc/kernels_sort.hip): the network's last stages run as one shifted
 * merge when a device-side certificate allows it, as per-stage launches otherwise.  Counts since create.  Blocking. */
typedef struct fs_sort_plan_info {
    uint32_t shifted;        /* sorts that took the shifted merge */
    uint32_t per_stage;      /* sorts whose certificate failed (per-stage plan: separate launches or the stand-by kernel) */
    uint32_t standby_runs;   /* of those, done by the single stand-by kernel */
    uint32_t stage;          /* first stage the shifted merge currently replaces (0: engine default) */
    uint32_t standby_single; /* 1: the single stand-by launch is in use instead of the per-stage launches */
    uint32_t timeouts;       /* stand-by grid-barrier time-outs (0 on a healthy device) */
    uint32_t wide_tiles;     /* 4096-element tiles whose keys spanned >= 2^20 - 1 cells (an uploaded, unordered state on a large grid):
                                left by the packed first kernel to the 64-bit one */
} fs_sort_plan_info;
fs_status fs_sort_plan_read(fs_sim* sim, fs_sort_plan_info* out);

/* ----------------------------------------------------------------- errors */
const char* fs_last_error(void);  /* thread-local, never NULL */
int fs_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FLUIDSIM_H */

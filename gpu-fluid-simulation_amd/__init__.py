"""gpu-fluid-simulation_amd — MI355X-native SPH fluid-step engine (hot path of
rookieCookies/gpu-fluid-simulation) behind the C ABI in include/fluidsim.h.

This module is the thin Python host mirror used by tests and bench.py; it only
marshals arguments into libfluidsim_hip.so (hand-written HIP kernels).  There is
no CPU fallback: without the built extension (or without a GPU) calls raise.

Mirrors, by name and argument meaning:
  FluidSimulation.new / .tick / .tick_count   src/simulation.rs:139,459,12
  SimulationSettings / TickSettings           src/simulation.rs:95-122
  ResizableBuffer                             src/buffer.rs:17-88
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import (  # noqa: F401
    FS_MATH_IEEE,
    FS_MATH_TOLERANCE,
    FS_MATH_WGSL_ULP,
    FS_SORT_BITONIC,
    FS_SORT_COUNTING,
    FS_SLAB_SERIAL,
    FS_SLAB_STRIPS,
    PARTICLE3_DTYPE,
    PARTICLE_DTYPE,
    PASS_NAMES,
    ExtensionMissing,
    Options,
    Settings,
    Settings3,
    SlabConfig,
    SlabCounters,
    SortStep,
    TickSettings,
    TickSettings3,
    Uniform,
    UVec2,
    Vec2,
    Vec3,
    load_library,
)

__all__ = [
    "FluidSimulation", "ResizableBuffer", "SimulationSettings", "default_tick_settings", "dam_break_2d",
    "FluidSimError", "load_library", "PARTICLE_DTYPE",
]


class FluidSimError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"fluidsim status {status}: {message}")
        self.status = status


def _check(lib, status):
    if status != _abi.FS_OK:
        raise FluidSimError(status, lib.fs_last_error().decode("utf-8", "replace"))


def SimulationSettings(particle_count=100_000, particle_spacing=0.1, smoothing_radius=0.2, size=(53.0, 53.0),
                       texture_size=(1024, 1024)):
    """Defaults: src/main.rs:48-54, src/renderer.rs:16."""
    return Settings(int(particle_count), float(particle_spacing), float(smoothing_radius),
                    Vec2(float(size[0]), float(size[1])), UVec2(int(texture_size[0]), int(texture_size[1])))


def default_tick_settings(**over):
    """Defaults: src/renderer.rs:374-388."""
    t = TickSettings(
        delta=np.float32(1.0) / np.float32(120.0), gravity=Vec2(0.0, 0.0), mass=1.0, pressure_constant=50.0,
        rest_density=0.0, damping_factor=0.1, viscosity_coefficient=25.0, surface_tension_treshold=0.1,
        surface_tension_coefficient=35.0, mouse_force_radius=5.0, mouse_force_power=150.0,
        mouse_pos=Vec2(0.0, 0.0), mouse_state=0)
    for k, v in over.items():
        if k in ("gravity", "mouse_pos"):
            v = Vec2(float(v[0]), float(v[1]))
        setattr(t, k, v)
    return t


def dam_break_2d(n):
    """The benchmark scene of SURVEY.md §8d: reference lattice + defaults, gravity on.

    Returns (settings, initial_offset, tick_settings).  All scene arithmetic in f32.
    """
    f = np.float32
    s, h = f(0.1), f(0.2)
    L = np.sqrt(f(n)) * s          # block side; exact for the square counts of the benchmark configs
    size = (f(2.0) * L, f(1.25) * L)
    off = (-size[0] / f(2) + L / f(2) + s / f(2), size[1] / f(2) - L / f(2) - s / f(2))
    settings = SimulationSettings(n, s, h, size, (1024, 1024))
    tick = default_tick_settings(gravity=(0.0, 9.81))
    return settings, (float(off[0]), float(off[1])), tick


def selftest_sort(keys, fuse_stage=-1, device=0):
    """The engine's sort (reference network, sort.wgsl:27-51) on bare u32 keys: (sorted_keys, perm, plan).

    plan = (calls that took the shifted late-stage merge, calls that took the per-stage plan); see
    csrc/kernels_sort.hip k_late_cert.  fuse_stage: -1 default plan, 0 per-stage only, k shifted merge from stage k.
    """
    lib = load_library()
    k = np.ascontiguousarray(keys, dtype=np.uint32)
    pairs = (k.astype(np.uint64) << np.uint64(32)) | np.arange(k.shape[0], dtype=np.uint64)
    plan = (C.c_uint32 * 2)()
    _check(lib, lib.fs_selftest_sort(int(device), pairs.ctypes.data_as(C.c_void_p), k.shape[0], int(fuse_stage), plan))
    return (pairs >> np.uint64(32)).astype(np.uint32), (pairs & np.uint64(0xFFFFFFFF)).astype(np.uint32), (plan[0], plan[1])


def selftest_sort_policy(required, log2_count=24, start_back=8, lag=4):
    """Replay the sort-plan policy (csrc/sort_policy.h) on the CPU against a per-step `required` stage; returns
    (stage chosen per step, single stand-by launch in the stream per step).  See fs_selftest_sort_policy."""
    lib = load_library()
    req = np.ascontiguousarray(required, dtype=np.uint32)
    stage = np.zeros(req.shape[0], dtype=np.uint32)
    single = np.zeros(req.shape[0], dtype=np.uint32)
    U = C.POINTER(C.c_uint32)
    _check(lib, lib.fs_selftest_sort_policy(int(log2_count), int(start_back), int(lag), req.ctypes.data_as(U), req.shape[0],
                                            stage.ctypes.data_as(U), single.ctypes.data_as(U)))
    return stage, single


class FluidSimulation:
    """FluidSimulation (src/simulation.rs:10-37) on one MI355X, driven through the C ABI."""

    def __init__(self, settings, device=0, sort_mode=FS_SORT_BITONIC, ref_quirks=True, initial_offset=(0.0, 0.0),
                 capacity=0, math_mode=FS_MATH_IEEE):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.settings = settings
        opts = Options()
        self._lib.fs_options_default(C.byref(opts))
        opts.device = int(device)
        opts.sort_mode = int(sort_mode)
        opts.ref_quirks = 1 if ref_quirks else 0
        opts.math_mode = int(math_mode)
        opts.initial_offset = Vec2(float(initial_offset[0]), float(initial_offset[1]))
        opts.capacity = int(capacity)
        _check(self._lib, self._lib.fs_create_ex(C.byref(settings), C.byref(opts), C.byref(self._h)))

    @classmethod
    def new(cls, settings, device=0, **kw):
        return cls(settings, device=device, **kw)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.fs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- stepping -----------------------------------------------------------
    def tick(self, tick_settings):
        _check(self._lib, self._lib.fs_step(self._h, C.byref(tick_settings)))

    def sync(self):
        _check(self._lib, self._lib.fs_sync(self._h))

    @property
    def tick_count(self):
        return int(self._lib.fs_tick_count(self._h))

    @property
    def particle_count(self):
        return int(self._lib.fs_particle_count(self._h))

    @property
    def grid_dims(self):
        w, h = C.c_uint32(), C.c_uint32()
        _check(self._lib, self._lib.fs_grid_dims(self._h, C.byref(w), C.byref(h)))
        return int(w.value), int(h.value)

    def timed_steps(self, tick_settings, steps):
        ms = C.c_double()
        _check(self._lib, self._lib.fs_timed_steps(self._h, C.byref(tick_settings), int(steps), C.byref(ms)))
        return float(ms.value)

    def profile(self, enable=True):
        _check(self._lib, self._lib.fs_profile_enable(self._h, 1 if enable else 0))

    def profile_read(self, reset=True):
        ms = (C.c_double * len(PASS_NAMES))()
        steps = C.c_uint64()
        _check(self._lib, self._lib.fs_profile_read(self._h, ms, C.byref(steps), 1 if reset else 0))
        return dict(zip(PASS_NAMES, [float(x) for x in ms])), int(steps.value)

    def sort_plan(self):
        """Diagnostics of the sort's late-stage plan (fs_sort_plan_read): dict of counts since create.  Blocking."""
        info = _abi.SortPlanInfo()
        _check(self._lib, self._lib.fs_sort_plan_read(self._h, C.byref(info)))
        return {k: int(getattr(info, k)) for k, _ in _abi.SortPlanInfo._fields_}

    # -- data ---------------------------------------------------------------
    def uniform(self):
        u = Uniform()
        _check(self._lib, self._lib.fs_get_uniform(self._h, C.byref(u)))
        return u

    def download_particles(self):
        out = np.empty(self.particle_count, dtype=PARTICLE_DTYPE)
        _check(self._lib, self._lib.fs_download_particles(self._h, out.ctypes.data_as(C.c_void_p), out.shape[0]))
        return out

    def upload_particles(self, arr):
        arr = np.ascontiguousarray(arr, dtype=PARTICLE_DTYPE)
        _check(self._lib, self._lib.fs_upload_particles(self._h, arr.ctypes.data_as(C.c_void_p), arr.shape[0]))

    def download_start_indices(self):
        w, h = self.grid_dims
        out = np.empty(w * h, dtype=np.uint32)
        _check(self._lib, self._lib.fs_download_start_indices(self._h, out.ctypes.data_as(C.c_void_p), out.shape[0]))
        return out

    def upload_start_indices(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.uint32)
        _check(self._lib, self._lib.fs_upload_start_indices(self._h, arr.ctypes.data_as(C.c_void_p), arr.shape[0]))

    def upload_force_field(self, field):
        field = np.ascontiguousarray(field, dtype=np.float32)
        h, w = field.shape[0], field.shape[1]
        _check(self._lib, self._lib.fs_upload_force_field(self._h, field.ctypes.data_as(C.c_void_p), w, h))

    def set_obstacle_image(self, image, want_field=False):
        """Obstacle mask (u8 [h, w], > 128 = obstacle) -> push-out field written into the simulation
        (generate_smooth_gradient_field + the renderer's write_buffer: src/main.rs:403-515, renderer.rs:497-502)."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        h, w = image.shape
        out = np.empty((h, w, 2), dtype=np.float32) if want_field else None
        _check(self._lib, self._lib.fs_generate_force_field(self._h, 0, image.ctypes.data_as(C.c_void_p), w, h,
                                                            out.ctypes.data_as(C.c_void_p) if want_field else None))
        return out

    def render_density(self, width, height, world_min=None, world_max=None):
        """Headless density-splat image (fluid_shader.wgsl:27-102): float32 [height, width, 4] RGBA.
        Default view = the reference camera: the whole domain, +y down (src/renderer.rs:558-561)."""
        sx, sy = self.settings.size.x, self.settings.size.y
        wmin = world_min if world_min is not None else (-sx / 2, -sy / 2)
        wmax = world_max if world_max is not None else (sx / 2, sy / 2)
        view = _abi.View(Vec2(float(wmin[0]), float(wmin[1])), Vec2(float(wmax[0]), float(wmax[1])), int(width), int(height))
        out = np.empty((int(height), int(width), 4), dtype=np.float32)
        _check(self._lib, self._lib.fs_render_density(self._h, C.byref(view), out.ctypes.data_as(C.c_void_p)))
        return out

    def export_handle(self, which=_abi.FS_EXPORT_PARTICLES):
        """fs_export_handle: interprocess handle of the AoS particle view (switches on the live view) or of start_indices."""
        h = _abi.MemHandle()
        _check(self._lib, self._lib.fs_export_handle(self._h, int(which), C.byref(h)))
        return h

    def particles_device_ptr(self):
        p = C.c_void_p()
        _check(self._lib, self._lib.fs_particles_device(self._h, C.byref(p)))
        return p.value

    def start_indices_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        _check(self._lib, self._lib.fs_start_indices_device(self._h, C.byref(p), C.byref(n)))
        return p.value, int(n.value)


def dam_break_3d(n):
    """3D benchmark scene (SURVEY.md §8d `dam_break_3d`; build-defined, no reference counterpart):
    side^3 cube lattice, s = 0.1, h = 0.2, box (2L, 1.25L, L + 2s), block one spacing off the -x wall
    and the +y floor, centred in z; reference tick defaults with gravity (0, 9.81, 0)."""
    f = np.float32
    side = int(round(n ** (1.0 / 3.0)))
    if side ** 3 != n:
        raise ValueError("dam_break_3d expects a cube particle count")
    s, h = f(0.1), f(0.2)
    L = f(side) * s
    size = (f(2.0) * L, f(1.25) * L, L + f(2.0) * s)
    off = (-size[0] / f(2) + L / f(2) + s / f(2), size[1] / f(2) - L / f(2) - s / f(2), f(0.0))
    st = Settings3(int(n), float(s), float(h), Vec3(float(size[0]), float(size[1]), float(size[2])))
    tick = TickSettings3(float(f(1.0) / f(120.0)), Vec3(0.0, 9.81, 0.0), 1.0, 50.0, 0.0, 0.1, 25.0)
    return st, (float(off[0]), float(off[1]), float(off[2])), tick


class FluidSimulation3D:
    """3D extension (include/fluidsim.h fs3_*); not in the reference."""

    def __init__(self, settings, device=0, initial_offset=(0.0, 0.0, 0.0), math_mode=FS_MATH_IEEE):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.settings = settings
        off = Vec3(*[float(x) for x in initial_offset])
        _check(self._lib, self._lib.fs3_create_ex(C.byref(settings), int(device), off, int(math_mode), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.fs3_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tick(self, t):
        _check(self._lib, self._lib.fs3_step(self._h, C.byref(t)))

    def sync(self):
        _check(self._lib, self._lib.fs3_sync(self._h))

    @property
    def tick_count(self):
        return int(self._lib.fs3_tick_count(self._h))

    @property
    def particle_count(self):
        return int(self._lib.fs3_particle_count(self._h))

    @property
    def grid_dims(self):
        w, h, d = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(self._lib, self._lib.fs3_grid_dims(self._h, C.byref(w), C.byref(h), C.byref(d)))
        return int(w.value), int(h.value), int(d.value)

    def download_particles(self):
        out = np.empty(self.particle_count, dtype=PARTICLE3_DTYPE)
        _check(self._lib, self._lib.fs3_download_particles(self._h, out.ctypes.data_as(C.c_void_p), out.shape[0]))
        return out

    def upload_particles(self, arr):
        arr = np.ascontiguousarray(arr, dtype=PARTICLE3_DTYPE)
        _check(self._lib, self._lib.fs3_upload_particles(self._h, arr.ctypes.data_as(C.c_void_p), arr.shape[0]))

    def timed_steps(self, t, steps):
        ms = C.c_double()
        _check(self._lib, self._lib.fs3_timed_steps(self._h, C.byref(t), int(steps), C.byref(ms)))
        return float(ms.value)

    def profile(self, enable=True):
        _check(self._lib, self._lib.fs3_profile_enable(self._h, 1 if enable else 0))

    def profile_read(self, reset=True):
        ms = (C.c_double * len(PASS_NAMES))()
        steps = C.c_uint64()
        _check(self._lib, self._lib.fs3_profile_read(self._h, ms, C.byref(steps), 1 if reset else 0))
        return dict(zip(PASS_NAMES, [float(x) for x in ms])), int(steps.value)


def reference_lattice_3d(settings, offset=(0.0, 0.0, 0.0)):
    lib = load_library()
    out = np.zeros(settings.particle_count, dtype=PARTICLE3_DTYPE)
    _check(lib, lib.fs3_reference_lattice(C.byref(settings), Vec3(*[float(x) for x in offset]),
                                          out.ctypes.data_as(C.c_void_p), out.shape[0]))
    return out


class SlabSimulation:
    """One rank of the multi-GPU slab decomposition (include/fluidsim.h fs_slab_*).

    Owns the global cell columns [own_lo, own_hi).  A step is pack() -> exchange the two
    messages with the slab neighbours -> step(); see multi.py for the driver.
    """

    def __init__(self, settings, own_lo, own_hi, has_left, has_right, capacity, recv_capacity, max_cols, device=0,
                 sort_mode=None, serial=False, strips=False):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.settings = settings
        mode = (0 if sort_mode is None else 1 + int(sort_mode)) | (FS_SLAB_SERIAL if serial else 0) | (FS_SLAB_STRIPS if strips else 0)
        self.cfg = SlabConfig(int(own_lo), int(own_hi), int(bool(has_left)), int(bool(has_right)), int(capacity),
                              int(recv_capacity), int(max_cols), mode)
        _check(self._lib, self._lib.fs_slab_create(C.byref(settings), int(device), C.byref(self.cfg), C.byref(self._h)))
        self.capacity = int(capacity)
        self.device_index = int(device)
        self.message_bytes = int(self._lib.fs_slab_message_bytes(self._h))
        # 0 serial step; 1 edge-first (default): step() advances the edge columns, builds the NEXT step's messages and only
        # then advances the interior — the exchange runs beside that on comm_stream_ptr; 2 strips: pack() also enqueues the
        # interior columns' whole step, step() the boundary strips (include/fluidsim.h)
        self.step_mode = int(self._lib.fs_slab_overlapped(self._h))
        self.overlapped = self.step_mode != 0

    def set_boundary_cols(self, cols):
        _check(self._lib, self._lib.fs_slab_set_boundary_cols(self._h, int(cols)))

    @property
    def boundary_cols(self):
        return int(self._lib.fs_slab_boundary_cols(self._h))

    @property
    def comm_stream_ptr(self):
        return self._lib.fs_slab_comm_stream(self._h)

    def comm_begin(self):
        _check(self._lib, self._lib.fs_slab_comm_begin(self._h))

    def comm_end(self):
        _check(self._lib, self._lib.fs_slab_comm_end(self._h))

    def wait_packed(self):
        """Block the host until the outgoing messages of the current pack() are complete."""
        _check(self._lib, self._lib.fs_slab_wait_packed(self._h))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.fs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload_owned(self, arr):
        arr = np.ascontiguousarray(arr, dtype=PARTICLE_DTYPE)
        _check(self._lib, self._lib.fs_slab_upload_owned(self._h, arr.ctypes.data_as(C.c_void_p), arr.shape[0]))

    def set_window(self, own_lo, own_hi):
        _check(self._lib, self._lib.fs_slab_set_window(self._h, int(own_lo), int(own_hi)))
        self.cfg.own_lo, self.cfg.own_hi = int(own_lo), int(own_hi)

    def pack(self, tick_settings, send_left_ptr, send_right_ptr):
        _check(self._lib, self._lib.fs_slab_pack(self._h, C.byref(tick_settings), send_left_ptr, send_right_ptr))

    def step(self, recv_left_ptr, recv_right_ptr):
        _check(self._lib, self._lib.fs_slab_step(self._h, recv_left_ptr, recv_right_ptr))

    def sync(self):
        _check(self._lib, self._lib.fs_sync(self._h))

    @property
    def stream_ptr(self):
        return self._lib.fs_stream(self._h)

    @property
    def tick_count(self):
        return int(self._lib.fs_tick_count(self._h))

    def counters(self):
        c = SlabCounters()
        _check(self._lib, self._lib.fs_slab_counters_read(self._h, C.byref(c)))
        return {"n_live": c.n_live, "lost": c.lost, "overflow": c.overflow, "far_halo": c.far_halo}

    def download(self):
        """(records with GLOBAL cell keys, owned mask) of the live slots."""
        out = np.empty(self.capacity, dtype=PARTICLE_DTYPE)
        owned = np.zeros(self.capacity, dtype=np.uint8)
        n = C.c_uint32()
        _check(self._lib, self._lib.fs_slab_download(self._h, out.ctypes.data_as(C.c_void_p),
                                                     owned.ctypes.data_as(C.c_void_p), self.capacity, C.byref(n)))
        return out[: n.value], owned[: n.value].astype(bool)

    def column_histogram(self, grid_w_global):
        h = np.zeros(int(grid_w_global), dtype=np.uint32)
        _check(self._lib, self._lib.fs_slab_column_histogram(self._h, h.ctypes.data_as(C.c_void_p), h.shape[0]))
        return h

    def max_speed(self):
        v = C.c_float()
        _check(self._lib, self._lib.fs_slab_max_speed(self._h, C.byref(v)))
        return float(v.value)

    def rebalance_stats(self, stats_dev_ptr, hist_dev_ptr, grid_w_global):
        """Enqueue the re-balancing inputs into two DEVICE buffers (4 x u32 stats, grid_w x u32 histogram); no read-back."""
        _check(self._lib, self._lib.fs_slab_rebalance_stats(self._h, stats_dev_ptr, hist_dev_ptr, int(grid_w_global)))

    def profile(self, enable=True):
        _check(self._lib, self._lib.fs_profile_enable(self._h, 1 if enable else 0))

    def profile_read(self, reset=True):
        ms = (C.c_double * len(PASS_NAMES))()
        steps = C.c_uint64()
        _check(self._lib, self._lib.fs_profile_read(self._h, ms, C.byref(steps), 1 if reset else 0))
        return dict(zip(PASS_NAMES, [float(x) for x in ms])), int(steps.value)


class ResizableBuffer:
    """ResizableBuffer<T> (src/buffer.rs:17-88) over HIP device memory."""

    def __init__(self, name, dtype, length, device=0):
        self._lib = load_library()
        self.dtype = np.dtype(dtype)
        self._h = C.c_void_p()
        st = self._lib.fs_buffer_create(int(device), self.dtype.itemsize, int(length), name.encode(), C.byref(self._h))
        _check(self._lib, st)

    def __len__(self):
        return int(self._lib.fs_buffer_len(self._h))

    def resize(self, new_cap):
        r = C.c_int()
        _check(self._lib, self._lib.fs_buffer_resize(self._h, int(new_cap), C.byref(r)))
        return bool(r.value)

    def write(self, offset, data):
        data = np.ascontiguousarray(data, dtype=self.dtype)
        _check(self._lib, self._lib.fs_buffer_write(self._h, int(offset), data.ctypes.data_as(C.c_void_p), data.shape[0]))

    def read(self, offset=0, count=None):
        count = len(self) - offset if count is None else count
        out = np.zeros(max(count, 0), dtype=self.dtype)
        _check(self._lib, self._lib.fs_buffer_read(self._h, int(offset), out.ctypes.data_as(C.c_void_p), out.shape[0]))
        return out

    @property
    def device_ptr(self):
        return self._lib.fs_buffer_device_ptr(self._h)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.fs_buffer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ImportedBuffer:
    """Consumer side of fs_export_handle in another process: maps the exported device range (fs_import_open)."""

    def __init__(self, handle_bytes, device=0):
        self._lib = load_library()
        self.handle = _abi.MemHandle.from_buffer_copy(handle_bytes)
        self._p = C.c_void_p()
        _check(self._lib, self._lib.fs_import_open(C.byref(self.handle), int(device), C.byref(self._p)))

    def read(self, dtype, count=None, offset=0):
        dtype = np.dtype(dtype)
        count = int(self.handle.bytes // dtype.itemsize) if count is None else int(count)
        out = np.empty(count, dtype=dtype)
        _check(self._lib, self._lib.fs_import_read(self._p, int(offset), out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def close(self):
        if self._p.value:
            _check(self._lib, self._lib.fs_import_close(self._p))
            self._p = C.c_void_p()


def generate_force_field(image, device=0):
    """Standalone generate_smooth_gradient_field (src/main.rs:403-515) on the GPU: u8 [h, w] -> f32 [h, w, 2]."""
    lib = load_library()
    image = np.ascontiguousarray(image, dtype=np.uint8)
    h, w = image.shape
    out = np.empty((h, w, 2), dtype=np.float32)
    _check(lib, lib.fs_generate_force_field(None, int(device), image.ctypes.data_as(C.c_void_p), w, h,
                                            out.ctypes.data_as(C.c_void_p)))
    return out


def write_png(path, rgba, background=(0.0, 0.0, 0.0)):
    """Minimal PNG writer (zlib only): composites straight-alpha RGBA f32 over `background`, 8-bit RGB."""
    import struct
    import zlib
    a = np.clip(rgba[..., 3:4], 0.0, 1.0)
    rgb = np.clip(rgba[..., :3], 0.0, 1.0) * a + np.asarray(background, dtype=np.float32) * (1.0 - a)
    img = (np.clip(rgb, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    h, w = img.shape[:2]
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def reference_lattice(settings, offset=(0.0, 0.0)):
    lib = load_library()
    out = np.zeros(settings.particle_count, dtype=PARTICLE_DTYPE)
    _check(lib, lib.fs_reference_lattice(C.byref(settings), Vec2(float(offset[0]), float(offset[1])),
                                         out.ctypes.data_as(C.c_void_p), out.shape[0]))
    return out


def sort_schedule(particle_count):
    lib = load_library()
    n = lib.fs_sort_schedule(int(particle_count), None, 0)
    arr = (SortStep * max(n, 1))()
    lib.fs_sort_schedule(int(particle_count), arr, n)
    return [(a.group_width, a.group_height, a.step_index, a.num_values) for a in arr[:n]]


def build_uniform(settings, tick_settings, tick_count):
    lib = load_library()
    u = Uniform()
    _check(lib, lib.fs_build_uniform(C.byref(settings), C.byref(tick_settings), int(tick_count), C.byref(u)))
    return u

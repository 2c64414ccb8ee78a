"""ctypes mirror of include/fluidsim.h (the C ABI) — struct layouts and prototypes.

The layouts follow the reference PODs: ParticleInstance (src/simulation.rs:126-135),
SimulationUniform (:53-90), SimulationSettings (:95-104), TickSettings (:107-122).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libfluidsim_hip.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)


class Vec2(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class UVec2(C.Structure):
    _fields_ = [("x", C.c_uint32), ("y", C.c_uint32)]


class Settings(C.Structure):
    _fields_ = [
        ("particle_count", C.c_uint32),
        ("particle_spacing", C.c_float),
        ("smoothing_radius", C.c_float),
        ("size", Vec2),
        ("texture_size", UVec2),
    ]


class TickSettings(C.Structure):
    _fields_ = [
        ("delta", C.c_float),
        ("gravity", Vec2),
        ("mass", C.c_float),
        ("pressure_constant", C.c_float),
        ("rest_density", C.c_float),
        ("damping_factor", C.c_float),
        ("viscosity_coefficient", C.c_float),
        ("surface_tension_treshold", C.c_float),
        ("surface_tension_coefficient", C.c_float),
        ("mouse_force_radius", C.c_float),
        ("mouse_force_power", C.c_float),
        ("mouse_pos", Vec2),
        ("mouse_state", C.c_int32),
    ]


class Uniform(C.Structure):
    _fields_ = [
        ("delta", C.c_float),
        ("particle_count", C.c_uint32),
        ("sqr_radius", C.c_float),
        ("frame_time", C.c_uint32),
        ("gravity", Vec2),
        ("bounds", Vec2),
        ("mouse_pos", Vec2),
        ("smoothing_radius", C.c_float),
        ("particle_mass", C.c_float),
        ("pressure_constant", C.c_float),
        ("rest_density", C.c_float),
        ("damping_factor", C.c_float),
        ("viscosity_coefficient", C.c_float),
        ("surface_tension_treshold", C.c_float),
        ("surface_tension_coefficient", C.c_float),
        ("poly6_kernel_volume", C.c_float),
        ("poly6_kernel_derivative", C.c_float),
        ("poly6_kernel_laplacian", C.c_float),
        ("spiky_kernel_derivative", C.c_float),
        ("viscosity_kernel", C.c_float),
        ("mouse_state", C.c_int32),
        ("mouse_force_radius", C.c_float),
        ("mouse_force_power", C.c_float),
        ("grid_w", C.c_uint32),
        ("grid_h", C.c_uint32),
        ("texture_size", Vec2),
    ]


class SortStep(C.Structure):
    _fields_ = [
        ("group_width", C.c_uint32),
        ("group_height", C.c_uint32),
        ("step_index", C.c_uint32),
        ("num_values", C.c_uint32),
    ]


class Options(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("sort_mode", C.c_int32),
        ("ref_quirks", C.c_int32),
        ("math_mode", C.c_int32),
        ("initial_offset", Vec2),
        ("capacity", C.c_uint32),
        ("reserved1", C.c_uint32),
    ]


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Settings3(C.Structure):
    _fields_ = [("particle_count", C.c_uint32), ("particle_spacing", C.c_float), ("smoothing_radius", C.c_float),
                ("size", Vec3)]


class TickSettings3(C.Structure):
    _fields_ = [("delta", C.c_float), ("gravity", Vec3), ("mass", C.c_float), ("pressure_constant", C.c_float),
                ("rest_density", C.c_float), ("damping_factor", C.c_float), ("viscosity_coefficient", C.c_float)]


PARTICLE3_DTYPE = np.dtype([("position", "<f4", (3,)), ("predicted_position", "<f4", (3,)), ("velocity", "<f4", (3,)),
                            ("density", "<f4"), ("grid", "<u4"), ("pad", "<u4")])
assert PARTICLE3_DTYPE.itemsize == 48


class View(C.Structure):
    _fields_ = [("world_min", Vec2), ("world_max", Vec2), ("width", C.c_uint32), ("height", C.c_uint32)]


class SlabConfig(C.Structure):
    _fields_ = [
        ("own_lo", C.c_uint32), ("own_hi", C.c_uint32),
        ("has_left", C.c_uint32), ("has_right", C.c_uint32),
        ("capacity", C.c_uint32), ("recv_capacity", C.c_uint32),
        ("max_cols", C.c_uint32), ("sort_mode", C.c_uint32),
    ]


class SlabCounters(C.Structure):
    _fields_ = [("n_live", C.c_uint32), ("lost", C.c_uint32), ("overflow", C.c_uint32), ("far_halo", C.c_uint32)]


# 32-byte AoS particle record as a numpy structured dtype (offsets 0/8/16/24/28).
PARTICLE_DTYPE = np.dtype(
    [
        ("position", "<f4", (2,)),
        ("predicted_position", "<f4", (2,)),
        ("velocity", "<f4", (2,)),
        ("density", "<f4"),
        ("grid", "<u4"),
    ]
)
assert PARTICLE_DTYPE.itemsize == 32
assert C.sizeof(Uniform) == 120
assert C.sizeof(Settings) == 28
assert C.sizeof(TickSettings) == 60

FS_OK = 0
FS_ERR_INVALID = 1
FS_ERR_DEVICE = 2
FS_ERR_OOM = 3
FS_ERR_UNSUPPORTED = 4
FS_ERR_COMM = 5

FS_SORT_BITONIC = 0
FS_SORT_COUNTING = 1
FS_MATH_IEEE = 0
FS_MATH_WGSL_ULP = 1
FS_MATH_TOLERANCE = 2

PASS_NAMES = ("predict_key", "sort", "reorder", "density", "force", "boundary")
FS_SLAB_SERIAL = 0x100
FS_SLAB_STRIPS = 0x200
FS_EXPORT_PARTICLES = 0
FS_EXPORT_START_INDICES = 1


class MemHandle(C.Structure):
    """fs_mem_handle (include/fluidsim.h): 80 bytes, safe to send to another process as raw bytes."""
    _fields_ = [("ipc", C.c_uint8 * 64), ("bytes", C.c_uint64), ("device", C.c_int32), ("dmabuf_fd", C.c_int32)]

class SortPlanInfo(C.Structure):
    """fs_sort_plan_info (include/fluidsim.h)."""
    _fields_ = [("shifted", C.c_uint32), ("per_stage", C.c_uint32), ("standby_runs", C.c_uint32), ("stage", C.c_uint32),
                ("standby_single", C.c_uint32), ("timeouts", C.c_uint32), ("wide_tiles", C.c_uint32)]


# name -> (restype, argtypes).  Every symbol include/fluidsim.h declares.
_P = C.c_void_p
PROTOTYPES = {
    "fs_create": (C.c_int, [C.POINTER(Settings), C.c_int, C.POINTER(_P)]),
    "fs_create_ex": (C.c_int, [C.POINTER(Settings), C.POINTER(Options), C.POINTER(_P)]),
    "fs_options_default": (None, [C.POINTER(Options)]),
    "fs_destroy": (None, [_P]),
    "fs_step": (C.c_int, [_P, C.POINTER(TickSettings)]),
    "fs_sync": (C.c_int, [_P]),
    "fs_tick_count": (C.c_uint32, [_P]),
    "fs_particle_count": (C.c_uint32, [_P]),
    "fs_grid_dims": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "fs_stream": (_P, [_P]),
    "fs_particles_device": (C.c_int, [_P, C.POINTER(_P)]),
    "fs_start_indices_device": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "fs_get_uniform": (C.c_int, [_P, C.POINTER(Uniform)]),
    "fs_upload_force_field": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32]),
    "fs_download_particles": (C.c_int, [_P, _P, C.c_size_t]),
    "fs_upload_particles": (C.c_int, [_P, _P, C.c_size_t]),
    "fs_download_start_indices": (C.c_int, [_P, _P, C.c_size_t]),
    "fs_upload_start_indices": (C.c_int, [_P, _P, C.c_size_t]),
    "fs_reference_lattice": (C.c_int, [C.POINTER(Settings), Vec2, _P, C.c_size_t]),
    "fs_sort_schedule": (C.c_size_t, [C.c_uint32, _P, C.c_size_t]),
    "fs_build_uniform": (C.c_int, [C.POINTER(Settings), C.POINTER(TickSettings), C.c_uint32, C.POINTER(Uniform)]),
    "fs_generate_force_field": (C.c_int, [_P, C.c_int, _P, C.c_uint32, C.c_uint32, _P]),
    "fs_render_density": (C.c_int, [_P, C.POINTER(View), _P]),
    "fs_profile_enable": (C.c_int, [_P, C.c_int]),
    "fs_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    "fs_timed_steps": (C.c_int, [_P, C.POINTER(TickSettings), C.c_uint32, C.POINTER(C.c_double)]),
    "fs_export_handle": (C.c_int, [_P, C.c_int, _P]),
    "fs_import_open": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "fs_import_read": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t]),
    "fs_import_close": (C.c_int, [_P]),
    "fs_comm_unique_id": (C.c_int, [_P]),
    "fs_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    "fs_comm_destroy": (None, [_P]),
    "fs_slab_exchange": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "fs_comm_allreduce": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int, C.c_int]),
    "fs_slab_create": (C.c_int, [C.POINTER(Settings), C.c_int, C.POINTER(SlabConfig), C.POINTER(_P)]),
    "fs_slab_upload_owned": (C.c_int, [_P, _P, C.c_size_t]),
    "fs_slab_set_window": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "fs_slab_message_bytes": (C.c_size_t, [_P]),
    "fs_slab_pack": (C.c_int, [_P, C.POINTER(TickSettings), _P, _P]),
    "fs_slab_step": (C.c_int, [_P, _P, _P]),
    "fs_slab_overlapped": (C.c_int, [_P]),
    "fs_slab_set_boundary_cols": (C.c_int, [_P, C.c_uint32]),
    "fs_slab_boundary_cols": (C.c_uint32, [_P]),
    "fs_slab_comm_stream": (_P, [_P]),
    "fs_slab_comm_begin": (C.c_int, [_P]),
    "fs_slab_comm_end": (C.c_int, [_P]),
    "fs_slab_wait_packed": (C.c_int, [_P]),
    "fs_slab_counters_read": (C.c_int, [_P, C.POINTER(SlabCounters)]),
    "fs_slab_download": (C.c_int, [_P, _P, _P, C.c_size_t, C.POINTER(C.c_uint32)]),
    "fs_slab_column_histogram": (C.c_int, [_P, _P, C.c_size_t]),
    "fs_slab_max_speed": (C.c_int, [_P, _P]),
    "fs_slab_rebalance_stats": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "fs3_create": (C.c_int, [C.POINTER(Settings3), C.c_int, Vec3, C.POINTER(_P)]),
    "fs3_create_ex": (C.c_int, [C.POINTER(Settings3), C.c_int, Vec3, C.c_int, C.POINTER(_P)]),
    "fs3_destroy": (None, [_P]),
    "fs3_step": (C.c_int, [_P, C.POINTER(TickSettings3)]),
    "fs3_sync": (C.c_int, [_P]),
    "fs3_tick_count": (C.c_uint32, [_P]),
    "fs3_particle_count": (C.c_uint32, [_P]),
    "fs3_grid_dims": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "fs3_download_particles": (C.c_int, [_P, _P, C.c_size_t]),
    "fs3_upload_particles": (C.c_int, [_P, _P, C.c_size_t]),
    "fs3_reference_lattice": (C.c_int, [C.POINTER(Settings3), Vec3, _P, C.c_size_t]),
    "fs3_timed_steps": (C.c_int, [_P, C.POINTER(TickSettings3), C.c_uint32, C.POINTER(C.c_double)]),
    "fs3_profile_enable": (C.c_int, [_P, C.c_int]),
    "fs3_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]),
    "fs_selftest_constdiv": (C.c_int, [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_uint32)]),
    "fs_sort_plan_read": (C.c_int, [C.c_void_p, C.POINTER(SortPlanInfo)]),
    "fs_selftest_sort_policy": (C.c_int, [C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_uint32), C.c_size_t,
                                          C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "fs_selftest_sort": (C.c_int, [C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]),
    "fs_constdiv_status": (C.c_int, [_P]),
    "fs_buffer_create": (C.c_int, [C.c_int, C.c_size_t, C.c_size_t, C.c_char_p, C.POINTER(_P)]),
    "fs_buffer_resize": (C.c_int, [_P, C.c_size_t, C.POINTER(C.c_int)]),
    "fs_buffer_write": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t]),
    "fs_buffer_read": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t]),
    "fs_buffer_len": (C.c_size_t, [_P]),
    "fs_buffer_device_ptr": (_P, [_P]),
    "fs_buffer_destroy": (None, [_P]),
    "fs_last_error": (C.c_char_p, []),
    "fs_abi_version": (C.c_int, []),
}

_lib = None


class ExtensionMissing(RuntimeError):
    """Raised when the HIP extension is not built; there is no CPU fallback."""


def load_library(path=None):
    """dlopen libfluidsim_hip.so and bind every prototype.  Fails loudly if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ExtensionMissing(
            f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
            "The product path has no CPU fallback."
        )
    lib = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.fs_abi_version() != 2:
        raise ExtensionMissing("ABI version mismatch between _abi.py and libfluidsim_hip.so")
    if path is None:
        _lib = lib
    return lib

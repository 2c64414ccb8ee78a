"""ctypes binding of the CPU oracle (oracle/sph_oracle.cpp).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libsph_oracle.so")

PARTICLE_DTYPE = np.dtype([("position", "<f4", (2,)), ("predicted_position", "<f4", (2,)),
                           ("velocity", "<f4", (2,)), ("density", "<f4"), ("grid", "<u4")])


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in ("sph_oracle.cpp", "sph_oracle3d.cpp", "Makefile")]
    srcs.append(os.path.join(HERE, "..", "include", "fluidsim.h"))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        P = C.c_void_p
        L.orc_create.restype = P
        L.orc_create.argtypes = [P, C.c_float, C.c_float, C.c_int]
        L.orc_destroy.argtypes = [P]
        for name in ("orc_predict", "orc_spatial_lookup", "orc_sort", "orc_cell_starts", "orc_move"):
            getattr(L, name).argtypes = [P]
            getattr(L, name).restype = None
        L.orc_begin_tick.argtypes = [P, P]
        L.orc_density.argtypes = [P, C.c_int]
        L.orc_step.argtypes = [P, P]
        L.orc_step_stable.argtypes = [P, P]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_set_threads(1)                      # the oracle is a scalar port unless a caller asks for more
        L.orc_gradient_field.argtypes = [P, C.c_uint32, C.c_uint32, P]
        L.orc_render.argtypes = [P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, P]
        L.orc_tick.restype = C.c_uint32
        L.orc_tick.argtypes = [P]
        L.orc_count.restype = C.c_uint32
        L.orc_count.argtypes = [P]
        L.orc_grid.argtypes = [P, P, P]
        L.orc_particles.restype = P
        L.orc_particles.argtypes = [P]
        L.orc_start_indices.restype = P
        L.orc_start_indices.argtypes = [P]
        L.orc_start_indices_len.restype = C.c_size_t
        L.orc_start_indices_len.argtypes = [P]
        L.orc_texture.restype = P
        L.orc_texture.argtypes = [P]
        L.orc_uniform.argtypes = [P, P]
        L.orc_poly6_norm.restype = C.c_float
        L.orc_poly6_norm.argtypes = [P]
        L.orc_lattice.argtypes = [P, C.c_float, C.c_float, P, C.c_size_t]
        L.orc_sort_schedule.restype = C.c_size_t
        L.orc_sort_schedule.argtypes = [C.c_uint32, P, C.c_size_t]
        L.orc_build_uniform.argtypes = [P, P, C.c_uint32, P]
        L.orc_grid_dims.argtypes = [P, P, P]
        L.orc_bitonic_keys.argtypes = [P, P, C.c_uint32]
        L.orc_poly6_value.restype = C.c_float
        L.orc_poly6_value.argtypes = [C.c_float, C.c_float]
        L.orc3_create.restype = P
        L.orc3_create.argtypes = [P, C.c_float, C.c_float, C.c_float]
        L.orc3_destroy.argtypes = [P]
        L.orc3_step.argtypes = [P, P]
        L.orc3_particles.restype = P
        L.orc3_particles.argtypes = [P]
        L.orc3_count.restype = C.c_uint32
        L.orc3_count.argtypes = [P]
        L.orc3_grid.argtypes = [P, P, P, P]
        L.orc3_constants.argtypes = [P, P]
        L.orc3_lattice.argtypes = [P, C.c_float, C.c_float, C.c_float, P, C.c_size_t]
        _lib = L
    return _lib


PARTICLE3_DTYPE = np.dtype([("position", "<f4", (3,)), ("predicted_position", "<f4", (3,)), ("velocity", "<f4", (3,)),
                            ("density", "<f4"), ("grid", "<u4"), ("pad", "<u4")])


class OracleSim3D:
    """3D oracle (oracle/sph_oracle3d.cpp); no reference counterpart."""

    def __init__(self, settings, initial_offset=(0.0, 0.0, 0.0)):
        self.L = lib()
        self.h = self.L.orc3_create(C.addressof(settings), *[float(x) for x in initial_offset])
        if not self.h:
            raise ValueError("oracle3d: invalid settings")
        self.n = int(self.L.orc3_count(self.h))

    def close(self):
        if self.h:
            self.L.orc3_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, tick):
        self.L.orc3_step(self.h, C.addressof(tick))

    def particles_view(self):
        buf = (C.c_char * (self.n * 48)).from_address(self.L.orc3_particles(self.h))
        return np.frombuffer(buf, dtype=PARTICLE3_DTYPE)

    def particles(self):
        return self.particles_view().copy()

    def set_particles(self, arr):
        self.particles_view()[:] = np.ascontiguousarray(arr, dtype=PARTICLE3_DTYPE)

    @property
    def grid_dims(self):
        w, h, d = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.L.orc3_grid(self.h, C.addressof(w), C.addressof(h), C.addressof(d))
        return int(w.value), int(h.value), int(d.value)

    def constants(self):
        out = (C.c_float * 3)()
        self.L.orc3_constants(self.h, C.addressof(out))
        return tuple(out)


class OracleSim:
    """CPU oracle simulation; `settings`/`tick` are the ctypes structs of the product ABI
    (same layout as include/fluidsim.h)."""

    def __init__(self, settings, initial_offset=(0.0, 0.0), ref_quirks=True):
        self.L = lib()
        self.settings = settings
        self.h = self.L.orc_create(C.addressof(settings), float(initial_offset[0]), float(initial_offset[1]),
                                   1 if ref_quirks else 0)
        if not self.h:
            raise ValueError("oracle: invalid settings (particle_count <= 1)")
        self.n = int(self.L.orc_count(self.h))

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, tick, stable_sort=False):
        (self.L.orc_step_stable if stable_sort else self.L.orc_step)(self.h, C.addressof(tick))

    # individual passes (reference dispatch order, src/simulation.rs:512-537)
    def begin_tick(self, tick): self.L.orc_begin_tick(self.h, C.addressof(tick))
    def predict(self): self.L.orc_predict(self.h)
    def spatial_lookup(self): self.L.orc_spatial_lookup(self.h)
    def sort(self): self.L.orc_sort(self.h)
    def cell_starts(self): self.L.orc_cell_starts(self.h)
    def density(self, reach=1): self.L.orc_density(self.h, int(reach))
    def move(self): self.L.orc_move(self.h)

    def render(self, width, height, world_min, world_max):
        out = np.empty((height, width, 4), dtype=np.float32)
        self.L.orc_render(self.h, float(world_min[0]), float(world_min[1]), float(world_max[0]), float(world_max[1]),
                          int(width), int(height), out.ctypes.data)
        return out

    @property
    def tick_count(self):
        return int(self.L.orc_tick(self.h))

    @property
    def grid_dims(self):
        w, h = C.c_uint32(), C.c_uint32()
        self.L.orc_grid(self.h, C.addressof(w), C.addressof(h))
        return int(w.value), int(h.value)

    def particles_view(self):
        """Writable numpy view of the oracle's particle array (no copy)."""
        ptr = self.L.orc_particles(self.h)
        buf = (C.c_char * (self.n * 32)).from_address(ptr)
        return np.frombuffer(buf, dtype=PARTICLE_DTYPE)

    def particles(self):
        return self.particles_view().copy()

    def set_particles(self, arr):
        self.particles_view()[:] = np.ascontiguousarray(arr, dtype=PARTICLE_DTYPE)

    def start_indices_view(self):
        ptr = self.L.orc_start_indices(self.h)
        n = int(self.L.orc_start_indices_len(self.h))
        buf = (C.c_char * (n * 4)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.uint32)

    def start_indices(self):
        return self.start_indices_view().copy()

    def texture_view(self):
        ptr = self.L.orc_texture(self.h)
        n = int(self.settings.texture_size.x) * int(self.settings.texture_size.y)
        buf = (C.c_char * (n * 8)).from_address(ptr)
        return np.frombuffer(buf, dtype=np.float32).reshape(int(self.settings.texture_size.y),
                                                            int(self.settings.texture_size.x), 2)

    def uniform_bytes(self):
        out = (C.c_char * 120)()
        self.L.orc_uniform(self.h, C.addressof(out))
        return bytes(out)


def set_threads(n):
    """OpenMP threads for the oracle's per-particle loops (results are identical for any count)."""
    lib().orc_set_threads(int(n))


def max_threads():
    return int(lib().orc_max_threads())


def gradient_field(image):
    """generate_smooth_gradient_field (src/main.rs:403-515) on the CPU: u8 [h, w] -> f32 [h, w, 2]."""
    image = np.ascontiguousarray(image, dtype=np.uint8)
    h, w = image.shape
    out = np.empty((h, w, 2), dtype=np.float32)
    lib().orc_gradient_field(image.ctypes.data, w, h, out.ctypes.data)
    return out


def bitonic_keys(keys):
    """Run the reference network on bare u32 keys; returns (sorted_keys, perm)."""
    k = np.ascontiguousarray(keys, dtype=np.uint32).copy()
    perm = np.empty_like(k)
    lib().orc_bitonic_keys(k.ctypes.data, perm.ctypes.data, k.shape[0])
    return k, perm

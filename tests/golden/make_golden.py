"""Generates tests/golden/*.npz from the CPU oracle (the reference cannot be run here and
ships no fixtures: SURVEY.md §4, §8c).  Run from the repo root: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpu_fluid_simulation_amd as g  # noqa: E402  (ABI structs only)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def dam_break(n, steps, name):
    st, off, tick = g.dam_break_2d(n)
    o = O.OracleSim(st, off)
    out = {"steps": np.int32(steps), "n": np.int32(n)}
    for s in range(steps):
        o.step(tick)
        out[f"particles_{s}"] = o.particles()
        out[f"start_indices_{s}"] = o.start_indices()
    np.savez_compressed(os.path.join(HERE, name), **out)


def jittered(n, steps, seed, name):
    """Jittered dam break with random velocities, a force field and the mouse on."""
    st, off, tick = g.dam_break_2d(n)
    tick.mouse_state = 1
    tick.mouse_pos = g.Vec2(-3.0, 2.0)
    rng = np.random.default_rng(seed)
    o = O.OracleSim(st, off)
    v = o.particles_view()
    v["position"] += rng.uniform(-0.025, 0.025, size=(n, 2)).astype(np.float32)
    v["predicted_position"] = v["position"]
    v["velocity"] = rng.uniform(-1, 1, size=(n, 2)).astype(np.float32)
    init = o.particles()
    field = np.zeros((st.texture_size.y, st.texture_size.x, 2), dtype=np.float32)
    field[600:700, 200:400] = (0.5, -1.0)
    o.texture_view()[:] = field
    out = {"steps": np.int32(steps), "n": np.int32(n), "initial": init,
           "field_box": np.array([600, 700, 200, 400], dtype=np.int32),
           "field_value": np.array([0.5, -1.0], dtype=np.float32)}
    for s in range(steps):
        o.step(tick)
        out[f"particles_{s}"] = o.particles()
        out[f"start_indices_{s}"] = o.start_indices()
    np.savez_compressed(os.path.join(HERE, name), **out)


if __name__ == "__main__":
    dam_break(4096, 8, "dam_break_4096.npz")
    jittered(3000, 6, 2024, "jitter_mouse_field_3000.npz")
    print("golden fixtures written")

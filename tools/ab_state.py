"""A/B of kernel variants on ONE frozen state (variants that break the physics cannot be stepped far).
  python tools/ab_state.py save 60            -> /tmp/fs_state.npy  (16M dam break after 60 default steps)
  python tools/ab_state.py run <lib.so|default> -> per-pass times of the one step from that state, repeated 4x"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import gpu_fluid_simulation_amd as g
from gpu_fluid_simulation_amd import _abi
mode = sys.argv[1]
n = 1 << 24
st, off, tick = g.dam_break_2d(n)
if mode == "save":
    sim = g.FluidSimulation(st, device=0, initial_offset=off)
    for _ in range(int(sys.argv[2])): sim.tick(tick)
    np.save("/tmp/fs_state.npy", sim.download_particles())
    print("saved")
else:
    variant = sys.argv[2]
    if variant != "default":
        _abi._lib = _abi.load_library(os.path.join("gpu-fluid-simulation_amd", variant))
    sim = g.FluidSimulation(st, device=0, initial_offset=off)
    state = np.load("/tmp/fs_state.npy")
    sim.profile(True)
    for i in range(4):
        sim.upload_particles(state)          # every timed step starts from the SAME state
        sim.sync(); sim.profile_read(True)
        ms = sim.timed_steps(tick, 1)
        p, k = sim.profile_read(True)
        print(variant, "step", i, round(ms, 4), {a: round(b, 4) for a, b in p.items()}, flush=True)

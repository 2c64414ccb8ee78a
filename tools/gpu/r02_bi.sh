#!/bin/bash
set -o pipefail
O=gpurun_out/r02bi; mkdir -p $O
python - > $O/out.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import gpu_fluid_simulation_amd as g
def run(n, steps=100, warm=10):
    st, off, tick = g.dam_break_2d(n)
    sim = g.FluidSimulation(st, device=0, initial_offset=off)
    for _ in range(warm): sim.tick(tick)
    sim.sync()
    ms = sim.timed_steps(tick, steps); sim.sync(); sim.close()
    return round(ms / steps, 4)
print("1M fresh", run(1 << 20), run(1 << 20))
print("16M", run(1 << 24))
print("1M after 16M", run(1 << 20), run(1 << 20))
print("64M", run(1 << 26, 20, 5))
print("1M after 64M", run(1 << 20), run(1 << 20))
PY
cat $O/out.txt

"""Render a few frames of the dam break to PNG (visual validation on a headless GPU box)."""
import os, sys
sys.path.insert(0, os.getcwd())
import gpu_fluid_simulation_amd as g
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
frames = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 150, 400, 800]
outdir = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/frames"
os.makedirs(outdir, exist_ok=True)
st, off, tick = g.dam_break_2d(n)
sim = g.FluidSimulation(st, device=0, initial_offset=off)
done = 0
for f in frames:
    while done < f:
        sim.tick(tick); done += 1
    img = sim.render_density(640, 400)
    g.write_png(os.path.join(outdir, f"dam_{n}_{f:05d}.png"), img)
    print("frame", f, "alpha coverage", float((img[..., 3] > 0.5).mean()), flush=True)

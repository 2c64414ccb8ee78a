"""One-off fuzz of the two "next" rows around the step: the obstacle-field producer (bit-exact against the oracle's
line-by-line restatement) on random image shapes and masks, and the density-splat image (1e-4 against the oracle)
on random scenes, step counts and views.  python tools/fuzz_field_render.py [first] [cases]"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import gpu_fluid_simulation_amd as fs
from oracle import oracle as orc
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time()
for case in range(first, first + cases):
    rng = np.random.default_rng(13000 + case)
    w, h = int(rng.integers(1, 1025)), int(rng.integers(1, 700))
    img = np.zeros((h, w), dtype=np.uint8)
    kind = case % 4
    if kind == 0:                                            # blobs
        yy, xx = np.mgrid[0:h, 0:w]
        for _ in range(int(rng.integers(1, 8))):
            cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(1, max(2, min(w, h) / 4))
            img[(xx - cx) ** 2 + (yy - cy) ** 2 < r * r] = rng.integers(129, 256)
    elif kind == 1:                                          # salt noise incl. the threshold values 128 / 129
        m = rng.random((h, w)) < rng.uniform(0.0005, 0.2)
        img[m] = rng.choice([128, 129, 200, 255], size=int(m.sum()))
    elif kind == 2:                                          # lines / borders
        img[rng.integers(0, h), :] = 255
        img[:, rng.integers(0, w)] = 130
        if rng.random() < 0.5:
            img[0, :] = 255; img[-1, :] = 255
    # kind 3: empty image (the border is the source set)
    got = fs.generate_force_field(img)
    want = orc.gradient_field(img)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"field case {case} ({w}x{h}, kind {kind})"
    if case % 5 == 0:                                        # render
        n = int(rng.integers(64, 9000))
        st, off, tick = fs.dam_break_2d(int(np.sqrt(n)) ** 2)
        sim = fs.FluidSimulation(st, device=0, initial_offset=off)
        ref = orc.OracleSim(st, off)
        for _ in range(int(rng.integers(1, 50))):
            sim.tick(tick); ref.step(tick)
        pw, ph = int(rng.integers(1, 200)), int(rng.integers(1, 160))
        cx, cy = rng.uniform(-st.size.x / 2, st.size.x / 2), rng.uniform(-st.size.y / 2, st.size.y / 2)
        sx, sy = rng.uniform(0.2, st.size.x), rng.uniform(0.2, st.size.y)
        wmin, wmax = (float(cx - sx / 2), float(cy - sy / 2)), (float(cx + sx / 2), float(cy + sy / 2))
        a = sim.render_density(pw, ph, world_min=wmin, world_max=wmax)
        b = ref.render(pw, ph, wmin, wmax)
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4, err_msg=f"render case {case}")
        sim.close(); ref.close()
    if (case - first) % 20 == 19:
        print(f"cases {first}..{case} ok ({time.time()-t0:.0f}s)", flush=True)
print("field/render fuzz ok:", cases, "cases")

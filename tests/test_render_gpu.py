"""Renderer hand-off: the headless density-splat image against the oracle's
restatement of fluid_shader.wgsl:27-102.  exp/log are device-library functions, so the bar is a
float tolerance (1e-4 abs on colours in [0, 3]); the underlying particle data is bit-identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_density_splat_matches_oracle(fs, orc, tmp_path):
    st, off, tick = fs.dam_break_2d(4096)
    sim = fs.FluidSimulation(st, device=0, initial_offset=off)
    ref = orc.OracleSim(st, off)
    with pytest.raises(fs.FluidSimError):
        sim.render_density(8, 8)                         # no cell table before the first step
    for _ in range(40):
        sim.tick(tick); ref.step(tick)
    wmin, wmax = (-st.size.x / 2, -st.size.y / 2), (st.size.x / 2, st.size.y / 2)
    got = sim.render_density(160, 100)
    want = ref.render(160, 100, wmin, wmax)
    assert got.shape == (100, 160, 4)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4)
    assert got[..., 3].max() == 1.0 and got[..., 3].min() == 0.0      # fluid and empty space both visible
    # zoomed view + PNG writer
    z = sim.render_density(64, 64, world_min=(-6.4, 2.0), world_max=(-4.4, 4.0))
    np.testing.assert_allclose(z, ref.render(64, 64, (-6.4, 2.0), (-4.4, 4.0)), rtol=0, atol=1e-4)
    path = tmp_path / "frame.png"
    fs.write_png(str(path), got)
    assert path.read_bytes()[:8] == b"\x89PNG\r\n\x1a\n"

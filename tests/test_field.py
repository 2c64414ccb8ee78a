"""Obstacle field producer: HIP wavefront version of generate_smooth_gradient_field
(src/main.rs:403-515) == the oracle's line-by-line restatement, bit for bit (integer coordinates
and exactly representable f32 arithmetic); then the produced field drives the push-out branch of
move_particle (compute.wgsl:127-140) identically on both sides."""
import numpy as np
import pytest


def _blobs(w, h, seed, k=5):
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(k):
        cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(3, max(4, min(w, h) / 6))
        img[(xx - cx) ** 2 + (yy - cy) ** 2 < r * r] = rng.integers(129, 256)
    img[rng.integers(0, h, 20), rng.integers(0, w, 20)] = 128            # exactly 128 is NOT a source (> 128)
    return img


def test_oracle_field_basic_properties(orc):
    img = np.zeros((32, 48), dtype=np.uint8)
    img[10, 20] = 255
    f = orc.gradient_field(img)
    assert f.shape == (32, 48, 2)
    assert tuple(f[10, 20]) == (0.0, 0.0)
    assert tuple(f[10, 25]) == (-5.0, 0.0)           # points from the pixel to the source
    assert tuple(f[4, 20]) == (0.0, 6.0)
    empty = orc.gradient_field(np.zeros((16, 16), dtype=np.uint8))      # no source -> the border is the source set
    assert tuple(empty[0, 5]) == (0.0, 0.0) and tuple(empty[8, 1]) == (-1.0, 0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,seed", [(64, 48, 1), (257, 131, 2), (1024, 1024, 3), (300, 1, 4), (1, 40, 5), (1024, 1000, 6)])
def test_field_matches_oracle_bitwise(fs, orc, w, h, seed):
    img = _blobs(w, h, seed)
    got = fs.generate_force_field(img)
    want = orc.gradient_field(img)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_field_without_sources_and_limits(fs, orc):
    img = np.zeros((200, 333), dtype=np.uint8)
    assert np.array_equal(fs.generate_force_field(img), orc.gradient_field(img))
    with pytest.raises(fs.FluidSimError):
        fs.generate_force_field(np.zeros((1025, 8), dtype=np.uint8))


@pytest.mark.gpu
def test_obstacle_image_drives_the_step(fs, orc):
    st, off, tick = fs.dam_break_2d(4096)
    sim = fs.FluidSimulation(st, device=0, initial_offset=off)
    ref = orc.OracleSim(st, off)
    img = np.zeros((1024, 1024), dtype=np.uint8)
    img[700:1024, 300:420] = 255                      # a pillar standing on the floor, right of the block
    field = sim.set_obstacle_image(img, want_field=True)
    want = orc.gradient_field(img)
    assert np.array_equal(field.view(np.uint32), want.view(np.uint32))
    ref.texture_view()[:] = want
    for s in range(30):
        sim.tick(tick); ref.step(tick)
    got, exp = sim.download_particles(), ref.particles()
    for f in ("position", "velocity", "density"):
        assert np.array_equal(got[f].view(np.uint32), exp[f].view(np.uint32)), f

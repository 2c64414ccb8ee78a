"""The force pass replaces the two divisions by loop-invariant constants (2h^3 and h^2, funcs.wgsl:119)
with a 3-instruction form ONLY after proving on the GPU, over all 2^32 f32 inputs, that it is
bit-identical to the IEEE division for the handle's constants.  This test runs that enumeration through
the ABI, checks that a wrong reciprocal is caught, and that the handle's status reflects the proof."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


LO = 2.0 ** -60


def _mismatches(fs, c, y, lo=LO, hi=None):
    lib = fs.load_library()
    bad = C.c_uint32(7)
    hi = c if hi is None else hi
    assert lib.fs_selftest_constdiv(0, C.c_float(c), C.c_float(y), C.c_float(lo), C.c_float(hi), C.byref(bad)) == 0, \
        lib.fs_last_error()
    return bad.value


def test_constant_division_proofs(fs):
    f = np.float32
    results = {}
    for h in (f(0.2), f(0.1), f(0.25), f(0.37), f(1.0)):
        for name, c in (("2h3", f(2.0) * h * h * h), ("h2", h * h)):
            y = f(1.0) / c
            results[(float(h), name)] = _mismatches(fs, float(c), float(y))
            # a wrong reciprocal must be caught by the enumeration (a 1-ulp error can still be repaired by
            # the correction step for some constants, so use a grossly wrong one)
            assert _mismatches(fs, float(c), float(y * f(1.001))) > 0
    print("mismatch counts for y = RN(1/c) on 2^-60 <= |x| <= c:", results)
    # outside the proven range the form really does fail (overflow: inf - inf), which is why the range matters
    assert _mismatches(fs, 0.016, float(f(1.0) / f(0.016)), lo=1e30, hi=3e38) > 0
    # the handle's status is exactly "proof succeeded" for its two constants
    st, off, tick = fs.dam_break_2d(4096)
    sim = fs.FluidSimulation(st, device=0, initial_offset=off)
    status = fs.load_library().fs_constdiv_status(sim._h)
    assert bool(status & 1) == (results[(float(f(0.2)), "2h3")] == 0)
    assert bool(status & 2) == (results[(float(f(0.2)), "h2")] == 0)
    # bits 2 / 3: the lean reciprocal and square root were proven over their whole ranges on this device
    assert status & 4 and status & 8, f"rcp/sqrt proofs failed on this device (status {status})"
    # bit 4: x / h of the cell coordinates (funcs.wgsl:212-214) for every numerator up to 4 x the larger bound —
    # the same enumeration through the self-test entry point must agree
    size = max(st.size.x, st.size.y)
    h32 = f(st.smoothing_radius)
    assert bool(status & 16) == (_mismatches(fs, float(h32), float(f(1.0) / h32), lo=2.0 ** -60, hi=float(f(4.0) * f(size))) == 0)
    assert status & 16, "the default scene's cell size is expected to pass (h = 0.2)"

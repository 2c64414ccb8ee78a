"""The late-stage plans of the engine's sort (csrc/kernels_sort.hip: shifted merge behind a device-side certificate,
per-stage launches otherwise) against the oracle's run of the reference network (sort.wgsl:27-51, schedule
src/simulation.rs:323-347) on adversarial key sets: the arrangement — ties included — must be the network's whichever
plan the certificate picks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _moved(n, rng, reach, ties=8, frac=0.3):
    """Sorted keys with `ties` equal keys per value, then a fraction of the elements re-keyed so that their sorted
    position moves by up to `reach` places (what one simulation step does to the previous step's order)."""
    k = (np.arange(n, dtype=np.int64) // ties)
    sel = rng.random(n) < frac
    k[sel] += rng.integers(-reach // ties, reach // ties + 1, size=int(sel.sum()))
    return np.clip(k, 0, None).astype(np.uint32)


def _check(fs, orc, keys, fuse, expect_plan=None):
    want_k, want_p = orc.bitonic_keys(keys)
    got_k, got_p, plan = fs.selftest_sort(keys, fuse_stage=fuse)
    assert np.array_equal(got_k, want_k)
    assert np.array_equal(got_p, want_p), f"tie arrangement differs (fuse_stage={fuse}, plan={plan})"
    if expect_plan is not None:
        assert plan == expect_plan, plan
    return plan


@pytest.mark.parametrize("n", [1 << 15, (1 << 16) + 5, 200_000, 1 << 18, (1 << 19) - 4097, 1 << 20])
@pytest.mark.parametrize("fuse", [-1, 0, 13, 14])
def test_small_moves_take_the_shifted_merge_and_match(fs, orc, n, fuse):
    rng = np.random.default_rng(n + fuse)
    keys = _moved(n, rng, reach=1500)
    plan = _check(fs, orc, keys, fuse)
    s = int(np.ceil(np.log2(n)))
    s0 = fuse if fuse > 0 else (max(s - 6, 13) if fuse < 0 else 0)
    if s0 and 13 <= s0 < s:
        assert plan == (1, 0)            # moves of 1500 places fit the +-4096 window of stage 13 and every later one
    else:
        assert plan == (0, 0)


@pytest.mark.parametrize("n", [1 << 16, 300_000, 1 << 20])
@pytest.mark.parametrize("fuse", [-1, 13, 15])
def test_random_keys_fail_the_certificate_and_match(fs, orc, n, fuse):
    rng = np.random.default_rng(7 * n + fuse)
    keys = rng.integers(0, 50_000, size=n, dtype=np.uint32)
    plan = _check(fs, orc, keys, fuse)
    assert plan[0] == 0


@pytest.mark.parametrize("reach", [3000, 4000, 4200, 6000, 9000, 20000])
def test_moves_around_the_window_size(fs, orc, reach):
    # stage 13: windows of +-4096 places; moves below, at and above that
    n = 1 << 18
    rng = np.random.default_rng(reach)
    keys = _moved(n, rng, reach=reach, ties=4, frac=0.05)
    for fuse in (13, 14, 15):
        _check(fs, orc, keys, fuse)


def test_one_long_run_of_equal_keys_across_windows(fs, orc):
    # equal keys never swap: a run of ties that covers whole windows and block boundaries must keep the network's order
    n = 1 << 18
    rng = np.random.default_rng(5)
    keys = _moved(n, rng, reach=800, ties=16)
    keys[40_000:140_000] = keys[40_000]
    keys[140_000:] = np.maximum(keys[140_000:], keys[40_000])
    for fuse in (13, 14, 16):
        _check(fs, orc, keys, fuse)


def test_sorted_and_reversed_inputs(fs, orc):
    n = 150_000
    keys = (np.arange(n, dtype=np.uint32) // 3)
    assert _check(fs, orc, keys, 13) == (1, 0)
    assert _check(fs, orc, keys[::-1].copy(), 13)[0] == 0


def test_a_single_far_mover_forces_the_per_stage_plan(fs, orc):
    n = 1 << 19
    keys = (np.arange(n, dtype=np.uint32) // 8)
    keys[1000] = keys[-1] + 5          # one element travels the whole array
    assert _check(fs, orc, keys, 13) == (0, 1)
    keys = (np.arange(n, dtype=np.uint32) // 8)
    keys[n - 7] = 0                    # ... and one the other way
    assert _check(fs, orc, keys, 14) == (0, 1)


@pytest.mark.parametrize("kind", ["moved", "random", "far_mover"])
def test_plans_at_4m(fs, orc, kind):
    # 1024 tiles, 22 stages: every kind of launch of both plans has work to do or to skip
    n = (1 << 22) + 12_345
    rng = np.random.default_rng(99)
    if kind == "moved":
        keys = _moved(n, rng, reach=5000)
    elif kind == "random":
        keys = rng.integers(0, 1 << 20, size=n, dtype=np.uint32)
    else:
        keys = _moved(n, rng, reach=5000)
        keys[123] = keys.max() + 1
    for fuse in (-1, 14, 17):
        plan = _check(fs, orc, keys, fuse)
        assert (plan[0] == 1) == (kind == "moved"), plan


STANDBY = 0x100      # fs_selftest_sort: the single stand-by launch instead of the per-stage ones


@pytest.mark.parametrize("n", [1 << 16, 300_000, (1 << 22) + 12_345])
@pytest.mark.parametrize("kind", ["moved", "random", "far_mover", "reversed"])
def test_standby_kernel_is_exact(fs, orc, n, kind):
    # the persistent stand-by kernel (grid barrier between passes) must produce the network's arrangement on its own
    rng = np.random.default_rng(n % 1000 + len(kind))
    keys = _moved(n, rng, reach=3000)
    if kind == "random":
        keys = rng.integers(0, 1 << 18, size=n, dtype=np.uint32)
    elif kind == "far_mover":
        keys[77] = keys.max() + 3
    elif kind == "reversed":
        keys = keys[::-1].copy()
    for stage in (13, 15):
        plan = _check(fs, orc, keys, stage | STANDBY)
        assert (plan[0] == 1) == (kind == "moved"), plan


def _run(fs, n, steps, env, monkeypatch, disturb=None):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    st, off, tick = fs.dam_break_2d(n)
    sim = fs.FluidSimulation(st, device=0, initial_offset=off)
    for i in range(steps):
        if disturb is not None:
            disturb(sim, i)
        sim.tick(tick)
    out = sim.download_particles(), sim.sort_plan()
    for k in env:
        monkeypatch.delenv(k)
    return out


@pytest.mark.parametrize("n,steps", [(1 << 16, 60), (1 << 18, 40), (1 << 20, 30)])
def test_plan_policy_never_changes_the_state(fs, monkeypatch, n, steps):
    # whichever launch sequence the host policy picks (stage, stand-by kind), the state is the per-stage plan's
    base, _ = _run(fs, n, steps, {"FS_SORT_POLICY": "0", "FS_SORT_FUSE_STAGE": "0"}, monkeypatch)
    got, info = _run(fs, n, steps, {}, monkeypatch)
    assert got.tobytes() == base.tobytes()
    assert info["shifted"] > 0 and info["timeouts"] == 0
    forced, info = _run(fs, n, steps, {"FS_SORT_TRUST": "1"}, monkeypatch)
    assert forced.tobytes() == base.tobytes()
    assert info["shifted"] + info["per_stage"] == steps and info["timeouts"] == 0
    assert info["standby_runs"] == info["per_stage"]               # every failed certificate was served by the stand-by kernel


def test_upload_of_a_shuffled_state_mid_run(fs, monkeypatch):
    n = 1 << 18

    def disturb(sim, i):
        if i == 25:
            p = sim.download_particles()
            rng = np.random.default_rng(4)
            sim.upload_particles(p[rng.permutation(n)])

    base, _ = _run(fs, n, 40, {"FS_SORT_POLICY": "0", "FS_SORT_FUSE_STAGE": "0"}, monkeypatch, disturb)
    got, info = _run(fs, n, 40, {}, monkeypatch, disturb)
    assert got.tobytes() == base.tobytes() and info["timeouts"] == 0
    forced, info = _run(fs, n, 40, {"FS_SORT_TRUST": "1"}, monkeypatch, disturb)
    assert forced.tobytes() == base.tobytes() and info["standby_runs"] >= 1 and info["timeouts"] == 0


def test_a_barrier_time_out_report_kills_the_handle(fs, monkeypatch):
    """include/fluidsim.h: after the stand-by kernel reports a grid-barrier time-out, fs_step fails with FS_ERR_DEVICE and
    keeps failing (the handle is dead; destroy it).  The report is injected (FS_SORT_INJECT_TIMEOUT: the barriers
    themselves hold), what is under test is the host's reaction; other handles of the process are not affected."""
    n = 1 << 18
    monkeypatch.setenv("FS_SORT_TRUST", "1")              # the single stand-by launch from the first step on
    monkeypatch.setenv("FS_SORT_INJECT_TIMEOUT", "1")
    st, off, tick = fs.dam_break_2d(n)
    sim = fs.FluidSimulation(st, device=0, initial_offset=off)
    monkeypatch.delenv("FS_SORT_TRUST")
    monkeypatch.delenv("FS_SORT_INJECT_TIMEOUT")
    rng = np.random.default_rng(11)
    p = sim.download_particles()
    sim.upload_particles(p[rng.permutation(n)])           # an arbitrary order: the certificate fails, the stand-by kernel works
    # a caller that steps, synchronises and downloads must not get FS_OK and a corrupt state — the time-out is
    # reported by the FIRST synchronisation after the offending step, not only by the plan of a later fs_step
    sim.tick(tick)                                        # the stand-by kernel runs (and "times out") in this very step
    with pytest.raises(fs.FluidSimError) as ei:
        sim.sync()
    assert ei.value.status == fs._abi.FS_ERR_DEVICE and "timed out" in str(ei.value)
    with pytest.raises(fs.FluidSimError):
        sim.download_particles()
    info = sim.sort_plan()                                # diagnostics stay readable
    assert info["timeouts"] >= 1 and info["standby_runs"] >= 1
    for _ in range(3):                                    # terminal: every later step fails the same way
        with pytest.raises(fs.FluidSimError) as ei:
            sim.tick(tick)
        assert ei.value.status == fs._abi.FS_ERR_DEVICE
    del sim
    # ... and when nobody synchronises in between, the certificate's report of a LATER step stops the run as before
    monkeypatch.setenv("FS_SORT_TRUST", "1"); monkeypatch.setenv("FS_SORT_INJECT_TIMEOUT", "1")
    sim = fs.FluidSimulation(st, device=0, initial_offset=off)
    monkeypatch.delenv("FS_SORT_TRUST"); monkeypatch.delenv("FS_SORT_INJECT_TIMEOUT")
    sim.upload_particles(p[rng.permutation(n)])
    failed_at = None
    for i in range(30):
        try:
            sim.tick(tick)
        except fs.FluidSimError as e:
            assert e.status == fs._abi.FS_ERR_DEVICE and "timed out" in str(e)
            failed_at = i
            break
    assert failed_at is not None and failed_at >= 1
    del sim
    other = fs.FluidSimulation(st, device=0, initial_offset=off)       # a fresh handle is healthy
    for _ in range(5):
        other.tick(tick)
    assert other.sort_plan()["timeouts"] == 0

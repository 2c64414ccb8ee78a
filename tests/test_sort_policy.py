"""The host policy that picks the sort's late-stage plan (csrc/sort_policy.h), replayed on the CPU through the C ABI
(fs_selftest_sort_policy) against a model of the flow.  Only the launch sequence depends on the policy — the kernels
produce the reference network's arrangement under any plan (tests/test_sort_gpu.py) — so what is checked here is the
COST side: failed certificates are rare, the stage follows the flow with a small margin, the expensive stand-by kernel
is only trusted when reports have been passing."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def fs():
    import gpu_fluid_simulation_amd as g
    g.load_library()
    return g


def _failures(stage, required):
    return int((stage < required).sum())


def test_steady_flow_settles_one_stage_above_the_requirement(fs):
    req = np.full(200, 16, dtype=np.uint32)
    stage, single = fs.selftest_sort_policy(req, log2_count=24)
    assert stage[0] == 16                                   # first guess: S - 8
    assert _failures(stage, req) == 0
    assert set(stage[40:]) == {17}                          # fit class 0 at 16 (no room) -> one stage up, and it stays
    assert single[:4].sum() == 0 and single[12:].all()      # trusted only after reports came in


def test_roomy_flow_descends_to_two_x_headroom(fs):
    req = np.full(200, 13, dtype=np.uint32)
    stage, single = fs.selftest_sort_policy(req, log2_count=24)
    assert _failures(stage, req) == 0
    assert stage[-1] == 14 and np.all(np.diff(stage.astype(int)) <= 0)      # 16 -> 15 -> 14, never below req + 1
    assert single[-1] == 1


def test_slow_growth_is_followed_without_a_failed_certificate(fs):
    # the dam break: the requirement creeps up a stage every ~40 steps
    req = np.repeat(np.arange(15, 20, dtype=np.uint32), 40)
    stage, single = fs.selftest_sort_policy(req, log2_count=24)
    assert _failures(stage, req) == 0
    assert np.all(stage >= req) and np.all(stage[10:] <= req[10:] + 2)
    assert single[20:].mean() > 0.95


def test_a_sudden_jump_costs_a_few_failures_then_recovers(fs):
    req = np.concatenate([np.full(60, 15), np.full(100, 19)]).astype(np.uint32)
    stage, single = fs.selftest_sort_policy(req, log2_count=24, lag=4)
    bad = np.flatnonzero(stage < req)
    assert 1 <= len(bad) <= 12 and bad.min() == 60           # reports are 4 steps old, two stages per failed report
    assert np.all(stage[bad.max() + 1:] >= 19)
    # the stand-by launch is withdrawn as soon as the first failure is reported, and comes back after passing reports
    assert single[64:bad.max() + 1].sum() == 0 and single[-1] == 1


def test_requirement_beyond_the_last_stage_is_clamped(fs):
    req = np.full(80, 30, dtype=np.uint32)                  # never fits (an arbitrary order every step)
    stage, single = fs.selftest_sort_policy(req, log2_count=20, start_back=8)
    assert stage.max() == 19 and single.sum() == 0          # S - 1, and never the stand-by kernel


@pytest.mark.parametrize("lag", [1, 2, 4, 8])
def test_random_walk_requirement(fs, lag):
    rng = np.random.default_rng(lag)
    walk = 16 + np.cumsum(rng.choice([-1, 0, 0, 0, 0, 0, 0, 1], size=600)) // 4
    req = np.clip(walk, 13, 21).astype(np.uint32)
    stage, single = fs.selftest_sort_policy(req, log2_count=24, lag=lag)
    assert _failures(stage, req) <= 6                       # a stage of slack absorbs single-stage moves
    assert stage.max() <= req.max() + 2

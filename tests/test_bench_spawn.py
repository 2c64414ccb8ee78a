"""bench.py as its own launcher: bare `python3 bench.py --gpus N` starts N rank processes (never exec,
nothing GPU-related in the parent), relays rank 0's JSON line and fails when any rank fails."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_spawner(tmp_path, child_src, n=3, timeout=None):
    child = tmp_path / "child.py"
    child.write_text(textwrap.dedent(child_src))
    drv = tmp_path / "drv.py"
    drv.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        assert "torch" not in sys.modules, "the launcher must not import torch"
        rc = bench.spawn_ranks({n}, ["--x", "1"], child=[sys.executable, {str(child)!r}], timeout={timeout!r})
        assert "torch" not in sys.modules
        sys.exit(rc)
    """))
    return subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, timeout=120)


def test_spawner_sets_rank_env_and_relays_rank0(tmp_path):
    r = _run_spawner(tmp_path, """
        import json, os, sys
        print(json.dumps({k: os.environ[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")} | {"argv": sys.argv[1:]}))
    """)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "only rank 0 prints on stdout"         # the others go to stderr
    assert lines[0]["RANK"] == "0" and lines[0]["WORLD_SIZE"] == "3" and lines[0]["MASTER_ADDR"] == "127.0.0.1"
    assert lines[0]["argv"] == ["--x", "1"]
    others = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")]
    assert sorted(o["RANK"] for o in others) == ["1", "2"]
    assert len({o["MASTER_PORT"] for o in others} | {lines[0]["MASTER_PORT"]}) == 1


def test_spawner_fails_when_a_rank_fails_and_stops_the_rest(tmp_path):
    r = _run_spawner(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)          # would hang in a collective: the launcher must terminate it
    """)
    assert r.returncode == 7
    assert "rank 1 exited with 7" in r.stderr


def test_spawner_timeout(tmp_path):
    r = _run_spawner(tmp_path, "import time; time.sleep(60)", n=2, timeout=1.0)
    assert r.returncode == 124


def test_bench_parent_branch_is_before_any_gpu_import():
    """The launcher branch of main() sits before `import torch` and before load_product()."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    i_spawn = main.index("spawn_ranks(args.gpus")
    assert i_spawn < main.index("import torch") and i_spawn < main.index("load_product(args.no_build)")
    assert "os.exec" not in src and "execv" not in src


def test_launcher_refuses_to_spawn_under_a_profiler():
    """under rocprofv3 the preloaded library has initialised the GPU in the parent; a bare `--gpus N` must not start
    ranks from it (fork + exec from a GPU process), it exits 4 with a message instead."""
    env = dict(os.environ, ROCPROF_TEST_MARKER="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 4 and "under a profiler" in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
def test_bare_bench_gpus2_runs_to_a_parsed_line():
    """`python3 bench.py --gpus 2` with no launcher: two ranks share the one GPU of the box over gloo."""
    env = dict(os.environ, FS_DIST_BACKEND="gloo", FS_FORCE_DEVICE0="1", FS_NO_SCALING_BASE="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3",
                        "--workload", "dam_break_2d_1M", "--no-build"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["value"] > 0
    assert out["checks"]["particles_conserved"] and out["checks"]["protocol_violations"] == 0
    assert out["roofline"]["bound"] != "hbm" or out["roofline"]["frac"] >= 0.45      # from evidence, not assumed
    assert "rank0_passes_ms" in out["roofline"]
    # the roofline's kernel is a compute pass; the pack + exchange interval is reported beside it, never as "the kernel"
    assert not out["roofline"]["kernel"].startswith("predict_key") and out["roofline"]["pack_exchange_ms"] > 0
    assert out["slab_step"].startswith("edge-first") and out["boundary_cols"] >= 4

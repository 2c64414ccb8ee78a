"""Renderer hand-off without a host round trip (reference consumer: src/renderer.rs:457-458 binding the
buffers of src/simulation.rs:552-564): a SECOND PROCESS opens the handles from fs_export_handle and reads the
cell-sorted ParticleInstance records / start_indices straight from the simulation's device allocations."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
out = sys.argv[2]
# nothing GPU-related has been imported or called yet: handles are read first, the library is loaded after
hp = bytes.fromhex(sys.stdin.readline().strip())
hs = bytes.fromhex(sys.stdin.readline().strip())
import numpy as np
import gpu_fluid_simulation_amd as g
bp = g.ImportedBuffer(hp, device=0)
bs = g.ImportedBuffer(hs, device=0)
k = 0
print("ready", flush=True)
for line in sys.stdin:
    if line.strip() != "read":
        break
    np.save(os.path.join(out, f"child_particles_{k}.npy"), bp.read(g.PARTICLE_DTYPE))
    np.save(os.path.join(out, f"child_starts_{k}.npy"), bs.read(np.uint32))
    k += 1
    print("done", flush=True)
bp.close(); bs.close()
'''


def test_mem_handle_layout(fs):
    import ctypes as C
    assert C.sizeof(fs._abi.MemHandle) == 80        # 64-byte hipIpcMemHandle_t + bytes + device + dmabuf_fd


@pytest.mark.gpu
def test_second_process_reads_exported_buffers(fs, tmp_path):
    n = 16384
    st, off, tick = fs.dam_break_2d(n)
    sim = fs.FluidSimulation(st, device=0, initial_offset=off)
    for _ in range(3):
        sim.tick(tick)
    hp = sim.export_handle(fs._abi.FS_EXPORT_PARTICLES)          # switches on the live AoS view
    hs = sim.export_handle(fs._abi.FS_EXPORT_START_INDICES)
    assert hp.bytes == n * 32 and hs.bytes == sim.grid_dims[0] * sim.grid_dims[1] * 4
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    child = subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path)], stdin=subprocess.PIPE,
                             stdout=subprocess.PIPE, text=True, cwd=ROOT)
    try:
        child.stdin.write(bytes(hp).hex() + "\n" + bytes(hs).hex() + "\n")
        child.stdin.flush()
        assert child.stdout.readline().strip() == "ready"
        for k in range(3):
            # k = 0: the state the export call materialised; k >= 1: records written by the force pass itself
            sim.sync()
            child.stdin.write("read\n"); child.stdin.flush()
            assert child.stdout.readline().strip() == "done"
            got_p = np.load(tmp_path / f"child_particles_{k}.npy")
            got_s = np.load(tmp_path / f"child_starts_{k}.npy")
            want_p, want_s = sim.download_particles(), sim.download_start_indices()
            assert np.array_equal(got_p.view(np.uint8), want_p.view(np.uint8)), f"round {k}: records differ"
            assert np.array_equal(got_s, want_s)
            for _ in range(2):
                sim.tick(tick)
        child.stdin.write("quit\n"); child.stdin.flush()
        assert child.wait(timeout=60) == 0
    finally:
        if child.poll() is None:
            child.kill()
    sim.close()


@pytest.mark.gpu
def test_live_view_equals_export_pass_and_survives_uploads(fs, orc):
    """The records the force pass writes are byte-identical to the ones the export kernel builds (and to the oracle),
    also after an upload invalidated the view."""
    n = 4096
    st, off, tick = fs.dam_break_2d(n)
    a = fs.FluidSimulation(st, device=0, initial_offset=off)     # live view
    b = fs.FluidSimulation(st, device=0, initial_offset=off)     # export pass on demand
    ref = orc.OracleSim(st, off)
    a.export_handle()
    for s in range(6):
        a.tick(tick); b.tick(tick); ref.step(tick)
        if s == 2:
            p = b.download_particles()
            p["velocity"] += np.float32(0.25)
            a.upload_particles(p); b.upload_particles(p); ref.set_particles(p)
            assert np.array_equal(a.download_particles().view(np.uint8), b.download_particles().view(np.uint8))
        ga, gb = a.download_particles(), b.download_particles()
        assert np.array_equal(ga.view(np.uint8), gb.view(np.uint8)), s
        assert np.array_equal(ga.view(np.uint8), ref.particles().view(np.uint8)), s

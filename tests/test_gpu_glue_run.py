"""The LAMMPS-side binding run against the real library.

tests/c/glue_run (built in the development container by tests/c/build_glue_run.sh from lammps_glue/, tests/c/glue_run_harness.cpp
and the reference's own unmodified base classes; the binary travels to the GPU box) creates the style objects of example/simple's
deck - pair_style dpd/meso | dpd/fast/meso, fix nve/meso, compute temp/meso, atom_style dpd/atomic/meso - and drives them through the
virtuals of the reference's Pair / Fix / Compute base classes (src/pair.h:130-160, pair_dpd_meso.h:30-41, fix_nve_meso.cu:97-199)
in the order of a host-driven timestep.  The same sequence through meso_amd.api (ctypes on the same C ABI) must give the same
positions, velocities, forces and temperatures bit for bit; and the library's own run loop (meso_run: fused kernels) the same
trajectory."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, DP_RUN
from meso_amd.datagen import make_box

pytestmark = pytest.mark.gpu

EXE = os.path.join(ROOT, "tests", "c", "glue_run")


@pytest.mark.skipif(not os.path.exists(EXE), reason="tests/c/glue_run not built (needs the reference tree: tests/c/build_glue_run.sh)")
@pytest.mark.parametrize("style", ["dpd/meso", "dpd/fast/meso"])
def test_glue_objects_drive_the_library_like_the_api(tmp_path, style):
    from meso_amd.api import Meso
    L, nsteps, every = 8, 12, 5
    x, v, lo, hi = make_box(L)
    n = len(x)
    hdr = struct.pack("<8i6d6d", n, nsteps, every, DP_RUN["seed"], 1 if "fast" in style else 0, 0, 0, 0, *lo, *hi, 1.0, 0.3, 0.005, 15.0, 4.5, 3.0)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(hdr + np.ascontiguousarray(x, np.float64).tobytes() + np.ascontiguousarray(v, np.float64).tobytes())
    r = subprocess.run([EXE, str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    raw = fout.read_bytes()
    assert struct.unpack_from("<i", raw)[0] == n
    tag = np.frombuffer(raw, np.int32, n, 4)
    off = 4 + 4 * n
    xg = np.frombuffer(raw, np.float64, 3 * n, off).reshape(n, 3); off += 24 * n
    vg = np.frombuffer(raw, np.float64, 3 * n, off).reshape(n, 3); off += 24 * n
    fg = np.frombuffer(raw, np.float64, 3 * n, off).reshape(n, 3); off += 24 * n
    tg = np.frombuffer(raw, np.float64, nsteps, off)
    order = np.argsort(tag)
    assert np.array_equal(tag[order], np.arange(1, n + 1))
    xg, vg, fg = xg[order], vg[order], fg[order]

    def deck(m):
        m.read_atoms(x, v, lo, hi)
        m.neighbor(0.3)
        m.neigh_modify(delay=0, every=every, check=False)
        m.pair_style(style, 1.0, DP_RUN["seed"])
        m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
        m.timestep(0.005)
        m.setup()

    with Meso() as m:
        deck(m)
        temps = []
        for it in range(nsteps):
            m.step_advance(it + 1)
            m.initial_integrate()
            if m.decide():
                m.reneighbor()
            else:
                m.forward_comm()
            m.force_clear()
            m.compute(0, 0)
            m.final_integrate()
            temps.append(m.temperature())
        xa, va, fa, _, _ = m.gather()
    assert np.array_equal(xg, xa) and np.array_equal(vg, va) and np.array_equal(fg, fa)
    assert np.array_equal(tg, np.array(temps))
    # ... and the library's own run loop (fused step boundary, epilogue): the same trajectory
    with Meso() as m:
        deck(m)
        m.run(nsteps)
        xr, vr, fr, _, _ = m.gather()
        t_run = m.temperature()
    prd = hi - lo
    d = xr - xg
    d -= np.round(d / prd) * prd
    tol = 1e-9 if style == "dpd/meso" else 1e-4
    assert np.abs(d).max() < tol and np.abs(vr - vg).max() < 100 * tol
    assert t_run == pytest.approx(tg[-1], rel=1e-6)

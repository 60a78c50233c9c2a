"""A host program written in C (tests/c/abi_smoke.c) links libmeso_hip.so directly - no Python, no ctypes in the data path -
and runs 10 timesteps of the dp.run settings; its trajectory must be the one the Python mirror of the same ABI produces."""
import os
import struct
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("style", [0, 1])
def test_c_host_links_the_library_and_runs_10_steps(style):
    from meso_amd.api import Meso
    from meso_amd.datagen import make_box
    x, v, lo, hi = make_box(8)
    n = len(x)
    with tempfile.TemporaryDirectory() as d:
        exe, fin, fout = os.path.join(d, "abi_smoke"), os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        r = subprocess.run(["gcc", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
                            "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "meso_amd"), "-lmeso_hip",
                            "-Wl,-rpath," + os.path.join(ROOT, "meso_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        with open(fin, "wb") as f:
            f.write(struct.pack("<q3d3d", n, *lo, *hi))
            f.write(np.ascontiguousarray(x).tobytes())
            f.write(np.ascontiguousarray(v).tobytes())
        r = subprocess.run([exe, fin, fout, "10", str(style)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr[-3000:]
        raw = np.fromfile(fout, dtype=np.float64)
    T, xc, vc, fc = raw[0], raw[1:1 + 3 * n].reshape(n, 3), raw[1 + 3 * n:1 + 6 * n].reshape(n, 3), raw[1 + 6 * n:].reshape(n, 3)
    with Meso(0) as m:
        m.read_atoms(x, v, lo, hi)
        m.neighbor(0.3)
        m.neigh_modify(delay=0, every=5, check=False)
        m.pair_style("dpd/fast/meso" if style else "dpd/meso", 1.0, 419084618)
        m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
        m.timestep(0.005)
        m.run(10)
        xp, vp, fp, _, _ = m.gather()
        Tp = m.temperature()
    assert np.array_equal(xc, xp) and np.array_equal(vc, vp) and np.array_equal(fc, fp)
    assert T == Tp and 0.5 < T < 2.0
    assert np.abs(fc.sum(axis=0)).max() < 1e-6 * np.abs(fc).max() * n ** 0.5

"""The LAMMPS-side binding (lammps_glue/) is plain C++ against LAMMPS' own headers: when the reference tree is
mounted, check that it compiles against the reference's src/*.h (the style-registration surface it plugs into)."""
import os
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
def test_glue_compiles_against_reference_headers():
    cmd = ["g++", "-fsyntax-only", "-std=c++11", "-DLAMMPS_GZIP", "-I" + REF, "-I" + REF + "/STUBS",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "lammps_glue"),
           os.path.join(ROOT, "lammps_glue", "meso_hip_glue.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_glue_registers_the_reference_style_keys():
    txt = open(os.path.join(ROOT, "lammps_glue", "meso_hip_glue.h")).read()
    for key in ("PairStyle(dpd/meso,", "PairStyle(dpd/fast/meso,", "FixStyle(nve/meso,", "ComputeStyle(temp/meso,",
                "IntegrateStyle(mvv/meso,", "IntegrateStyle(verlet/meso,", "PairStyle(dpd/mini/meso,",
                "PairStyle(dpd/polyforce/meso,", "PairStyle(dpd/tableforce/meso,", "BondStyle(harmonic/meso,",
                "BondStyle(fene/meso,", "AngleStyle(harmonic/meso,"):
        assert key in txt

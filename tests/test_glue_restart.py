"""SURVEY.md 8(f) row 4, external check of the restart record: the glue's MesoHipPairDPD::write_restart / read_restart run against
LAMMPS' own Pair base class (src/pair.cpp compiled unmodified) and must emit / accept exactly the bytes MesoPairDPD::write_restart
and write_restart_settings emit (src/USER-MESO/pair_dpd_meso.cu:363-447): cut_global (f64) seed (i32) mix_flag (i32), then for
every i <= j the setflag (i32) followed - only when set - by a0 gamma sigma expw cut (f64 each).  The expected bytes are packed
here from that description; the C ABI behind the glue is a recording stub, so the test also sees what read_restart hands to
meso_pair_dpd_settings / meso_pair_dpd_coeff.  CPU only; needs the reference tree for LAMMPS' headers and pair.cpp."""
import os
import struct
import subprocess
import tempfile

import pytest

from conftest import ROOT

REF = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")
def test_pair_restart_record_has_the_reference_layout():
    inc = ["-I" + REF, "-I" + REF + "/STUBS", "-I" + REF + "/MOLECULE", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "lammps_glue")]
    with tempfile.TemporaryDirectory() as d:
        objs = []
        for src in (os.path.join(ROOT, "tests", "c", "glue_restart_harness.cpp"), os.path.join(ROOT, "lammps_glue", "meso_hip_glue.cpp"),
                    REF + "/pair.cpp", REF + "/memory.cpp", REF + "/error.cpp", REF + "/universe.cpp"):
            o = os.path.join(d, os.path.basename(src) + ".o")
            r = subprocess.run(["g++", "-std=c++11", "-O1", "-w", "-fPIC", "-DLAMMPS_GZIP"] + inc + ["-c", src, "-o", o], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            objs.append(o)
        for src in (os.path.join(ROOT, "tests", "c", "meso_stub.c"), REF + "/STUBS/mpi.c"):
            o = os.path.join(d, os.path.basename(src) + ".o")
            r = subprocess.run(["gcc", "-O1", "-w", "-fPIC", "-I" + REF + "/STUBS", "-c", src, "-o", o], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            objs.append(o)
        exe = os.path.join(d, "glue_restart")
        r = subprocess.run(["g++", "-o", exe] + objs + ["-Wl,--unresolved-symbols=ignore-all", "-lm"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        rec, log = os.path.join(d, "pair.restart"), os.path.join(d, "calls.log")
        subprocess.run([exe, "write", rec], check=True, timeout=60)
        got = open(rec, "rb").read()
        # MesoPairDPD::write_restart_settings :423-428, then MesoPairDPD::write_restart :363-380 for ntypes = 2
        exp = struct.pack("<dii", 1.0, 419084618, 0)                       # mix_flag GEOMETRIC = 0 (Pair::Pair, src/pair.cpp)
        exp += struct.pack("<i5d", 1, 15.0, 4.5, 3.0, 1.0, 1.0)            # 1-1
        exp += struct.pack("<i5d", 1, 40.0, 6.0, 3.4641016151377544, 0.5, 1.25)   # 1-2
        exp += struct.pack("<i", 0)                                        # 2-2 not set: the flag alone
        assert got == exp, (len(got), len(exp))
        # read_restart: consumes exactly that record, hands every value to the C ABI, and writes the same bytes again
        subprocess.run([exe, "read", rec, log], check=True, timeout=60)
        calls = open(log).read().split("\n")
        assert "meso_pair_dpd_settings 0 1 419084618" in calls
        assert "meso_pair_dpd_coeff 1 1 15 4.5 3 1 1" in calls
        assert "meso_pair_dpd_coeff 1 2 40 6 3.4641016151377544 0.5 1.25" in calls
        assert not any(c.startswith("meso_pair_dpd_coeff 2 2") for c in calls)
        assert open(log + ".rewrite", "rb").read() == exp

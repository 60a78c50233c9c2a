"""Host-side pieces of bench.py and of the RCCL stand-in that need no GPU: argument defaults (the driver's plain `python bench.py`
must name the 64^3 workload and the two other boxes of the reference's protocol), the hash that ties profiles/force_kernel_profile.json
to the force kernel's sources, and the stand-in library (tests/c/rccl_stand_in.cpp) exporting exactly the librccl entry points the
engine calls."""
import os
import re
import subprocess
import sys

from conftest import ROOT


def test_bench_defaults_and_source_hash(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.gpus, a.box, a.style, a.every) == (1, 64, "dpd/fast/meso", 5) and a.other_boxes == "25,48" and a.opt == []
    h = bench._kernel_source_hash()
    assert re.fullmatch(r"[0-9a-f]{16}", h) and h == bench._kernel_source_hash()
    assert bench._norm_kernel("void meso::k_pair_dpd_ring<true, 0, true, true, 1, true, 0>(meso::PairArgs)") == \
        bench._norm_kernel("k_pair_dpd_ring<true, 0, true, true, 1, true, 0>")


def test_rccl_stand_in_exports_what_the_engine_calls():
    src = os.path.join(ROOT, "tests", "c", "rccl_stand_in.cpp")
    lib = os.path.join(ROOT, "tests", "c", "librccl_stand_in.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        r = subprocess.run(["hipcc", "-O1", "-shared", "-fPIC", "-std=c++17", src, "-o", lib], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    have = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    used = set()
    for f in os.listdir(os.path.join(ROOT, "meso_amd", "csrc")):
        if f.endswith(".hip"):
            used |= set(re.findall(r"\b(nccl[A-Z]\w*)\s*\(", open(os.path.join(ROOT, "meso_amd", "csrc", f)).read()))
    assert used and used <= have, (sorted(used - have), sorted(have))

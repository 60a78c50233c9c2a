#!/usr/bin/env python3
"""Floors of the fp32 force kernel's mandatory work, measured on the real table (meso_pair_floor, meso_amd/csrc/pair_floor.hip):
python3 tools/pair_floor.py [--box 64] [--out profiles/r06_floor.txt]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from meso_amd.api import Meso
from meso_amd.datagen import make_box

ap = argparse.ArgumentParser()
ap.add_argument("--box", type=int, default=64)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--out", default=None)
a = ap.parse_args()
x, v, lo, hi = make_box(a.box)
m = Meso(0)
m.set_option("pair_npart", 1); m.set_option("row_part", 1)      # (the floor kernels are written for the one-lane form: 256-atom groups)
m.read_atoms(x, v, lo, hi)
m.neighbor(0.3)
m.neigh_modify(delay=0, every=5, check=False)
m.pair_style("dpd/fast/meso", 1.0, 419084618)
m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
m.timestep(0.005)
m.setup()
m.run(300)             # a thermalised configuration: the table of step 300
# the real kernel alone, same session
m.set_option("fuse_pair", 0); m.set_option("profile", 1); m.timer_reset()
m.run(50)
ms, calls = m.timer("pair")
m.set_option("profile", 0); m.set_option("fuse_pair", 1)
real_us = 1e3 * ms / calls
n = len(x)
lines = []
res = {}
for rep in range(2):
    for mode in (1, 2, 3, 6, 7, 802, 1202, 52, 56, 40002, 60002, 80002, 10000002, 12000002):
        us, nt, ne = m.pair_floor(mode, a.reps)
        res.setdefault(mode, []).append(us)
info = m.neigh_info()
nbar = info["avg_count"]
b_pair = n * (36 + 4 * nbar + 24)
lines.append("# tools/pair_floor.py --box %d: floors of the fp32 force kernel's mandatory work on the table of step 300 (MI355X)" % a.box)
lines.append("atoms %d, row entries walked (front sections) %d = %.2f per atom, pairs evaluated %d = %.2f per atom" % (n, nt, nt / n, ne, ne / n))
lines.append("B_pair (SURVEY.md 8d) %.1f MB -> 0.50 of 8 TB/s = %.1f us" % (b_pair / 1e6, b_pair / 4e12 * 1e6))
lines.append("%-34s %8s %8s   %s" % ("kernel", "us", "us (2nd)", "B_pair / t / 8 TB/s"))
for mode, name in ((1, "(a) arithmetic only"), (2, "(b) loads only"), (3, "(c) both, independent"), (6, "(b') loads, 16 gathers in flight"), (7, "(c') both, 16 gathers in flight"),
                   (802, "(b) with every gather inside the group's own 256 atoms"), (1202, "(b) with every gather inside a 4096-atom window"),
                   (52, "(b) first-round workgroups of a CU adjacent"), (56, "(b') first-round workgroups of a CU adjacent"),
                   (40002, "(b) 4/16 of the coordinate gathers not fetched"), (60002, "(b) 6/16 not fetched"), (80002, "(b) 8/16 not fetched"),
                   (10000002, "(b) 10/16 of every row gathered, the rest read from LDS"), (12000002, "(b) 12/16 gathered, rest from LDS")):
    u = res[mode]
    lines.append("%-34s %8.1f %8.1f   %.3f" % (name, u[0], u[1], b_pair / (min(u) * 1e-6) / 8e12))
lines.append("%-34s %8.1f %8s   %.3f" % ("k_pair_dpd_ring alone (same session)", real_us, "", b_pair / (real_us * 1e-6) / 8e12))
try:
    us8, kept, _ = m.pair_floor(8, a.reps)
    ms_n, calls_n = m.timer("neigh")
    lines.append("the least an incremental list build costs - every stored row (front + back, %.1f entries per atom) walked, one gather and one"
                 % (kept / n))
    lines.append("list-cutoff test per entry, rows written back in place of compaction: %.1f us per pass (the bin scan: see k_tile_build in the kernel statistics)" % us8)
except Exception as e:
    lines.append("refilter floor skipped: %s" % e)
out = "\n".join(lines)
print(out)
if a.out:
    open(a.out, "w").write(out + "\n")
m.close()

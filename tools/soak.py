import sys, time
sys.path.insert(0, ".")
import numpy as np
from meso_amd.api import Meso
from meso_amd.datagen import make_box, make_polymer_box
def run(name, L, poly, steps, chunk):
    m = Meso()
    if poly:
        x, v, types, bonds, lo, hi = make_polymer_box(L, frac=poly)
        m.read_atoms(x, v, lo, hi, types=types, ntypes=2); m.special_bonds(0.0, 1.0, 1.0); m.read_bonds(bonds)
        m.bond_style("harmonic/meso", 1); m.bond_coeff(1, 50.0, 0.5)
    else:
        x, v, lo, hi = make_box(L); m.read_atoms(x, v, lo, hi)
    m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style("dpd/fast/meso", 1.0, 419084618)
    if poly:
        for (i, j), a in {(1, 1): 15.0, (2, 2): 15.0, (1, 2): 40.0}.items(): m.pair_coeff(i, j, a, 4.5, 3.0, 1.0, 1.0)
    else:
        m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
    m.timestep(0.005); m.setup()
    t0 = time.time(); done = 0
    while done < steps:
        m.run(chunk); done += chunk
        T = m.temperature(); info = m.neigh_info()
        print("%s step %6d  T %.4f  avg_count %.2f max_count %d  %.0f steps/s" % (name, done, T, info["avg_count"], info["max_count"], done / (time.time() - t0)), flush=True)
    xg, vg, fg, tag, typ = m.gather()
    print(name, "momentum", np.abs(vg.sum(0)).max(), "tags ok", bool(np.array_equal(tag, np.arange(1, len(tag) + 1))), flush=True)
    m.close()
run("fluid64", 64, 0, 20000, 4000)
run("melt64", 64, 1.0, 6000, 2000)
run("poly128", 128, 0.1, 2000, 1000)

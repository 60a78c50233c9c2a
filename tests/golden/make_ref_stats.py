"""Generates tests/golden/ref_lmp_stats25.json from the REFERENCE ITSELF (oracle/_ref/ref_lmp = the reference's own stock-CPU
sources compiled unmodified, oracle/build_ref.sh): thermostat-ON statistics of the 25^3 rho=4 box (the generator's deck,
62 500 atoms; pair_style dpd 1.0 1.0 419084618, pair_coeff 1 1 15 4.5 -> sigma = 3, neighbor 0.3 bin, every 5, dt 0.005):
temperature, pair energy per atom and pressure sampled every 10 steps over steps 1000-1100.  Must run where /root/reference is
mounted (about 8 minutes on one core).  The fixture is numbers only."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from meso_amd.datagen import make_box  # noqa: E402
from oracle import ref  # noqa: E402


def main():
    assert ref.build(), "reference sources not mounted"
    L = 25
    x, v, lo, hi = make_box(L)
    n = len(x)
    sample = list(range(1000, 1101, 10))
    recs = ref.run(x, v, lo, hi, nsteps=1100, sample=sample, T=1.0, cut=1.0, seed=419084618, coeff=[(1, 1, 15.0, 4.5, 0.0)],
                   every=5, timeout=3600)
    vol = float(np.prod(hi - lo))
    out = {"L": L, "natoms": n, "steps": sample, "T": [], "pe_per_atom": [], "press": [],
           "deck": "make_box(25); pair_style dpd 1.0 1.0 419084618; pair_coeff 1 1 15 4.5; neighbor 0.3 bin; "
                   "neigh_modify delay 0 every 5 check no; timestep 0.005; fix nve; Atom::sort off",
           "source": "oracle/_ref/ref_lmp (reference's own pair_dpd.cpp, neigh_half_bin.cpp, comm.cpp, fix_nve.cpp, random_mars.cpp)"}
    for r in recs:
        ke2 = float((r["v"] ** 2).sum())                       # mass 1
        T = ke2 / (3.0 * n - 3.0)                              # compute temp: dof = 3N - 3
        # compute pressure (src/compute_pressure.cpp:213-231, lj units nktv2p = 1): (dof kT + virial_xx+yy+zz) / (3 V)
        P = ((3.0 * n - 3.0) * T + float(r["virial"][:3].sum())) / (3.0 * vol)
        out["T"].append(T); out["pe_per_atom"].append(float(r["eng_vdwl"]) / n); out["press"].append(P)
    for k in ("T", "pe_per_atom", "press"):
        out["mean_" + k] = float(np.mean(out[k]))
    with open(os.path.join(HERE, "ref_lmp_stats25.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("written: <T> %.5f <PE/atom> %.5f <P> %.4f" % (out["mean_T"], out["mean_pe_per_atom"], out["mean_press"]))


if __name__ == "__main__":
    main()

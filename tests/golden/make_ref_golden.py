"""Generates tests/golden/ref_lmp_*.npz from the REFERENCE ITSELF: oracle/_ref/ref_lmp is the reference's own stock-CPU
sources compiled unmodified (oracle/build_ref.sh); this script must run where /root/reference is mounted.

The fixtures hold inputs and the reference's outputs (data, not source):
  ref_rng.npz          RanMars / RanPark streams for the seeds the input decks use
  ref_lmp_L6.npz       864 atoms, 1 type, the dp.run parameters in their stock CPU form, 20 steps (4 rebuilds)
  ref_lmp_L5_2types.npz  500 atoms, 2 types with different masses and per-pair cutoffs, rebuild every 3 steps
Atom::sort is off in all of them (atom.cpp cannot be compiled unmodified; see oracle/ref_harness.cpp).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from meso_amd.datagen import make_box  # noqa: E402
from oracle import ref  # noqa: E402

CASES = {
    "ref_lmp_L6": dict(L=6, nsteps=20, sample=[0, 1, 5, 10, 20], every=5, T=1.0, cut=1.0, seed=419084618,
                       coeff=[(1, 1, 15.0, 4.5, 0.0)]),
    "ref_lmp_L5_2types": dict(L=5, nsteps=12, sample=[0, 3, 7, 12], every=3, T=0.8, cut=1.0, seed=90210,
                              coeff=[(1, 1, 15.0, 4.5, 0.0), (2, 2, 25.0, 4.5, 0.9), (1, 2, 40.0, 6.0, 1.1)],
                              mass=[1.0, 2.5]),
}


def inputs(c):
    x, v, lo, hi = make_box(c["L"])
    types = None
    if "mass" in c:
        types = (np.arange(len(x)) % 3 == 0).astype(np.int32) + 1
    return x, v, lo, hi, types


def main():
    assert ref.build(), "reference sources not mounted"
    rng = {}
    for kind, seed in (("mars", 419084618), ("mars", 90210), ("park", 788662042), ("park", 1)):
        u, g = ref.rng(kind, seed, 2000)
        rng["%s_%d_uniform" % (kind, seed)] = u
        rng["%s_%d_gaussian" % (kind, seed)] = g
    np.savez_compressed(os.path.join(HERE, "ref_rng.npz"), **rng)
    for name, c in CASES.items():
        x, v, lo, hi, types = inputs(c)
        recs = ref.run(x, v, lo, hi, nsteps=c["nsteps"], sample=c["sample"], T=c["T"], cut=c["cut"], seed=c["seed"],
                       coeff=c["coeff"], every=c["every"], types=types, mass=c.get("mass"))
        out = dict(x0=x, v0=v, lo=lo, hi=hi, steps=np.array([r["step"] for r in recs]),
                   nghost=np.array([r["nghost"] for r in recs]), nneigh=np.array([r["nneigh"] for r in recs]),
                   eng_vdwl=np.array([r["eng_vdwl"] for r in recs]), virial=np.stack([r["virial"] for r in recs]),
                   x=np.stack([r["x"] for r in recs]), v=np.stack([r["v"] for r in recs]),
                   f=np.stack([r["f"] for r in recs]))
        if types is not None:
            out["types"] = types
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "written:", len(x), "atoms,", len(recs), "records")


if __name__ == "__main__":
    main()

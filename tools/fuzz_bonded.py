#!/usr/bin/env python3
"""tools/fuzz_bonded.py [NCASES] [SEED]: random bonded decks (cube edge, fraction of beads in A2B4 chains, special-bond weights, bond
style, pair style) run 23 steps through the default path and through the plain one (every fusion of the rebuild and of the step boundary
switched off): positions, velocities and forces must be equal bit for bit.  Companion of tools/fuzz_paths.py for decks with topology
lists (the streaming gather that moves them, the ghost tiles in its launch, the special-bond filter on two-section rows)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from meso_amd.api import Meso
from meso_amd.datagen import make_polymer_box

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
PLAIN = (("fuse_count", 0), ("merge_ghosts", 0), ("lean_boundary", 0), ("fused_rebuild", 0), ("ghost_epilogue", 0), ("xcd_balance", 0),
         ("row_part", 0), ("async_counts", 0), ("fuse_bonds", 0), ("split_gather", 0))
bad = 0
for case in range(ncases):
    L = int(rng.integers(8, 34))
    frac = float(rng.choice([0.1, 0.4, 1.0]))
    special = tuple(float(s) for s in rng.choice([0.0, 1.0], 3))
    bond = str(rng.choice(["harmonic", "fene"]))
    style = str(rng.choice(["dpd/meso", "dpd/fast/meso"]))
    every = int(rng.choice([1, 3, 5]))
    split = int(rng.choice([-1, 1]))
    x, v, types, bonds, lo, hi = make_polymer_box(L, frac=frac)
    res = []
    # (third run, round 6: the default path with an outgrown ghost capacity planted at a random step - redone inside run())
    k_plant = int(rng.integers(1, 20))
    for idx, opts in enumerate(((("split_gather", split),), PLAIN, (("split_gather", split),))):
        planted = idx == 2
        m = Meso()
        for k, val in opts:
            m.set_option(k, val)
        m.read_atoms(x, v, lo, hi, types=types, ntypes=2)
        m.special_bonds(*special); m.read_bonds(bonds)
        m.bond_style(bond + "/meso", 1)
        if bond == "fene":
            m.bond_coeff(1, 40.0, 1.2, 0.5, 0.4)      # (K R0 epsilon sigma of tests/test_gpu_bonds.py: WCA core below 0.45, bonds start near 0.5)
        else:
            m.bond_coeff(1, 50.0, 0.5)
        m.neighbor(0.3); m.neigh_modify(delay=0, every=every, check=False)
        m.pair_style(style, 1.0, 419084618)
        for (i, j), a in {(1, 1): 15.0, (2, 2): 15.0, (1, 2): 40.0}.items():
            m.pair_coeff(i, j, a, 4.5, 3.0, 1.0, 1.0)
        m.timestep(0.004); m.setup()
        m.force_clear(); m.compute(0, 0); m.bond_compute(0)
        if planted:
            m.run(k_plant)
            m.set_option("debug_ghost_cap", 3)
            m.run(23 - k_plant)
        else:
            m.run(23)
        res.append(m.gather()[:3]); m.close()
    same = all(np.array_equal(a, b) for a, b in zip(res[0], res[1])) and all(np.array_equal(a, b) for a, b in zip(res[0], res[2]))
    fin = bool(np.isfinite(res[0][0]).all())
    print("case %2d  L %2d  n %6d  chains %.1f  special %s  %-8s  every %d  split_gather %2d  %-13s  %s" % (
        case, L, len(x), frac, special, bond, every, split, style, "equal" if same and fin else ("NOT FINITE (deck blew up)" if not fin else "DIFFERENT")), flush=True)
    bad += not (same and fin)
sys.exit(1 if bad else 0)

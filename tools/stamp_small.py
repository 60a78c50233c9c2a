"""tools/stamp_small.py [BOX]: per-phase shader-clock cycles of the ring kernel's waves, step boundary fused and not (run with
MESO_LIB=meso_amd/libmeso_hip_stamp.so from tools/build_variant.sh stamp "-DRG_STAMP -DRG_FEW" pair_ring.hip; profiles/r06_notes.md section 5)."""
import sys, os
sys.path.insert(0, os.getcwd())
from meso_amd.api import Meso
from meso_amd.datagen import make_box
L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for fuse in (1, 0):
    x, v, lo, hi = make_box(L)
    m = Meso()
    m.set_option("fuse_pair", fuse)
    m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style("dpd/fast/meso", 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
    m.setup()
    print("== box %d fuse_pair %d" % (L, fuse), file=sys.stderr, flush=True)
    m.run(400)
    m.close()

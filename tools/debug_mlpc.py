import sys, numpy as np
sys.path.insert(0, '.')
from meso_amd.api import Meso
from meso_amd.datagen import make_box
for L in (12, 16, 25, 32):
    for style in ("dpd/meso", "dpd/fast/meso"):
        res = {}
        for pk in (4, 3):
            x, v, lo, hi = make_box(L)
            m = Meso(); m.set_option("pair_kernel", pk)
            m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
            m.pair_style(style, 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0); m.timestep(0.005)
            m.setup(); m.force_clear(); m.compute()
            f = m.gather()[2]; res[pk] = f
            bad = np.isnan(f).any(axis=1).sum()
            m.close()
        d = np.abs(res[4] - res[3]); 
        print(L, style, "nan rows:", np.isnan(res[3]).any(axis=1).sum(), "maxdiff", np.nanmax(d), "argmax", np.nanargmax(d.max(axis=1)) if not np.isnan(d).all() else -1)

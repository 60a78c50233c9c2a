import itertools, numpy as np, sys
from meso_amd.api import Meso, MesoError
from meso_amd.datagen import make_box
x, v, lo, hi = make_box(9)
def run(style, opts, sigma, steps):
    m = Meso(0)
    for k, val in opts.items(): m.set_option(k, val)
    m.read_atoms(x, v, lo, hi); m.neighbor(0.3); m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style(style, 1.0, 419084618); m.pair_coeff(1, 1, 15.0, 4.5, sigma, 1.0, 1.0); m.timestep(0.005)
    m.setup(); m.run(steps)
    out = m.gather(); T = m.temperature(); m.close()
    return out, T
bad = 0
for style in ("dpd/fast/meso", "dpd/meso"):
    ref, Tref = run(style, {}, 0.0, 12)
    tol = 3e-5 if style == "dpd/fast/meso" else 1e-9
    for layout, pk, nk, fs, fp, sh in itertools.product((0, 1, 2), (0, 1, 2, 3, 4, 5), (0, 1), (0, 1), (0, 1), (0, 1)):
        if (fp, sh) != (1, 1) and (layout, pk) not in ((2, 2), (2, 5)): continue     # ring-only switches
        opts = {"layout": layout, "pair_kernel": pk, "neigh_kernel": nk, "fuse_step": fs, "fuse_pair": fp, "pair_share": sh}
        try:
            out, T = run(style, opts, 0.0, 12)
        except MesoError as e:
            print("ERR", style, opts, e); bad += 1; continue
        d = out[0] - ref[0]; d -= np.round(d / (hi - lo)) * (hi - lo)
        dx, dv = np.abs(d).max(), np.abs(out[1] - ref[1]).max()
        if not (dx < tol and dv < 50 * tol):
            print("MISMATCH", style, opts, dx, dv); bad += 1
print("done, bad =", bad)

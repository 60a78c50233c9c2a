#!/usr/bin/env python3
"""Driver of tools/profile_window_check.sh: a 20-step dpd/meso run with `-profile interval 5 12` (MesoDevice::configure_profiler,
/root/reference/src/USER-MESO/engine_meso.cu:155-177: collection starts when ntimestep == 5 and stops when ntimestep == 12, i.e. the
launches of timesteps 6..12).  Started by rocprofv3 directly (the program after `--`)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from meso_amd.api import Meso
from meso_amd.datagen import make_box

x, v, lo, hi = make_box(12)
with Meso() as m:
    m.profile_window("interval", 5, 12)
    m.read_atoms(x, v, lo, hi)
    m.neighbor(0.3)
    m.neigh_modify(delay=0, every=5, check=False)
    m.pair_style("dpd/meso", 1.0, 419084618)
    m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
    m.timestep(0.005)
    m.setup()
    m.run(20)
    print("ran", m.ntimestep, "steps, T", m.temperature())

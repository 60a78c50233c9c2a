#!/usr/bin/env python3
"""Child process of tests/test_gpu_rccl_branch.py, started with LD_PRELOAD=tests/c/librccl_stand_in.so: several in-process ranks
(one host thread + one engine context each) run the engine's RCCL branch - meso_comm_init(..., "rccl", ncclUniqueId) and the grouped
send / receive schedule of Engine::xchg - against the in-process stand-in, and the same deck again over the LOCAL transport; the two
runs must agree bit for bit (same kernels, same message schedule; only the copy primitive differs).

    rccl_stand_in_run.py NRANKS GX GY GZ L STYLE STEPS [key=value ...]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402  (before libmeso_hip.so: the HIP runtime loaded first serves both)
from meso_amd.api import Meso, nccl_unique_id  # noqa: E402
from meso_amd.datagen import make_box  # noqa: E402


EXPECT_DIFFER = any(kv.startswith("expect_differ=") for kv in sys.argv[8:])


def run(nranks, grid, L, style, steps, transport, opts):
    x, v, lo, hi = make_box(L)
    uid = nccl_unique_id() if transport == "rccl" else np.frombuffer(np.random.default_rng(7 * nranks + L).bytes(8), np.uint8)
    out, errs = [None] * nranks, []

    def work(r):
        try:
            m = Meso()
            for k, val in opts:
                m.set_option(k, val)
            m.comm_init(nranks, r, grid, transport, uid)
            m.read_atoms(x, v, lo, hi)
            m.neighbor(0.3)
            m.neigh_modify(delay=0, every=5, check=False)
            m.pair_style(style, 1.0, 419084618)
            m.pair_coeff(1, 1, 15.0, 4.5, 3.0, 1.0, 1.0)
            m.timestep(0.005)
            m.setup()
            m.run(steps)
            out[r] = (m.gather(by_tag=False), m.counts(), m.comm_count(), m.xchg_stats())
            m.close()
        except Exception as e:   # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(nranks)]
    [t.start() for t in th]
    t_end = time.time() + 240
    while any(t.is_alive() for t in th) and time.time() < t_end and not errs:
        time.sleep(0.05)
    if (errs or any(o is None for o in out)) and EXPECT_DIFFER and transport == "rccl":
        print("PLANTED HAZARD SEEN: the run ended with", errs)
        sys.stdout.flush()
        os._exit(0)
    if errs or any(o is None for o in out):
        print("FAILED", transport, errs)
        sys.stdout.flush()
        os._exit(2)
    cols = [np.concatenate([o[0][k] for o in out]) for k in range(4)]
    order = np.argsort(cols[3], kind="stable")
    return [c[order] for c in cols], [o[1] for o in out], [o[2] for o in out], [o[3] for o in out]


def main():
    nranks, gx, gy, gz, L = (int(t) for t in sys.argv[1:6])
    style, steps = sys.argv[6], int(sys.argv[7])
    opts = [(kv.split("=")[0], float(kv.split("=")[1])) for kv in sys.argv[8:]]
    # expect_differ=1: a hazard planted on purpose (option debug_early_reuse) must show as a trajectory that differs from the LOCAL one
    expect_differ = bool(dict(opts).pop("expect_differ", 0)) if any(k == "expect_differ" for k, _ in opts) else False
    opts = [(k, val) for k, val in opts if k != "expect_differ"]
    a, ca, na, xa = run(nranks, (gx, gy, gz), L, style, steps, "rccl", opts)
    b, cb, _, _ = run(nranks, (gx, gy, gz), L, style, steps, "local", [(k, val) for k, val in opts if k != "debug_early_reuse"])
    if expect_differ:
        same = all(np.array_equal(a[k], b[k], equal_nan=True) for k in range(3))
        print("PLANTED HAZARD %s" % ("NOT SEEN" if same else "SEEN: trajectories differ"))
        return
    if dict(opts).get("profile"):
        # option profile: every RCCL group sits between two HIP events on the exchange stream and is booked per kind of exchange
        # (meso_xchg_stats; the host / in-process transports keep a host-side account instead)
        for r, st in enumerate(xa):
            assert st and all(v["calls"] > 0 and v["ms_wire"] > 0.0 and v["bytes"] > 0 for v in st.values()), (r, st)
        print("exchange kinds timed on rank 0:", {k: (v["calls"], round(v["ms_wire"], 3)) for k, v in xa[0].items()})
    assert all(n == nranks for n in na), na                      # ncclCommCount of every rank's communicator
    assert ca == cb, (ca, cb)
    n = 4 * L ** 3
    assert np.array_equal(a[3], np.arange(1, n + 1)), "an atom was lost or duplicated"
    for k in range(3):
        assert np.array_equal(a[k], b[k]), "RCCL branch and LOCAL transport differ in array %d" % k
    print("OK ranks %d steps %d: x, v, f bit-identical over both transports; ghosts per rank %s" % (nranks, steps, [c[1] for c in ca]))


if __name__ == "__main__":
    main()

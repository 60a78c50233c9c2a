"""Build libmeso_hip.so for gfx950 with hipcc (in-tree; the .so travels to the GPU box)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libmeso_hip.so")
SOURCES = ["pair_ring.hip", "pair_ring_dp.hip", "pair_floor.hip", "kernels.hip", "brick.hip", "rebuild.hip", "bond.hip", "sort.hip", "engine.hip", "restart.hip", "comm.hip", "script.hip", "capi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result"]


def _deps(src):
    d = [os.path.join(CSRC, src)]
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            d.append(os.path.join(CSRC, f))
    d.append(os.path.join(HERE, "..", "include", "meso_hip.h"))
    if src == "pair_ring_dp.hip":
        d.append(os.path.join(CSRC, "pair_ring.hip"))
    return d


# per-file flags.  pair_ring.hip: the SLP vectoriser packs two of the three coordinate differences of a distance into v_pk_*_f32
# and pays two v_mov per entry for the register pairs (7 instead of 6 instructions, profiles/r03_isa_pair_ring.txt): 103 -> 101 us
EXTRA = {"pair_ring.hip": ["-fno-slp-vectorize"], "pair_ring_dp.hip": ["-fno-slp-vectorize"]}


def _compile(src):
    obj = os.path.join(OBJ, src + ".o")
    if os.path.exists(obj) and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in _deps(src)):
        return obj, False
    cmd = ["hipcc"] + FLAGS + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=5) as ex:
        res = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lrccl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

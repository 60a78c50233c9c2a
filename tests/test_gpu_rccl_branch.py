"""The engine's RCCL branch with SEVERAL ranks on the one GPU of the test box.

RCCL itself refuses two ranks on one device (tests/test_gpu_multiproc.py records the refusal), so Engine::comm_init("rccl") and the
grouped ncclSend / ncclRecv schedule of Engine::xchg (meso_amd/csrc/comm.hip; replaces MesoComm::borders / exchange /
forward_comm, /root/reference/src/USER-MESO/comm_meso.cu:41-186,256-550) never executed with more than one rank before round 4.
Here a child process preloads tests/c/librccl_stand_in.so - the nine librccl entry points the engine calls, re-implemented for
in-process ranks with RCCL's matching rules (per-pair FIFO, equal byte counts, grouped posting) - and runs the 2x2x2 and 2x1x1
decks over transport "rccl" and again over "local": trajectories must be bit-identical.  What this does NOT cover is RCCL itself
(its kernels, its IPC set-up): that needs the driver's multi-GPU node.  Nor does it cover stream-ordering or buffer-reuse hazards of an
asynchronous transport IN ITS DEFAULT MODE: there the stand-in synchronises the stream before it posts and after every copy
(tests/c/rccl_stand_in.cpp), so what the comparison validates is the sizes, offsets, order and matching of the engine's messages.  Round 6:
a second, stream-ordered mode (RCCL_STAND_IN_ASYNC=1) enqueues the copies on the callers' streams behind events - RCCL's semantics - and
every deck runs in both."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

LIB = os.path.join(ROOT, "tests", "c", "librccl_stand_in.so")


def _build():
    src = os.path.join(ROOT, "tests", "c", "rccl_stand_in.cpp")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O1", "-shared", "-fPIC", "-std=c++17", src, "-o", LIB], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize("nranks,grid,L,style,steps,opts", [
    (8, (2, 2, 2), 12, "dpd/meso", 23, ()),                        # 4 rebuilds with migration, refresh received straight into the merged arrays
    (8, (2, 2, 2), 12, "dpd/fast/meso", 23, ()),
    (2, (2, 1, 1), 10, "dpd/fast/meso", 12, ()),                   # a rank that is its own neighbour in y and z
    (8, (2, 2, 2), 12, "dpd/meso", 12, ("refresh_direct=0", "refresh_epilogue=0")),     # pack / scatter kernels around the exchange
    (4, (2, 2, 1), 12, "dpd/meso", 12, ("async_counts=0",)),       # the synchronous two-phase border exchange
    (8, (2, 2, 2), 12, "dpd/fast/meso", 12, ("profile=1",)),       # every RCCL group between two HIP events: exchange times per kind
    (8, (2, 2, 2), 12, "dpd/fast/meso", 200, ("overlap=1", "refresh_epilogue=1")),      # 40 rebuilds: refresh on the side stream under the bulk launch
])
@pytest.mark.parametrize("mode", ["host-synchronous", "stream-ordered"])
def test_rccl_branch_equals_local_transport(nranks, grid, L, style, steps, opts, mode):
    _build()
    # (stream-ordered mode: every receive held back by 50 us on the receiver's stream - transfers that arrive late, as over a real link)
    env = dict(os.environ, LD_PRELOAD=LIB, HSA_ENABLE_IPC_MODE_LEGACY="0", RCCL_STAND_IN_ASYNC="1" if mode == "stream-ordered" else "0",
               RCCL_STAND_IN_DELAY_US="50")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "rccl_stand_in_run.py"), str(nranks)] + [str(g) for g in grid] + [str(L), style, str(steps)] + list(opts)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "OK ranks" in r.stdout, (r.stdout + r.stderr)[-3000:]
    if "profile=1" in opts:
        assert "exchange kinds timed on rank 0" in r.stdout, r.stdout[-2000:]


def test_stream_ordered_stand_in_sees_a_send_buffer_reused_too_early():
    """The stand-in's second mode (RCCL_STAND_IN_ASYNC=1, tests/c/rccl_stand_in.cpp run_ops_async) enqueues its copies on the callers'
    streams behind events, as RCCL does: a send staging that is rewritten from an unordered stream right behind the exchange (option
    debug_early_reuse, planted for this test) corrupts what the peers receive and the trajectory differs from the LOCAL transport's -
    while the host-synchronous mode, which has finished every transfer before ncclGroupEnd returns, cannot see the hazard."""
    _build()
    seen = {}
    for mode in ("0", "1"):
        # (every receive held back by 2 ms: the scribble, posted right behind the exchange on a stream that waits for nothing, has
        # certainly run when the bytes move - the outcome does not depend on how the race goes)
        env = dict(os.environ, LD_PRELOAD=LIB, HSA_ENABLE_IPC_MODE_LEGACY="0", RCCL_STAND_IN_ASYNC=mode, RCCL_STAND_IN_DELAY_US="2000")
        cmd = [sys.executable, os.path.join(ROOT, "tests", "rccl_stand_in_run.py"), "8", "2", "2", "2", "12", "dpd/fast/meso", "23",
               "debug_early_reuse=1", "expect_differ=1"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0 and "PLANTED HAZARD" in r.stdout, (r.stdout + r.stderr)[-3000:]
        seen[mode] = "SEEN" in r.stdout and "NOT SEEN" not in r.stdout
    assert seen["1"] and not seen["0"], seen

"""HOST transport over torch.distributed point-to-point calls (any backend that moves CPU tensors, i.e. gloo).

The engine's HOST transport (include/meso_hip.h: meso_comm_set_host_exchange) stages every message through host buffers
and hands the exchange to a caller-supplied function.  This is the debugging / rehearsal transport: several ranks may
share one GPU (RCCL refuses that), so the multi-process flow of bench.py and the decomposition can be exercised on a
one-GPU box.  The production transport is RCCL (transport "rccl")."""
from __future__ import annotations

import ctypes as C

from . import _lib


def make_exchange(dist, rank):
    """Returns the ctypes callback (keep a reference to it for the lifetime of the context)."""
    import torch

    def exchange(_user, npeer, peer, sendbuf, sendbytes, recvbuf, recvbytes):
        try:
            reqs, recvs = [], []
            for k in range(npeer):
                p, ns, nr = peer[k], sendbytes[k], recvbytes[k]
                if p == rank:
                    if nr:
                        C.memmove(recvbuf[k], sendbuf[k], nr)
                    continue
                if nr:
                    t = torch.empty(nr, dtype=torch.uint8)
                    reqs.append(dist.irecv(t, src=p))
                    recvs.append((k, t))
                if ns:
                    s = torch.frombuffer((C.c_ubyte * ns).from_address(sendbuf[k]), dtype=torch.uint8).clone()
                    reqs.append(dist.isend(s, dst=p))
            for r in reqs:
                r.wait()
            for k, t in recvs:
                C.memmove(recvbuf[k], t.data_ptr(), t.numel())
            return 0
        except Exception as e:      # noqa: BLE001 - reported through the engine's status code
            print("host exchange failed:", repr(e), flush=True)
            return 1

    return _lib.HOST_EXCHANGE_FN(exchange)

"""Host-staged transport for several ranks that cannot use RCCL among themselves (processes sharing one GPU in tests
and in `bench.py --gpus N` on a box with fewer devices; MPI hosts without GPU-aware transport): the callbacks of
`tsx_comm_set_callbacks` on top of a torch.distributed process group (gloo).  Message matching follows
`exchange_diffuse_boundary` (src/pprts_explicit.F90:769-843): recv[W] <- peer W's send[E], recv[E] <- peer E's send[W],
recv[S] <- peer S's send[N], recv[N] <- peer N's send[S]."""
from __future__ import annotations

import numpy as np


def attach(solver, rank: int, group=None):
    """install exchange / allreduce callbacks on a DiffuseSolver / PprtsSolver; the default process group must exist
    (group: a gloo group to use instead, e.g. beside a default group on the nccl backend)"""
    import torch
    import torch.distributed as dist

    def exchange(send, recv, peers):
        want_tag = [1, 0, 3, 2]  # the tag is the sender's face index: W = 0, E = 1, S = 2, N = 3
        reqs, keep = [], []
        for q in range(4):
            if len(recv[q]) == 0 or peers[q] == rank:
                continue
            t = torch.from_numpy(recv[q])
            keep.append(t)
            reqs.append(dist.irecv(t, src=peers[q], tag=want_tag[q], group=group))
        for q in range(4):
            if len(send[q]) == 0 or peers[q] == rank:
                continue
            t = torch.from_numpy(np.array(send[q], copy=True))
            keep.append(t)
            reqs.append(dist.isend(t, dst=peers[q], tag=q, group=group))
        for q in range(4):  # self neighbours
            if len(recv[q]) and peers[q] == rank:
                recv[q][...] = send[q ^ 1]
        for r in reqs:
            r.wait()

    def allreduce(buf):
        dist.all_reduce(torch.from_numpy(buf), group=group)

    solver.comm_set_callbacks(exchange, allreduce)


def attach_peer(solver):
    """device-resident peer transport (tsx_peer.hip): the mailboxes' IPC handles are all-gathered over the default process
    group (any backend); after that no exchange touches the host"""
    import torch.distributed as dist

    def allgather(blob):
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, blob)
        return out

    solver.comm_peer_init(allgather)


def attach_peer_checked(solver, rounds=64):
    """attach_peer + the transport's self test, agreed over the process group: True if every rank passed (the transport stays),
    False if any rank failed (it is disabled on every rank; the caller attaches RCCL or the host-staged callbacks instead)"""
    import torch
    import torch.distributed as dist

    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"

    def agreed(ok):
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item()) >= 1.0

    attached = True
    try:
        attach_peer(solver)
    except Exception:   # attach refused (no IPC between these ranks ...): every rank still takes part in the agreements below
        attached = False
    if not agreed(attached):
        if attached:
            solver.comm_peer_disable()
        return False
    # first with the light ordering that uncached mailboxes allow, then -- should a rank have seen stale or missing data -- once
    # more with full system-scope fences (tsx_peer_dev.hpp) before the transport is given up
    for heavy in (0, 1):
        try:
            if heavy:
                dist.barrier()          # nobody still sends
                solver.comm_peer_reset()
                solver.comm_peer_set_fences(1)
                dist.barrier()          # nobody sends into a mailbox that is being cleared
            ok = solver.comm_peer_selftest(rounds) == 0.0
        except Exception:
            ok = False
        if agreed(ok):
            return True
    try:
        solver.comm_peer_disable()
    except Exception:
        pass
    return False

"""N > 1 path on CPU: two/four `gloo` ranks shard a periodic domain with tenstream_amd.coord (the
restatement of setup_coord_native), exchange halos the way halo_fill_5pt / halo_reduce_5pt do
(src/pprts_base.F90:1622-1731) and apply the oracle's *local* operator; the gathered result must equal
the single-rank operator.  This pins the decomposition, the neighbour table and the message semantics
the RCCL path implements (the HIP halo kernels themselves are covered by the force_halo GPU tests)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, Nx, Ny, Nz, ret):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import oracle as O
    from tenstream_amd import coord, synthetic

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz, n1d=1)
        co = coord.coord(rank, world, Nx, Ny)
        sl = (slice(co.ys, co.ys + co.ym), slice(co.xs, co.xs + co.xm))
        lay = O.layout("3_10", Nz, co.xm, co.ym)
        c64 = np.ascontiguousarray(P["coeff"][sl].astype(np.float64))
        a11, a12, alb = (np.ascontiguousarray(P[k][sl]) for k in ("a11", "a12", "albedo"))
        xg = np.random.default_rng(7).standard_normal((Ny, Nx, Nz + 1, 10))
        x = np.ascontiguousarray(xg[sl])

        def exchange(send_w, send_e, send_s, send_n):
            """returns (from_w, from_e, from_s, from_n); tags keep same-peer messages apart"""
            outs = {}
            reqs = []
            bufs = {"w": np.empty_like(send_e), "e": np.empty_like(send_w), "s": np.empty_like(send_n), "n": np.empty_like(send_s)}
            plan = [(co.west, send_w, 1), (co.east, send_e, 2), (co.south, send_s, 3), (co.north, send_n, 4)]
            recv_plan = [(co.west, "w", 2), (co.east, "e", 1), (co.south, "s", 4), (co.north, "n", 3)]
            tens = {}
            for peer, key, tag in recv_plan:
                tens[key] = torch.from_numpy(bufs[key])
                if peer == rank:
                    continue
                reqs.append(dist.irecv(tens[key], src=peer, tag=tag))
            for peer, arr, tag in plan:
                if peer == rank:
                    continue
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(arr)), dst=peer, tag=tag))
            for r in reqs:
                r.wait()
            # self neighbours (1 rank along an axis): periodic wrap
            if co.west == rank:
                bufs["w"][...] = send_e
                bufs["e"][...] = send_w
            if co.south == rank:
                bufs["s"][...] = send_n
                bufs["n"][...] = send_s
            return bufs["w"], bufs["e"], bufs["s"], bufs["n"]

        # halo_fill_5pt: owner -> ghost
        lx = np.zeros((co.ym + 2, co.xm + 2, Nz + 1, 10))
        lx[1:-1, 1:-1] = x
        fw, fe, fs, fn = exchange(x[:, 0], x[:, -1], x[0, :], x[-1, :])
        lx[1:-1, 0], lx[1:-1, -1], lx[0, 1:-1], lx[-1, 1:-1] = fw, fe, fs, fn
        lb = O.op_local(lay, c64, P["l1d"], a11, a12, alb, lx)
        # halo_reduce_5pt: ghost -> owner ADD
        fw, fe, fs, fn = exchange(lb[1:-1, 0], lb[1:-1, -1], lb[0, 1:-1], lb[-1, 1:-1])
        y = lb[1:-1, 1:-1].copy()
        y[:, 0] += fw
        y[:, -1] += fe
        y[0, :] += fs
        y[-1, :] += fn
        y += x
        # reference: the single-rank operator on the global domain
        layg = O.layout("3_10", Nz, Nx, Ny)
        yg = O.diff_apply(layg, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], P["albedo"], xg)
        err = float(np.abs(y - yg[sl]).max() / np.abs(yg).max())
        ret[rank] = err
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,Nx,Ny", [(2, 6, 8), (4, 8, 6)])
def test_sharded_operator_equals_global(world, Nx, Ny):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, Nx, Ny, 5, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=240)
        for p in procs:
            assert p.exitcode == 0
        assert len(ret) == world and max(ret.values()) < 1e-14, dict(ret)


def _hostcomm_worker(rank, world, port, use_group, ret):
    """the host-staged callbacks of tenstream_amd.hostcomm on a 2 x 1 periodic process grid (W and E neighbour are the same
    rank, S and N the rank itself), and comm_peer_init's all-gather when one rank cannot export its mailbox"""
    sys.path.insert(0, ROOT)
    import ctypes as C

    import torch.distributed as dist

    from tenstream_amd import _lib, hostcomm
    from tenstream_amd.solver import DiffuseSolver

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        class Fake:
            nranks = world
            h = None

            def comm_set_callbacks(self, exchange, allreduce):
                self.exchange, self.allreduce = exchange, allreduce

        f = Fake()
        hostcomm.attach(f, rank, group=dist.new_group(backend="gloo") if use_group else None)
        other = 1 - rank
        peers = [other, other, rank, rank]
        send = [np.full(5 + q, 100.0 * rank + q) for q in range(2)] + [np.full(3, 100.0 * rank + q) for q in (2, 3)]
        recv = [np.zeros(6), np.zeros(5), np.zeros(3), np.zeros(3)]   # recv[W] <- peer's send[E] (6 long), recv[E] <- peer's send[W]
        f.exchange(send, recv, peers)
        ok = (np.all(recv[0] == 100.0 * other + 1) and np.all(recv[1] == 100.0 * other + 0) and np.all(recv[2] == 100.0 * rank + 3)
              and np.all(recv[3] == 100.0 * rank + 2))
        v = np.array([1.0 + rank, 10.0])
        f.allreduce(v)
        ok = ok and v[0] == 3.0 and v[1] == 20.0

        # comm_peer_init: rank 1's export fails; both ranks must leave the all-gather, and both must refuse the transport
        class FakeLib:
            def tsx_comm_peer_export(self, h, buf):
                if rank == 1:
                    return 7
                C.memset(buf, 0x5A, _lib.PEER_BLOB_BYTES)
                return 0

            def tsx_comm_peer_attach(self, h, blobs):
                raise AssertionError("attach must not be reached")

        f.lib = FakeLib()

        def allgather(blob):
            out = [None] * world
            dist.all_gather_object(out, blob)
            return out

        try:
            DiffuseSolver.comm_peer_init(f, allgather)
            ok = False
        except (_lib.TsxError, RuntimeError) as e:
            ok = ok and isinstance(e, _lib.TsxError if rank == 1 else RuntimeError)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_group", [False, True])
def test_hostcomm_callbacks_and_peer_bootstrap_failure(use_group):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_hostcomm_worker, args=(r, 2, port, use_group, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=240)
        for p in procs:
            assert p.exitcode == 0
        assert dict(ret) == {0: True, 1: True}, dict(ret)

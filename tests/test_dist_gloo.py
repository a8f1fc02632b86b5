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

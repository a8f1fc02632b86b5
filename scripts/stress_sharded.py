"""Stress of the sharded g-point pipeline (GPU box): 4 ranks on cuda:0, the same solar g-point solved REPS times from a zero guess;
every repetition must reproduce the first one.  usage: python scripts/stress_sharded.py [host|peer] [REPS]
Prints, per deviating repetition and rank: iterations of the direct sweep / diffuse solve and where the result differs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, transport, reps, ret):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from tenstream_amd import coord, lut, synthetic
    from tenstream_amd.pprts import PprtsSolver

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Nx, Ny, Nz, phi0, theta0, tall_top = 12, 10, 8, 30.0, 55.0, 1
    kabs, ksca, g = synthetic.cloud_field(Nx, Ny, Nz, seed=5)
    kabs *= 20.0
    dz = np.full((Ny, Nx, Nz), 50.0)
    dz[:, :, :tall_top] = 400.0
    planck = np.linspace(2.0, 6.0, Nz + 1)[None, None, :] * (1 + 0.05 * np.random.default_rng(0).random((Ny, Nx, 1)))
    dax = lut.direct_axes()
    Tdir, Sdir = lut.synthetic_direct_tables(dax)
    co = coord.coord(rank, world, Nx, Ny)
    sl = (slice(co.ys, co.ys + co.ym), slice(co.xs, co.xs + co.xm))
    P = PprtsSolver(Nz, co.xm, co.ym, 100.0, 100.0, phi0, theta0, device=0, xs=co.xs, ys=co.ys, glob_xm=Nx, glob_ym=Ny, rank=rank,
                    nranks=world, neighbors=(co.west, co.east, co.south, co.north))
    P.set_lut_diffuse(lut.synthetic_diffuse_table("3_10"), lut.diffuse_axes("3_10"))
    P.set_lut_direct(Tdir, Sdir, dax)

    def exchange(send, recv, peers):
        want_tag = [1, 0, 3, 2]
        reqs, keep = [], []
        for q in range(4):
            if len(recv[q]) == 0 or peers[q] == rank:
                continue
            t = torch.from_numpy(recv[q])
            keep.append(t)
            reqs.append(dist.irecv(t, src=peers[q], tag=want_tag[q]))
        for q in range(4):
            if len(send[q]) == 0 or peers[q] == rank:
                continue
            t = torch.from_numpy(np.array(send[q], copy=True))
            keep.append(t)
            reqs.append(dist.isend(t, dst=peers[q], tag=q))
        for q in range(4):
            if len(recv[q]) and peers[q] == rank:
                recv[q][...] = send[q ^ 1]
        for r in reqs:
            r.wait()

    def allreduce(buf):
        dist.all_reduce(torch.from_numpy(buf))

    from tenstream_amd import hostcomm

    if transport == "peer":
        hostcomm.attach_peer(P.core)
    else:
        P.core.comm_set_callbacks(exchange, allreduce)
    loc = lambda a: np.ascontiguousarray(a[sl])
    first, bad = None, []
    if os.environ.get("STRESS_FRESH"):   # a fresh 4-rank solver and a fresh one-rank periodic solver (rank 0) per repetition, first solves only
        P.close()
        firstG = None
        for rep in range(reps):
            P = PprtsSolver(Nz, co.xm, co.ym, 100.0, 100.0, phi0, theta0, device=0, xs=co.xs, ys=co.ys, glob_xm=Nx, glob_ym=Ny, rank=rank,
                            nranks=world, neighbors=(co.west, co.east, co.south, co.north))
            P.set_lut_diffuse(lut.synthetic_diffuse_table("3_10"), lut.diffuse_axes("3_10"))
            P.set_lut_direct(Tdir, Sdir, dax)
            if transport == "peer":
                hostcomm.attach_peer(P.core)
            else:
                P.core.comm_set_callbacks(exchange, allreduce)
            P.set_optical_properties(0.15, loc(kabs), loc(ksca), loc(g), loc(dz))
            info = P.solve(1000.0, rtol=1e-10, atol=1e-30, maxit=3000)
            res = [np.array(a, copy=True) for a in P.get_result()]
            P.set_optical_properties(0.15, loc(kabs), loc(ksca), loc(g), loc(dz), planck=loc(planck))   # the test's thermal g-point
            P.solve(0.0, rtol=1e-10, atol=1e-30, maxit=3000)
            P.get_result()
            if rank == 0:
                G = PprtsSolver(Nz, Nx, Ny, 100.0, 100.0, phi0, theta0, device=0)
                G.set_lut_diffuse(lut.synthetic_diffuse_table("3_10"), lut.diffuse_axes("3_10"))
                G.set_lut_direct(Tdir, Sdir, dax)
                G.set_optical_properties(0.15, kabs, ksca, g, dz)
                gi = G.solve(1000.0, rtol=1e-10, atol=1e-30, maxit=3000)
                gres = [np.array(a, copy=True) for a in G.get_result()]
                G.close()
                if firstG is None:
                    firstG = (gres, gi.niter)
                dg = [float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)) for a, b in zip(gres, firstG[0])]
                if max(dg) > 1e-6 or gi.niter != firstG[1]:
                    bad.append(("ONE-RANK", rep, gi.niter, firstG[1], dg))
            if first is None:
                first = (res, info.niter)
            dev = [float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)) for a, b in zip(res, first[0])]
            if max(dev) > 1e-6 or info.niter != first[1]:
                d = np.abs(res[3] - first[0][3])
                bad.append(("SHARDED", rep, info.niter, first[1], dev, np.argwhere(d > 1e-3 * np.abs(first[0][3]).max())[:12].tolist()))
            P.close()
        ret[rank] = bad
        dist.destroy_process_group()
        return
    for rep in range(reps):
        P.set_optical_properties(0.15, loc(kabs), loc(ksca), loc(g), loc(dz))
        info = P.solve(1000.0, rtol=1e-10, atol=1e-30, maxit=3000, zero_guess=True)
        res = [np.array(a, copy=True) for a in P.get_result()]
        nd = P.core.niter_direct() if hasattr(P.core, "niter_direct") else -1
        if first is None:
            first = (res, info.niter, nd)
            continue
        dev = [float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)) for a, b in zip(res, first[0])]
        if max(dev) > 1e-6 or info.niter != first[1] or nd != first[2]:
            d = np.abs(res[3] - first[0][3])
            bad.append((rep, info.niter, first[1], nd, first[2], dev, np.argwhere(d > 1e-3 * np.abs(first[0][3]).max())[:12].tolist()))
    ret[rank] = bad
    P.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket

    import torch.multiprocessing as mp

    transport = sys.argv[1] if len(sys.argv) > 1 else "host"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        ps = [ctx.Process(target=worker, args=(r, 4, port, transport, reps, ret)) for r in range(4)]
        for p in ps:
            p.start()
        for p in ps:
            p.join(timeout=900)
        out = dict(ret)
    print(transport, reps, "repetitions; exit codes", [p.exitcode for p in ps])
    for r in sorted(out):
        print("rank", r, "deviating repetitions:", len(out[r]))
        for b in out[r][:6]:
            print("   ", b)

"""Multi-rank solve on ONE GPU: 2/4 processes share cuda:0, each owns one x/y block of a periodic domain and
runs the HIP path with the host-staged callback transport (gloo underneath).  Checks the sharded operator and
the sharded preconditioned BiCGStab against the single-rank oracle on the global domain.  (The RCCL transport
itself is exercised by test_rccl_transport_with_one_rank_communicator; 8-GPU runs belong to the driver.)"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _guarded(fn, rank, *args):
    """run a worker; report its exception text through the shared dict instead of only an exit code"""
    ret = args[-1]
    try:
        fn(rank, *args)
    except Exception:  # noqa: BLE001
        import traceback

        ret[("error", rank)] = traceback.format_exc()
        raise


def _spawn(fn, world, args):
    """world processes on cuda:0, one attempt: a rank that fails or hangs fails the test (no silent retry -- a flaky
    halo-ordering or tag bug must show); the numerical assertions are made by the caller on the returned results."""
    import time

    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    with ctx.Manager() as m:
        ret = m.dict()
        port = _free_port()
        procs = [ctx.Process(target=_guarded, args=(fn, r, world, port) + tuple(args) + (ret,)) for r in range(world)]
        for p in procs:
            p.start()
        deadline = time.time() + 300
        while any(p.is_alive() for p in procs) and time.time() < deadline:
            if any(p.exitcode not in (None, 0) for p in procs):
                break  # one rank died: the others would wait in a collective for ever
            time.sleep(0.05)
        for p in procs:
            if p.is_alive():
                p.join(timeout=2)
            if p.is_alive():
                p.kill()
                p.join()
        codes = [p.exitcode for p in procs]
        out = dict(ret)
        errs = {k: v for k, v in out.items() if isinstance(k, tuple) and k[0] == "error"}
        assert all(c == 0 for c in codes) and not errs and len(out) == world, f"workers failed: exit codes {codes}, errors {errs}"
        return out


def _worker(rank, world, port, solver, Nx, Ny, Nz, transport, ret):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("TSX_PEER_TIMEOUT_S", "10")
    import torch
    import torch.distributed as dist

    from oracle import oracle as O
    from tenstream_amd import DiffuseSolver, coord, synthetic

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=1)
        if transport.endswith("+distinct"):   # every cell a block of its own (0.1 % apart): nothing bit-identical to share, the
            transport = transport.split("+")[0]   # preconditioner groups near-identical blocks, each rank for itself
            P["coeff"] = (P["coeff"] * (1 + 1e-3 * np.random.default_rng(5).random(P["coeff"].shape))).astype(np.float32)
        co = coord.coord(rank, world, Nx, Ny)
        sl = (slice(co.ys, co.ys + co.ym), slice(co.xs, co.xs + co.xm))
        s = DiffuseSolver(solver, Nz, co.xm, co.ym, xs=co.xs, ys=co.ys, glob_xm=Nx, glob_ym=Ny, rank=rank, nranks=world,
                          neighbors=(co.west, co.east, co.south, co.north), device=0)

        def exchange(send, recv, peers):
            want_tag = [1, 0, 3, 2]  # recv[W] <- peer W's send[E] (tag 1), recv[E] <- tag 0, recv[S] <- tag 3, recv[N] <- tag 2
            reqs, keep = [], []
            for q in range(4):
                if len(recv[q]) == 0:
                    continue
                if peers[q] == rank:
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
            for q in range(4):  # self neighbours
                if len(recv[q]) and peers[q] == rank:
                    recv[q][...] = send[q ^ 1]
            for r in reqs:
                r.wait()

        def allreduce(buf):
            t = torch.from_numpy(buf)
            dist.all_reduce(t)

        if transport == "peer":   # device-resident: IPC-mapped mailboxes, no host in any exchange (tsx_peer.hip)
            from tenstream_amd import hostcomm

            hostcomm.attach_peer(s)
        else:
            s.comm_set_callbacks(exchange, allreduce)
        loc = lambda k: np.ascontiguousarray(P[k][sl])
        s.set_coeffs(loc("coeff"), P["l1d"], loc("a11"), loc("a12"), loc("albedo"))
        layg = O.layout(solver, Nz, Nx, Ny)
        c64 = P["coeff"].astype(np.float64)
        xg = np.random.default_rng(11).standard_normal((Ny, Nx, Nz + 1, s.D))
        y = s.apply(np.ascontiguousarray(xg[sl]))
        if solver == "3_10":
            yg = O.diff_apply(layg, c64, P["l1d"], P["a11"], P["a12"], P["albedo"], xg)
        else:
            A = O.assemble_csr(layg, c64, P["l1d"], P["a11"], P["a12"], P["albedo"])
            yg = (A @ xg.ravel()).reshape(xg.shape)
        e_apply = float(np.abs(y - yg[sl]).max() / np.abs(yg).max())

        import scipy.sparse.linalg as spla

        A = O.assemble_csr(layg, c64, P["l1d"], P["a11"], P["a12"], P["albedo"])
        x_ref = spla.spsolve(A.tocsc(), P["b"].ravel()).reshape(P["b"].shape)
        x = np.zeros(s.vec_shape)
        info = s.solve(np.ascontiguousarray(P["b"][sl]), x, rtol=1e-10, atol=1e-30, pc=1, pc_sweeps=1)
        e_solve = float(np.abs(x - x_ref[sl]).max() / np.abs(x_ref).max())
        # and with the library's default preconditioner (red-black where the local block allows it, else zebra rows;
        # couplings across rank faces are dropped from M, the fixed point is the same)
        x2 = np.zeros(s.vec_shape)
        info2 = s.solve(np.ascontiguousarray(P["b"][sl]), x2, rtol=1e-10, atol=1e-30)
        e_solve = max(e_solve, float(np.abs(x2 - x_ref[sl]).max() / np.abs(x_ref).max()))
        assert info2.reason == 2, info2
        # the explicit (stationary) solver on several ranks: same fixed point; its residual is the mean over ranks of the
        # local change norms (imp_allreduce_mean, src/pprts_explicit.F90:620), so every rank reports the same history
        x3 = np.zeros(s.vec_shape)
        info3 = s.solve(np.ascontiguousarray(P["b"][sl]), x3, explicit_solver=1, rtol=1e-11, atol=1e-30, maxit=3000, pc_sweeps=5)
        assert info3.reason == 2, info3
        e_solve = max(e_solve, float(np.abs(x3 - x_ref[sl]).max() / np.abs(x_ref).max()))
        h3 = torch.tensor([float(info3.res_hist[0]), float(info3.niter)], dtype=torch.float64)
        hmax, hmin = h3.clone(), h3.clone()
        dist.all_reduce(hmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(hmin, op=dist.ReduceOp.MIN)
        assert torch.equal(hmax, hmin), (hmax, hmin)
        ret[rank] = (e_apply, e_solve, info.reason, info.niter, float(info.res_hist[0]), float(np.linalg.norm(P["b"])))
        s.close()
    finally:
        dist.destroy_process_group()


# (2, 10, 13): ym = 6 | 7 and (3, 8, 8): ym = 2 | 3 | 3 -- uneven splits of setup_coord_native (xs = (xi * Nx) / nxp,
# src/pprts_base.F90:789-790) give neighbouring ranks different parities: the preconditioner's halo exchange must then be off
# on EVERY rank (tsx_pc_global_agree), not just on the odd ones (an even rank's messages would find no partner)
@pytest.mark.parametrize("world,solver,Nx,Ny", [(2, "3_10", 10, 12), (4, "3_10", 12, 10), (2, "8_16", 6, 8), (2, "3_10", 10, 13),
                                                (3, "3_10", 8, 8)])
@pytest.mark.parametrize("transport", ["host", "peer", "peer+distinct"])   # (uneven splits: mailbox slots sized by the global extents)
def test_sharded_hip_solve_equals_global_oracle(gpu, world, solver, Nx, Ny, transport):
    """transport "host": the callbacks (gloo underneath); "peer": the device-resident transport -- halos stored by the sender's
    kernel into the receiver's IPC-mapped mailbox, all-reduces as all-to-all stores (tsx_peer.hip), the rank processes sharing
    cuda:0.  Same checks, same iteration counts."""
    ret = _spawn(_worker, world, (solver, Nx, Ny, 6, transport))
    its = {v[3] for v in ret.values()}
    assert len(its) == 1  # every rank saw the same (all-reduced) scalars
    for e_apply, e_solve, reason, niter, r0, bn in ret.values():
        assert e_apply < 1e-13 and reason == 2 and e_solve < 1e-8
        assert abs(r0 - bn) <= 1e-12 * bn  # the initial residual is the *global* norm of b


def test_peer_transport_with_self_neighbours_in_one_process(gpu, monkeypatch):
    """One rank whose four neighbours are itself (force_halo): every face message goes through the rank's own mailbox --
    send kernel (payload + sequence word), receive kernel (wait, copy out, acknowledge) -- instead of a device-to-device copy.
    Same arithmetic, so apply and residual history are bit-identical to the copy path; the double-buffered slots and the
    acknowledgements are exercised by the 40 exchanges per iteration of the default solver."""
    from tenstream_amd import DiffuseSolver, synthetic

    monkeypatch.setenv("TSX_PEER_TIMEOUT_S", "5")
    for solver, shape in (("3_10", (12, 10, 8)), ("8_16", (8, 6, 8))):
        Nx, Ny, Nz = shape
        P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=1)
        out = []
        # copies; then the three ways the passes' boundary records travel through the mailbox (TSX_PEER_INPLACE: 0 pack / send /
        # receive kernels, 1 fused pack + send and the next pass reading in place, 2 the pass sends its records itself), the last
        # one also with full system-scope fences around every flag
        for peer, inplace, fences in ((False, "2", "0"), (True, "0", "0"), (True, "1", "0"), (True, "2", "0"), (True, "2", "1")):
            monkeypatch.setenv("TSX_PEER_INPLACE", inplace)
            monkeypatch.setenv("TSX_PEER_FENCES", fences)
            s = DiffuseSolver(solver, Nz, Nx, Ny, force_halo=True)
            if peer:
                s.comm_peer_init(lambda blob: [blob])
            s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
            x = np.random.default_rng(2).standard_normal(s.vec_shape)
            y = s.apply(x)
            sol = np.zeros(s.vec_shape)
            info = s.solve(P["b"], sol, rtol=1e-10, atol=1e-30)
            out.append((y, sol, info))
            s.close()
        y0, x0, i0 = out[0]
        assert i0.reason == 2
        for y1, x1, i1 in out[1:]:
            assert np.array_equal(y0, y1)
            assert i1.reason == 2 and i0.niter == i1.niter
            assert np.array_equal(i0.res_hist, i1.res_hist) and np.array_equal(x0, x1)


def _selftest_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("TSX_PEER_TIMEOUT_S", "10")
    import torch.distributed as dist

    from tenstream_amd import DiffuseSolver, coord, hostcomm, synthetic

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Nx, Ny, Nz = 12, 8, 6
        P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz)
        co = coord.coord(rank, world, Nx, Ny)
        sl = (slice(co.ys, co.ys + co.ym), slice(co.xs, co.xs + co.xm))
        s = DiffuseSolver("3_10", Nz, co.xm, co.ym, xs=co.xs, ys=co.ys, glob_xm=Nx, glob_ym=Ny, rank=rank, nranks=world,
                          neighbors=(co.west, co.east, co.south, co.north), device=0)
        ok = hostcomm.attach_peer_checked(s, rounds=40)
        loc = lambda k: np.ascontiguousarray(P[k][sl])
        s.set_coeffs(loc("coeff"), P["l1d"], loc("a11"), loc("a12"), loc("albedo"))
        x = np.zeros(s.vec_shape)
        i1 = s.solve(np.ascontiguousarray(P["b"][sl]), x, rtol=1e-10, atol=1e-30)
        # now give the transport up everywhere (what bench.py does when a rank's self test fails) and solve over the callbacks
        s.comm_peer_disable()
        hostcomm.attach(s, rank)
        x2 = np.zeros(s.vec_shape)
        i2 = s.solve(np.ascontiguousarray(P["b"][sl]), x2, rtol=1e-10, atol=1e-30)
        ret[rank] = (ok, i1.reason, i1.niter, i2.reason, i2.niter, float(np.abs(x - x2).max() / np.abs(x).max()))
        s.close()
    finally:
        dist.destroy_process_group()


def test_peer_transport_self_test_and_fallback(gpu):
    """tsx_comm_peer_selftest (patterned exchanges of varying length + all-reduces, verified on the receiver) passes with two rank
    processes on one device, agreed over the process group (hostcomm.attach_peer_checked, what `bench.py --transport auto` runs
    before it trusts the transport); after tsx_comm_peer_disable the same solver continues over the host-staged callbacks and
    arrives at the same solution with the same iteration count."""
    ret = _spawn(_selftest_worker, 2, ())
    for ok, r1, n1, r2, n2, diff in ret.values():
        assert ok is True and r1 == 2 and r2 == 2 and n1 == n2 and diff < 1e-9, (ok, r1, n1, r2, n2, diff)


# ---- 8 ranks, 2 x 4: config 3's decomposition (src/pprts_base.F90:747-790: dims = [nyp, nxp] = [4, 2], ranks x-fastest) ----
def _worker8(rank, world, port, Nx, Ny, Nz, ref_path, transport, ret):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("TSX_PEER_TIMEOUT_S", "20")
    import torch.distributed as dist

    from tenstream_amd import DiffuseSolver, coord, hostcomm, synthetic

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ref = np.load(ref_path)
        P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz)
        co = coord.coord(rank, world, Nx, Ny)
        sl = (slice(co.ys, co.ys + co.ym), slice(co.xs, co.xs + co.xm))
        s = DiffuseSolver("3_10", Nz, co.xm, co.ym, xs=co.xs, ys=co.ys, glob_xm=Nx, glob_ym=Ny, rank=rank, nranks=world,
                          neighbors=(co.west, co.east, co.south, co.north), device=0)
        if transport == "peer":
            hostcomm.attach_peer(s)
        else:
            hostcomm.attach(s, rank)
        loc = lambda k: np.ascontiguousarray(P[k][sl])
        s.set_coeffs(loc("coeff"), P["l1d"], loc("a11"), loc("a12"), loc("albedo"))
        y = s.apply(np.ascontiguousarray(ref["xg"][sl]))
        e_apply = float(np.abs(y - ref["yg"][sl]).max() / np.abs(ref["yg"]).max())
        x = np.zeros(s.vec_shape)
        info = s.solve(np.ascontiguousarray(P["b"][sl]), x, rtol=1e-10, atol=1e-30)   # the library's default preconditioner
        e_solve = float(np.abs(x - ref["x_ref"][sl]).max() / np.abs(ref["x_ref"]).max())
        xd = np.zeros(s.vec_shape)
        infod = s.solve(np.ascontiguousarray(P["b"][sl]), xd)                          # reference default tolerances
        ret[rank] = (e_apply, e_solve, info.reason, info.niter, infod.niter, (co.nxp, co.nyp, co.xi, co.yi, co.xm, co.ym),
                     s.pc_info())
        s.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("transport", ["host", "peer"])
def test_eight_ranks_two_by_four_equal_the_global_oracle(gpu, tmp_path, transport):
    """Config 3's process grid on one device: 8 rank processes share cuda:0 (host-staged face exchange, or the device-resident
    peer transport with all 8 mailboxes IPC-mapped into every process), a 64 x 64 x 16
    domain -> 2 x 4 blocks of 32 x 16 columns.  Operator and default solve against the oracle on the global domain; the
    preconditioner's halo exchange keeps the one-rank iteration count."""
    from oracle import oracle as O
    from tenstream_amd import DiffuseSolver, synthetic

    Nx, Ny, Nz = 64, 64, 16
    P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz)
    lay = O.layout("3_10", Nz, Nx, Ny)
    c64 = P["coeff"].astype(np.float64)
    xg = np.random.default_rng(11).standard_normal((Ny, Nx, Nz + 1, 10))
    yg = O.diff_apply(lay, c64, P["l1d"], P["a11"], P["a12"], P["albedo"], xg)
    x_ref, oi = O.solve_ilu(lay, c64, P["l1d"], P["a11"], P["a12"], P["albedo"], P["b"], rtol=1e-13, atol=1e-30, maxit=3000)
    assert oi["reason"] == 2
    ref_path = str(tmp_path / "ref.npz")
    np.savez(ref_path, xg=xg, yg=yg, x_ref=x_ref)
    # one periodic rank for the iteration counts
    s1 = DiffuseSolver("3_10", Nz, Nx, Ny)
    s1.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
    x1 = np.zeros(s1.vec_shape)
    its1_tight = s1.solve(P["b"], x1, rtol=1e-10, atol=1e-30).niter
    x1[...] = 0
    its1 = s1.solve(P["b"], x1).niter
    s1.close()
    ret = _spawn(_worker8, 8, (Nx, Ny, Nz, ref_path, transport))
    grids = set()
    for rank, (e_apply, e_solve, reason, niter, niter_d, grid, pci) in ret.items():
        assert e_apply < 1e-13 and reason == 2 and e_solve < 1e-8, (rank, e_apply, e_solve, reason)
        assert grid[:2] == (2, 4) and grid[2:4] == (rank % 2, rank // 2) and grid[4:] == (32, 16), grid
        assert abs(niter - its1_tight) <= 1 and abs(niter_d - its1) <= 1, (niter, its1_tight, niter_d, its1)
        assert pci[0] == 3 and pci[2], pci   # red-black scan passes on every rank
        grids.add(grid[2:4])
    assert len(grids) == 8


# ---- the whole g-point pipeline on several ranks ------------------------------------------------------------------
def _pipeline_worker(rank, world, port, Nx, Ny, Nz, phi0, theta0, tall_top, transport, ret):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from tenstream_amd import coord, lut, synthetic
    from tenstream_amd.pprts import PprtsSolver

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kabs, ksca, g = synthetic.cloud_field(Nx, Ny, Nz, seed=5)
        kabs *= 20.0
        dz = np.full((Ny, Nx, Nz), 50.0)
        if tall_top:
            dz[:, :, :tall_top] = 400.0
        planck = np.linspace(2.0, 6.0, Nz + 1)[None, None, :] * (1 + 0.05 * np.random.default_rng(0).random((Ny, Nx, 1)))
        dax = lut.direct_axes()
        Tdir, Sdir = lut.synthetic_direct_tables(dax)

        def make(nx, ny, **kw):
            P = PprtsSolver(Nz, nx, ny, 100.0, 100.0, phi0, theta0, device=0, **kw)
            P.set_lut_diffuse(lut.synthetic_diffuse_table("3_10"), lut.diffuse_axes("3_10"))
            P.set_lut_direct(Tdir, Sdir, dax)
            return P

        co = coord.coord(rank, world, Nx, Ny)
        sl = (slice(co.ys, co.ys + co.ym), slice(co.xs, co.xs + co.xm))
        P = make(co.xm, co.ym, xs=co.xs, ys=co.ys, glob_xm=Nx, glob_ym=Ny, rank=rank, nranks=world,
                 neighbors=(co.west, co.east, co.south, co.north))

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

        if transport == "peer":
            from tenstream_amd import hostcomm

            hostcomm.attach_peer(P.core)
        else:
            P.core.comm_set_callbacks(exchange, allreduce)
        loc = lambda a: np.ascontiguousarray(a[sl])
        out = {}
        for kind in ("solar", "thermal"):
            lsolar = kind == "solar"
            P.set_optical_properties(0.15, loc(kabs), loc(ksca), loc(g), loc(dz), planck=None if lsolar else loc(planck))
            info = P.solve(1000.0 if lsolar else 0.0, rtol=1e-10, atol=1e-30, maxit=3000)
            out[kind] = (info.reason, info.niter) + tuple(P.get_result())
        # the same problem on one periodic rank (rank 0 only; same process, same GPU)
        errs = {}
        if rank == 0:
            G = make(Nx, Ny)
        ref = {}
        for kind in ("solar", "thermal"):
            lsolar = kind == "solar"
            if rank == 0:
                G.set_optical_properties(0.15, kabs, ksca, g, dz, planck=None if lsolar else planck)
                gi = G.solve(1000.0 if lsolar else 0.0, rtol=1e-10, atol=1e-30, maxit=3000)
                ref[kind] = [np.ascontiguousarray(a) for a in G.get_result()]
            objs = [ref.get(kind)]
            dist.broadcast_object_list(objs, src=0)
            full = objs[0]
            e, where = [], []
            for got, want in zip(out[kind][2:], full):
                scale = max(np.abs(want).max(), 1e-30)
                d = np.abs(got - want[sl])
                e.append(float(d.max() / scale))
                where.append(tuple(int(v) for v in np.unravel_index(int(d.argmax()), d.shape)) + (int((d > 1e-3 * scale).sum()),))
            # (diagnostics for a failure: rank's block, and per quantity the local (j, i, k) of the largest deviation + how many cells deviate)
            errs[kind] = (out[kind][0], out[kind][1], e, {"block": (co.xs, co.xm, co.ys, co.ym), "worst": where})
            ref[kind] = full
        # a deviation on any rank: solve both decompositions once more in the same processes and say in the diagnostics which of the
        # two reproduces itself (the sharded result `again_sharded`, the one-rank reference `again_one_rank`: largest relative change)
        flag = torch.tensor([float(max(errs["solar"][2]) >= 3e-4 or max(errs["thermal"][2]) >= 1e-7)])
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if flag.item() > 0:
            for kind in ("solar", "thermal"):
                lsolar = kind == "solar"
                P.set_optical_properties(0.15, loc(kabs), loc(ksca), loc(g), loc(dz), planck=None if lsolar else loc(planck))
                P.solve(1000.0 if lsolar else 0.0, rtol=1e-10, atol=1e-30, maxit=3000, zero_guess=True)
                rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
                again = {"again_sharded": [rel(a, b) for a, b in zip(P.get_result(), out[kind][2:])]}
                if rank == 0:
                    G.set_optical_properties(0.15, kabs, ksca, g, dz, planck=None if lsolar else planck)
                    G.solve(1000.0 if lsolar else 0.0, rtol=1e-10, atol=1e-30, maxit=3000, zero_guess=True)
                    again["again_one_rank"] = [rel(a, b) for a, b in zip(G.get_result(), ref[kind])]
                errs[kind][3].update(again)
        ret[rank] = errs
        P.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,Nx,Ny,phi0,theta0,tall_top", [(2, 10, 12, 200.0, 40.0, 0), (4, 12, 10, 30.0, 55.0, 1),
                                                               (4, 10, 12, 300.0, 20.0, 0)])
@pytest.mark.parametrize("transport", ["host", "peer"])
def test_sharded_pipeline_equals_one_rank_pipeline(gpu, world, Nx, Ny, phi0, theta0, tall_top, transport):
    """set_optical_properties -> direct sweep (face exchange per sweep) -> setup_b -> solve -> flux divergence on 2/4
    ranks against the same g-point on one periodic rank; host-staged and device-resident (peer) transport.

    History (profiles/r06/DEFECT.md): the parameter set (4 ranks, 12 x 10, one thick 1-D top layer) deviated in 1-2 % of its fresh
    four-process runs in rounds 4 and 5.  Round 6 caught the event with scripts/fresh_loop.py: the contents of device blocks that a
    rank process had been handed by hipMalloc -- and had filled -- a few hundred microseconds earlier (the block-sharing build's
    scratch, the representatives dd_ent_cell) read back as zeros; the library's device code was intact, a relaunch gave the same.
    The library now takes its device memory from a pool of driver allocations that are quarantined before first use and never
    handed back (tsx_pool.hip): no driver allocation lies on a solver's path after tsx_create's.  The test stays strict and
    single-attempt; on a deviation the workers solve once more and the message (and TSX_TEST_DIAG_DIR) says which decomposition
    reproduces itself."""
    ret = _spawn(_pipeline_worker, world, (Nx, Ny, 8, phi0, theta0, tall_top, transport))
    for rank, errs in ret.items():
        reason, _, e, diag = errs["solar"]
        everyone = {r: (v["solar"][:3], v["solar"][3], v["thermal"][:3], v["thermal"][3]) for r, v in ret.items()}
        if os.environ.get("TSX_TEST_DIAG_DIR") and rank == min(ret):   # iteration counts of every run: a damaged preconditioner shows there first
            with open(os.path.join(os.environ["TSX_TEST_DIAG_DIR"], "sharded_pipeline_iterations.txt"), "a") as fh:
                fh.write(f"{transport} {world} {Nx}x{Ny} solar {errs['solar'][1]} thermal {errs['thermal'][1]}\n")
        if os.environ.get("TSX_TEST_DIAG_DIR") and (reason != 2 or max(e) >= 3e-4 or errs["thermal"][0] != 2 or max(errs["thermal"][2]) >= 1e-7):
            import json   # (debugging a rare deviation: the assertion message is shortened by pytest)

            with open(os.path.join(os.environ["TSX_TEST_DIAG_DIR"], f"sharded_pipeline_{transport}_{os.getpid()}_{rank}.json"), "w") as fh:
                json.dump({str(k): v for k, v in everyone.items()}, fh, indent=1)
        assert reason == 2
        # edn, eup, abso, edir: the direct sweep stops at rtol 1e-5 on both decompositions (different iterates)
        assert max(e) < 3e-4, (rank, e, diag, everyone)
        reason, _, e, diag = errs["thermal"]
        assert reason == 2 and max(e) < 1e-7, (rank, e, diag, everyone)


# ---- RCCL with two real peers ---------------------------------------------------------------------------------------
def _rccl_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from tenstream_amd import DiffuseSolver, coord, synthetic
    from tenstream_amd._lib import TsxError

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Nx, Ny, Nz = 12, 8, 6
        P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz)
        co = coord.coord(rank, world, Nx, Ny)
        sl = (slice(co.ys, co.ys + co.ym), slice(co.xs, co.xs + co.xm))
        s = DiffuseSolver("3_10", Nz, co.xm, co.ym, xs=co.xs, ys=co.ys, glob_xm=Nx, glob_ym=Ny, rank=rank, nranks=world,
                          neighbors=(co.west, co.east, co.south, co.north), device=0)
        uid = [s.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        try:
            s.comm_init(uid[0])
        except TsxError as e:
            ret[rank] = ("refused", str(e))
            return
        loc = lambda k: np.ascontiguousarray(P[k][sl])
        s.set_coeffs(loc("coeff"), P["l1d"], loc("a11"), loc("a12"), loc("albedo"))
        x = np.zeros(s.vec_shape)
        info = s.solve(np.ascontiguousarray(P["b"][sl]), x, rtol=1e-10, atol=1e-30)
        ret[rank] = ("ok", info.reason, info.niter, x)
        s.close()
    finally:
        dist.destroy_process_group()


def test_rccl_transport_with_two_ranks_on_one_device(gpu):
    """Two ranks, both on cuda:0, one RCCL communicator: exercises the grouped send/recv ordering of the face exchange
    (W and E are the same peer with 2 ranks along a periodic axis) and the 3-double all-reduces with a real peer.  RCCL
    (like NCCL) refuses two ranks of one communicator on the same device; where it does, the test is skipped with RCCL's
    own message -- the driver's multi-GPU run is then the first execution with real peers."""
    import scipy.sparse.linalg as spla

    from oracle import oracle as O
    from tenstream_amd import synthetic

    ret = _spawn(_rccl_worker, 2, ())
    if any(v[0] == "refused" for v in ret.values()):
        pytest.skip("RCCL refuses two ranks on one device: " + "; ".join(str(v[1])[:200] for v in ret.values() if v[0] == "refused"))
    Nx, Ny, Nz = 12, 8, 6
    P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz)
    lay = O.layout("3_10", Nz, Nx, Ny)
    A = O.assemble_csr(lay, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], P["albedo"])
    x_ref = spla.spsolve(A.tocsc(), P["b"].ravel()).reshape(P["b"].shape)
    from tenstream_amd import coord

    for rank, v in ret.items():
        co = coord.coord(rank, 2, Nx, Ny)
        sl = (slice(co.ys, co.ys + co.ym), slice(co.xs, co.xs + co.xm))
        assert v[1] == 2 and np.abs(v[3] - x_ref[sl]).max() <= 1e-8 * np.abs(x_ref).max()


@pytest.mark.parametrize("solver", ["3_10", "8_16"])
def test_preconditioner_halo_keeps_the_couplings_across_rank_faces(gpu, monkeypatch, solver):
    """On several ranks the red-black passes exchange the boundary columns' side-stream records after every pass
    (tsx_k_pcs_halo_pack + the operator's exchange pattern), so M^-1 keeps the couplings across rank faces.  One rank whose
    faces are routed through the exchange buffers (force_halo) then preconditions exactly like the periodic rank that reads
    its neighbours in place: same iteration count, same solution.  With TSX_PC_HALO=0 (couplings across faces dropped,
    block-Jacobi over ranks like the reference's PCBJACOBI) the small domain needs more iterations."""
    from tenstream_amd import DiffuseSolver, synthetic

    Nx, Ny, Nz = (24, 16, 12) if solver == "3_10" else (12, 8, 6)
    P = synthetic.make_problem(solver, Nx=Nx, Ny=Ny, Nz=Nz, n1d=1)
    its, xs = {}, {}
    for name, fh, env in (("wrap", 0, "1"), ("halo", 1, "1"), ("dropped", 1, "0")):
        monkeypatch.setenv("TSX_PC_HALO", env)
        s = DiffuseSolver(solver, Nz, Nx, Ny, force_halo=fh)
        s.set_coeffs(P["coeff"], P["l1d"], P["a11"], P["a12"], P["albedo"])
        x = np.zeros(s.vec_shape)
        info = s.solve(P["b"], x, rtol=1e-10, atol=1e-30)
        assert info.reason == 2
        its[name], xs[name] = info.niter, x
        s.close()
    assert abs(its["halo"] - its["wrap"]) <= 1
    assert its["dropped"] > its["halo"]
    for name in ("halo", "dropped"):
        assert np.abs(xs[name] - xs["wrap"]).max() <= 1e-8 * np.abs(xs["wrap"]).max()


# ---- the flow kernel with rank faces inside it ----------------------------------------------------------------------------
def _flow_faces_worker(rank, world, port, Nx, Ny, Nz, ret):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("TSX_PEER_TIMEOUT_S", "20")
    import torch.distributed as dist

    from tenstream_amd import DiffuseSolver, coord, hostcomm, synthetic

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz)
        co = coord.coord(rank, world, Nx, Ny)
        sl = (slice(co.ys, co.ys + co.ym), slice(co.xs, co.xs + co.xm))
        loc = lambda k: np.ascontiguousarray(P[k][sl])
        out = {}
        for env in ("1", "0"):   # the intermediate passes as ONE launch with the faces inside it / a launch per pass
            os.environ["TSX_FLOW_PEER"] = env
            s = DiffuseSolver("3_10", Nz, co.xm, co.ym, xs=co.xs, ys=co.ys, glob_xm=Nx, glob_ym=Ny, rank=rank, nranks=world,
                              neighbors=(co.west, co.east, co.south, co.north), device=0)
            hostcomm.attach_peer(s)
            s.set_coeffs(loc("coeff"), P["l1d"], loc("a11"), loc("a12"), loc("albedo"))
            res = []
            for rtol in (1e-5, 1e-9):
                x = np.zeros(s.vec_shape)
                info = s.solve(loc("b"), x, rtol=rtol, atol=1e-30)
                res += [info.reason, info.niter, x, np.asarray(info.res_hist)]
            fl = s.flow_info()
            out[env] = (res, fl["in_use"], fl["fat"])
            s.close()
        same = all(np.array_equal(a, b) for a, b in zip(out["1"][0], out["0"][0]))
        ret[rank] = (same, out["1"][1], out["0"][1], out["1"][0][0], out["1"][0][1], out["1"][0][4], out["1"][0][5])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,Nx,Ny,Nz,body", [(2, 128, 32, 16, "auto"), (4, 128, 64, 12, "auto"), (8, 128, 128, 8, "auto"),
                                                 (4, 128, 64, 12, "lean")])
def test_flow_kernel_with_rank_faces_equals_launch_per_pass(gpu, monkeypatch, world, Nx, Ny, Nz, body):
    """Round 5: on several ranks the flow kernel (the intermediate passes of an application of M^-1 in one launch) keeps the
    exchange of the boundary records inside the launch: a face tile stores its records into the neighbour rank's mailbox slot and,
    after its drain, the tags of its row / columns; the neighbour's tile of the next pass polls those tags and reads the records
    in place; no acknowledgements inside the launch, the launch's last workgroup posts them and the faces' sequence numbers
    (tsx_k_pcs_flow FPEER, src/pprts_explicit.F90:769-843's exchange pattern).  Same message numbering, slots and arithmetic as a
    launch per pass (TSX_FLOW_PEER=0): solutions and residual histories must be bit-identical on every rank -- rank processes
    sharing cuda:0, 2 x 1, 2 x 2 and 2 x 4 grids (W and E the same peer on the first)."""
    if body == "lean":   # round 6: the rank faces inside the lean body (shards whose passes are not resident at once), forced here
        monkeypatch.setenv("TSX_FLOW_FAT", "0")
    ret = _spawn(_flow_faces_worker, world, (Nx, Ny, Nz))
    for rank, (same, used1, used0, r5, n5, r9, n9) in ret.items():
        assert used1 and not used0, (rank, used1, used0)   # the flow kernel ran with its faces / did not
        assert same, rank
        assert r5 == 2 and r9 == 2, (rank, r5, r9)
    assert len({v[4] for v in ret.values()}) == 1 and len({v[6] for v in ret.values()}) == 1   # every rank the same counts


# ---- the RCCL transport's host code with real ranks, through a test double of librccl -------------------------------------
def _build_fake_rccl(tmp_path):
    import subprocess

    so = str(tmp_path / "libfake_rccl.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "c", "fake_rccl.cpp"), "-lrt"],
                   check=True)
    return so


def _rccl_double_worker(rank, world, port, Nx, Ny, Nz, fake_so, ret):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from tenstream_amd import DiffuseSolver, coord, hostcomm, synthetic

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = synthetic.make_problem("3_10", Nx=Nx, Ny=Ny, Nz=Nz, n1d=1)
        co = coord.coord(rank, world, Nx, Ny)
        sl = (slice(co.ys, co.ys + co.ym), slice(co.xs, co.xs + co.xm))
        loc = lambda k: np.ascontiguousarray(P[k][sl])
        xg = np.random.default_rng(11).standard_normal((Ny, Nx, Nz + 1, 10))

        def run(attach):
            s = DiffuseSolver("3_10", Nz, co.xm, co.ym, xs=co.xs, ys=co.ys, glob_xm=Nx, glob_ym=Ny, rank=rank, nranks=world,
                              neighbors=(co.west, co.east, co.south, co.north), device=0)
            attach(s)
            s.set_coeffs(loc("coeff"), P["l1d"], loc("a11"), loc("a12"), loc("albedo"))
            res = [s.apply(np.ascontiguousarray(xg[sl]))]
            for kw in (dict(rtol=1e-10, atol=1e-30, pc=1, pc_sweeps=1), dict(rtol=1e-9, atol=1e-30), dict()):
                x = np.zeros(s.vec_shape)
                info = s.solve(loc("b"), x, **kw)
                res += [info.reason, info.niter, x, np.asarray(info.res_hist)]
            s.close()
            return res

        def attach_double(s):   # the RCCL transport, librccl replaced by the double (TSX_RCCL_LIB is read at the first bind)
            uid = [s.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            s.comm_init(uid[0])

        os.environ["TSX_RCCL_LIB"] = fake_so
        out = {}
        for ov in ("1", "0"):   # interior / frame split around the exchanges on comm_stream (the default with RCCL), and without
            os.environ["TSX_OVERLAP"] = ov
            out["rccl" + ov] = run(attach_double)
            out["host" + ov] = run(lambda s: hostcomm.attach(s, rank))
        del os.environ["TSX_OVERLAP"]
        def agree(a, b):   # the apply bit for bit; the solves up to the order in which the transports sum the dots over the ranks
            ok = np.array_equal(a[0], b[0])
            for q in range(1, len(a), 4):
                ok = ok and a[q] == b[q] and abs(a[q + 1] - b[q + 1]) <= 1 and np.abs(a[q + 2] - b[q + 2]).max() <= 1e-7 * np.abs(b[q + 2]).max()
            return bool(ok)

        same = {k: agree(out["rccl" + k], out["host" + k]) for k in ("1", "0")}
        r = out["rccl1"]
        ret[rank] = (same["1"], same["0"], r[1], r[2], r[5], r[6], r[9], r[10])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,Nx,Ny,Nz", [(2, 24, 8, 6), (4, 24, 16, 6), (8, 32, 32, 6)])
def test_rccl_transport_host_code_with_real_ranks_through_a_test_double(gpu, tmp_path, world, Nx, Ny, Nz):
    """librccl refuses two ranks of a communicator on one device (the test above skips), so on this pool the RCCL transport's
    host code -- ncclCommInitRank, the second communicator from ncclCommSplit for the exchanges on comm_stream, the grouped
    ncclSend / ncclRecv in the order W, E, S, N against E, W, N, S, the 3-double all-reduces between the scalar stages, the
    preconditioner's exchanges after every pass, with and without the interior / frame overlap -- never ran with a second rank.
    tests/c/fake_rccl.cpp implements exactly the entry points libtsx binds over shared memory with NCCL's matching rule
    (messages between two ranks of a communicator in issue order, groups issued together); TSX_RCCL_LIB selects it.  2 ranks
    along a periodic axis (W and E the same peer -- the case the receive order exists for), 2 x 2 and 2 x 4: applies, solves and
    applies bit-identical to the host-staged callbacks on the same ranks, solves equal up to the order in which the two transports
    sum the dots over the ranks (same reasons, iteration counts within one, solutions to 1e-7 at the loosest tolerance)."""
    fake = _build_fake_rccl(tmp_path)
    ret = _spawn(_rccl_double_worker, world, (Nx, Ny, Nz, fake))
    for rank, v in ret.items():
        assert v[0] and v[1], (rank, v[:2])
        assert v[2] == 2 and v[4] == 2 and v[6] == 2, (rank, v)
    for k in (3, 5, 7):   # every rank the same iteration counts
        assert len({v[k] for v in ret.values()}) == 1

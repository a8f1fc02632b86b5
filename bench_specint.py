#!/usr/bin/env python3
"""Config 4 of BASELINE.json: the many-g-point spectral loop (rrtmg-like: 112 SW + 140 LW g-points) on the
256x256x64 domain, every g-point running the whole device pipeline

    set_optical_properties -> (direct sweep -> setup_b | thermal setup_b) -> diffuse solve -> flux divergence ->
    get_result, accumulated on the device            (rrtmg/rrtmg/pprts_rrtmg.F90:999-1055 drives the same sequence)

driven the way the reference's rrtmg wrapper drives pprts:
  * the column is a merged grid (src/tenstr_atm.F90): thin dynamics layers (dz = 50 m at dx = 100 m) below thick
    background-atmosphere layers (dz/dx > twostr_ratio = 2) -- those become 1-D (Eddington) rows inside the same operator
    (src/pprts.F90:669-677), so every solve exercises the l1d branch of the operator, the preconditioner and setup_b;
  * every g-point has its own solution uid (`opt_solution_uid=ib`, rrtmg/rrtmg/pprts_rrtmg.F90:1027): the first radiation
    call starts each g-point from the previous g-point's solution (-initial_guess_from_last_uid, src/pprts.F90:2540-2552),
    later calls from the g-point's own previous solution (tsx_pprts_select_solution; 252 slots of 0.22 GB stay resident
    in the 288 GB of HBM);
  * `--calls 2` (default) runs two radiation calls: the second one after the cloud field has moved by one column, like a
    model time step.
Optical properties per g-point: the benchmark cloud field with kabs/ksca scaled by a log-uniform factor in [1e-3, 30]
(SURVEY 8(d) config 4); LW g-points get a Planck surrogate.  Multi-GPU: g-points -- independent solves -- are dealt in
contiguous blocks to the ranks and only the four accumulated result arrays are all-reduced at the end: no data-path
collective.  Not the headline metric: `bench.py` is; this prints one JSON line of its own.

    python bench_specint.py [--gpus N] [--sw 112] [--lw 140] [--calls 2]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--sw", type=int, default=112)
    ap.add_argument("--lw", type=int, default=140)
    ap.add_argument("--nx", type=int, default=256)
    ap.add_argument("--ny", type=int, default=256)
    ap.add_argument("--nz", type=int, default=64)
    ap.add_argument("--nz-background", type=int, default=16, help="thick background layers on top of the dynamics grid (1-D rows)")
    ap.add_argument("--streams", type=int, default=4, help="solver instances (HIP streams + host threads) working on different "
                    "g-points at the same time")
    ap.add_argument("--calls", type=int, default=2, help="radiation calls (time steps); the first one is cold")
    ap.add_argument("--pc-sweeps", type=int, default=0, help="0 = library default")
    ap.add_argument("--phi0", type=float, default=180.0)
    ap.add_argument("--theta0", type=float, default=40.0)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline (the oracle's restatement of the reference's "
                    "default CPU path on ONE g-point's diffuse system, extrapolated to the loop)")
    return ap.parse_args(argv)


def column_dz(Nz, n_bg, dz_dyn=50.0):
    """k = 0 is the top of the atmosphere (the rrtmg wrapper reverses its bottom-up columns, pprts_rrtmg.F90:1006-1008):
    n_bg background layers growing from 300 m to 3 km, then the dynamics layers"""
    dz = np.full(Nz, dz_dyn)
    if n_bg > 0:
        dz[:n_bg] = np.geomspace(3000.0, 300.0, n_bg)
    return dz


def run_loop(args, dev, rank=0, world=1, all_reduce=None):
    import torch

    from tenstream_amd import lut as LUT
    from tenstream_amd import synthetic as S
    from tenstream_amd.pprts import PprtsSolver

    Nx, Ny, Nz = args.nx, args.ny, args.nz
    n_bg = min(args.nz_background, Nz - 1)
    dx, albedo = 100.0, 0.1
    dz1d = column_dz(Nz, n_bg)
    # clouds live in the dynamics grid only; the background layers carry a thin gas extinction that falls off with height
    kabs = np.full((Ny, Nx, Nz), 1e-5)
    ksca = np.full((Ny, Nx, Nz), 1e-5)
    g = np.zeros((Ny, Nx, Nz))
    kd, sd, gd = S.cloud_field(Nx, Ny, Nz - n_bg, seed=20240611)
    kabs[:, :, n_bg:], ksca[:, :, n_bg:], g[:, :, n_bg:] = kd, sd, gd
    if n_bg:
        fall = np.geomspace(0.02, 0.6, n_bg)
        kabs[:, :, :n_bg] *= fall
        ksca[:, :, :n_bg] *= fall
    t64 = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
    kabs0, ksca0, g0 = t64(kabs), t64(ksca), t64(g)
    dz_d = t64(np.broadcast_to(dz1d, (Ny, Nx, Nz)).copy())
    alb = torch.full((Ny, Nx), albedo, dtype=torch.float64, device=dev)
    # Planck surrogate: 288 K at the surface to 220 K at the top, band-integrated ~ sigma T^4 / pi / n_lw
    T = torch.linspace(220.0, 288.0, Nz + 1, dtype=torch.float64, device=dev)
    planck0 = (5.670374419e-8 * T**4 / np.pi).expand(Ny, Nx, Nz + 1).contiguous()
    # the surface's own emission (atm%tskin -> Bsrfc, rrtmg/rrtmg/pprts_rrtmg.F90:609-642): a skin 0-3 K warmer than the air at
    # the lowest level, varying over the ground; handed over as planck_srfc with every LW g-point like :681 does
    tskin = 288.0 + 3.0 * torch.rand((Ny, Nx), dtype=torch.float64, device=dev, generator=torch.Generator(device=dev).manual_seed(11))
    planck_srfc0 = (5.670374419e-8 * tskin**4 / np.pi).contiguous()

    # --streams K: K solver instances, each with its own HIP stream and host thread, work on different g-points at the same
    # time (the library is re-entrant per instance; ctypes calls release the GIL).  On 256 x 256 columns one g-point fills the
    # chip; on small domains several in flight do
    K = max(1, args.streams)
    dax = LUT.direct_axes()
    Tdir, Sdir = LUT.synthetic_direct_tables(dax)
    Ps = []
    for _ in range(K):
        Pk = PprtsSolver(Nz, Nx, Ny, dx, dx, args.phi0, args.theta0, device=dev.index)
        Pk.set_lut_diffuse(LUT.synthetic_diffuse_table("3_10"), LUT.diffuse_axes("3_10"))
        Pk.set_lut_direct(Tdir, Sdir, dax)
        Ps.append(Pk)
    P = Ps[0]

    rng = np.random.default_rng(7)
    ng = args.sw + args.lw
    factors = np.exp(rng.uniform(np.log(1e-3), np.log(30.0), ng))
    weights = (rng.dirichlet(np.ones(args.sw)).tolist() if args.sw else []) + (rng.dirichlet(np.ones(args.lw)).tolist() if args.lw else [])
    lo, hi = (rank * ng) // world, ((rank + 1) * ng) // world   # contiguous blocks: uid - 1 stays on the same rank
    mine = list(range(lo, hi))

    L = Nz + 1
    new_acc = lambda: [torch.zeros((Ny, Nx, L), dtype=torch.float64, device=dev), torch.zeros((Ny, Nx, L), dtype=torch.float64, device=dev),
                       torch.zeros((Ny, Nx, Nz), dtype=torch.float64, device=dev), torch.zeros((Ny, Nx, L), dtype=torch.float64, device=dev)]
    acc = new_acc()
    accs = [acc] + [new_acc() for _ in range(K - 1)]          # one accumulator and one scratch set per instance
    tmps = [[torch.empty_like(a) for a in acc] for _ in range(K)]
    tstreams = [torch.cuda.Stream(device=dev) for _ in range(K)] if K > 1 else [None]
    mu0 = float(np.cos(np.deg2rad(args.theta0)))
    shift = {"n": 0}

    def run(q, w=0):
        P, tmp, acc = Ps[w], tmps[w], accs[w]
        f = float(factors[q])
        lsolar = q < args.sw
        roll = lambda a: torch.roll(a, shifts=shift["n"], dims=1) if shift["n"] else a
        P.set_optical_properties(alb, roll(kabs0) * f, roll(ksca0) * f, roll(g0), dz_d, planck=None if lsolar else planck0 * weights[q],
                                 planck_srfc=None if lsolar else planck_srfc0 * weights[q])
        kw = dict(pc_sweeps=args.pc_sweeps) if args.pc_sweeps > 0 else {}
        e0 = 1361.0 * weights[q] if lsolar else 0.0
        info = P.solve(e0, lsolar=lsolar, uid=q, **kw)
        P.get_result(out=tmp)
        for a, t in zip(acc, tmp):
            a += t
        # energy balance of this g-point: what the atmosphere absorbs is the convergence of the net flux (periodic domain).
        # Left on the device (four sums in one small tensor): the host reads all of them after the loop, not per g-point
        edn, eup, abso, edir = tmp
        bal = torch.stack(((edn[:, :, 0] + edir[:, :, 0] - eup[:, :, 0]).sum(), (edn[:, :, -1] + edir[:, :, -1] - eup[:, :, -1]).sum(),
                           (abso * dz_d).sum(), eup[:, :, -1].sum()))
        return info, bal

    def sync():
        torch.cuda.synchronize()

    calls = []
    for Pk in Ps:
        Pk.set_optical_properties(alb, kabs0, ksca0, g0, dz_d)   # allocations and first touch outside the timed loop
    n1d = int(P.l1d.sum())
    for call in range(args.calls):
        shift["n"] = call
        for ac in accs:
            for a in ac:
                a.zero_()
        sync()
        t0 = time.perf_counter()
        if K == 1:
            out = [run(q) for q in mine]
        else:
            import concurrent.futures as cf

            def work(w):   # a contiguous block of g-points per instance: the guess of uid - 1 stays with it
                blk = mine[(w * len(mine)) // K:((w + 1) * len(mine)) // K]
                with torch.cuda.stream(tstreams[w]):
                    r = [run(q, w) for q in blk]
                    tstreams[w].synchronize()
                return r

            with cf.ThreadPoolExecutor(K) as ex:
                out = [o for part in ex.map(work, range(K)) for o in part]
            for other in accs[1:]:
                for a, t in zip(acc, other):
                    a += t
        if all_reduce is not None:
            for a in acc:
                all_reduce(a)
        sync()
        dt = time.perf_counter() - t0
        def balance(t):   # |absorbed - (net TOA - net surface)| relative to the largest of the fluxes involved
            net_toa, net_srf, atm, up_srf = (float(v) for v in t.tolist())
            return abs(atm - (net_toa - net_srf)) / max(abs(net_toa), abs(net_srf), up_srf, 1e-30)

        out = [(o[0], balance(o[1])) for o in out]
        infos = [o[0] for o in out]
        its = np.array([i.niter for i in infos]) if infos else np.zeros(1)
        calls.append(dict(seconds=dt, iterations_min_med_max=[int(its.min()), float(np.median(its)), int(its.max())],
                          reasons=sorted({int(i.reason) for i in infos}), diffuse_solve_ms_total=float(sum(i.solve_ms for i in infos)),
                          energy_balance_max=float(max((o[1] for o in out), default=0.0)),
                          toa_net_down_Wm2=float((acc[0][:, :, 0] + acc[3][:, :, 0] - acc[1][:, :, 0]).mean())))
    # ---- where a g-point's time goes, and the dominant kernel against its roofline (round 5; outside every timed call): a sample
    # of the g-points once more on instance 0, the device synchronised between the phases of a g-point
    breakdown = roofline = None
    if rank == 0 and mine:
        sample = mine[::max(1, len(mine) // 24)][:24]
        ph = {"set_optical_properties": 0.0, "solve (direct sweep, setup_b, diffuse solve)": 0.0, "get_result + accumulate + balance sums": 0.0}
        dsolve, its_s = 0.0, []
        P0, tmp0, acc0 = Ps[0], tmps[0], accs[0]
        kw = dict(pc_sweeps=args.pc_sweeps) if args.pc_sweeps > 0 else {}
        P0.core.log_enable(True)   # the reference's log events (solver%logs): device time per phase, roctx ranges for a trace
        for q in sample:
            f = float(factors[q])
            lsolar = q < args.sw
            sync()
            t0 = time.perf_counter()
            P0.set_optical_properties(alb, kabs0 * f, ksca0 * f, g0, dz_d, planck=None if lsolar else planck0 * weights[q],
                                      planck_srfc=None if lsolar else planck_srfc0 * weights[q])
            sync()
            t1 = time.perf_counter()
            info = P0.solve(1361.0 * weights[q] if lsolar else 0.0, lsolar=lsolar, uid=q, **kw)
            sync()
            t2 = time.perf_counter()
            P0.get_result(out=tmp0)
            for a, t in zip(acc0, tmp0):
                a += t
            (tmp0[2] * dz_d).sum()
            sync()
            t3 = time.perf_counter()
            ph["set_optical_properties"] += t1 - t0
            ph["solve (direct sweep, setup_b, diffuse solve)"] += t2 - t1
            ph["get_result + accumulate + balance sums"] += t3 - t2
            dsolve += info.solve_ms * 1e-3
            its_s.append(info.niter)
        tot = sum(ph.values())
        events = P0.core.log_get()
        P0.core.log_enable(False)
        breakdown = {"sample_gpoints": len(sample), "ms_per_gpoint": tot / len(sample) * 1e3,
                     "log_events_device_ms_per_gpoint": {k: round(v[1] / len(sample), 4) for k, v in events.items()},
                     "log_events_counts": {k: v[0] for k, v in events.items()},
                     "phases_ms_per_gpoint": {k: v / len(sample) * 1e3 for k, v in ph.items()},
                     "diffuse_solve_ms_per_gpoint_device": dsolve / len(sample) * 1e3,
                     "fractions": dict({k: v / tot for k, v in ph.items()}, diffuse_solve_of_total=dsolve / tot),
                     "iterations_mean": float(np.mean(its_s)),
                     "note": "one instance, the device synchronised after every phase (the timed calls above overlap the phases of "
                             "different g-points and do not synchronise): a breakdown, not a rate"}
        core = P0.core
        try:
            pc_ran, sweeps, scan, _ = core.pc_info()
            if scan:
                pass_ms = core.bench_kernel(3, 80)
                fl = core.flow_info()
                if fl["in_use"]:
                    ms = core.bench_kernel(4, 20)
                    fl = core.flow_info()
                    nb = core.algorithmic_bytes(4)
                    kname = (f"tsx_k_pcs_flow (passes {fl['first_pass']}..{fl['end_pass'] - 1} of one M^-1 application in one launch)")
                else:
                    ms, nb, kname = pass_ms, core.algorithmic_bytes(3), "tsx_k_pcs_rb<..., GS, MODE 0, RQ 2> (one intermediate red-black pass of M^-1)"
                ach = nb / (ms * 1e-3) / 1e9
                roofline = {"bound": "hbm", "kernel": kname, "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                            "traffic": None, "bytes_per_launch": nb, "ms_per_launch": ms, "frac_of_achievable": ach / 6300.0,
                            "the_same_pass_as_its_own_launch_us": pass_ms * 1e3, "passes_per_application": sweeps + 1,
                            "basis": "the launch the diffuse solves spend most of their time in, timed alone with HIP events on the "
                                     "solver's stream on the last sampled g-point's coefficients (tsx_bench_kernel); traffic: "
                                     "profiles/r05/traffic_3_10_256x256x64.json holds the PMC bytes of the same kernel on the headline field"}
        except Exception as e:   # noqa: BLE001
            roofline = {"error": str(e)}
    cpu = None
    if rank == 0 and not args.no_cpu_baseline and args.sw > 0:
        # CPU baseline, one g-point (the solar one with the median optical-depth factor): the very diffuse system the device
        # solved -- its blocks, 1-D layers, Eddington coefficients and right-hand side read back -- through the oracle's
        # restatement of the reference's default CPU path (assembled AIJ + FBCGS + PCBJACOBI / ILU(0), one subdomain per usable
        # core), zero guess, reference default tolerances; the loop is 252 such solves plus the rest of pprts()
        import bench
        from oracle import oracle as O
        from tenstream_amd.coord import decompose

        q = int(np.argsort(factors[:args.sw])[args.sw // 2])
        shift["n"] = 0
        info_d, _ = run(q)
        P0 = Ps[0]
        coeff = P0.core.get_coeffs()
        nz0 = lambda a: np.nan_to_num(a, nan=0.0)
        l1d = P0.l1d
        threads = min(bench.usable_cores(), 128)
        npx, npy = decompose(threads)
        lay = O.layout("3_10", Nz, Nx, Ny)
        frac = 1.0 - float(l1d.sum()) / Nz
        rt, at, mx = O.default_tolerances(Nx, Ny, Nz + 1, frac)
        _, oi = O.solve_bjacobi_ilu_mt(lay, coeff, l1d, nz0(P0.get_field("a11")), nz0(P0.get_field("a12")),
                                       np.full((Ny, Nx), albedo), P0.get_field("b"), npx, npy, rtol=rt, atol=at, maxit=mx)
        del coeff
        cpu = {"value": 1.0 / oi["t_solve"], "unit": "g-points/s", "cores": threads, "kind": "port",
               "sample": f"ONE of the {ng} g-points (solar, median factor {factors[q]:.3g}) on {Nx}x{Ny}x{Nz}: the device's own blocks and "
                         f"right-hand side; assembled CSR + FBCGS + block-Jacobi / ILU(0) on {npx}x{npy} subdomains, zero guess, "
                         f"{oi['niter']} its, reason {oi['reason']}, solve {oi['t_solve']:.2f}s (assembly {oi['t_assemble']:.2f}s and "
                         f"factorisation {oi['t_factor']:.2f}s per g-point not counted; the device's solve of the same g-point: "
                         f"{info_d.niter} its, {info_d.solve_ms:.1f} ms warm-started)",
               "loop_extrapolated_s": ng * oi["t_solve"],
               "loop_extrapolated_with_assembly_s": ng * (oi["t_solve"] + oi["t_assemble"] + oi["t_factor"])}
    for Pk in Ps:
        Pk.close()
    return dict(ng=ng, rank_gpoints=len(mine), n1d_layers=n1d, calls=calls, mu0=mu0, cpu_baseline=cpu, breakdown=breakdown, roofline=roofline)


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    R = run_loop(args, dev, rank, world, all_reduce=(dist.all_reduce if world > 1 else None))
    secs = [c["seconds"] for c in R["calls"]]
    if world > 1:
        tt = torch.tensor(secs, dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        secs = [float(v) for v in tt.tolist()]
    if rank == 0:
        Nx, Ny, Nz, ng = args.nx, args.ny, args.nz, R["ng"]
        last = len(secs) - 1
        print(json.dumps({
            "metric": "pprts 3_10 spectral loop g-points/s", "value": ng / secs[last], "unit": "g-points/s", "n_gpus": world,
            "seconds": secs[last], "higher_is_better": True, "scaling": "strong", "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.sw} SW + {args.lw} LW g-points on {Nx}x{Ny}x{Nz} ({R['n1d_layers']} thick background layers "
                                   f"as 1-D rows), whole pipeline per g-point on the device, one solution uid per g-point, "
                                   f"g-points dealt in blocks to the GPUs" + (f" and to {args.streams} concurrent solver instances per GPU" if args.streams > 1 else "") + ", results all-reduced once; value = radiation call {last + 1} "
                                   f"of {len(secs)} (call 1 is cold: guess from the previous g-point)",
                       "cells_gpoints_per_s": ng * Nx * Ny * Nz / secs[last], "rank0_gpoints": R["rank_gpoints"],
                       "calls": [dict(c, seconds=s_, gpoints_per_s=ng / s_) for c, s_ in zip(R["calls"], secs)],
                       "breakdown": R["breakdown"]},
            "roofline": R["roofline"],
            "cpu_baseline": R["cpu_baseline"]}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

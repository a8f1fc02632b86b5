#!/usr/bin/env python3
"""Config 4 of BASELINE.json: the many-g-point spectral loop (rrtmg-like: 112 SW + 140 LW g-points) on the
256x256x64 domain, every g-point running the whole device pipeline

    set_optical_properties -> (direct sweep -> setup_b | thermal setup_b) -> diffuse solve -> flux divergence ->
    get_result, accumulated on the device            (rrtmg/rrtmg/pprts_rrtmg.F90:999-1055 drives the same sequence)

Optical properties per g-point: the benchmark cloud field with kabs/ksca scaled by a log-uniform factor in [1e-3, 30]
(SURVEY 8(d) config 4); LW g-points get a Planck surrogate.  The previous solution is the initial guess (the reference
keeps `solution` per uid).  Multi-GPU: the 288 GB of an MI355X hold the whole domain, so g-points -- independent
solves -- are dealt round-robin to the ranks and only the four accumulated result arrays are all-reduced at the end:
no data-path collective.  Not the headline metric: `bench.py` is; this prints one JSON line of its own.

    python bench_specint.py [--gpus N] [--sw 112] [--lw 140]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--sw", type=int, default=112)
    ap.add_argument("--lw", type=int, default=140)
    ap.add_argument("--nx", type=int, default=256)
    ap.add_argument("--ny", type=int, default=256)
    ap.add_argument("--nz", type=int, default=64)
    ap.add_argument("--pc-sweeps", type=int, default=0, help="0 = library default")
    ap.add_argument("--phi0", type=float, default=180.0)
    ap.add_argument("--theta0", type=float, default=40.0)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    from tenstream_amd import lut as LUT
    from tenstream_amd import synthetic as S
    from tenstream_amd.pprts import PprtsSolver

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    Nx, Ny, Nz = args.nx, args.ny, args.nz
    dx, dz, albedo = 100.0, 50.0, 0.1

    kabs, ksca, g = S.cloud_field(Nx, Ny, Nz, seed=20240611)
    t64 = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
    kabs0, ksca0, g0 = t64(kabs), t64(ksca), t64(g)
    dz_d = torch.full((Ny, Nx, Nz), dz, dtype=torch.float64, device=dev)
    alb = torch.full((Ny, Nx), albedo, dtype=torch.float64, device=dev)
    # Planck surrogate: 288 K at the surface to 220 K at the top, band-integrated ~ sigma T^4 / pi / n_lw
    T = torch.linspace(220.0, 288.0, Nz + 1, dtype=torch.float64, device=dev)
    planck0 = (5.670374419e-8 * T**4 / np.pi).expand(Ny, Nx, Nz + 1).contiguous()

    P = PprtsSolver(Nz, Nx, Ny, dx, dx, args.phi0, args.theta0, device=local_rank)
    P.set_lut_diffuse(LUT.synthetic_diffuse_table("3_10"), LUT.diffuse_axes("3_10"))
    dax = LUT.direct_axes()
    Tdir, Sdir = LUT.synthetic_direct_tables(dax)
    P.set_lut_direct(Tdir, Sdir, dax)

    rng = np.random.default_rng(7)
    ng = args.sw + args.lw
    factors = np.exp(rng.uniform(np.log(1e-3), np.log(30.0), ng))
    weights = rng.dirichlet(np.ones(args.sw)).tolist() + rng.dirichlet(np.ones(args.lw)).tolist()
    mine = [q for q in range(ng) if q % world == rank]

    L = Nz + 1
    acc = [torch.zeros((Ny, Nx, L), dtype=torch.float64, device=dev), torch.zeros((Ny, Nx, L), dtype=torch.float64, device=dev),
           torch.zeros((Ny, Nx, Nz), dtype=torch.float64, device=dev), torch.zeros((Ny, Nx, L), dtype=torch.float64, device=dev)]
    tmp = [torch.empty_like(a) for a in acc]

    def run(q):
        f = float(factors[q])
        lsolar = q < args.sw
        P.set_optical_properties(alb, kabs0 * f, ksca0 * f, g0, dz_d, planck=None if lsolar else planck0 * weights[q])
        kw = dict(pc_sweeps=args.pc_sweeps) if args.pc_sweeps > 0 else {}
        info = P.solve(1361.0 * weights[q] if lsolar else 0.0, lsolar=lsolar, **kw)
        P.get_result(out=tmp)
        for a, t in zip(acc, tmp):
            a += t
        return info

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(mine[0])  # warm-up (allocations, first-touch)
    for a in acc:
        a.zero_()
    barrier()
    t0 = time.perf_counter()
    infos = [run(q) for q in mine]
    if world > 1:
        for a in acc:
            dist.all_reduce(a)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    its = np.array([i.niter for i in infos])
    reasons = sorted({int(i.reason) for i in infos})
    solve_ms = float(sum(i.solve_ms for i in infos))
    if rank == 0:
        print(json.dumps({
            "metric": "pprts 3_10 spectral loop g-points/s", "value": ng / dt, "unit": "g-points/s", "n_gpus": world,
            "seconds": dt, "higher_is_better": True, "scaling": "strong", "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.sw} SW + {args.lw} LW g-points on {Nx}x{Ny}x{Nz}, whole pipeline per g-point on the "
                                   f"device, g-points dealt round-robin to GPUs, results all-reduced once",
                       "cells_gpoints_per_s": ng * Nx * Ny * Nz / dt, "rank0_gpoints": len(mine),
                       "rank0_iterations_min_med_max": [int(its.min()), float(np.median(its)), int(its.max())],
                       "rank0_reasons": reasons, "rank0_diffuse_solve_ms_total": solve_ms,
                       "toa_net_down_Wm2": float((acc[0][:, :, 0] + acc[3][:, :, 0] - acc[1][:, :, 0]).mean())}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

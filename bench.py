#!/usr/bin/env python3
"""bench.py -- pprts 3_10 diffuse-solve throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one diffuse solve (I - T) x = b with the reference's stop rule (rtol 1e-5, atol
1e-4*Nx*Ny*(Nz+1), src/pprts_base.F90:1126-1131) from a zero initial guess on one synthetic solar
g-point; inputs (coefficient blocks, RHS) are resident in HBM before the timed region.  At N > 1 the
domain is sharded 2-D in x/y exactly like the reference's DMDA (src/pprts_base.F90:747-790, 972-990), one rank
per GPU, face halos + dot products over RCCL (host-staged gloo when several ranks have to share a device).
Domain:  --scaling strong (the default at N > 1): the global domain is --nx x --ny whatever N is -- BASELINE.json's metric,
         "256x256x64 at 1/2/4/8 GPU"; config.baseline_config names the BASELINE line a run corresponds to;
         --scaling weak: every GPU owns --nx x --ny columns (256 x 256 x 64 per GPU);
         --global-nx / --global-ny: an explicit global domain, e.g. config 3 = `--gpus 8 --global-nx 512 --global-ny 512`
         (2 x 4 ranks of 256 x 128 columns).
Prints ONE JSON line on rank 0.
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
try:  # taken before any OpenMP runtime gets a chance to bind this thread to a single place
    _AFFINITY = len(os.sched_getaffinity(0))
except AttributeError:
    _AFFINITY = os.cpu_count() or 1

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nx", type=int, default=256, help="columns in x: per GPU (weak scaling) or of the global domain (strong)")
    ap.add_argument("--ny", type=int, default=256, help="columns in y, likewise")
    ap.add_argument("--scaling", choices=("auto", "weak", "strong"), default="auto",
                    help="auto = strong: BASELINE.json's metric is the 256x256x64 domain at 1/2/4/8 GPUs (fixed total work)")
    ap.add_argument("--global-nx", type=int, default=0, help="explicit global domain (implies fixed total work)")
    ap.add_argument("--global-ny", type=int, default=0)
    ap.add_argument("--transport", choices=("auto", "rccl", "peer", "host"), default="auto",
                    help="auto: with a device per rank the device-resident peer transport (IPC-mapped mailboxes over xGMI, "
                         "tsx_peer.hip) if its self test passes on every rank, else RCCL; with ranks sharing a device host-staged "
                         "(gloo).  peer / rccl / host force one (peer also works with ranks sharing a device)")
    ap.add_argument("--nz", type=int, default=64)
    ap.add_argument("--solver", default="3_10")
    ap.add_argument("--pc", type=int, default=3,
                    help="0 none, 1 column-block Jacobi, 2 zebra line-GS over column blocks, 3 red-black GS over column blocks")
    ap.add_argument("--pc-sweeps", type=int, default=0, help="half-grid passes - 1; 0 = the library's choice")
    ap.add_argument("--explicit", action="store_true", help="the explicit (stationary) solver instead of flexible BiCGStab "
                    "(-solar_diff_explicit, src/pprts.F90:2799); a side measurement, the headline is the Krylov solve")
    ap.add_argument("--seed", type=int, default=20240611, help="seed of the synthetic cloud field (the headline uses the default)")
    ap.add_argument("--cover", type=float, default=0.3, help="cloud cover of the synthetic field (the headline uses 0.3)")
    ap.add_argument("--field", choices=("clouds", "heterogeneous"), default="clouds",
                    help="clouds: SURVEY 8(d)'s field (homogeneous clear-sky background + cloud layer; the headline). "
                         "heterogeneous: every cell its own kabs / ksca (log-normal noise on background and clouds), so no two "
                         "cells share a transport block")
    ap.add_argument("--skip-no-sharing", action="store_true",
                    help="skip the second (reported, never `value`) leg that repeats the solves with every block stored per cell")
    ap.add_argument("--skip-extra-legs", action="store_true",
                    help="skip the reported-only legs config.all_fp64 (fp64 recurrence / no reduced precision anywhere) and "
                         "config.heterogeneous (every cell its own block)")
    ap.add_argument("--check-every", type=int, default=None, help="host looks at the convergence flag every n iterations (library default 4)")
    ap.add_argument("--kernel-reps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="edge of the CPU-baseline sample domain (columns); 0 = 256 with >= 16 threads, else 112")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads (= subdomains) of the CPU baseline; 0 = usable cores (affinity, cgroup quota), at most 128")
    return ap.parse_args()


def _free_port():
    import socket

    so = socket.socket()
    so.bind(("127.0.0.1", 0))
    port = so.getsockname()[1]
    so.close()
    return port


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N rank processes from here.  This parent never touches
    the GPU (no torch import, no HIP call); it relays rank 0's JSON line and exits with the first non-zero exit code."""
    import subprocess

    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TSX_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        out0 = procs[0].communicate()[0]
        for pr in procs:
            pr.wait()
            rc = rc or pr.returncode
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    for line in out0.decode().splitlines():   # ONE JSON line: libraries (gloo) chat on the ranks' stdout
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    import torch
    import torch.distributed as dist

    from tenstream_amd import DiffuseSolver
    from tenstream_amd import synthetic as S
    from tenstream_amd.coord import decompose

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count must equal --gpus")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no GPU visible (libtsx has no CPU fallback)")
    transport = args.transport
    try_peer_first = False
    if transport == "auto":
        transport = "rccl" if ndev >= world else "host"
        try_peer_first = transport == "rccl" and world > 1
    if transport == "rccl" and ndev < world:
        raise SystemExit(f"bench.py: RCCL needs one device per rank ({world} ranks, {ndev} devices); use --transport host")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        sys.stdout.flush()
        keep_fd = os.dup(1)
        os.dup2(2, 1)   # gloo / RCCL print their banners on stdout: the line contract wants exactly one JSON line there
        try:
            if transport == "rccl":
                dist.init_process_group("nccl", device_id=dev)
            else:   # host-staged exchanges, or only the bootstrap of the peer transport
                dist.init_process_group("gloo", rank=rank, world_size=world)
        finally:
            os.dup2(keep_fd, 1)
            os.close(keep_fd)

    # ---- domain -------------------------------------------------------------------------------------
    npx, npy = decompose(world)
    Nz = args.nz
    if args.global_nx or args.global_ny:
        Nx, Ny, scaling = args.global_nx or args.nx, args.global_ny or args.ny, "strong"
    elif args.scaling in ("strong", "auto") and world > 1:
        Nx, Ny, scaling = args.nx, args.ny, "strong"
    else:   # one GPU (either reading gives the same domain; the line says "weak" as the contract's example does), or --scaling weak
        Nx, Ny, scaling = args.nx * npx, args.ny * npy, "weak"
    co = decompose.coord(rank, world, Nx, Ny)
    dx, dz, albedo = 100.0, 50.0, 0.1
    solver = args.solver

    # ---- synthetic optical properties for the owned block (same seed on all ranks -> one global field).  Only the cloud
    # mask is generated globally; delta scaling and the source term are evaluated on the owned block plus one periodic
    # halo column/row (the source of a side stream comes from the neighbouring column), so set-up cost does not grow with N
    def make_field(field):
        kabs, ksca, g = S.cloud_field(Nx, Ny, Nz, seed=args.seed, cover=args.cover, heterogeneous=field == "heterogeneous")
        jj = np.arange(co.ys - 1, co.ys + co.ym + 1) % Ny
        ii = np.arange(co.xs - 1, co.xs + co.xm + 1) % Nx
        kabs, ksca, g = (np.ascontiguousarray(a[np.ix_(jj, ii)]) for a in (kabs, ksca, g))
        kabs, ksca, g = S.delta_scale(kabs, ksca, g)
        b_slab = S.solar_source(solver, kabs, ksca, g, dz, dx, np.full((co.ym + 2, co.xm + 2), albedo))
        loc = tuple(np.ascontiguousarray(a[1:-1, 1:-1]) for a in (kabs, ksca, g))
        return loc, torch.tensor(np.ascontiguousarray(b_slab[1:-1, 1:-1]), device=dev)

    (kabs_l, ksca_l, g_l), b = make_field(args.field)
    l1d = torch.zeros(Nz, dtype=torch.uint8, device=dev)
    a11 = torch.zeros((co.ym, co.xm, Nz), dtype=torch.float64, device=dev)
    a12 = torch.zeros_like(a11)
    alb = torch.full((co.ym, co.xm), albedo, dtype=torch.float64, device=dev)

    s = DiffuseSolver(solver, Nz, co.xm, co.ym, xs=co.xs, ys=co.ys, glob_xm=Nx, glob_ym=Ny, rank=rank, nranks=world,
                      neighbors=(co.west, co.east, co.south, co.north), device=dev_index)
    if try_peer_first:
        from tenstream_amd import hostcomm

        if hostcomm.attach_peer_checked(s):   # agreed over the process group: every rank passed the self test
            transport = "peer"
    if world > 1 and transport == "rccl":
        uid = [s.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ok = torch.ones(1, device=dev)
        try:
            s.comm_init(uid[0])
        except Exception as e:   # no second communicator on this node: agreed below, then host-staged exchanges over gloo
            print(f"bench.py: rank {rank}: RCCL communicator refused ({e})", file=sys.stderr)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 1.0:
            from tenstream_amd import hostcomm

            transport = "host"
            hostcomm.attach(s, rank, group=dist.new_group(backend="gloo"))
    elif world > 1 and transport == "peer" and not try_peer_first:
        from tenstream_amd import hostcomm

        hostcomm.attach_peer(s)
    elif world > 1:
        from tenstream_amd import hostcomm

        hostcomm.attach(s, rank)
    # coefficient blocks come from the product path: LUT (synthetic stand-in table in the reference's exact
    # shape/ordering) uploaded once, then N-linear interpolation per cell on the device (tsx_diff_set_optprop)
    from tenstream_amd import lut as LUT

    s.set_lut_diffuse(LUT.synthetic_diffuse_table(solver), LUT.diffuse_axes(solver))
    dev_f = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
    dz_l = torch.full((co.ym, co.xm, Nz), dz, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t_setup = time.perf_counter()
    optprop = (dev_f(kabs_l), dev_f(ksca_l), dev_f(g_l), dz_l, dx, l1d, a11, a12, alb)
    s.set_optprop(*optprop)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    torch.cuda.empty_cache()
    x = torch.zeros_like(b)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up + timed steps -------------------------------------------------------------------------
    kw = dict(pc=args.pc, pc_sweeps=args.pc_sweeps, check_every=args.check_every, explicit_solver=int(args.explicit),
              maxit=10000 if args.explicit else None)

    def timed_steps(rhs=None, **override):
        """W untimed + K timed solves from a zero guess, barrier + synchronize on both sides, max over ranks.  A solve that
        fails (a bounded wait of the peer transport expired ...) is recorded, not raised: every rank still reaches the barriers"""
        got, failed = [], None
        rhs = b if rhs is None else rhs
        kws = dict(kw, **override)

        def one():
            nonlocal failed
            if failed is None:
                try:
                    return s.solve(rhs, x, initial_guess_zero=1, **kws)
                except Exception as e:   # noqa: BLE001
                    failed = e
            return None

        first = None
        for q in range(args.warmup):
            tw = time.perf_counter()
            iw = one()   # (solve() returns after the stream has been synchronised)
            if q == 0 and iw is not None:   # the handle's first solve with these options: no iteration-count hint yet
                first = {"wall_ms": (time.perf_counter() - tw) * 1e3, "solve_ms_device": iw.solve_ms, "iterations": iw.niter}
        barrier()
        t0 = time.perf_counter()
        marks = [t0]
        for _ in range(args.steps):   # "zero initial guess" is part of the workload: the caller says so, x is not read
            got.append(one())
            marks.append(time.perf_counter())
        barrier()
        el = time.perf_counter() - t0
        per = sorted((marks[i + 1] - marks[i]) * 1e3 for i in range(len(marks) - 1))
        timed_steps.last = {"first_solve": first,
                            "step_ms_min_med_max": [per[0], per[len(per) // 2], per[-1]] if per else None}
        if world > 1:
            tt = torch.tensor([el, 0.0 if failed is None else 1.0], dtype=torch.float64,
                              device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt[0].item())
            if float(tt[1].item()) > 0 and failed is None:
                failed = RuntimeError("a solve failed on another rank")
        return el, got, failed

    dt, infos, failed = timed_steps()
    if failed is not None and world > 1 and transport == "peer":
        # the peer transport passed its self test and then lost a message: give it up on every rank (the all-reduce above made
        # the failure known everywhere) and measure again over RCCL, or host-staged where ranks share a device
        print(f"bench.py: rank {rank}: peer transport failed during the solves ({failed}); falling back", file=sys.stderr)
        from tenstream_amd import hostcomm

        s.comm_peer_disable()
        barrier()
        try_peer_first = True
        if dist.get_backend() == "nccl":
            uid = [s.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            s.comm_init(uid[0])
            transport = "rccl"
        else:
            hostcomm.attach(s, rank)
            transport = "host"
        dt, infos, failed = timed_steps()
    if failed is not None:
        raise failed
    cells_total = Nx * Ny * Nz
    value = cells_total * args.steps / dt
    info = infos[-1]
    # SURVEY 8(d) also asks for a tight run (rtol 1e-8, atol 1e-30 as in tests/test_pprts_symmetry/tenstream.options) and
    # a warm start (previous solution as initial guess, default tolerances): reported in `config`, never part of `value`
    spread = dict(timed_steps.last)
    # the timed region repeats ONE system, so the host's first look at the convergence flag comes where the previous solve ended
    # (check_every = 0, krylov_run): one synchronisation per solve.  Reported beside it: the same steps with the hint off
    # (check_every fixed at 2, the library's non-adaptive cadence: a look every second iteration) -- never part of `value`
    no_hint = None
    if args.check_every is None and not args.explicit and world == 1:
        dt_nh, infos_nh, failed_nh = timed_steps(check_every=2)
        if failed_nh is None:
            no_hint = {"cells_per_s": cells_total * args.steps / dt_nh, "ms_per_step": dt_nh / args.steps * 1e3,
                       "iterations": infos_nh[-1].niter, "step_ms_min_med_max": timed_steps.last["step_ms_min_med_max"],
                       "how": "check_every = 2: no iteration-count hint, the host looks at the flag every second iteration"}
    x.zero_()
    tight = s.solve(b, x, rtol=1e-8, atol=1e-30, **kw)
    warm = s.solve(b, x, **kw)

    # ---- rooflines, HIP events on the solver's stream (tsx_bench_kernel): the operator apply, one whole iteration, and
    # the preconditioner (one application = pc_sweeps + 1 half-grid passes; one intermediate pass of the scan kernels)
    spmv_ms = s.bench_kernel(0, args.kernel_reps)
    iter_ms = s.bench_kernel(1, max(4, args.kernel_reps // 4))
    bytes_spmv = s.algorithmic_bytes(0)
    bytes_iter = s.algorithmic_bytes(1)
    pc_ms = pass_ms = None
    dd_on, dd_nent = s.dedup_info()
    pc_shared = dd_on or bool(getattr(s, "dedup_mode", 0) & 2)   # the preconditioner reads per-block records through an index
    pc_ran, sweeps, scan, rec_shared = s.pc_info()   # what the solves above actually ran (automatic pass count, zebra on odd grids)
    flow = flow_ms = bytes_flow = None
    if scan and args.pc_sweeps == 0:
        pc_ms = s.bench_kernel(2, args.kernel_reps)
        pass_ms = s.bench_kernel(3, 4 * args.kernel_reps)
        flow = s.flow_info()   # of the application just timed: do the intermediate passes run as one launch (tsx_k_pcs_flow)?
        if flow["in_use"]:
            try:
                flow_ms = s.bench_kernel(4, args.kernel_reps)
                flow = s.flow_info()   # (as the Krylov loop issues it: from pass 1)
                bytes_flow = s.algorithmic_bytes(4)
            except Exception:   # several ranks: the launch holds the rank faces and cannot be timed alone (tsx_bench_kernel says so)
                flow_ms = bytes_flow = None
    bw = s.probe_bandwidth(1 << 30, 5)   # what plain streaming kernels reach on this box: copy and read-only, best variant each
    copy_gbps = bw["copy_GBps"]
    # the byte counts follow the storage format in use: take them while the solver is in the state that was timed
    bytes_survey_spmv = s.algorithmic_bytes(10)
    bytes_pass, bytes_pc = (s.algorithmic_bytes(3), s.algorithmic_bytes(2)) if pass_ms is not None else (None, None)

    # ---- second leg, reported under config.no_sharing and never part of `value`: the same solves with every cell's block
    # stored (TSX_DEDUP=0 / TSX_PC_RECSHARE=0: what any field whose cells all differ gets, e.g. --field heterogeneous)
    no_sharing = None
    if (dd_on or getattr(s, "dedup_mode", 0) & 2) and not args.skip_no_sharing and not args.explicit:
        keep = {k: os.environ.get(k) for k in ("TSX_DEDUP", "TSX_PC_RECSHARE")}
        os.environ["TSX_DEDUP"] = "0"
        os.environ["TSX_PC_RECSHARE"] = "0"
        s.set_optprop(*optprop)
        dt_ns, infos_ns, failed_ns = timed_steps()
        if failed_ns is not None:
            raise failed_ns
        it_ns = s.bench_kernel(1, max(4, args.kernel_reps // 4))
        pass_ns = s.bench_kernel(3, 4 * args.kernel_reps) if scan and args.pc_sweeps == 0 else None
        spmv_ns = s.bench_kernel(0, args.kernel_reps)
        b_it = s.algorithmic_bytes(1)
        no_sharing = {"cells_per_s": cells_total * args.steps / dt_ns, "ms_per_step": dt_ns / args.steps * 1e3,
                      "iterations": infos_ns[-1].niter, "reason": infos_ns[-1].reason, "iter_ms": it_ns,
                      "iter_frac_of_B_iter": b_it / (it_ns * 1e-3) / 1e9 / 8000.0, "B_iter_bytes": b_it,
                      "spmv_ms": spmv_ns, "spmv_frac": s.algorithmic_bytes(0) / (spmv_ns * 1e-3) / 1e9 / 8000.0,
                      "pass_ms": pass_ns,
                      "pass_frac": None if pass_ns is None else s.algorithmic_bytes(3) / (pass_ns * 1e-3) / 1e9 / 8000.0,
                      "how": "TSX_DEDUP=0 TSX_PC_RECSHARE=0, same field, same solves"}
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        s.set_optprop(*optprop)   # back to the default storage for what follows

    # ---- third leg (reported under config.all_fp64, never `value`): the same solves without the fp32 recurrence --
    # "recurrence_fp64": r, s, v, t in fp64, only the preconditioned directions in fp32 (fp32_directions = 1);
    # "everything_fp64": no reduced precision anywhere, the preconditioner on the exact fp64 blocks (fp32_directions = 0,
    # pc_coeff_fp16 = 0) -- the reference's default arithmetic (ireals = real64) end to end
    all_fp64 = None
    if world == 1 and not args.skip_extra_legs and not args.explicit:   # (one rank: the driver's N = 1 line carries it)
        all_fp64 = {}
        for name, ov in (("recurrence_fp64", dict(fp32_directions=1)), ("everything_fp64", dict(fp32_directions=0, pc_coeff_fp16=0))):
            dt_f, infos_f, failed_f = timed_steps(**ov)
            if failed_f is not None:
                raise failed_f
            pcf = s.pc_info()
            all_fp64[name] = {"cells_per_s": cells_total * args.steps / dt_f, "ms_per_step": dt_f / args.steps * 1e3,
                              "iterations": infos_f[-1].niter, "reason": infos_f[-1].reason,
                              "rel_residual": infos_f[-1].rnorm / infos_f[-1].rnorm0,
                              "preconditioner": {0: "none", 1: "column-jacobi", 2: f"column-zebra({pcf[1] + 1} passes)",
                                                 3: f"column-red-black({pcf[1] + 1} passes)"}.get(pcf[0], str(pcf[0])),
                              "options": ov}
        all_fp64["which_one_a_real64_build_gets"] = (
            "neither by default: tsx_default_ksp_opts sets fp32_directions = 2 whatever the caller's ireals (the headline: x, b, dots, "
            "operator and stop rule fp64, recurrence vectors fp32 with fp64 residual replacement); recurrence_fp64 is what "
            "fp32_directions = 1 -- and any rtol < 1e-7 -- selects; everything_fp64 (fp32_directions = 0, pc_coeff_fp16 = 0) is the "
            "reference's arithmetic end to end and the retry solver")

    # ---- fourth leg (config.heterogeneous, never `value`): the general field -- log-normal noise on every cell's kabs / ksca,
    # so no two cells share a transport block (what an LES humidity / aerosol field + gas optics delivers); same generator
    # seed, same solves.  The operator reads every cell's own block, the preconditioner groups near-identical ones
    heterogeneous = None
    if world == 1 and args.field == "clouds" and not args.skip_extra_legs and not args.explicit:
        (ka_h, ks_h, g_h), b_h = make_field("heterogeneous")
        opt_h = (dev_f(ka_h), dev_f(ks_h), dev_f(g_h)) + optprop[3:]
        s.set_optprop(*opt_h)
        dt_h, infos_h, failed_h = timed_steps(rhs=b_h)
        if failed_h is not None:
            raise failed_h
        on_h, nent_h = s.dedup_info()
        _, sw_h, scan_h, _ = s.pc_info()
        it_h = s.bench_kernel(1, max(4, args.kernel_reps // 4))
        sp_h = s.bench_kernel(0, args.kernel_reps)
        pass_h = s.bench_kernel(3, 4 * args.kernel_reps) if scan_h and args.pc_sweeps == 0 else None
        pc_h = s.bench_kernel(2, args.kernel_reps) if scan_h and args.pc_sweeps == 0 else None
        tag_h = f"{co.xm}x{co.ym}x{Nz}_heterogeneous"
        kname_h = "tsx_k_pcs_rb" if solver == "3_10" else "tsx_k_pcsh_rb"
        tr_sp, src_h = pmc_traffic(solver, tag_h, ["tsx_k_spmv", (",0,1,double,double", ",0,2,double,double"), ",false,double>"])
        tr_pass, _ = pmc_traffic(solver, tag_h, [kname_h, (",true,0,true,2,", ",true,0,true,2>")])
        tr_it, _ = pmc_traffic(solver, tag_h, None)
        heterogeneous = {
            "cells_per_s": cells_total * args.steps / dt_h, "ms_per_step": dt_h / args.steps * 1e3,
            "iterations": infos_h[-1].niter, "reason": infos_h[-1].reason, "rel_residual": infos_h[-1].rnorm / infos_h[-1].rnorm0,
            "distinct_blocks_or_groups": nent_h, "dedup_mode": int(getattr(s, "dedup_mode", 0)),
            "iter_ms": it_h, "iter_frac_of_B_iter": s.algorithmic_bytes(1) / (it_h * 1e-3) / 1e9 / 8000.0,
            "iter_traffic_bytes": tr_it,
            "spmv_ms": sp_h, "spmv_bytes": s.algorithmic_bytes(0), "spmv_frac": s.algorithmic_bytes(0) / (sp_h * 1e-3) / 1e9 / 8000.0,
            "spmv_traffic_bytes": tr_sp,
            "pass_ms": pass_h, "pass_bytes": None if pass_h is None else s.algorithmic_bytes(3),
            "pass_frac": None if pass_h is None else s.algorithmic_bytes(3) / (pass_h * 1e-3) / 1e9 / 8000.0,
            "pass_traffic_bytes": tr_pass, "pc_ms": pc_h, "passes": sw_h + 1, "traffic_source": src_h,
            "how": "--field heterogeneous of the same seed: log-normal noise on every cell's kabs / ksca"}
        del opt_h, b_h
        s.set_optprop(*optprop)

    def roof(kernel, ms, nbytes, patterns, full_storage_bytes=None):
        ach = nbytes / (ms * 1e-3) / 1e9
        traffic, src = pmc_traffic(solver, f"{co.xm}x{co.ym}x{Nz}" + ("" if args.field == "clouds" else "_" + args.field), patterns)
        r = {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
             "traffic": traffic, "traffic_source": src, "bytes_per_launch": nbytes, "ms_per_launch": ms,
             "traffic_GBps": None if traffic is None else traffic / (ms * 1e-3) / 1e9,
             # beside the nominal peak: the guide's achievable rate (MI355X_MICROARCH.md, HBM: about 6.3 TB/s) and this box's own
             # best streaming copy (tsx_probe_bandwidth) -- algorithmic bytes, and the PMC traffic where a profile is committed
             "frac_of_achievable": ach / 6300.0,
             "frac_of_measured_copy": ach / copy_gbps,
             "traffic_frac_of_achievable": None if traffic is None else traffic / (ms * 1e-3) / 1e9 / 6300.0}
        if full_storage_bytes is not None and full_storage_bytes != nbytes:
            # `achieved` counts the bytes of the storage format in use (distinct blocks once + a per-cell index); SURVEY
            # 8(d)'s figure for every cell's block stored is kept beside it (an equivalent rate, not a bandwidth)
            r["survey_bytes_per_launch"] = full_storage_bytes
            r["survey_equivalent_GBps"] = full_storage_bytes / (ms * 1e-3) / 1e9
        return r

    out = None
    if rank == 0:
        r_spmv = roof("tsx_k_spmv_w (y = (I - T) x, fp64 x and y)", spmv_ms, bytes_spmv,
                      ["tsx_k_spmv", (",0,1,double,double", ",0,2,double,double"), ",true,double>" if dd_on else ",false,double>"],
                      bytes_survey_spmv)
        r_iter = roof("one BiCGStab iteration (2 M^-1, 2 SpMV, 3 vector updates)", iter_ms, bytes_iter, None)
        if not dd_on:   # the committed PMC profile is of the shared-block kernels
            r_iter["traffic"] = r_iter["traffic_GBps"] = r_iter["traffic_source"] = None
        r_iter["basis"] = ("SURVEY 8(d)'s 2*B_spmv + 16*N*sv: the reference's iteration without M^-1, every cell's block "
                           "stored; this iteration also runs M^-1 twice and shares identical blocks, see `traffic`")
        r_pass = None
        if pass_ms is not None:
            kname = "tsx_k_pcs_rb" if solver == "3_10" else "tsx_k_pcsh_rb"
            r_pass = roof(f"{kname}<..., GS, MODE 0, RQ 2> (one intermediate red-black pass of M^-1)", pass_ms,
                          bytes_pass, [kname, (",true,0,true,2,", ",true,0,true,2>") if pc_shared else (",true,0,false,2,", ",true,0,false,2>")])
        r_pc = None
        if pc_ms is not None:
            r_pc = {"kernel": f"M^-1: {sweeps + 1} half-grid passes", "ms_per_application": pc_ms,
                    "bytes_per_application": bytes_pc,
                    "achieved": bytes_pc / (pc_ms * 1e-3) / 1e9, "unit": "GB/s"}
        # the kernel the solve spends most of its time in: the preconditioner pass (about half of an iteration) when the
        # scan kernels run, else the operator apply
        dominant = r_pass if r_pass is not None else r_spmv
        r_flow = None
        if flow_ms is not None:
            # the intermediate passes of an application run as ONE launch: that launch is where the solve spends its time
            npass = flow["end_pass"] - flow["first_pass"]
            r_flow = roof(f"tsx_k_pcs_flow (passes {flow['first_pass']}..{flow['end_pass'] - 1} of one M^-1 application in one launch: "
                          f"{npass} intermediate red-black passes, work items (pass, tile) behind a ticket counter)", flow_ms,
                          bytes_flow, ["tsx_k_pcs_flow"])
            r_flow["passes_per_launch"] = npass
            r_flow["us_per_pass"] = flow_ms * 1e3 / npass
            r_flow["the_same_pass_as_its_own_launch_us"] = pass_ms * 1e3
            dominant = r_flow
        out = {
            "metric": f"pprts {solver} diffuse-solve cells/s" + (" (explicit solver)" if args.explicit else ""),
            "value": value,
            "unit": "cells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            # x, b, every dot product, the operator's arithmetic and the stop rule (on the true residual) are f64; the recurrence
            # vectors and directions are stored in f32 (fp32_directions = 2) unless --explicit / a tight rtol selects the f64 recurrence
            "dtype": "f64" if args.explicit else "f64/f32",
            "data": "synthetic",
            "config": {
                "workload": f"pprts {solver} diffuse solve, {Nx}x{Ny}x{Nz} cells global ({co.xm}x{co.ym}x{Nz} on rank 0), "
                            f"{scaling} scaling, single solar g-point, rtol 1e-5 / reference atol, zero initial guess",
                "process_grid": f"{npx}x{npy}",
                "transport": "none (1 rank)" if world == 1 else {"rccl": "RCCL" + (" (after the peer transport failed its self test or a solve)" if try_peer_first else ""),
                                                                    "peer": "device-resident peer mailboxes (HIP IPC)",
                                                                    "host": "host-staged (gloo)" + (" (after the peer transport failed)" if try_peer_first else "")}[transport],
                "coeff_storage": "fp32 blocks (lossless); x, b, dots, stop rule fp64; recurrence vectors r, s, v, t, directions p, "
                                 "p-hat, s-hat and shadow residual fp32, the residual replaced by b - A x in fp64 before "
                                 "convergence is declared (fp32_directions = 2)",
                "preconditioner_storage": "inside M^-1 only: fp16 column blocks and side-to-top couplings, fp8 side-to-side "
                                          "couplings, fp32/bf16 iterates; the operator always uses the exact blocks",
                "coeff_source": "device N-linear LUT interpolation (tsx_diff_set_optprop), synthetic table",
                "coeff_dedup": dict(in_use=dd_on, distinct_blocks=dd_nent, cells_local=co.xm * co.ym * Nz,
                                    preconditioner_groups_near_identical_blocks=bool(getattr(s, "dedup_mode", 0) & 2),
                                    note="bit-identical blocks are stored once behind a per-cell index (lossless; TSX_DEDUP=0 "
                                         "disables); the rooflines count the bytes of this format"),
                "preconditioner_records_shared": rec_shared,
                "coeff_setup_ms": t_setup * 1e3,
                "preconditioner": {0: "none", 1: "column-jacobi", 2: f"column-zebra({sweeps + 1} passes)",
                                   3: f"column-red-black({sweeps + 1} passes)"}.get(pc_ran, str(pc_ran)),
                "iterations": info.niter,
                "reason": info.reason,
                "rel_residual": info.rnorm / info.rnorm0,
                "solve_ms_device": info.solve_ms,
                "import_ms": info.import_ms,
                "export_ms": info.export_ms,
                "iter_ms": iter_ms,
                "first_solve": spread["first_solve"],
                "step_ms_min_med_max": spread["step_ms_min_med_max"],
                "no_hint": no_hint,
                "flow_kernel": flow,
                "tight_run": {"rtol": 1e-8, "iterations": tight.niter, "reason": tight.reason, "solve_ms": tight.solve_ms},
                "warm_start": {"iterations": warm.niter, "reason": warm.reason, "solve_ms": warm.solve_ms},
                "iter_GBps": bytes_iter / (iter_ms * 1e-3) / 1e9,
                "copy_GBps_measured": copy_gbps,
                "read_GBps_measured": bw["read_GBps"],
                "bandwidth_probe": dict(bw, note="best of 1 / 4 / 8 sixteen-byte accesses per lane in flight, plain and "
                                                 "non-temporal, grids of 2048 / 4096 / 16384 workgroups, 1 GiB"),
                "field": args.field,
                "no_sharing": no_sharing,
                "all_fp64": all_fp64,
                "heterogeneous": heterogeneous,
                "baseline_config": baseline_config(solver, Nx, Ny, Nz, world, npx, npy, scaling),
            },
            "roofline": dominant,
            "roofline_pass": r_pass,
            "roofline_spmv": r_spmv,
            "roofline_iter": r_iter,
            "roofline_pc": r_pc,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, solver, dx, dz, albedo, s, b, x, Nx, Ny, kw)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def pmc_traffic(solver, size, patterns):
    """HBM-side bytes per launch of one kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN/traffic_<solver>_<local size>.json, written by scripts/pmc_summary.py: separate FETCH_SIZE / WRITE_SIZE
    passes, gfx950 FETCH_SIZE x2 correction, launches that exit at once after convergence excluded).  PMC passes cannot run
    inside the timed bench, so the figure is looked up; None if no profile of this solver and size is committed.
    patterns: substrings the kernel name must contain; None = the sum over one iteration (entry "iteration")."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, "profiles", "r*", f"traffic_{solver}_{size}.json")), reverse=True):
        try:
            doc = json.load(open(path))
        except Exception:
            continue
        if patterns is None:
            it = doc.get("iteration")
            if it:
                return it["traffic_bytes"], os.path.relpath(path, here)
            continue
        ok = lambda name, p: any(q in name for q in p) if isinstance(p, tuple) else p in name
        hit = [v for name, v in doc["kernels"].items() if all(ok(name, p) for p in patterns) and v["launches"] > 0]
        if hit:
            n = sum(v["launches"] for v in hit)
            return sum(v["traffic_bytes_per_launch"] * v["launches"] for v in hit) / n, os.path.relpath(path, here)
    return None, None


def usable_cores():
    """CPUs this process may actually use: affinity mask and the cgroup CPU quota (the GPU boxes are slices of a node:
    256 logical CPUs visible, cpu.max = 16 CPUs' worth of time), not just os.cpu_count()."""
    n = min(os.cpu_count() or 1, _AFFINITY)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def baseline_config(solver, Nx, Ny, Nz, world, npx, npy, scaling):
    """Which line of BASELINE.json `configs` / `metric` this run is (None: a workload BASELINE.json does not list)."""
    if solver == "3_10" and (Nx, Ny, Nz) == (256, 256, 64):
        return f"metric: pprts 3_10 diffuse-solve cells/s on 256x256x64 at {world} GPU" + ("s" if world > 1 else "")
    if solver == "3_10" and (Nx, Ny, Nz) == (128, 128, 64) and world == 1:
        return "configs[1]: pprts 3_10, 128x128x64 single solar g-point, 1xMI355X"
    if solver == "3_10" and (Nx, Ny, Nz) == (512, 512, 64) and world == 8 and (npx, npy) == (2, 4):
        return "configs[2]: pprts 3_10, 512x512x64 domain-decomposed 2x4 across 8xMI355X"
    if solver == "8_16" and (Nx, Ny, Nz) == (256, 256, 64) and world == 1:
        return "configs[4]: pprts 8_16 higher-order streams, 256x256x64, 1xMI355X"
    return None


def cpu_baseline(args, solver, dx, dz, albedo, dev_solver=None, dev_b=None, dev_x=None, Nx=0, Ny=0, solve_kw=None):
    """The reference's default CPU path as restated by the oracle, timed on this box's host cores on a bounded sample of
    the same generator: assembled AIJ + KSPFBCGS + PCBJACOBI/ILU(0), one subdomain per core like one MPI rank per core
    (src/pprts.F90:4342-4371, 4415-4425).  With one thread this is the 1-rank default (plain ILU(0)).

    When the sample IS the benchmark domain (>= 16 usable cores), the CPU solves the very system the device solved -- the
    coefficient blocks are read back from the device (tsx_diff_get_coeffs), same b -- and `parity_check` compares the two
    converged solutions (both tightened to ~1e-10; outside every timed region)."""
    from oracle import oracle as O
    from tenstream_amd import synthetic as S
    from tenstream_amd.coord import decompose

    cores = usable_cores()
    threads = args.cpu_threads if args.cpu_threads > 0 else min(cores, 128)
    n = args.cpu_sample if args.cpu_sample > 0 else (256 if threads >= 16 else 112)
    npx, npy = decompose(threads)
    lay = O.layout(solver, args.nz, n, n)
    rt, at, mx = O.default_tolerances(n, n, args.nz + 1)
    same = dev_solver is not None and n == Nx and n == Ny and args.nz == dev_solver.Nz
    parity = None
    if same:
        coeff = dev_solver.get_coeffs()
        bh = dev_b.cpu().numpy()
        Pz = dict(l1d=np.zeros(args.nz, dtype=np.uint8), a11=np.zeros((n, n, args.nz)), a12=np.zeros((n, n, args.nz)),
                  albedo=np.full((n, n), albedo))
        x, info = O.solve_bjacobi_ilu_mt(lay, coeff, Pz["l1d"], Pz["a11"], Pz["a12"], Pz["albedo"], bh, npx, npy, rtol=rt,
                                         atol=at, maxit=mx, tighten=(1e-6, 1e-30, 2000))
        del coeff
        dev_x.zero_()
        di = dev_solver.solve(dev_b, dev_x, **dict(solve_kw or {}, rtol=1e-10, atol=1e-30))
        xd = dev_x.cpu().numpy()
        xt = info["x_tight"]
        scale = float(np.abs(xt).max())
        dev_x.zero_()
        dd = dev_solver.solve(dev_b, dev_x, **(solve_kw or {}))
        xdd = dev_x.cpu().numpy()
        parity = {"max_rel_err": float(np.abs(xd - xt).max() / scale),
                  "device": {"rtol": 1e-10, "its": di.niter, "reason": di.reason},
                  "cpu": {"its": info["niter"] + info["niter_tight"], "reason": info["reason_tight"],
                          "how": "the timed default-tolerance solve continued to 1e-6 of its final residual with the same factors"},
                  "default_tolerance_max_rel_err": {"device_vs_tight": float(np.abs(xdd - xt).max() / scale),
                                                    "cpu_vs_tight": float(np.abs(x - xt).max() / scale)},
                  "system": "identical: the device's coefficient blocks (tsx_diff_get_coeffs) and right-hand side",
                  "norm": "max |x_device - x_cpu| / max |x_cpu| over all unknowns"}
        sample_desc = f"{n}x{n}x{args.nz}: the benchmark domain itself (the device's blocks and right-hand side)"
    else:
        P = S.make_problem(solver, Nx=n, Ny=n, Nz=args.nz, dx=dx, dz=dz, albedo=albedo)
        x, info = O.solve_bjacobi_ilu_mt(lay, P["coeff"].astype(np.float64), P["l1d"], P["a11"], P["a12"], P["albedo"],
                                         P["b"], npx, npy, rtol=rt, atol=at, maxit=mx)
        sample_desc = f"{n}x{n}x{args.nz} periodic domain of the same generator"
    cells = n * n * args.nz
    # SURVEY 8(d) B2 / B3 beside it, one core each on a smaller sample of the same generator (both are serial codes in the
    # oracle: the matrix-free FBCGS without preconditioner -- the algorithm of the GPU path minus M^-1 --, and the
    # reference's PETSc-free explicit SOR, src/pprts_explicit.F90:461-713), same tolerances rule
    m = 32
    Pm = S.make_problem(solver, Nx=m, Ny=m, Nz=args.nz, dx=dx, dz=dz, albedo=albedo)
    laym = O.layout(solver, args.nz, m, m)
    rtm, atm, mxm = O.default_tolerances(m, m, args.nz + 1)
    cm = Pm["coeff"].astype(np.float64)
    t0 = time.perf_counter()
    _, i2 = O.solve_matfree(laym, cm, Pm["l1d"], Pm["a11"], Pm["a12"], Pm["albedo"], Pm["b"], rtol=rtm, atol=atm, maxit=mxm)
    t2 = time.perf_counter() - t0
    others = [{"name": "B2 matrix-free FBCGS, no preconditioner", "value": m * m * args.nz / t2, "unit": "cells/s", "cores": 1,
               "sample": f"{m}x{m}x{args.nz}, {i2['niter']} its, reason {i2['reason']}, {t2:.2f}s"}]
    if solver == "3_10":
        t0 = time.perf_counter()
        _, i3 = O.solve_sor(laym, cm, Pm["l1d"], Pm["a11"], Pm["a12"], Pm["albedo"], Pm["b"], rtol=rtm, atol=atm)
        t3 = time.perf_counter() - t0
        others.append({"name": "B3 explicit SOR (explicit_ediff)", "value": m * m * args.nz / t3, "unit": "cells/s", "cores": 1,
                       "sample": f"{m}x{m}x{args.nz}, {i3['niter']} sweeps, converged {i3['converged']}, {t3:.2f}s"})
    return {
        "others": others,
        "value": cells / info["t_solve"],
        "unit": "cells/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{sample_desc}; assembled CSR + FBCGS + block-Jacobi/ILU(0) on "
                  f"{npx}x{npy} subdomains (one thread each), {info['niter']} its, reason {info['reason']}, solve "
                  f"{info['t_solve']:.2f}s (assembly {info['t_assemble']:.2f}s, factor {info['t_factor']:.2f}s not counted)",
        "parity_check": parity,
        "host_cores_available": cores,
        "host_cores_logical": os.cpu_count(),
    }


if __name__ == "__main__":
    main()

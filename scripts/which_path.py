"""Which sharing / preconditioner path every rank of the four-process pipeline case takes (GPU box): python scripts/which_path.py
Prints per rank and g-point: iterations, tsx_dedup_info mode and entry count, tsx_pc_info, tsx_flow_info."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def worker(rank, world, port):
    sys.path.insert(0, ROOT)
    import torch, torch.distributed as dist
    from tenstream_amd import coord, lut, synthetic, hostcomm
    from tenstream_amd.pprts import PprtsSolver
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    Nx, Ny, Nz, phi0, theta0, tall_top = 12, 10, 8, 30.0, 55.0, 1
    kabs, ksca, g = synthetic.cloud_field(Nx, Ny, Nz, seed=5); kabs *= 20.0
    dz = np.full((Ny, Nx, Nz), 50.0); dz[:, :, :tall_top] = 400.0
    planck = np.linspace(2.0, 6.0, Nz + 1)[None, None, :] * (1 + 0.05 * np.random.default_rng(0).random((Ny, Nx, 1)))
    dax = lut.direct_axes(); Tdir, Sdir = lut.synthetic_direct_tables(dax)
    co = coord.coord(rank, world, Nx, Ny); sl = (slice(co.ys, co.ys + co.ym), slice(co.xs, co.xs + co.xm))
    P = PprtsSolver(Nz, co.xm, co.ym, 100.0, 100.0, phi0, theta0, device=0, xs=co.xs, ys=co.ys, glob_xm=Nx, glob_ym=Ny, rank=rank, nranks=world,
                    neighbors=(co.west, co.east, co.south, co.north))
    P.set_lut_diffuse(lut.synthetic_diffuse_table("3_10"), lut.diffuse_axes("3_10")); P.set_lut_direct(Tdir, Sdir, dax)
    hostcomm.attach(P.core, rank)
    loc = lambda a: np.ascontiguousarray(a[sl])
    for kind in ("solar", "thermal"):
        ls = kind == "solar"
        P.set_optical_properties(0.15, loc(kabs), loc(ksca), loc(g), loc(dz), planck=None if ls else loc(planck))
        info = P.solve(1000.0 if ls else 0.0, rtol=1e-10, atol=1e-30, maxit=3000)
        on, nent = P.core.dedup_info()
        print(rank, kind, "its", info.niter, "dedup mode", P.core.dedup_mode, "nent", nent, "of", co.xm * co.ym * Nz, "pc", P.core.pc_info(), "flow", P.core.flow_info(), flush=True)
        P.get_result()
    P.close(); dist.destroy_process_group()
if __name__ == "__main__":
    import socket, torch.multiprocessing as mp
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 4, port)) for r in range(4)]
    [p.start() for p in ps]; [p.join(300) for p in ps]

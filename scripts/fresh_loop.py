"""Fresh-process loop for the rare wrong result of the four-process pipeline test (rounds 4-6; GPU box).

Every iteration starts FOUR fresh torch-free Python processes at once (numpy + ctypes only: 0.3 s each instead of the 15 s of
a pytest run with torch and gloo), one per rank of the failing case (12 x 10 columns, 8 levels, one thick 1-D top layer, 2 x 2
ranks sharing cuda:0).  Each builds its rank's PprtsSolver, waits at a file barrier so that the four first launches coincide, calls
set_optical_properties (solar), and then checks
  * the representatives of the shared-block storage (TSX_DEBUG_CHECKS: `index_check ... LOST` = ent_cell all zero behind the launch
    that writes it, and what a second launch of the same kernel leaves),
  * every device code object of libtsx.so against the file (scripts/code_verify.py),
  * (--solve, needs no peers: rank-local one-rank periodic solver of the same columns) nothing else.
usage:  python scripts/fresh_loop.py N [outdir]        -> summary on stdout, details under outdir (default gpurun_out/r06/fresh)
"""
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, it, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import numpy as np

    from tenstream_amd import coord, lut, synthetic
    from tenstream_amd.pprts import PprtsSolver

    world, Nx, Ny, Nz, phi0, theta0, tall_top = 4, 12, 10, 8, 30.0, 55.0, 1
    kabs, ksca, g = synthetic.cloud_field(Nx, Ny, Nz, seed=5)
    kabs *= 20.0
    dz = np.full((Ny, Nx, Nz), 50.0)
    dz[:, :, :tall_top] = 400.0
    dax = lut.direct_axes()
    Tdir, Sdir = lut.synthetic_direct_tables(dax)
    co = coord.coord(rank, world, Nx, Ny)
    sl = (slice(co.ys, co.ys + co.ym), slice(co.xs, co.xs + co.xm))

    def barrier(tag):
        open(os.path.join(outdir, f"{tag}.{it}.{rank}"), "w").close()
        t0 = time.time()
        while len(glob.glob(os.path.join(outdir, f"{tag}.{it}.*"))) < world and time.time() - t0 < 60:
            time.sleep(0.0005)

    if os.environ.get("FRESH_EARLY_BARRIER", "1") != "0":
        barrier("early")   # the four processes take their first device memory (tsx_create: the pool's first slabs) at the same time
    P = PprtsSolver(Nz, co.xm, co.ym, 100.0, 100.0, phi0, theta0, device=0, xs=co.xs, ys=co.ys, glob_xm=Nx, glob_ym=Ny, rank=rank,
                    nranks=world, neighbors=(co.west, co.east, co.south, co.north))
    P.set_lut_diffuse(lut.synthetic_diffuse_table("3_10"), lut.diffuse_axes("3_10"))
    P.set_lut_direct(Tdir, Sdir, dax)

    def exchange(send, recv, peers):
        raise RuntimeError("no exchange expected before the solve")

    # The test's allreduce (gloo) lines the four ranks up INSIDE set_optical_properties, a few hundred microseconds before the block-
    # sharing build takes its fresh device memory; a spin barrier over a shared page does the same here (the values need no exchange:
    # the 1-D layer flags are the same on every rank of this case, OR over ranks = identity)
    import mmap
    import struct

    shm = None
    if os.environ.get("FRESH_TIGHT", "1") != "0":
        fd = os.open(f"/dev/shm/tsx_fresh_{os.path.basename(outdir)}_{it}", os.O_RDWR)
        shm = mmap.mmap(fd, 64)
    gen = [0]

    def allreduce(buf):
        if shm is None:
            return None
        gen[0] += 1
        struct.pack_into("<q", shm, 8 * rank, gen[0])
        t0 = time.time()
        while True:
            if all(struct.unpack_from("<q", shm, 8 * r)[0] >= gen[0] for r in range(world)):
                return None
            if time.time() - t0 > 30:
                raise RuntimeError("spin barrier timed out")

    P.core.comm_set_callbacks(exchange, allreduce)
    # file barrier: the four processes enter set_optical_properties together (as they do behind gloo's rendezvous in the test)
    barrier("ready")
    loc = lambda a: np.ascontiguousarray(a[sl])
    P.set_optical_properties(0.15, loc(kabs), loc(ksca), loc(g), loc(dz), planck=None)
    import code_verify

    bad = code_verify.verify(P.lib)
    import ctypes

    st = (ctypes.c_int64 * 8)()
    P.lib.tsx_pool_stats(-1, st)
    res = {"rank": rank, "it": it, "pid": os.getpid(), "code_bad": bad[:50], "n_code_bad": len(bad), "pool": [int(v) for v in st]}
    print(json.dumps(res), flush=True)
    P.close()


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    outdir = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "r06", "fresh")
    os.makedirs(outdir, exist_ok=True)
    lost, codebad, crashed, wiped = [], [], [], []
    nproc = guard_us = 0
    t0 = time.time()
    for it in range(N):
        env = dict(os.environ, TSX_DEBUG_CHECKS=os.path.join(outdir, f"chk.{it}"))
        shm_path = f"/dev/shm/tsx_fresh_{os.path.basename(outdir)}_{it}"
        with open(shm_path, "wb") as fh:
            fh.write(b"\0" * 64)
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(it), outdir], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(4)]
        for r, p in enumerate(procs):
            try:
                out, err = p.communicate(timeout=180)
            except subprocess.TimeoutExpired:
                p.kill()
                out, err = p.communicate()
            if p.returncode != 0:
                crashed.append((it, r, p.returncode, err[-600:]))
            for line in out.splitlines():
                if line.startswith("{"):
                    d = json.loads(line)
                    if d["n_code_bad"]:
                        codebad.append(d)
                    if d.get("pool") and d["pool"][4]:
                        wiped.append(d)
                    nproc += 1
                    guard_us += d.get("pool", [0] * 8)[7]
        os.remove(shm_path)
        hit = False
        for fn in glob.glob(os.path.join(outdir, f"chk.{it}.*")):
            txt = open(fn).read()
            if "LOST" in txt or "a31e272015f12c43" in txt:
                hit = True
                lost.append((it, [l for l in txt.splitlines() if "index_check" in l or "detail" in l][:12]))
        for fn in glob.glob(os.path.join(outdir, f"ready.{it}.*")) + glob.glob(os.path.join(outdir, f"early.{it}.*")) + ([] if hit else glob.glob(os.path.join(outdir, f"chk.{it}.*"))):
            os.remove(fn)
    summary = {"iterations": N, "seconds": round(time.time() - t0, 1), "lost": lost, "code_bad": codebad, "crashed": crashed,
               "pool_quarantine_caught_a_wipe": wiped, "processes": nproc, "mean_quarantine_us_per_process": guard_us / max(nproc, 1),
               "TSX_POOL": os.environ.get("TSX_POOL", "1")}
    with open(os.path.join(outdir, "summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(f"fresh_loop: {N} iterations x 4 processes in {summary['seconds']} s: {len(lost)} with lost representatives, "
          f"{len(codebad)} processes with damaged code, {len(crashed)} crashed; the pool's quarantine caught fresh memory losing its "
          f"contents in {len(wiped)} processes (mean quarantine {guard_us / max(nproc, 1):.0f} us per process)")
    for x in wiped[:8]:
        print("WIPE-IN-QUARANTINE", json.dumps(x["pool"]), "rank", x["rank"], "it", x["it"])
    for x in lost[:5]:
        print("LOST", x)
    for x in codebad[:5]:
        print("CODE", json.dumps(x)[:1500])
    for x in crashed[:5]:
        print("CRASH", x)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    else:
        main()

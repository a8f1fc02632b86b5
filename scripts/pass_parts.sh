# usage: in the container  bash scripts/pass_parts.sh build   (hipcc, ~3 min);  on the GPU box  bash scripts/pass_parts.sh
# What an intermediate 3_10 pass spends on each kind of access: builds of the library that leave one kind out (TSX_PCS_PROBE bits,
# tsx_kernels_pcs.hpp; the results are meaningless), the pass timed alone on the metric domain, builds side by side on one box.
cd ${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = build ]; then
  cd tenstream_amd/csrc
  for b in 1 2 4 8 16 32 33 31; do
    mkdir -p ../lib_p$b
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=on -DTSX_PCS_PROBE=$b -c -o ../lib_p$b/tsx_pcs.o tsx_pcs.hip &&
      /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o ../lib_p$b/libtsx.so ../lib_p$b/tsx_pcs.o $(ls ../lib/obj/*.o | grep -v tsx_pcs.o) -ldl && rm ../lib_p$b/tsx_pcs.o ) &
  done
  wait
  exit 0
fi
for v in lib lib_p1 lib_p2 lib_p4 lib_p8 lib_p16 lib_p32 lib_p33 lib_p31 lib; do
  [ -f tenstream_amd/$v/libtsx.so ] || continue
  TSX_LIB=$PWD/tenstream_amd/$v/libtsx.so python3 - <<'PY' 2>&1 | grep -v amdgpu
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from tenstream_amd import DiffuseSolver, synthetic as S, lut as LUT
Nx = Ny = 256; Nz = 64; dev = torch.device("cuda", 0)
kabs, ksca, g = S.cloud_field(Nx, Ny, Nz); kabs, ksca, g = S.delta_scale(kabs, ksca, g)
alb = np.full((Ny, Nx), 0.1)
b = torch.tensor(S.solar_source("3_10", kabs, ksca, g, 50.0, 100.0, alb), device=dev)
t = lambda a: torch.tensor(a, dtype=torch.float64, device=dev)
s = DiffuseSolver("3_10", Nz, Nx, Ny)
s.set_lut_diffuse(LUT.synthetic_diffuse_table("3_10"), LUT.diffuse_axes("3_10"))
s.set_optprop(t(kabs), t(ksca), t(g), torch.full((Ny, Nx, Nz), 50.0, dtype=torch.float64, device=dev), 100.0,
              torch.zeros(Nz, dtype=torch.uint8, device=dev), t(np.zeros_like(kabs)), t(np.zeros_like(kabs)), t(alb))
x = torch.zeros_like(b)
try:
    s.solve(b, x, maxit=3)
except Exception as e:
    pass
print(f"{os.path.basename(os.path.dirname(os.environ['TSX_LIB'])):8s} pass {1e3 * s.bench_kernel(3, 200):6.2f} us")
PY
done

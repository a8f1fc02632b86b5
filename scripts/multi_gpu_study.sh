# For the first run on a multi-GPU node: scaling of the headline solve and the knobs of the face exchanges.
#   usage: bash scripts/multi_gpu_study.sh [max gpus, default 8]
# Prints one line per run: ranks, mode, knobs, cells/s, ms per solve, iterations.
MAXN=${1:-8}
run() {  # label, then environment assignments, then -- and bench.py arguments
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" > /tmp/mg.json 2>/tmp/mg.err
  python - "$label" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("/tmp/mg.json") if l.startswith("{")][-1]); c = d["config"]
    print(f"{sys.argv[1]:58s} {d['value']/1e6:8.1f} M cells/s {d['ms_per_step']:8.2f} ms  its {c['iterations']}  grid {c['process_grid']}  {c['transport']}")
except Exception as e:
    print(sys.argv[1], "failed:", e, open("/tmp/mg.err").read()[-300:])
PY
}
for n in 1 2 4 8; do
  [ $n -gt $MAXN ] && break
  run "weak   N=$n (256x256 columns per GPU)" A=1 -- --gpus $n
  run "strong N=$n (256x256 columns in total)" A=1 -- --gpus $n --scaling strong
  if [ $n -gt 1 ]; then
    run "weak   N=$n, preconditioner halo not overlapped" TSX_PC_OVERLAP=0 -- --gpus $n
    run "weak   N=$n, no overlap at all (TSX_OVERLAP=0)" TSX_OVERLAP=0 -- --gpus $n
    run "weak   N=$n, preconditioner halo every 2nd pass" TSX_PC_HALO_EVERY=2 -- --gpus $n
    run "weak   N=$n, no preconditioner halo (block-Jacobi over ranks)" TSX_PC_HALO=0 -- --gpus $n
    run "strong N=$n, preconditioner halo every 2nd pass" TSX_PC_HALO_EVERY=2 -- --gpus $n --scaling strong
  fi
done
if [ $MAXN -ge 8 ]; then run "config 3: 512x512x64 on 2x4" A=1 -- --gpus 8 --global-nx 512 --global-ny 512; fi

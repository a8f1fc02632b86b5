"""Writes profiles/<round>/README.md from the committed line / traffic files of that round.  usage: python scripts/profiles_readme.py r04"""
import json
import os
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda f: os.path.join(root, "profiles", rnd, f)
L = {t: json.load(open(P(f"{t}_line.json"))) for t in ("bench", "heterogeneous", "config2", "config5")}
T = {t: json.load(open(P(f"{t}_traffic.json"))) for t in L}
b = L["bench"]; c = b["config"]
ns, fp, het, cb = c["no_sharing"], c["all_fp64"], c["heterogeneous"], b["cpu_baseline"]
s4, s1 = json.load(open(P("config4_specint_252gpoints.json"))), json.load(open(P("config4_specint_252gpoints_one_instance.json")))
cb4 = s4.get('cpu_baseline') or s1.get('cpu_baseline')
c3p, c3h = json.load(open(P("config3_8ranks_one_device_peer_line.json"))), json.load(open(P("config3_8ranks_one_device_host_line.json")))


def pass_traffic(t):
    ks = [v for k, v in T[t]["kernels"].items() if "pcs_flow" in k]   # round 5: the dominant launch where it runs
    if L[t]["roofline"]["kernel"].startswith("tsx_k_pcs_flow") and ks:
        return ks[0]["traffic_bytes_per_launch"]
    ks = [v for k, v in T[t]["kernels"].items() if ("pcs_rb" in k or "pcsh_rb" in k) and ",true,0,true,2" in k and v["launches"] > 100]
    return ks[0]["traffic_bytes_per_launch"] if ks else None


def row(t):
    d = L[t]; r = d["roofline"]; cc = d["config"]; tr = pass_traffic(t) or r["traffic"]
    return (f"{d['value'] / 1e6:.1f} M cells/s, {d['ms_per_step']:.2f} ms, {cc['iterations']} iterations of {cc['iter_ms']:.3f} ms ({cc['preconditioner']}); "
            f"`roofline`: {r['ms_per_launch'] * 1e3:.1f} µs for {r['bytes_per_launch'] / 1e6:.1f} MB algorithmic = {r['frac']:.3f} of 8 TB/s "
            f"({r['frac_of_achievable']:.2f} of 6.3 TB/s), PMC traffic {tr / 1e6:.1f} MB = {tr / (r['ms_per_launch'] * 1e-3) / 1e12:.2f} TB/s; one iteration "
            f"{T[t]['iteration']['traffic_bytes'] / 1e9:.2f} GB of traffic")


import re
sh = {}
for line in open(P("shard_study.txt")):
    m = re.match(r"(\d+x\d+)x\d+ (\w+)\s+wall\s+([\d.]+) ms", line)
    if m:
        sh[(m.group(1), m.group(2))] = float(m.group(3))
shard = "; ".join(f"{k.replace('x', ' × ')} columns wrap {sh[(k, 'wrap')]:.2f} ms, peer {sh[(k, 'peer')]:.2f} ms" for k in ("128x64", "128x128", "256x256") if (k, "wrap") in sh)
txt = f"""# profiles/{rnd} — round-4 rocprofv3 summaries (one MI355X, gfx950)

Collected by `ROUND={rnd} bash scripts/profile_round.sh <tag> [bench.py arguments]` on the GPU box (`cd /tmp && export TMPDIR=/tmp`, the program
directly after `--`, the two `--pmc` passes on their own), final code of the round (28 passes per `M⁻¹`); condensed by `scripts/kt_summary.py`
(durations of *working* launches: the kernels of an iteration enqueued beyond convergence return at once and rocprofv3's listed average
mixes them in) and `scripts/pmc_summary.py` (`2·FETCH_SIZE·1024 + WRITE_SIZE·1024` per working launch, the gfx950 correction of
`MI355X_MICROARCH.md`). Per tag: `<tag>_line.json` (`python3 bench.py --steps 20 --warmup 5 [args]`, with the CPU baseline and the
reported-only legs for `bench`), `<tag>_kernel_stats.csv` + `_kernel_stats_line.json` (`rocprofv3 --kernel-trace --stats --output-format
csv -- python3 bench.py --no-cpu-baseline --skip-no-sharing --skip-extra-legs --steps 5 --warmup 1 [args]`), `<tag>_working_launches.json`,
`<tag>_traffic.json` (the two PMC passes of `… --steps 2 --warmup 1 --kernel-reps 4 [args]`). `bench.py` looks `roofline.traffic` up in
`traffic_<solver>_<local size>[_<field>].json` (copies of the per-tag files). This file: `python scripts/profiles_readme.py {rnd}`.

| tag | arguments | what it shows |
|---|---|---|
| `bench` | — (the driver's command) | {row('bench')}. Reported-only legs: `config.no_sharing` {ns['cells_per_s'] / 1e6:.1f} M ({ns['iterations']} its, pass {ns['pass_ms'] * 1e3:.1f} µs = {ns['pass_frac']:.2f}); `config.all_fp64`: fp64 recurrence {fp['recurrence_fp64']['cells_per_s'] / 1e6:.1f} M ({fp['recurrence_fp64']['iterations']} its), no reduced precision anywhere {fp['everything_fp64']['cells_per_s'] / 1e6:.1f} M ({fp['everything_fp64']['iterations']} its, {fp['everything_fp64']['preconditioner']}); `config.heterogeneous` {het['cells_per_s'] / 1e6:.1f} M ({het['iterations']} its, pass {het['pass_ms'] * 1e3:.1f} µs = {het['pass_frac']:.2f}, operator {het['spmv_ms']:.3f} ms = {het['spmv_frac']:.2f}); tight run {c['tight_run']['iterations']} its {c['tight_run']['solve_ms']:.1f} ms, warm start {c['warm_start']['iterations']} it {c['warm_start']['solve_ms']:.2f} ms. `tsx_probe_bandwidth`: copy {c['copy_GBps_measured'] / 1e3:.2f} TB/s, read {c['read_GBps_measured'] / 1e3:.2f} TB/s. `cpu_baseline`: B1 {cb['value'] / 1e6:.2f} M cells/s on {cb['cores']} threads, `parity_check.max_rel_err` {cb['parity_check']['max_rel_err']:.1e} on the full 256×256×64 system |
| `heterogeneous` | `--field heterogeneous` | {row('heterogeneous')} (per-block records entry-major for the near-identical grouping) |
| `config2` | `--nx 128 --ny 128` | {row('config2')} |
| `config5` | `--solver 8_16` | {row('config5')} |
| `config4_specint_252gpoints.json`, `…_one_instance.json` | `python3 bench_specint.py`, `--streams 1` | 252 g-points warm: **{s4['value']:.1f} g-points/s** with four instances in flight, **{s1['value']:.1f}** with one (cold call {s4['config']['calls'][0]['gpoints_per_s']:.1f} / {s1['config']['calls'][0]['gpoints_per_s']:.1f}; box to box 124–126 / 135–142); LW half with `planck_srfc`; `cpu_baseline`: the oracle's port of the reference's CPU path on ONE g-point's diffuse system, {cb4['value']:.2f} g-points/s on {cb4['cores']} threads (solve only) |
| `config3_8ranks_one_device_{{peer,host}}_line.json` | `python3 bench.py --gpus 8 --global-nx 512 --global-ny 512 --transport peer \\| host` | configs[2] at its own size with the 8 rank processes sharing the box's one GPU (first half of the round, 22 passes): {c3p['ms_per_step'] / 1e3:.1f} s / {c3h['ms_per_step'] / 1e3:.1f} s per step, {c3p['config']['iterations']} iterations on both (8 processes time-slice the device; the peer kernels spin on mailboxes of ranks that are not scheduled): functional evidence for `tests/test_gpu_config3.py`, not a rate |

`sweeps_grid.txt` (`python scripts/sweeps_grid.py`, first half of the round): iterations, final residual and time per solve for 20 … 32 passes per `M⁻¹`
on ten workloads — why 28 is the default. `shard_study.txt` (`python scripts/shard_study.py xm ym`, modes in `profiles/r03/README.md`): {shard}
(round 3: 3.53 / 5.30 ms at 128 × 64). `shard_128x64_wrap_kernel_stats.csv`: kernel trace of the periodic 128 × 64 × 64 shard at 22 passes, before the
pass kernels were rewritten (intermediate pass 10.9 µs, 59 % of the solve). `pass_pmc_3_10.txt` (`bash scripts/pass_pmc.sh`): SQ / TA / TCP / TCC counters
of the intermediate 3_10 pass on the metric domain, one `--pmc` run per set, whole-device sums per launch: 944 vector instructions per wave, `VALUBusy`
31 %, the texture addresser busy 60–68 % of the kernel (`TA_BUSY_avr` / `SQ_BUSY_CYCLES` per unit), waves waiting on memory 55 % of their cycles — what
DESIGN §4 "what bounds a pass" quotes.
"""
open(P("README.md"), "w").write(txt)
print(txt)

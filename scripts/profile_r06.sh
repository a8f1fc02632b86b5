# usage (GPU box): bash scripts/profile_r06.sh  -- the round-6 profile set: bench line + rocprofv3 kernel trace + PMC traffic per tag (profile_round.sh),
# config 4 with one and four instances, the shard study + scale projection, a marker trace of a small spectral loop (roctx ranges per phase)
export ROUND=r06
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
bash $R/scripts/profile_round.sh bench
bash $R/scripts/profile_round.sh config5 --solver 8_16
bash $R/scripts/profile_round.sh config2 --nx 128 --ny 128
bash $R/scripts/profile_round.sh heterogeneous --field heterogeneous
cd $R
python3 bench_specint.py 2>/dev/null | tail -1 > $O/config4_specint_252gpoints.json
python3 bench_specint.py --streams 1 2>/dev/null | tail -1 > $O/config4_specint_252gpoints_one_instance.json
for sz in "128 64" "128 128" "256 128" "256 256"; do python3 scripts/shard_study.py $sz 64 2>&1 | grep -v amdgpu.ids; done > $O/shard_study.txt
python3 scripts/scale_projection.py $O/scale_projection.json > $O/scale_projection.txt 2>&1
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/mk
TSX_LOG=1 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d /tmp/mk -- python3 $R/bench_specint.py --sw 4 --lw 4 --streams 1 --nx 128 --ny 128 --no-cpu-baseline > /tmp/mk_line.json 2>/tmp/mk_err.log
ls /tmp/mk/*/ > $O/marker_trace_files.txt 2>&1
f=$(ls /tmp/mk/*/*marker_api_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && { head -1 $f > $O/marker_trace_head.csv; grep -m 60 -E "set_optprop|solve_Mdiff|compute_Edir|get_result|compute_Ediff|setup_Mdiff" $f >> $O/marker_trace_head.csv; }
tail -3 /tmp/mk_err.log

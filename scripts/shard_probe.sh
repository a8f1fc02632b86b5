#!/bin/bash
# where a small shard's time goes: kernel trace of the periodic 128 x 64 x 64 shard, then a sweep over the pass count
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/shard
SHARD_MODES=wrap rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/shard/kt -- python3 $R/scripts/shard_study.py 128 64 > $R/gpurun_out/shard/kt.log 2>&1
f=$(find $R/gpurun_out/shard/kt -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/shard/kernel_stats_128x64_wrap.csv
find $R/gpurun_out/shard/kt -name "*.csv" ! -name "*kernel_stats.csv" -delete
cd $R
for sw in 9 13 17 21 25; do echo "PC_SWEEPS=$sw"; SHARD_MODES=wrap,peer PC_SWEEPS=$sw python scripts/shard_study.py 128 64; done

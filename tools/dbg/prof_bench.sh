#!/bin/bash
# rocprofv3 kernel stats of bench.py, normalised per bench step; summary under gpurun_out/prof_$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/prof_$1
mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D -o out --output-format csv -- python3 $R/bench.py --steps 8 --warmup 2 --no-precision-check > $D/bench.json 2>/dev/null
python3 $R/tools/dbg/prof_summary.py $D/out_kernel_stats.csv

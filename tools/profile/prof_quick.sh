TAG=$1
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles
mkdir -p $O
D=$R/gpurun_out/prof_$TAG
mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D -o out --output-format csv -- python3 $R/bench.py --steps 8 --warmup 2 --no-precision-check --no-cpu-baseline --shapes $O/${TAG}_conv_shapes_one_step.txt > $D/bench.json 2>/dev/null
cp $D/out_kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv
python3 $R/tools/dbg/prof_summary.py $D/out_kernel_stats.csv > $O/${TAG}_per_step_summary.txt
head -45 $O/${TAG}_per_step_summary.txt

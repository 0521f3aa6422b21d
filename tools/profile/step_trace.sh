cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$R/gpurun_out/prof_trace
rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace -d $D -o out --output-format csv -- python3 $R/bench.py --steps 12 --warmup 3 --no-precision-check --no-cpu-baseline --phase-iters 2 > $D/bench.json 2>/dev/null
python3 $R/tools/dbg/step_trace.py $D/out_kernel_trace.csv full > $R/gpurun_out/step_trace.txt
head -60 $R/gpurun_out/step_trace.txt

#!/bin/bash
# Every measured artefact the docs / the bench line cite, from ONE build on ONE GPU box:
#   gpurun --timeout 2400 -- 'bash tools/profile/refresh_profiles.sh r06'
# then copy gpurun_out/profiles/* into profiles/ and commit.  (rocprofv3: kernel trace + stats in one pass; --pmc passes separately.)
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles
mkdir -p $O
SHA=$(cd $R && python3 -c "from aod_meh_hua_amd.build import source_digest; print(source_digest())")
echo "kernels_sha16 $SHA" > $O/${TAG}_build.txt
# 1. the default bench line (what the driver runs) + per-shape conv listing of its instrumented step
python3 $R/bench.py --shapes $O/${TAG}_conv_shapes_one_step.txt > $O/${TAG}_bench_default.json 2>/dev/null
# 2. rocprofv3 kernel trace + stats of the same command, normalised per step
D=$R/gpurun_out/prof_$TAG
mkdir -p $D
rocprofv3 --kernel-trace --stats -d $D -o out --output-format csv -- python3 $R/bench.py --steps 8 --warmup 2 --no-precision-check > $D/bench.json 2>/dev/null
cp $D/out_kernel_stats.csv $O/${TAG}_bench_kernel_stats.csv
tail -1 $D/bench.json > $O/${TAG}_bench_line_under_rocprof.json
python3 $R/tools/dbg/prof_summary.py $D/out_kernel_stats.csv > $O/${TAG}_per_step_summary.txt
# 3. secondary configurations: configs[3] (10k on-device pool), configs[4] (R101 / 80 classes / 800x1344), two ranks on this one GPU (gloo)
python3 $R/bench.py --mode pool --pool 10000 --no-cpu-baseline --no-precision-check > $O/${TAG}_bench_pool10k.json 2>/dev/null
python3 $R/bench.py --precision bf16 --mode pool --pool 10000 --no-cpu-baseline --no-precision-check > $O/${TAG}_bench_pool10k_bf16.json 2>/dev/null
python3 $R/bench.py --precision bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-precision-check --shapes $O/${TAG}_conv_shapes_one_step_bf16.txt > $O/${TAG}_bench_bf16.json 2>/dev/null
python3 $R/bench.py --config r101coco --steps 6 --warmup 2 --no-cpu-baseline --no-precision-check > $O/${TAG}_bench_r101coco.json 2>/dev/null
AOD_BENCH_ONE_GPU=1 python3 $R/bench.py --gpus 2 --steps 6 --warmup 2 --batch 8 --no-cpu-baseline --no-precision-check > $O/${TAG}_bench_2ranks_one_gpu.json 2>/dev/null
AOD_BENCH_ONE_GPU=1 python3 $R/bench.py --gpus 2 --mode pool --pool 2000 --no-cpu-baseline --no-precision-check > $O/${TAG}_bench_2ranks_one_gpu_pool.json 2>/dev/null
# 3b. the persistent x3 conv kernel against the general one, launch by launch (GPU-bound, interleaved); same-box step-level A/B
python3 $R/tools/dbg/x3p_micro.py > /dev/null 2>&1; cp $R/gpurun_out/x3p_micro.txt $O/${TAG}_x3p_micro.txt
MODES=0,1,1r768 python3 $R/tools/dbg/x3p_grouped_micro.py > $O/${TAG}_x3p_grouped_micro.txt 2>/dev/null
for m in 0 1 0 1; do AOD_X3P=$m python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-precision-check 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('AOD_X3P=$m', d['value'], 'img/s', d['ms_per_step'], 'ms/step  train', d['phase_rates']['train_ms_per_batch'], 'score', d['phase_rates']['score_ms_per_batch'], ' backbone_fpn ms', r['backbone_fpn']['total']['ms'], 'frac', r['backbone_fpn']['total']['frac'], ' heads ms', r['heads']['total']['ms'])" >> $O/${TAG}_x3p_step_ab.txt; done
# 4. PMC passes (tagged with the build digest; bench.py merges them only for this build)
bash $R/tools/profile/pmc_passes.sh > $O/${TAG}_pmc_passes.txt 2>&1
ls -la $O

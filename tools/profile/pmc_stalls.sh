#!/bin/bash
# What do the conv kernels' waves wait on?  Separate rocprofv3 --pmc passes (kernel trace only) of three eager bench steps; per kernel the averages of
# the SQ wait / busy counters.   gpurun -- 'bash tools/profile/pmc_stalls.sh'  ->  gpurun_out/profiles/r06_pmc_stalls.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
D=$R/gpurun_out/pmc_stalls
rm -rf $D; mkdir -p $D $R/gpurun_out/profiles
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $D/p$i -o out --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-precision-check --no-graph --phase-iters 2 > $D/bench$i.json 2> $D/err$i.txt
  tail -2 $D/err$i.txt | cut -c1-200
done
python3 - <<PY > $R/gpurun_out/profiles/r06_pmc_stalls.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('$D/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:64]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
names = sorted({c for d in agg.values() for c in d})
print('per-launch averages of SQ counters (rocprofv3 --pmc, eager bench steps); ratios to SQ_WAVE_CYCLES where that makes sense')
keep = [k for k in agg if any(x in k for x in ('conv_igemm', 'conv_wgrad', 'bottleneck', 'halo_x3', 'stem_pool', 'conv_x3p'))]
for k in sorted(keep, key=lambda k: -agg[k].get('SQ_WAVE_CYCLES', 0)):
    d = {c: agg[k][c] / cnt[(k, c)] for c in agg[k]}
    wc = d.get('SQ_WAVE_CYCLES', 0) or 1
    print()
    print(k, ' launches', cnt[(k, 'SQ_WAVE_CYCLES')] // max(1, sum(1 for _ in glob.glob('$D/p*'))))
    for c in names:
        if c in d:
            print('   %-34s %16.0f   %6.3f of wave cycles' % (c, d[c], d[c] / wc))
PY
head -60 $R/gpurun_out/profiles/r06_pmc_stalls.txt

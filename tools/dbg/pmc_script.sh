#!/bin/bash
# PMC passes over one python script: tools/dbg/pmc_script.sh <script.py> <kernel-name substring> <tag>; per-launch means are printed
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=$1; F=$2; T=$3
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU"; do
  d=$R/gpurun_out/pmc_$T/$(echo $grp | tr ' ' '_' | cut -c1-40)
  mkdir -p $d
  rocprofv3 --pmc $grp --kernel-trace -d $d -o out --output-format csv -- python3 $R/$S > $d/log.txt 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('$R/gpurun_out/pmc_$T/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70] + ' grid=' + r.get('Grid_Size', '?')
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    if '$F' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:36s} {v / cnt[(k, c)]:16.0f}  (per launch, n={cnt[(k, c)]})')
PY

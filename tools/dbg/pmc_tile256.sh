#!/bin/bash
# PMC passes over tools/dbg/tile256.py (256 x 256 conv tile on full-round shapes); per-launch means under gpurun_out/pmc_tile256/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES"; do
  d=$R/gpurun_out/pmc_tile256/$(echo $grp | tr ' ' '_' | cut -c1-40)
  mkdir -p $d
  rocprofv3 --pmc $grp --kernel-trace -d $d -o out --output-format csv -- python3 $R/tools/dbg/tile256.py > $d/log.txt 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('$R/gpurun_out/pmc_tile256/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60] + ' grid=' + r.get('Grid_Size', '?')
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    if 'conv_igemm' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:36s} {v / cnt[(k, c)]:16.0f}  (per launch, n={cnt[(k, c)]})')
PY

#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_BUFFER_LOAD_WAVEFRONTS_sum TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  d=$R/gpurun_out/pmcl2_$tag/$(echo $grp | tr ' ' '_' | cut -c1-40)
  mkdir -p $d
  rocprofv3 --pmc $grp --kernel-trace -d $d -o out --output-format csv -- python3 $R/tools/dbg/conv_micro.py "$@" > $d/log.txt 2>&1
  tail -2 $d/log.txt
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('$R/gpurun_out/pmcl2_$tag/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, d in agg.items():
    if 'conv' not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:36s} {v / cnt[(k, c)]:16.0f}  (per launch, n={cnt[(k, c)]})')
PY

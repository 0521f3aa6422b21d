#!/bin/bash
# rocprofv3 kernel stats of the secondary bench configurations; outputs under gpurun_out/prof_cfg_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "pool:--mode pool --pool 2000" "r101coco:--config r101coco --steps 6 --warmup 2"; do
  name=${cfg%%:*}; args=${cfg#*:}
  D=$R/gpurun_out/prof_cfg_$name
  mkdir -p $D
  rocprofv3 --kernel-trace --stats -d $D -o out --output-format csv -- python3 $R/bench.py $args --no-precision-check --no-cpu-baseline > $D/bench.json 2>/dev/null
  echo "== $name"; tail -1 $D/bench.json | cut -c1-300
  python3 - <<PY
import csv
rows=list(csv.DictReader(open('$D/out_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print('%-70s calls %6s  %8.2f ms  %5.1f%%  avg %7.1f us'%(r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e6, 100*float(r['TotalDurationNs'])/tot, float(r['AverageNs'])/1e3))
PY
done

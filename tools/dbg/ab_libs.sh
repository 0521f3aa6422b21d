#!/bin/bash
# same-box A/B of two builds of the library:  tools/dbg/ab_libs.sh old.so new.so [pairs] [steps]   (extra environment applies to both)
for i in $(seq 1 ${3:-3}); do for v in "$1" "$2"; do
  AOD_HIP_LIB=$v timeout 600 python bench.py --steps ${4:-40} --warmup 5 --no-cpu-baseline --no-precision-check 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print('$v', d['value'], 'img/s', d['ms_per_step'], 'ms/step  train', d['phase_rates']['train_ms_per_batch'], 'score', d['phase_rates']['score_ms_per_batch'], ' backbone_fpn ms', r['backbone_fpn']['total']['ms'], 'frac', r['backbone_fpn']['total']['frac'])"
done; done

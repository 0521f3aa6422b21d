#!/bin/bash
# same-box A/B of one environment switch:  tools/dbg/ab_env.sh VAR valueA valueB
for v in "$2" "$3" "$2" "$3"; do
  env $1=$v python bench.py --steps 8 --warmup 2 --no-cpu-baseline --shapes gpurun_out/shapes_$1_$v.txt > gpurun_out/b_$v.log 2>&1
  echo "$1=$v $(tail -1 gpurun_out/b_$v.log | cut -c60-140)"
done

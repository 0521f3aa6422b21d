#!/bin/bash
# same-box A/B of the 1x1 tile override
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -2
for t in none 64x64 64x128 128x64; do
  AOD_TILE_1X1=$t python bench.py --steps 8 --warmup 2 --no-cpu-baseline --shapes gpurun_out/shapes_$t.txt > gpurun_out/b_$t.log 2>&1
  echo "$t $(tail -1 gpurun_out/b_$t.log | cut -c60-140)"
done

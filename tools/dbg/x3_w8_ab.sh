for shape in "16 32 32 256 256 3" "16 64 64 128 128 3" "16 32 32 1024 256 1" "16 16 16 512 512 3" "16 64 64 512 128 1" "16 64 64 256 256 3"; do
  for w8 in 0 1; do AOD_X3_128_W8=$w8 python tools/dbg/x3_shape.py $shape 2>&1 | grep -v amdgpu | sed "s/^/W8=$w8 /"; done
done
AOD_X3_128_W8=1 python -m pytest tests/test_gpu_x3_kernels.py -m gpu -q -k "forward_dgrad_wgrad or grouped" -p no:cacheprovider 2>&1 | tail -3

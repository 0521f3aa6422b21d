for shape in "16 32 32 256 256 3" "16 64 64 128 128 3" "16 32 32 1024 256 1" "16 32 32 256 1024 1" "16 16 16 512 512 3" "16 64 64 512 128 1"; do
  for st3 in 0 1; do AOD_X3_128_ST3=$st3 python tools/dbg/x3_shape.py $shape 2>&1 | grep -v amdgpu | sed "s/^/ST3=$st3 /"; done
done

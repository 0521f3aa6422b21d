"""Per-tile phase timing of bottleneck64_fwd_kernel (debug build with -DAOD_TILE_TIMING: python tools/dbg/tile_timing.py build).
  run on GPU:   AOD_HIP_LIB=tools/dbg/_build/libaodhip_dbg.so python tools/dbg/bn_timing.py [Cin]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from aod_meh_hua_amd import hipops as ho
from aod_meh_hua_amd._C import lib
Cin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B, H, W = 16, 128, 128
M = B * H * W
x = torch.randn(M, Cin, device='cuda').relu().bfloat16()
res = x if Cin == 256 else torch.randn(M, 256, device='cuda').bfloat16()
w1 = (torch.randn(64, Cin, device='cuda') * 0.1).bfloat16(); w2 = (torch.randn(64, 576, device='cuda') * 0.05).bfloat16(); w3 = (torch.randn(256, 64, device='cuda') * 0.1).bfloat16()
v = lambda n: (torch.rand(n, device='cuda') + 0.5, torch.randn(n, device='cuda') * 0.1)
(s1, b1), (s2, b2), (s3, b3) = v(64), v(64), v(256)
out = torch.empty(M, 256, device='cuda', dtype=torch.bfloat16)
f = lambda: ho.bottleneck64_fwd(x, B, H, W, w1, s1, b1, w2, s2, b2, w3, s3, b3, res, out)
for _ in range(200): f()
nt = B * 8 * 8
st = torch.zeros(nt * 16, dtype=torch.int64, device='cuda')
lib.aod_dbg_set_bn_stamps.argtypes = [ctypes.c_void_p]
assert lib.aod_dbg_set_bn_stamps(st.data_ptr()) == 0
torch.cuda.synchronize(); f(); torch.cuda.synchronize()
t = st.cpu().numpy().reshape(nt, 16).astype(np.float64) * 0.01
print(f'{nt} tiles, kernel span {t[:, 7].max() - t[:, 0].min():.1f} us; per-tile mean phase durations (us):')
for k, name in enumerate(['setup + first loads issued', 'phase 1 K loop (conv1 on halo)', 'epilogue 1 + filter loads', 'phase 2 (conv2, 9 taps)', 'epilogue 2', 'phase 3 (conv3 + res + stores)', 'store drain']):
    d = t[:, k + 1] - t[:, k]
    print(f'  {name:34s} mean {d.mean():6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}')
print(f'  inside epilogue 1: loads issued after {(t[:, 8] - t[:, 2]).mean():.2f} us, t1 written after {(t[:, 9] - t[:, 2]).mean():.2f} us, barrier passed after {(t[:, 3] - t[:, 2]).mean():.2f} us')
tot = t[:, 7] - t[:, 0]
print(f'  tile total {tot.mean():.2f} us; tile start times percentiles 0/25/50/75/100: {[round(float(np.percentile(t[:, 0] - t[:, 0].min(), q)), 1) for q in (0, 25, 50, 75, 100)]}')

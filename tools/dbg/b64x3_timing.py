"""Per-tile phase timing of the fused 64-plane x3 block (frozen layer 1; debug build with -DAOD_TILE_TIMING: python tools/dbg/tile_timing.py build).
  run on GPU:   AOD_HIP_LIB=tools/dbg/_build/libaodhip_dbg.so python tools/dbg/b64x3_timing.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np      # noqa: E402
import torch            # noqa: E402
from aod_meh_hua_amd import functional as AF      # noqa: E402
from aod_meh_hua_amd import hipops as ho          # noqa: E402
from aod_meh_hua_amd._C import lib                # noqa: E402

AF.set_precision('bf16x3')
B, H, W = 16, 128, 128
M = B * H * W
g = torch.Generator(device='cuda').manual_seed(1)
rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)
X = lambda t: ho.x3_split(t)
x = X(rnd(M, 256).relu())
w1, w2, w3 = X(rnd(64, 256) * 0.05), X(rnd(64 * 9, 64) * 0.03).view(64, 9 * 128), X(rnd(256, 64) * 0.05)
v = lambda n: (torch.rand(n, device='cuda', generator=g) + 0.5, rnd(n) * 0.1)
(s1, b1), (s2, b2), (s3, b3) = v(64), v(64), v(256)
out = torch.empty(M, 512, dtype=torch.bfloat16, device='cuda')
f = lambda: ho.bottleneck64_fwd(x, B, H, W, w1, s1, b1, w2, s2, b2, w3, s3, b3, x, out=out)
for _ in range(30):
    f()
nt = B * ((H + 7) // 8) * ((W + 15) // 16)
st = torch.zeros(nt * 8, dtype=torch.int64, device='cuda')
lib.aod_dbg_set_b64_stamps.argtypes = [ctypes.c_void_p]
assert lib.aod_dbg_set_b64_stamps(st.data_ptr()) == 0
torch.cuda.synchronize(); f(); torch.cuda.synchronize()
t = st.cpu().numpy().reshape(nt, 8).astype(np.float64) * 0.01
print(f'64x3 identity block, {nt} tiles, kernel span {t[:, 7].max() - t[:, 0].min():.1f} us; per-tile phase durations (us):')
names = ['phase 1 K loop (conv1 on halo, 8 K-steps)', 'epilogue 1 (t1 -> LDS)', 'phase 2 (conv2, 18 steps)', 'epilogue 2 + residual loads + barrier',
         'phase 3 half 0 (conv3 + res + stores)', 'phase 3 half 1', 'store drain']
for k, name in enumerate(names):
    d = t[:, k + 1] - t[:, k]
    print(f'  {name:44s} mean {d.mean():6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}')
tot = t[:, 7] - t[:, 0]
print(f'  tile total {tot.mean():.2f} us ({nt / 256:.0f} tiles per CU); tile start percentiles 0/25/50/75/100: '
      f'{[round(float(np.percentile(t[:, 0] - t[:, 0].min(), q)), 1) for q in (0, 25, 50, 75, 100)]}')

"""Per-tile phase timing of the fused 128-plane x3 block (debug build with -DAOD_TILE_TIMING: python tools/dbg/tile_timing.py build).
  run on GPU:   AOD_HIP_LIB=tools/dbg/_build/libaodhip_dbg.so python tools/dbg/b128x3_timing.py [fwd|bwd]"""
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
mode = sys.argv[1] if len(sys.argv) > 1 else 'fwd'
B, H, W = 16, 64, 64
M = B * H * W
g = torch.Generator(device='cuda').manual_seed(1)
rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)
X = lambda t: ho.x3_split(t)
x = X(rnd(M, 512).relu())
w1, w2, w3 = X(rnd(128, 512) * 0.05), X(rnd(128 * 9, 128) * 0.03).view(128, 9 * 256), X(rnd(512, 128) * 0.05)
v = lambda n: (torch.rand(n, device='cuda', generator=g) + 0.5, rnd(n) * 0.1)
(s1, b1), (s2, b2), (s3, b3) = v(128), v(128), v(512)
if mode == 'fwd':
    f = lambda: ho.bottleneck128_fwd(x, B, H, W, w1, s1, b1, w2, s2, b2, w3, s3, b3, keep=True)
else:
    t1, t2 = X(rnd(M, 128).relu()), X(rnd(M, 128).relu())
    gy = X(rnd(M, 512) * 0.1)
    f = lambda: ho.bottleneck_bwd(gy, B, H, W, w1, w2, w3, t2, t1, x)
for _ in range(50):
    f()
nt = B * ((H + 7) // 8) * ((W + 15) // 16)
st = torch.zeros(nt * 16, dtype=torch.int64, device='cuda')
lib.aod_dbg_set_bw3_stamps.argtypes = [ctypes.c_void_p]
assert lib.aod_dbg_set_bw3_stamps(st.data_ptr()) == 0
torch.cuda.synchronize(); f(); torch.cuda.synchronize()
raw = st.cpu().numpy().reshape(nt, 16).astype(np.float64)
t = raw * 0.01
print(f'128x3 {mode}: {nt} tiles, kernel span {t[:, 6].max() - t[:, 0].min():.1f} us; per-tile phase durations (us):')
for k, name in enumerate(['phase 1 K loop (conv1 on halo)', 'epilogue 1', 'phase 2 (conv2)', 'epilogue 2', 'phase 3 (conv3 + res + stores)', 'store drain']):
    d = t[:, k + 1] - t[:, k]
    print(f'  {name:34s} mean {d.mean():6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}')
tot = t[:, 6] - t[:, 0]
print(f'  tile total {tot.mean():.2f} us; tile start percentiles 0/25/50/75/100: {[round(float(np.percentile(t[:, 0] - t[:, 0].min(), q)), 1) for q in (0, 25, 50, 75, 100)]}')
for w, o in ((0, 8), (5, 12)):
    c = raw[:, o:o + 4]
    print(f'  phase 2, wave {w}: shader cycles per step: wait for the slice {c[:, 0].mean() / 36:7.1f}  own LDS reads {c[:, 1].mean() / 36:7.1f}  barrier {c[:, 2].mean() / 36:7.1f}  '
          f'issue + reads + MFMA {c[:, 3].mean() / 36:7.1f}  (sum {c.sum(1).mean() / 36:7.1f}; 24 MFMAs of 16 cycles per wave, two waves per SIMD)')

"""Where a tile of the persistent x3 conv kernel spends its time (debug build with -DAOD_TILE_TIMING: stamps of consumer wave 0 at kernel entry,
ring primed, K loop done, epilogue done -- first tile -- and kernel exit).
  build here:  python tools/dbg/tile_timing.py build
  run on GPU:  AOD_HIP_LIB=tools/dbg/_build/libaodhip_dbg.so python tools/dbg/x3p_timing.py [C O H W R]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from aod_meh_hua_amd import functional as AF  # noqa: E402
from aod_meh_hua_amd import hipops as ho  # noqa: E402
from aod_meh_hua_amd._C import lib  # noqa: E402

AF.set_precision('bf16x3')
os.environ['AOD_X3P'] = '1'; os.environ['AOD_X3P_MIN_TILES'] = '1'; os.environ['AOD_X3P_MIN_STEPS'] = '1'
C, O, H, W, R = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (256, 256, 32, 32, 3)
B = int(os.environ.get('B', '16'))
g = torch.Generator(device='cuda').manual_seed(1)
x = AF.as_nchw(ho.x3_split(torch.randn(B * H * W, C, device='cuda', generator=g)), B, H, W)
w = torch.randn(O, C, R, R, device='cuda', generator=g) / (C * R * R) ** 0.5
bias = torch.randn(O, device='cuda', generator=g)


RES = os.environ.get('RES', '0') == '1'            # RES=1: a residual operand (the expand 1x1 convs; PRE=0 switches the operand prefetch form off)
if 'PRE' in os.environ:
    os.environ['AOD_X3P_PRE'] = os.environ['PRE']
os.environ['AOD_X3P_PRE_MIN_STEPS'] = '1'
res = AF.as_nchw(ho.x3_split(torch.randn(B * H * W, O, device='cuda', generator=g)), B, H, W) if RES else None


def f():
    with torch.no_grad():
        return AF.conv_bn_act(x, w, bias=bias, res=res, stride=1, pad=R // 2, relu=True)


for _ in range(300):       # (clock readings need the steady state)
    f()
st = torch.zeros(256 * 16, dtype=torch.int64, device='cuda')
lib.aod_dbg_set_x3p_stamps.argtypes = [ctypes.c_void_p]
assert lib.aod_dbg_set_x3p_stamps(st.data_ptr()) == 0
torch.cuda.synchronize()
f()
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(256, 16)
s = s[s[:, 0] > 0]
t = s[:, :5].astype(np.float64) * 0.01          # us (100 MHz wall clock)
c = s[:, 8:13].astype(np.float64)               # shader cycles
t0 = t[:, 0].min()
nk = (9 if R == 3 else 1) * (2 * C // 64)
print(f'{len(s)} workgroups; K-steps per tile {nk}; kernel span (first entry -> last exit) {t[:, 4].max() - t0:.1f} us; entry skew {t[:, 0].max() - t0:.2f} us')
for k, name in enumerate(['ring primed (decode + first stage landed)', 'K loop of the first tile', 'epilogue of the first tile', 'rest (further tiles)']):
    d = t[:, k + 1] - t[:, k]
    dc = c[:, k + 1] - c[:, k]
    print(f'  {name:44s} mean {d.mean():7.2f} us  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f}   {dc.mean():9.0f} cycles  clock {np.median(dc / np.maximum(d, 1e-3)) / 1e3:.2f} GHz')
if (s[:, 5] > 0).any():
    # second tile of the workgroups that had one: K loop = epilogue of tile 0 done -> K loop of tile 1 done, then its epilogue
    m = s[:, 5] > 0
    t2 = s[m][:, [3, 5, 6]].astype(np.float64) * 0.01
    print(f'  second tile ({int(m.sum())} workgroups): K loop {np.mean(t2[:, 1] - t2[:, 0]):7.2f} us   epilogue {np.mean(t2[:, 2] - t2[:, 1]):7.2f} us')
kl = c[:, 2] - c[:, 1]
print(f'  cycles per K-step in the loop: mean {kl.mean() / nk:.0f} (48 MFMAs = 768 matrix-pipe cycles per consumer wave)')

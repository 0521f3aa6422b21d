"""Per-tile phase timing of conv_igemm_kernel (debug build with -DAOD_TILE_TIMING).
  build here:   python tools/dbg/tile_timing.py build
  run on GPU:   AOD_HIP_LIB=tools/dbg/_build/libaodhip_dbg.so python tools/dbg/tile_timing.py run B H W C N RS [res]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, 'tools', 'dbg', '_build')
if sys.argv[1] == 'build':
    os.makedirs(OUT, exist_ok=True)
    csrc = os.path.join(ROOT, 'aod_meh_hua_amd', 'csrc')
    srcs = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(('.hip', '.cpp')))
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-DAOD_TILE_TIMING', *(['-DAOD_WGRAD_NO_EPI'] if 'noepi' in sys.argv else []), *(['-DAOD_WGRAD_NO_TLOAD'] if 'notload' in sys.argv else []), '-shared', '-x', 'hip',
                           '-o', os.path.join(OUT, 'libaodhip_noepi.so' if 'noepi' in sys.argv else ('libaodhip_notload.so' if 'notload' in sys.argv else 'libaodhip_dbg.so'))] + srcs)
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np, torch
from aod_meh_hua_amd import hipops as ho
from aod_meh_hua_amd._C import lib
B, H, W, Cc, N, RS = map(int, sys.argv[2:8])
use_res = len(sys.argv) > 8
M = B * H * W
segs = [ho.Seg(B, H, W, 0)]
R = 3 if RS == 9 else 1
x = torch.randn(M, Cc, device='cuda').bfloat16()
w = torch.randn(N, Cc, R, R, device='cuda') * 0.05
wp = ho.pack_weight_fwd(w)
res = torch.randn(M, N, device='cuda').bfloat16() if use_res else None
bias = torch.randn(N, device='cuda')
out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
f = lambda: ho.conv2d_rows(x, segs, wp, N, R, R, 1, R // 2, 1, relu=True, pre_shift=bias, res=res, out=out)
for _ in range(int(os.environ.get('AOD_TT_WARM', '3'))): f()       # (thousands of launches: clock readings need the DVFS steady state)
ntile = ((M + 127) // 128) * ((N + 127) // 128)
st = torch.zeros(ntile * 16, dtype=torch.int64, device='cuda')
lib.aod_dbg_set_tile_stamps.argtypes = [__import__('ctypes').c_void_p]
assert lib.aod_dbg_set_tile_stamps(st.data_ptr()) == 0
torch.cuda.synchronize()
f()
torch.cuda.synchronize()
s = st.cpu().numpy().reshape(ntile, 16)
t = s[:, :7].astype(np.float64) * 0.01      # us
t0 = t[:, 0].min()
span = t[:, 6].max() - t0
names = ['prologue (to stamp 1)', 'decode + first load', 'main loop', 'acc->LDS', 'epilogue stores issued', 'store drain']
print(f'{ntile} tiles, kernel span {span:.1f} us; per-tile mean phase durations (us):')
for k, nme in enumerate(names):
    d = t[:, k + 1] - t[:, k]
    print(f'  {nme:24s} mean {d.mean():6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}')
it = s[:, 8:16].astype(np.float64) * 0.01
print('  prologue stamps relative to kernel entry (mean us): 8 (args + tile index), 10 (drow table), 1 (lambdas set up), 9 (rows decoded):', [round(float((x - t[:, 0]).mean()), 2) for x in (it[:, 0], it[:, 2], t[:, 1], it[:, 1])])
dclk = (s[:, 13] - s[:, 12]).astype(np.float64)
dwall = (t[:, 3] - t[:, 2])
ok = dwall > 1.0
if ok.any():
    print(f'  in-kernel shader clock over the K loop (median over workgroups): {np.median(dclk[ok] / dwall[ok]) / 1e3:.2f} GHz')
tot = t[:, 6] - t[:, 0]
print(f'  tile total               mean {tot.mean():6.2f}  p10 {np.percentile(tot, 10):6.2f}  p90 {np.percentile(tot, 90):6.2f}')
hw = s[:, 7]
cu = (hw & 0xffffffff) >> 8 & 0xf; se = (hw & 0xffffffff) >> 13 & 0x7; sh = (hw & 0xffffffff) >> 12 & 1; xcc = (hw >> 32) & 0xf
key = xcc * 1000 + se * 100 + sh * 50 + cu
ks = np.unique(key)
print(f'{len(ks)} distinct CUs seen; tiles per CU min {min((key == k).sum() for k in ks)} max {max((key == k).sum() for k in ks)}')
# per-CU occupancy: fraction of the span during which 0/1/2 tiles are resident
occ = []
gaps = []
for k in ks[:64]:
    idx = np.where(key == k)[0]
    ev = sorted([(t[i, 0], 1) for i in idx] + [(t[i, 6], -1) for i in idx])
    cur, last, acc = 0, t0, [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    for tm, dlt in ev:
        acc[min(cur, 5)] += tm - last; last = tm; cur += dlt
    occ.append([a / span for a in acc])
occ = np.array(occ).mean(0)
print('mean fraction of the kernel span with k resident workgroups on a CU: ' + ' '.join(f'{k}:{v:.2f}' for k, v in enumerate(occ)))
start = np.sort(t[:, 0] - t0)
print('tile start times (us) percentiles 0/25/50/75/100:', [round(float(np.percentile(start, q)), 1) for q in (0, 25, 50, 75, 100)])

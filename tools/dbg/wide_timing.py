"""Per-tile phase timing of the wide bottleneck kernels (debug build with -DAOD_TILE_TIMING: python tools/dbg/tile_timing.py build).
  run on GPU:   AOD_HIP_LIB=tools/dbg/_build/libaodhip_dbg.so python tools/dbg/wide_timing.py [128|256] [fwd|bwd|frag]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from aod_meh_hua_amd import hipops as ho
from aod_meh_hua_amd._C import lib
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = sys.argv[2] if len(sys.argv) > 2 else 'fwd'
B, H, W = (16, 64, 64) if P == 128 else (16, 32, 32)
M, C4 = B * H * W, 4 * P
g = torch.Generator(device='cuda').manual_seed(1)
rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)
x = rnd(M, C4).relu().bfloat16()
w1 = (rnd(P, C4) * 0.05).bfloat16(); w2 = (rnd(P, 9 * P) * 0.03).bfloat16(); w3 = (rnd(C4, P) * 0.05).bfloat16()
v = lambda n: (torch.rand(n, device='cuda', generator=g) + 0.5, rnd(n) * 0.1)
(s1, b1), (s2, b2), (s3, b3) = v(P), v(P), v(C4)
t1 = rnd(M, P).relu().bfloat16(); t2 = rnd(M, P).relu().bfloat16()
if mode == 'frag':       # the register-streamed 256-plane kernel on fragment-major images
    import ctypes as C
    from aod_meh_hua_amd._C import call, ptr, stream

    class Rec(C.Structure):
        _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('rows', C.c_int32), ('K', C.c_int32), ('blk0', C.c_int32), ('pad_', C.c_int32)]
    fr = []
    for wt in (w1, w2, w3):
        o = torch.empty(wt.numel(), dtype=torch.bfloat16, device='cuda')
        r = (Rec * 1)()
        r[0].src, r[0].dst, r[0].rows, r[0].K, r[0].blk0 = wt.data_ptr(), o.data_ptr(), wt.shape[0], wt.shape[1], 0
        tab = torch.frombuffer(bytearray(bytes(r)), dtype=torch.uint8).cuda()
        call('aod_frag_pack', ptr(tab), 1, (wt.numel() + 2047) // 2048, stream())
        fr.append(o)
    f = lambda: ho.bottleneck128_fwd(x, B, H, W, fr[0], s1, b1, fr[1], s2, b2, fr[2], s3, b3, keep=True, frag=True)
elif mode == 'fwd':
    f = lambda: ho.bottleneck128_fwd(x, B, H, W, w1, s1, b1, w2, s2, b2, w3, s3, b3, keep=True)
else:
    gy = (rnd(M, C4) * 0.1).bfloat16()
    f = lambda: ho.bottleneck_bwd(gy, B, H, W, w1, w2, w3, t2, t1, x)
for _ in range(100): f()
th, tw = (8, 16) if P == 128 else (4, 16)
nt = B * ((H + th - 1) // th) * ((W + tw - 1) // tw)
st = torch.zeros(nt * 16, dtype=torch.int64, device='cuda')
lib.aod_dbg_set_bnw_stamps.argtypes = [ctypes.c_void_p]
assert lib.aod_dbg_set_bnw_stamps(st.data_ptr()) == 0
torch.cuda.synchronize(); f(); torch.cuda.synchronize()
t = st.cpu().numpy().reshape(nt, 16).astype(np.float64) * 0.01
print(f'P={P} {mode}: {nt} tiles, kernel span {t[:, 6].max() - t[:, 0].min():.1f} us; per-tile phase durations (us):')
for k, name in enumerate(['phase 1 K loop (conv1 on halo)', 'epilogue 1', 'phase 2 (conv2)', 'epilogue 2', 'phase 3 (conv3 + res + stores)', 'store drain']):
    d = t[:, k + 1] - t[:, k]
    print(f'  {name:34s} mean {d.mean():6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}')
tot = t[:, 6] - t[:, 0]
print(f'  tile total {tot.mean():.2f} us; tile start percentiles 0/25/50/75/100: {[round(float(np.percentile(t[:, 0] - t[:, 0].min(), q)), 1) for q in (0, 25, 50, 75, 100)]}')

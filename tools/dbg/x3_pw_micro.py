"""Reference-precision 1x1 convs of the backbone one by one: time per launch, activation bytes / time, and (debug build) tile phases.
  python tools/dbg/tile_timing.py build     (here)
  python tools/dbg/x3_pw_micro.py                                                   (product library: times)
  AOD_HIP_LIB=tools/dbg/_build/libaodhip_dbg.so python tools/dbg/x3_pw_micro.py tt   (debug library: tile phases)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aod_meh_hua_amd import functional as AF, hipops as ho
from aod_meh_hua_amd._C import lib
AF.set_precision('bf16x3')
TT = 'tt' in sys.argv
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
# (B, H, W, Cin, Cout, residual, R)
shapes = [(16, 64, 64, 128, 512, 1, 1), (16, 32, 32, 256, 1024, 1, 1), (16, 32, 32, 1024, 256, 0, 1), (16, 16, 16, 512, 2048, 1, 1), (16, 64, 64, 512, 128, 0, 1),
          (16, 128, 128, 256, 128, 0, 1), (16, 128, 128, 64, 256, 0, 1), (16, 16, 16, 2048, 512, 0, 1), (16, 64, 64, 512, 256, 0, 1), (16, 32, 32, 256, 256, 0, 3)]
for B, H, W, Cc, N, use_res, R in shapes:
    M = B * H * W
    segs = [ho.Seg(B, H, W, 0)]
    x = ho.x3_split(torch.randn(M, Cc, device='cuda'))
    w = torch.randn(N, Cc, R, R, device='cuda') * 0.05
    wp = ho.x3_split(w.permute(0, 2, 3, 1).reshape(N * R * R, Cc).contiguous()).view(N, R, R, -1)
    res = ho.x3_split(torch.randn(M, N, device='cuda')) if use_res else None
    bias = torch.randn(N, device='cuda')
    out = torch.empty(M, ho.xw(N), device='cuda', dtype=torch.bfloat16)
    f = lambda: ho.conv2d_rows(x, segs, wp, N, R, R, 1, R // 2, 1, relu=True, pre_shift=bias, res=res, out=out)
    us = t(f)
    mb = (M * Cc + M * N * (2 if use_res else 1)) * 4 / 1e6
    fl = 2.0 * M * N * Cc * R * R
    print(f'M={M:7d} C={Cc:5d} N={N:5d} R={R} res={use_res}: {us:7.1f} us  {mb / us * 1e3:6.0f} GB/s  {fl / us / 1e6:6.1f} TF alg', flush=True)
    if TT:
        ntile = ((M + 63) // 64) * ((N + 63) // 64)
        st = torch.zeros(ntile * 16, dtype=torch.int64, device='cuda')
        lib.aod_dbg_set_tile_stamps.argtypes = [__import__('ctypes').c_void_p]
        assert lib.aod_dbg_set_tile_stamps(st.data_ptr()) == 0
        torch.cuda.synchronize(); f(); torch.cuda.synchronize()
        lib.aod_dbg_set_tile_stamps(None)
        s = st.cpu().numpy().reshape(ntile, 16)
        s = s[s[:, 0] != 0]
        tt = s[:, :7].astype(np.float64) * 0.01
        names = ['prologue', 'decode+first load', 'main loop', 'acc->LDS', 'epilogue stores issued', 'store drain']
        print(f'   {len(s)} tiles, span {tt[:, 6].max() - tt[:, 0].min():.1f} us; phases (mean us): ' +
              '  '.join(f'{n} {float((tt[:, k + 1] - tt[:, k]).mean()):.2f}' for k, n in enumerate(names)) + f'  | tile total {float((tt[:, 6] - tt[:, 0]).mean()):.2f}')

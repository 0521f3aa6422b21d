"""Calibration: what the vendor GEMM (hipBLASLt through torch.matmul, bf16, random data) reaches on the GEMM shapes the
conv layers reduce to -- a practical ceiling for the implicit-GEMM kernel on the same device in the same call."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aod_meh_hua_amd import hipops as ho
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
shapes = [('tower 3x3 as GEMM', 87296, 256, 2304), ('big square', 8192, 8192, 8192), ('l1 1x1 64->256', 262144, 256, 64),
          ('l2 1x1 512->128', 65536, 128, 512), ('l3 3x3 256', 16384, 256, 2304), ('l4 1x1 2048->512', 4096, 512, 2048),
          ('l1 3x3 64', 262144, 64, 576)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(K, N, device='cuda').bfloat16()
    us = t(lambda: torch.matmul(a, b))
    print(f'{name:22s} M={M:7d} N={N:5d} K={K:5d}  matmul {us:8.1f} us {2.0*M*N*K/us/1e6:7.1f} TF   {(M*K+K*N+M*N)*2/us/1e3:7.1f} GB/s')
# ours on the tower
B = 16
sizes = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
segs, r = [], 0
for h, w in sizes:
    segs.append(ho.Seg(B, h, w, r)); r += B * h * w
M = r
x = torch.randn(M, 256, device='cuda').bfloat16()
w = torch.randn(256, 256, 3, 3, device='cuda') * 0.02
wp = ho.pack_weight_fwd(w)
us = t(lambda: ho.conv2d_rows(x, segs, wp, 256, 3, 3, 1, 1, 1, relu=True))
print(f'ours tower fwd {us:8.1f} us {2.0*M*256*2304/us/1e6:7.1f} TF')

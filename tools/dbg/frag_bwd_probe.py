"""which operand of the register-streamed backward kernel's third product goes wrong?  constant residual rows / all-positive masks remove one
source of per-pixel data at a time"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
exec(open(os.path.join(os.path.dirname(__file__), 'frag_check.py')).read().split("reps = int")[0])
B, H, W = 3, 32, 32
M = B * H * W
for name, gy_c, x_pos, a_pos in (('plain', 0, 0, 0), ('const G', 1, 0, 0), ('mask x all positive', 0, 1, 0), ('const G + x positive', 1, 1, 0), ('all masks positive', 0, 1, 1)):
    gy = (torch.full((M, C4), 0.125, device='cuda') if gy_c else rnd(M, C4) * 0.1).bfloat16()
    x = (torch.ones(M, C4, device='cuda') if x_pos else rnd(M, C4)).bfloat16()
    a1 = (torch.ones(M, P, device='cuda') if a_pos else rnd(M, P)).bfloat16()
    a2 = (torch.ones(M, P, device='cuda') if a_pos else rnd(M, P)).bfloat16()
    r0 = ho.bottleneck_bwd(gy, B, H, W, w1, w2, w3, a2, a1, x)
    r1 = ho.bottleneck_bwd(gy, B, H, W, f1, f2, f3, a2, a1, x, frag=True)
    torch.cuda.synchronize()
    print(name, [int((~(a.float() == b.float())).sum()) for a, b in zip(r0[:3], r1[:3])])

"""register-streamed 256-plane bottleneck (fragment-major filters) against the LDS-ring form on random operands: where do they differ?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from aod_meh_hua_amd import hipops as ho
from aod_meh_hua_amd._C import call, ptr, stream
import ctypes as C
P = 256; C4 = 1024
g = torch.Generator(device='cuda').manual_seed(1)
rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)
w1 = (rnd(P, C4) * 0.05).bfloat16(); w2 = (rnd(P, 9 * P) * 0.03).bfloat16(); w3 = (rnd(C4, P) * 0.05).bfloat16()
v = lambda n: (torch.rand(n, device='cuda', generator=g) + 0.5, rnd(n) * 0.1)
(s1, b1), (s2, b2), (s3, b3) = v(P), v(P), v(C4)


class Rec(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('rows', C.c_int32), ('K', C.c_int32), ('blk0', C.c_int32), ('pad_', C.c_int32)]


def frag(w):
    out = torch.empty(w.numel(), dtype=torch.bfloat16, device='cuda')
    r = (Rec * 1)()
    r[0].src, r[0].dst, r[0].rows, r[0].K, r[0].blk0 = w.data_ptr(), out.data_ptr(), w.shape[0], w.shape[1], 0
    tab = torch.frombuffer(bytearray(bytes(r)), dtype=torch.uint8).cuda()
    call('aod_frag_pack', ptr(tab), 1, (w.numel() + 2047) // 2048, stream())
    torch.cuda.synchronize()
    return out


f1, f2, f3 = frag(w1), frag(w2), frag(w3)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nbad = 0
for B, H, W in ((1, 4, 16), (2, 13, 37), (3, 32, 32), (16, 32, 32)) * reps:
    x = rnd(B * H * W, C4).relu().bfloat16()
    y0, t10, t20 = ho.bottleneck128_fwd(x, B, H, W, w1, s1, b1, w2, s2, b2, w3, s3, b3, keep=True)
    y1, t11, t21 = ho.bottleneck128_fwd(x, B, H, W, f1, s1, b1, f2, s2, b2, f3, s3, b3, keep=True, frag=True)
    torch.cuda.synchronize()
    for n, a, b in (('t1', t10, t11), ('t2', t20, t21), ('y', y0, y1)):
        d = (a.float() - b.float()).abs()
        bad = (d > 0).nonzero()
        print(B, H, W, n, 'max diff', float(d.max()), 'n bad', len(bad), 'first', bad[:3].tolist(), 'bad cols mod 8', sorted(set((bad[:, 1] % 8).tolist())) if len(bad) else '', 'rows mod 16', sorted(set((bad[:, 0] % 16).tolist())) if len(bad) else '', 'chunks', sorted(set((bad[:, 1] // 256).tolist())) if len(bad) else '', 'waves', sorted(set(((bad[:, 1] % 256) // 32).tolist())) if len(bad) else '')
    # backward form: same operands read as gradient / masks
    gy = (rnd(B * H * W, C4) * 0.1).bfloat16()
    a1 = rnd(B * H * W, P).relu().bfloat16(); a2 = rnd(B * H * W, P).relu().bfloat16()
    r0 = ho.bottleneck_bwd(gy, B, H, W, w1, w2, w3, a2, a1, x)
    r1 = ho.bottleneck_bwd(gy, B, H, W, f1, f2, f3, a2, a1, x, frag=True)
    torch.cuda.synchronize()
    for n, a, b in zip(('gx', 'gt2', 'gt1'), r0[:3], r1[:3]):
        d = (a.float() - b.float()).abs()
        if not bool((d == 0).all()):
            nbad += 1
            bad = (~(d == 0)).nonzero()
            print(B, H, W, n, 'BWD n bad', len(bad), 'cols mod 8', sorted(set((bad[:, 1] % 8).tolist())), 'rows mod 16', sorted(set((bad[:, 0] % 16).tolist())), 'chunks', sorted(set((bad[:, 1] // 256).tolist())), 'waves', sorted(set(((bad[:, 1] % 256) // 32).tolist())), 'nan', int(torch.isnan(b.float()).sum()), 'zeros where ref nonzero', int(((b.float() == 0) & (a.float() != 0)).sum()))
    for n, a, b in zip(('cx', 'c2', 'c1'), r0[3:], r1[3:]):
        if not torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(a.abs().max())):
            nbad += 1
            print(B, H, W, n, 'BWD colsum diff', float((a - b).abs().max()), float(a.abs().max()))
    nbad += int(not torch.equal(y0, y1)) + int(not torch.equal(t10, t11)) + int(not torch.equal(t20, t21))
print('TOTAL BAD', nbad)

"""Race screen for the kernels with hand-counted waits / raw barriers (bottleneck, stem, pointwise, asm-read wgrad): many back-to-back
launches on fresh random data at several sizes, every result compared bit for bit with the reference path (or with a repeat run)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from aod_meh_hua_amd import functional as AF, hipops as ho
from aod_meh_hua_amd._C import lib
dev = torch.device('cuda')
model, cfg = B.build_model(dev, B.CONFIGS['voc512'])
model.eval()
bb = model.backbone
torch.manual_seed(3)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
with torch.no_grad():
    for it in range(iters):
        Bn, H, W = [(16, 128, 128), (2, 50, 37), (8, 72, 104), (1, 16, 16), (4, 128, 64)][it % 5]
        x = torch.randn(Bn, 64, H, W, device=dev).relu().bfloat16().contiguous(memory_format=torch.channels_last)
        for blk in bb.layer1:
            os.environ['AOD_FUSE_BOTTLENECK'] = '1'
            ys = [blk(x) for _ in range(3)]                       # back to back
            os.environ['AOD_FUSE_BOTTLENECK'] = '0'
            y0 = blk(x)
            for y in ys:
                if not torch.equal(y, y0): bad += 1; print('bottleneck mismatch', it, Bn, H, W, flush=True)
            x = y0
        img = torch.randn(Bn, 3, 2 * H, 2 * W + (2 if it % 3 == 0 else 0), device=dev)
        a = [AF.stem_pool_s2d(img, bb.conv1, bb.norm1) for _ in range(3)]
        b = AF.max_pool_3x3_s2(AF.stem_conv_s2d(img, bb.conv1, bb.norm1))
        for t in a:
            if not torch.equal(t, b): bad += 1; print('stem mismatch', it, flush=True)
        # pointwise streaming kernel vs general kernel
        M = Bn * H * W
        segs = [ho.Seg(Bn, H, W, 0)]
        xx = torch.randn(M, 256, device=dev).bfloat16(); wp = ho.pack_weight_fwd(torch.randn(64, 256, 1, 1, device=dev) * 0.06)
        res = torch.randn(M, 64, device=dev).bfloat16(); sc = torch.rand(64, device=dev) + 0.5; sh = torch.randn(64, device=dev)
        outs = []
        for mode in (0, 1, 1):
            lib.aod_set_pointwise_mode(mode)
            o = torch.empty(M, 64, device=dev, dtype=torch.bfloat16)
            ho.conv2d_rows(xx, segs, wp, 64, 1, 1, 1, 0, 1, pre_scale=sc, pre_shift=sh, res=res, relu=True, out=o)
            outs.append(o)
        lib.aod_set_pointwise_mode(-1)
        if not (torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])): bad += 1; print('pointwise mismatch', it, flush=True)
        # wgrad (asm fragment reads; 256 x 256 tile at the big size): repeat-run determinism
        if it % 5 == 0:
            lv = [(16, 64, 64), (16, 32, 32)]
            sg, r0 = [], 0
            for b_, h_, w_ in lv:
                sg.append(ho.Seg(b_, h_, w_, r0)); r0 += b_ * h_ * w_
            xr = torch.randn(r0, 256, device=dev).bfloat16(); dz = torch.randn(r0, 256, device=dev).bfloat16()
            zs = ho.out_segs(sg, 3, 3, 1, 1, 1)
            g1 = ho.unpack_wgrad(ho.conv2d_wgrad_rows(xr, sg, dz, zs, 3, 3, 1, 1, 1), 256, 256).clone()
            g2 = ho.unpack_wgrad(ho.conv2d_wgrad_rows(xr, sg, dz, zs, 3, 3, 1, 1, 1), 256, 256)
            if not torch.equal(g1, g2): bad += 1; print('wgrad mismatch', it, flush=True)
torch.cuda.synchronize()
print('race screen:', iters, 'iterations,', bad, 'mismatches')

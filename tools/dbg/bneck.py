"""Fused 64-channel bottleneck (csrc/bottleneck.hip) against the three-launch form on the layer1 blocks: max deviation, us per block."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from aod_meh_hua_amd import functional as AF
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
dev = torch.device('cuda')
model, cfg = B.build_model(dev, B.CONFIGS['voc512'])
model.eval()
torch.manual_seed(1)
for bn in model.backbone.modules():
    if isinstance(bn, torch.nn.BatchNorm2d):
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5); bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2)
layer1 = model.backbone.layer1
for (Bn, H, W) in ((2, 50, 37), (1, 16, 16), (16, 128, 128)):
    x = (torch.randn(Bn, 64, H, W, device=dev) * 1.0).relu().bfloat16().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for bi, blk in enumerate(layer1):
            os.environ['AOD_FUSE_BOTTLENECK'] = '0'
            y0 = blk(x)
            torch.cuda.synchronize(); print('  a', flush=True)
            us0 = t(lambda: blk(x))
            print('  b', flush=True)
            os.environ['AOD_FUSE_BOTTLENECK'] = '1'
            assert AF.bottleneck64_applies(blk, x)
            y1 = blk(x)
            torch.cuda.synchronize(); print('  c', flush=True)
            us1 = t(lambda: blk(x))
            print('  d', flush=True)
            torch.cuda.synchronize()
            d = (y1.float() - y0.float()).abs()
            sc = float(y0.float().abs().max())
            nbad = int((d > 2e-2 * sc).sum())
            print(f'{Bn}x{H}x{W} block {bi} Cin={x.shape[1]:3d}: 3 launches {us0:7.1f} us  fused {us1:7.1f} us   max|diff| {float(d.max()):.4f} (max |y| {sc:.2f}, mean |y| {float(y0.float().abs().mean()):.3f}) '
                  f'elements off by > 2% of max: {nbad}', flush=True)
            x = y0

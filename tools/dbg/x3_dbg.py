"""debug: which ingredient of a reference-precision conv's backward deviates (BN fold / ReLU mask / plain)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from aod_meh_hua_amd import functional as AF, hipops as ho
from aod_meh_hua_amd.mmcv_lite import BatchNorm2d

AF.set_precision('bf16x3')
g = torch.Generator(device='cuda').manual_seed(5)
rnd = lambda *sh: torch.randn(*sh, device='cuda', generator=g)
X = lambda t: ho.x3_split(t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous())
Fm = lambda rows, B, H, W, C: ho.x3_merge(rows, C).view(B, H, W, C).permute(0, 3, 1, 2)
for use_bn, relu in ((True, False), (False, True), (True, True), (False, False)):
    B, C, O, H, W, R = 2, 64, 64, 32, 32, 3
    x = rnd(B, C, H, W)
    w = (rnd(O, C, R, R) / (C * R * R) ** 0.5).requires_grad_()
    bn = bias = None
    if use_bn:
        bn = BatchNorm2d(O).cuda().eval()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(O, device='cuda', generator=g) + 0.5); bn.bias.copy_(rnd(O) * 0.1)
            bn.running_mean.copy_(rnd(O) * 0.1); bn.running_var.copy_(torch.rand(O, device='cuda', generator=g) + 0.5)
    else:
        bias = (rnd(O) * 0.1).requires_grad_()
    xx = AF.as_nchw(X(x), B, H, W).requires_grad_()
    y = AF.conv_bn_act(xx, w, bn=bn, bias=bias, stride=1, pad=1, relu=relu)
    xr = x.clone().requires_grad_()
    z = F.conv2d(xr, w.detach(), bias.detach() if bias is not None else None, 1, 1)
    if bn is not None:
        z = F.batch_norm(z, bn.running_mean, bn.running_var, bn.weight.detach(), bn.bias.detach(), False, 0.0, bn.eps)
    if relu:
        z = torch.relu(z)
    gy = rnd(B, O, H, W)
    y.backward(AF.as_nchw(X(gy), B, H, W))
    z.backward(gy)
    gx = Fm(AF.as_rows(xx.grad), B, H, W, C)
    d = (gx - xr.grad).abs()
    print('bn', use_bn, 'relu', relu, 'max err', float(d.max()), 'ref max', float(xr.grad.abs().max()), 'frac elems > 1e-3:', float((d > 1e-3).float().mean()),
          'fwd err', float((Fm(AF.as_rows(y), B, H, W, O) - z).abs().max()))
    bad = (d > 1e-3).nonzero()
    if len(bad):
        print('  first bad idx', bad[:5].tolist(), 'by channel', torch.bincount(bad[:, 1], minlength=C).tolist()[:16], 'by y', torch.bincount(bad[:, 2], minlength=H).tolist()[:8])

"""scoring forward of the same images inside batches of different sizes: which tensors stop being bit-identical?  PREC=bf16x3|bf16"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from aod_meh_hua_amd import functional as AF
from aod_meh_hua_amd.datasets import DevicePhiloxPool
AF.set_precision(os.environ.get('PREC', 'bf16x3'))
dev = torch.device('cuda', 0)
SIZE = int(os.environ.get('SIZE', '512'))
model, _ = bench.build_model(dev, dict(bench.CONFIGS['voc512']))
model.eval()
ds = DevicePhiloxPool(64, (SIZE, SIZE), seed=20)
def fwd(ids):
    img = ds.device_batch(ids, dev)['img'][0]
    with torch.no_grad():
        feats_b = model.backbone(img)
        feats = model.neck(feats_b)
        (cls, box), L = model.bbox_head.test_heads(feats)
    return dict(backbone=[AF.x3_to_f32(f, f.shape[1] // 2) if AF.get_precision() == 'bf16x3' else f.float() for f in feats_b],
                neck=[AF.x3_to_f32(f, 256) if AF.get_precision() == 'bf16x3' else f.float() for f in feats],
                cls=[c.float() for c in cls], box=[b.float() for b in box], L=[l.float() for l in L])
ref = fwd(list(range(16)))
for B in (5, 8, 3, 1):
    got = fwd(list(range(B)))
    msg = []
    for k in ref:
        for l, (a, b) in enumerate(zip(ref[k], got[k])):
            a = a[:B]
            if not torch.equal(a, b):
                msg.append(f'{k}[{l}] max abs diff {float((a - b).abs().max()):.2e} ({int((a != b).sum())} of {a.numel()} elements)')
    print('batch', B, 'vs the same images in a batch of 16:', msg or 'bit-identical everywhere')

import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from aod_meh_hua_amd import functional as AF
dev = torch.device('cuda')
model, cfg = B.build_model(dev, B.CONFIGS['voc512'])
model.eval()
layer1 = model.backbone.layer1
for shp in ((1, 32, 32), (1, 16, 48), (2, 50, 37), (16, 128, 128)):
  print('shape', shp, flush=True)
  x = torch.randn(shp[0], 64, shp[1], shp[2], device=dev).relu().bfloat16().contiguous(memory_format=torch.channels_last)
  with torch.no_grad():
    for bi, blk in enumerate(layer1):
        print('block', bi, 'Cin', x.shape[1], flush=True)
        os.environ['AOD_FUSE_BOTTLENECK'] = '0'
        y0 = blk(x); torch.cuda.synchronize(); print(' unfused ok', flush=True)
        os.environ['AOD_FUSE_BOTTLENECK'] = '1'
        y1 = blk(x); torch.cuda.synchronize(); print(' fused ok', float((y1.float()-y0.float()).abs().max()), flush=True)
        x = y0

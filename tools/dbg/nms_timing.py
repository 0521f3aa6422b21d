"""phase stamps of nms_kernel (image 0) on the bench's scoring batch: debug build (python tools/dbg/tile_timing.py build), then
   AOD_HIP_LIB=tools/dbg/_build/libaodhip_dbg.so python tools/dbg/nms_timing.py"""
import sys, os, ctypes, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from aod_meh_hua_amd._C import lib
dev = torch.device('cuda')
cd = B.CONFIGS['voc512']
model, cfg = B.build_model(dev, cd)
pool = B.synth_batch(16, 512, 512, dev, 1)
pm = copy.deepcopy(model)
B.calibrate_head(pm, pool['img'])
ids = torch.arange(16, device=dev)
pm.eval()
st = torch.zeros(16, dtype=torch.int64, device=dev)
lib.aod_dbg_set_nms_stamps.argtypes = [ctypes.c_void_p]
with torch.no_grad():
    for _ in range(3):
        pm(img=[pool['img']], img_metas=[pool['img_metas']], image_ids=ids, **B.SCORE_KW)
    assert lib.aod_dbg_set_nms_stamps(st.data_ptr()) == 0
    torch.cuda.synchronize()
    pm(img=[pool['img']], img_metas=[pool['img_metas']], image_ids=ids, **B.SCORE_KW)
    torch.cuda.synchronize()
t = st.cpu().numpy().astype(float)
print('nvalid', int(t[8]), 'kept', int(t[9]))
names = ['compaction', 'radix select (last tranche)', 'gather + sort', 'greedy scan', 'tail']
for k, n in enumerate(names):
    print(f'  {n:30s} {(t[k + 1] - t[k]) * 0.01:8.2f} us')
print('  total', (t[5] - t[0]) * 0.01)

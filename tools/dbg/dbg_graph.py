import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_graphs import _build, _batch, _eager_iter
from aod_meh_hua_amd.graphs import GraphedTrainStep
from aod_meh_hua_amd import functional as AF
b0 = _batch(31)
model2, opt2, opt_L2 = _build()
sd0 = {k: v.detach().clone() for k, v in model2.state_dict().items()}
l_e = float(model2.train_step(b0, Labeled=True, Pseudo=False)[0]['loss'].detach())
print('eager loss at sd0', l_e)
gs = GraphedTrainStep(model2, opt2, opt_L2, warmup=1, Labeled=True, Pseudo=False)
o = gs(b0); print('first graphed call loss', float(o['loss']))
model2.load_state_dict(sd0, strict=True)
for st in (opt2.state, opt_L2.state):
    for s in st.values(): s['momentum_buffer'].zero_()
w = model2.bbox_head.retina_cls.weight
it = AF.PREP.items[id(w)]
print('wf before replay matches w?', float((it.wf.float().permute(0, 3, 1, 2)[:, :w.shape[1]] - w.detach()).abs().max()))
o = gs(b0); print('replay after reload loss', float(o['loss']), 'expected', l_e)
print('wf after replay vs sd0 w', float((it.wf.float().permute(0, 3, 1, 2)[:, :w.shape[1]] - sd0['bbox_head.retina_cls.weight']).abs().max()))
print('n items', len(AF.PREP.items), 'order', len(AF.PREP.order))
print('---- more checks')
model2.load_state_dict(sd0, strict=True)
ptr_before = w.data_ptr()
it.wf.zero_()
print('w vs sd0 before replay', float((w.detach() - sd0['bbox_head.retina_cls.weight']).abs().max()), 'table ptr', AF.PREP.table.data_ptr())
gs.graphs[0].replay(); torch.cuda.synchronize()
print('wf rewritten by replay? absmax', float(it.wf.float().abs().max()), ' vs sd0', float((it.wf.float().permute(0, 3, 1, 2)[:, :w.shape[1]] - sd0['bbox_head.retina_cls.weight']).abs().max()))
print('w moved by replay', float((w.detach() - sd0['bbox_head.retina_cls.weight']).abs().max()), 'same ptr', w.data_ptr() == ptr_before)
# is the refresh the FIRST thing in the graph?  emulate: eager refresh from sd0 then compare wf
model2.load_state_dict(sd0, strict=True)
AF.PREP.refresh(); torch.cuda.synchronize()
print('eager refresh wf vs sd0', float((it.wf.float().permute(0, 3, 1, 2)[:, :w.shape[1]] - sd0['bbox_head.retina_cls.weight']).abs().max()))
print('---- which weights does wf hold after a replay?')
model2.load_state_dict(sd0, strict=True)
it.wf.zero_()
gs.graphs[0].replay(); torch.cuda.synchronize()
def perr(ref): return float((it.wf.float().permute(0, 3, 1, 2)[:, :w.shape[1]] - ref).abs().max())
print('wf vs w_after', perr(w.detach()), ' wf vs sd0', perr(sd0['bbox_head.retina_cls.weight']))
# other layers
for name in ('backbone.layer2.0.conv1.weight', 'neck.fpn_convs.0.conv.weight', 'bbox_head.cls_convs.0.conv.weight', 'bbox_head.L_convs.0.conv.weight'):
    p = dict(model2.named_parameters())[name]; i2 = AF.PREP.items[id(p)]
    e_after = float((i2.wf.float().permute(0, 3, 1, 2)[:, :p.shape[1]] - p.detach()).abs().max())
    e_sd0 = float((i2.wf.float().permute(0, 3, 1, 2)[:, :p.shape[1]] - sd0[name]).abs().max())
    print(name, 'wf vs w_after', e_after, 'wf vs sd0', e_sd0)

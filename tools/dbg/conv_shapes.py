"""Per-shape conv timing of one train+score step (HIP events around every conv launch)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from aod_meh_hua_amd import hipops as ho
dev = torch.device('cuda', 0)
model, cfg = bench.build_model(dev)
opt, opt_L = bench.make_optimizers(model, cfg)
B, H = 16, 512
data = bench.synth_batch(B, H, H, dev, 20)
kw = dict(return_loss=False, rescale=True, isEval=False, isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum',
          scaleUnc=False, showNMS=False, saveUnc=False, saveMaxConf=False, clsW=False, batchIdx=0)
def step(score):
    model.train()
    out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    opt.zero_grad(); out['loss'].backward(); opt.step()
    lossL = model.train_step_L(prev, head_out, feat_out)
    opt_L.zero_grad(); lossL['loss'].backward(); opt_L.step()
    if score:
        model.eval()
        with torch.no_grad():
            model(img=[data['img']], img_metas=[data['img_metas']], **kw)
for _ in range(3): step(False)
ho.PROFILE = []
step(False)
torch.cuda.synchronize()
agg = {}
for kind, shape, flops, e0, e1, _scope in ho.PROFILE:
    a = agg.setdefault((kind, shape), [0, 0.0, flops])
    a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print('total conv ms', round(tot, 3))
for (kind, shape), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f'{kind:6s} M={shape[0]:7d} N={shape[1]:5d} K={shape[2]:6d}  n={n:3d}  ms={ms:7.3f}  avg_us={ms/n*1e3:8.1f}  TF={fl*n/ms/1e9:7.1f}')

"""Host enqueue time vs GPU time of one bench step (is the step launch-bound?)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device('cuda', 0)
model, cfg = bench.build_model(dev)
opt, opt_L = bench.make_optimizers(model, cfg)
data = bench.synth_batch(16, 512, 512, dev, 20)
kw = dict(return_loss=False, rescale=True, isEval=False, isUnc='Epistemic', uPool='Entropy_NMS', uPool2='objectSum_scaleMax_classSum',
          scaleUnc=False, showNMS=False, saveUnc=False, saveMaxConf=False, clsW=False, batchIdx=0)
def step():
    model.train()
    out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    opt.zero_grad(); out['loss'].backward(); opt.step()
    lossL = model.train_step_L(prev, head_out, feat_out)
    opt_L.zero_grad(); lossL['loss'].backward(); opt_L.step()
    model.eval()
    with torch.no_grad():
        model(img=[data['img']], img_metas=[data['img_metas']], **kw)
for _ in range(3): step()
torch.cuda.synchronize()
hs, ts = [], []
for _ in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    hs.append((t1 - t0) * 1e3); ts.append((t2 - t0) * 1e3)
print('host enqueue ms', [round(h, 2) for h in hs], 'total ms', [round(t, 2) for t in ts])

"""Which tensors still take a stand-alone activation-backward pass (aod_act_bwd) or an aten gradient add in one eager training step:
shape + the conv (weight shape) whose backward asked for it."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from aod_meh_hua_amd import hipops as ho, functional as AF
dev = torch.device('cuda')
cd = B.CONFIGS['voc512']
model, cfg = B.build_model(dev, cd)
opt, opt_L = B.make_optimizers(model, cfg)
data = B.synth_batch(16, 512, 512, dev, 0)
names = {id(p): n for n, p in model.named_parameters()}
log = []
orig = ho.act_bwd
cur = [None]
bw = AF.ConvFn.backward


def act_bwd(g, a, *args, **kw):
    log.append((tuple(g.shape), str(g.dtype)[6:], kw.get('relu'), cur[0]))
    return orig(g, a, *args, **kw)


def backward(ctx, *gouts):
    cur[0] = names.get(id(ctx.saved_tensors[0]), '?')
    return bw(ctx, *gouts)


def step():
    model.train()
    out, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    opt.zero_grad(); out['loss'].backward()
    outL = model.train_step_L(prev, head_out, feat_out, Labeled=True, Pseudo=False)
    opt_L.zero_grad(); outL['loss'].backward()
    opt.step(); opt_L.step()


step()
ho.act_bwd = act_bwd
AF.ConvFn.backward = staticmethod(backward)
step()
torch.cuda.synchronize()
for l in log:
    print(l)

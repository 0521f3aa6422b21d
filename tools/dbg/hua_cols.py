"""Debug: per-pair epistemic of aod_hua_score (scale mode) vs the numpy Philox oracle for several Dirichlet widths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from aod_meh_hua_amd import scoring
from oracle import hua as ohua
for nd, spread in ((20, 8), (21, 8), (23, 8), (80, 8), (79, 8), (81, 8), (81, 40), (20, 40)):
    B, n = 1, 48
    gen = torch.Generator().manual_seed(5)
    sc = torch.rand(B, n, nd, generator=gen) ** spread
    hot = torch.randint(0, nd, (B, n), generator=gen)
    sc[torch.arange(B)[:, None], torch.arange(n)[None], hot] += 12.0 * torch.rand(B, n, generator=gen)
    sc = sc / sc.sum(-1, keepdim=True)
    lam = torch.rand(B, n, generator=gen) * 0.3 + 0.01
    anchor = torch.arange(n, dtype=torch.int32)[None].repeat(B, 1) * 3 + 11
    cand = scoring.Candidates(torch.zeros(B, n, 4).cuda(), sc.cuda().contiguous(), lam.cuda(), anchor.cuda(), [0, n], torch.ones(1, B, dtype=torch.int32).cuda(), [None])
    ids = torch.tensor([3], device='cuda')
    unc, pc, pout = scoring.hua_score(cand, None, None, ids, 1, (0, 0, 0), want_pairs=True, seed=20, scale_mode=True, dirichlet_cols=nd, num_samples=200)
    torch.cuda.synchronize()
    pout = pout.cpu().numpy()
    fg = (sc[0].max(-1)[0] > 0.3).nonzero()[:, 0]
    lhat = lam[0].mean() / (lam[0][fg] + 1e-7) * 25
    alpha = (sc[0][fg] * lhat[:, None]).numpy()
    ale, epi = ohua.philox_dirichlet_stats(alpha, 3, anchor[0][fg].numpy(), np.zeros(len(fg), np.int64), 20, num_samples=200)
    got = pout[0, :int(pc[0])]
    err = np.abs(got[:, 3] - epi)
    print(nd, spread, 'pairs', int(pc[0]), len(fg), 'median', np.median(err), 'max', err.max(), 'epi mean', epi.mean(), 'ale err', np.abs(got[:, 2] - ale).max())

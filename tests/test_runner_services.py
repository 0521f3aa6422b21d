"""CPU tests of the runner services on the hot path's caller side: StepLrUpdaterHook (by-epoch step schedule on the MAIN
optimizer only, + SSD's linear warm-up), checkpoint / resume round trip with the reference's file layout (epoch_N.pth + latest.pth,
{meta, state_dict, optimizer}; mmdet/utils/Epoch_Based_Runner_Lambda.py:144-169, mmdet/apis/train_Lambda.py:85-88), and the distributed
aspect-ratio group sampler (mmdet/datasets/samplers/group_sampler.py:53-148) with its per-epoch reshuffle."""
import os

import numpy as np
import torch

from aod_meh_hua_amd.mmcv_lite import BaseRunner, StepLrUpdaterHook
from aod_meh_hua_amd.utils.Epoch_Based_Runner_Lambda import MyEpochBasedRunnerLambda


class _Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(3, 2)
        self.retina_L = torch.nn.Linear(2, 1)


class _Loader:
    sampler = None

    def __init__(self, it):
        self.items = list(it)

    def __iter__(self):
        return iter(self.items)

    def __len__(self):
        return len(self.items)


class _EpochRunner(MyEpochBasedRunnerLambda):
    """run_iter replaced by a plain SGD step so that the services can be tested without a GPU"""

    def run_iter(self, data_batch, train_mode, **kwargs):
        self.lrs.append((self.epoch, self.optimizer.param_groups[0]['lr'], self.optimizer_L.param_groups[0]['lr']))
        for opt in (self.optimizer, self.optimizer_L):
            opt.zero_grad()
        loss = (self.model.retina_L(self.model.a(data_batch)) ** 2).mean()
        loss.backward()
        self.optimizer.step(), self.optimizer_L.step()
        self.outputs = dict(loss=loss.detach(), log_vars=dict(loss=float(loss)), num_samples=1)


def _runner(tmp, lr_config, seed=0):
    torch.manual_seed(seed)
    m = _Tiny()
    r = _EpochRunner(m, optimizer=torch.optim.SGD(m.a.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4), work_dir=str(tmp), max_epochs=3)
    r.optimizer_L = torch.optim.SGD(m.retina_L.parameters(), lr=1e-3, momentum=0.9, weight_decay=1e-4)
    r.lrs = []
    r.register_training_hooks(lr_config, None, dict(interval=1), None)
    return r, m


def test_step_lr_schedule_touches_only_the_main_optimizer(tmp_path):
    r, m = _runner(tmp_path, dict(policy='step', step=[2]))
    g = torch.Generator().manual_seed(1)
    loader = _Loader(torch.randn(4, 3, generator=g) for _ in range(5))
    r.run_SSL([loader], [('train', 1)], 3, onlyEval=False)
    by_epoch = {e: {(a, b) for ee, a, b in r.lrs if ee == e} for e in range(3)}
    assert by_epoch[0] == by_epoch[1] == {(1e-3, 1e-3)}
    assert len(by_epoch[2]) == 1 and np.isclose(list(by_epoch[2])[0][0], 1e-4) and list(by_epoch[2])[0][1] == 1e-3      # lr x 0.1 at epoch 2
    assert r.epoch == 3 and r.iter == 15
    # the driver's default schedule (lr_config.step = [1000], tools/train_RetinaNet.py:179-183) never fires
    r2, _ = _runner(tmp_path / 'b', dict(policy='step', step=[1000]))
    r2.run_SSL([loader], [('train', 1)], 3, onlyEval=False)
    assert {a for _, a, _ in r2.lrs} == {1e-3}


def test_linear_warmup_of_the_ssd_config(tmp_path):
    """Config_SSD.py lr_config: warmup='linear', warmup_iters=500, warmup_ratio=0.001 (mmcv LrUpdaterHook.get_warmup_lr)."""
    h = StepLrUpdaterHook(step=[16, 22], warmup='linear', warmup_iters=4, warmup_ratio=0.001)
    r, _ = _runner(tmp_path, None)
    r.register_hook(h)
    loader = _Loader(torch.randn(4, 3) for _ in range(6))
    r.run_SSL([loader], [('train', 1)], 1, onlyEval=False)
    got = [a for _, a, _ in r.lrs]
    exp = [1e-3 * (1 - (1 - i / 4) * (1 - 0.001)) for i in range(4)] + [1e-3, 1e-3]
    assert np.allclose(got, exp, rtol=1e-12), (got, exp)


def test_checkpoint_files_and_resume_round_trip(tmp_path):
    g = torch.Generator().manual_seed(2)
    loader = _Loader(torch.randn(4, 3, generator=g) for _ in range(4))
    # uninterrupted run: 3 epochs
    r_full, m_full = _runner(tmp_path / 'full', dict(policy='step', step=[2]))
    r_full.run_SSL([loader], [('train', 1)], 3, onlyEval=False)
    # interrupted after epoch 2 + resumed from latest.pth
    r1, m1 = _runner(tmp_path / 'part', dict(policy='step', step=[2]))
    r1.run_SSL([loader], [('train', 1)], 2, onlyEval=False)
    wd = tmp_path / 'part'
    assert sorted(os.listdir(wd)) == ['epoch_1.pth', 'epoch_2.pth', 'latest.pth'] and os.path.islink(wd / 'latest.pth')
    assert os.readlink(wd / 'latest.pth') == 'epoch_2.pth'
    ck = torch.load(wd / 'latest.pth', map_location='cpu', weights_only=False)
    assert set(ck) == {'meta', 'state_dict', 'optimizer'} and ck['meta']['epoch'] == 2 and ck['meta']['iter'] == 8
    assert list(ck['state_dict']) == list(m1.state_dict())
    r2, m2 = _runner(tmp_path / 'part', dict(policy='step', step=[2]), seed=99)          # different init: everything must come from the file
    r2.optimizer_L.load_state_dict(r1.optimizer_L.state_dict())                          # (the reference checkpoints the main optimizer only)
    r2.resume(str(wd / 'latest.pth'))
    assert r2.epoch == 2 and r2.iter == 8
    r2.run_SSL([loader], [('train', 1)], 3, onlyEval=False)
    assert [e for e, _, _ in r2.lrs] == [2] * 4 and np.isclose(r2.lrs[0][1], 1e-4)        # resumes INTO the decayed epoch
    for (k, a), b in zip(m_full.state_dict().items(), m2.state_dict().values()):
        if k.startswith('a.'):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-8), k                             # momentum buffers restored too
    assert os.readlink(wd / 'latest.pth') == 'epoch_3.pth'


def test_distributed_group_sampler_shares_and_epochs():
    from aod_meh_hua_amd.datasets import DistributedGroupSampler

    class DS:
        flag = np.array([0] * 7 + [1] * 10, dtype=np.uint8)

        def __len__(self):
            return 17
    shares = {}
    for rank in range(2):
        s = DistributedGroupSampler(DS(), samples_per_gpu=2, num_replicas=2, rank=rank, seed=5)
        s.set_epoch(0)
        shares[rank] = list(s)
        assert len(shares[rank]) == len(s) == 10                # ceil(7/4)*2 + ceil(10/4)*2 per rank
        for i in range(0, 10, 2):                               # a batch never mixes aspect-ratio groups
            assert DS.flag[shares[rank][i]] == DS.flag[shares[rank][i + 1]]
    assert set(shares[0]) | set(shares[1]) == set(range(17))    # together the ranks cover the data set (with padding repeats)
    s = DistributedGroupSampler(DS(), samples_per_gpu=2, num_replicas=2, rank=0, seed=5)
    s.set_epoch(1)
    assert list(s) != shares[0]                                 # reshuffled every epoch
    s.set_epoch(0)
    assert list(s) == shares[0]                                 # deterministic given (seed, epoch)

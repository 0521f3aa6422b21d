"""CPU tests of `init_cfg=dict(type='Pretrained', ...)` (configs/_base_/Config_RetinaNet.py:33 `torchvision://resnet50`,
Config_SSD.py:32 `open-mmlab://vgg16_caffe`; mmdet/models/backbones/resnet.py:405-423, ssd_vgg.py:80-96): the scheme strings resolve against
a local directory, a torchvision-keyed ResNet-50 / caffe-keyed VGG16 `state_dict` lands in `backbone.*` (classifier tensors ignored), and an
unresolvable string WARNS instead of silently leaving random weights.  The checkpoints are synthesised here with the published key layouts
(torchvision.models.resnet50: conv1 / bn1 / layerL.B.{convK,bnK,downsample.{0,1}} / fc; mmcv VGG16 caffe: features.{idx} / classifier.{0,3,6})."""
import os
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bn(sd, name, c, g):
    sd[name + '.weight'] = torch.rand(c, generator=g) + 0.5
    sd[name + '.bias'] = torch.randn(c, generator=g) * 0.1
    sd[name + '.running_mean'] = torch.randn(c, generator=g) * 0.1
    sd[name + '.running_var'] = torch.rand(c, generator=g) + 0.5
    sd[name + '.num_batches_tracked'] = torch.tensor(0)


def torchvision_resnet50_state_dict(seed=3):
    g = torch.Generator().manual_seed(seed)
    sd = {'conv1.weight': torch.randn(64, 3, 7, 7, generator=g) * 0.05}
    _bn(sd, 'bn1', 64, g)
    inpl = 64
    for li, (planes, nb) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), 1):
        for b in range(nb):
            p = f'layer{li}.{b}'
            for k, (ci, co, ks) in enumerate(((inpl, planes, 1), (planes, planes, 3), (planes, planes * 4, 1)), 1):
                sd[f'{p}.conv{k}.weight'] = torch.randn(co, ci, ks, ks, generator=g) * 0.02
                _bn(sd, f'{p}.bn{k}', co, g)
            if b == 0:
                sd[f'{p}.downsample.0.weight'] = torch.randn(planes * 4, inpl, 1, 1, generator=g) * 0.02
                _bn(sd, f'{p}.downsample.1', planes * 4, g)
            inpl = planes * 4
    sd['fc.weight'] = torch.randn(1000, 2048, generator=g) * 0.01
    sd['fc.bias'] = torch.zeros(1000)
    return sd


def caffe_vgg16_state_dict(seed=4):
    g = torch.Generator().manual_seed(seed)
    sd, idx, inpl = {}, 0, 3
    for planes, nb in zip((64, 128, 256, 512, 512), (2, 2, 3, 3, 3)):
        for _ in range(nb):
            sd[f'features.{idx}.weight'] = torch.randn(planes, inpl, 3, 3, generator=g) * 0.02
            sd[f'features.{idx}.bias'] = torch.randn(planes, generator=g) * 0.01
            inpl, idx = planes, idx + 2
        idx += 1
    for i, (ci, co) in zip((0, 3, 6), ((64, 32), (32, 32), (32, 10))):         # (classifier shapes are irrelevant: the keys are ignored)
        sd[f'classifier.{i}.weight'] = torch.randn(co, ci, generator=g)
        sd[f'classifier.{i}.bias'] = torch.zeros(co)
    return sd


def _retina_cfg():
    from aod_meh_hua_amd.mmcv_lite import Config
    return Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))


def test_torchvision_resnet50_lands_in_backbone(tmp_path, monkeypatch, capsys):
    from aod_meh_hua_amd.models import build_detector
    sd = torchvision_resnet50_state_dict()
    torch.save(sd, tmp_path / 'resnet50-0676ba61.pth')                     # the file name mmcv's torchvision:// download would have
    monkeypatch.setenv('AOD_PRETRAINED_DIR', str(tmp_path))
    cfg = _retina_cfg()
    assert cfg.model.backbone.init_cfg.checkpoint == 'torchvision://resnet50'
    model = build_detector(cfg.model)
    before = model.neck.lateral_convs[0].conv.weight.clone()
    with warnings.catch_warnings():
        warnings.simplefilter('error')                                     # a resolvable checkpoint must not warn
        model.init_weights()
    own = model.backbone.state_dict()
    assert set(own) == {k for k in sd if not k.startswith('fc.')}           # same key layout as torchvision, minus the classifier
    for k, v in own.items():
        assert torch.equal(v, sd[k]), k
    assert torch.equal(model.state_dict()['backbone.layer1.0.conv1.weight'], sd['layer1.0.conv1.weight'])
    assert 'fc.weight' in capsys.readouterr().out                          # the ignored keys are reported, as mmcv logs them
    assert not torch.equal(model.neck.lateral_convs[0].conv.weight, before)  # the other modules still ran their own init_cfg
    # a loaded backbone keeps the pretrained BN gammas (zero_init_residual only applies without a checkpoint: resnet.py:405-423)
    assert float(model.backbone.layer2[0].bn3.weight.abs().min()) > 0


def test_torch_home_hub_checkpoints_is_searched(tmp_path, monkeypatch):
    from aod_meh_hua_amd.mmcv_lite import resolve_pretrained
    monkeypatch.delenv('AOD_PRETRAINED_DIR', raising=False)
    monkeypatch.setenv('TORCH_HOME', str(tmp_path))
    assert resolve_pretrained('torchvision://resnet50') is None
    d = tmp_path / 'hub' / 'checkpoints'
    d.mkdir(parents=True)
    (d / 'resnet50-19c8e357.pth').write_bytes(b'')
    assert resolve_pretrained('torchvision://resnet50') == str(d / 'resnet50-19c8e357.pth')
    assert resolve_pretrained('https://download.pytorch.org/models/resnet50-19c8e357.pth') == str(d / 'resnet50-19c8e357.pth')
    assert resolve_pretrained(str(d / 'resnet50-19c8e357.pth')) == str(d / 'resnet50-19c8e357.pth')
    assert resolve_pretrained('open-mmlab://vgg16_caffe') is None


def test_caffe_vgg16_lands_in_ssd_backbone(tmp_path, monkeypatch):
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    sd = caffe_vgg16_state_dict()
    torch.save(dict(state_dict=sd, meta={}), tmp_path / 'vgg16_caffe-292e1171.pth')
    monkeypatch.setenv('AOD_PRETRAINED_DIR', str(tmp_path))
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_SSD.py'))
    assert cfg.model.backbone.init_cfg.checkpoint == 'open-mmlab://vgg16_caffe'
    model = build_detector(cfg.model)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        model.init_weights()
    own = model.backbone.state_dict()
    loaded = [k for k in sd if k.startswith('features.')]
    assert len(loaded) == 26 and all(torch.equal(own[k], sd[k]) for k in loaded)
    # fc6 / fc7 (features.31 / .33) are not in the caffe file: they keep the constructor's initialisation and are finite
    assert set(own) - set(sd) == {'features.31.weight', 'features.31.bias', 'features.33.weight', 'features.33.bias'}
    assert all(bool(torch.isfinite(own[k]).all()) for k in set(own) - set(sd))


def test_missing_pretrained_file_warns_and_names_the_string(tmp_path, monkeypatch):
    from aod_meh_hua_amd import mmcv_lite
    from aod_meh_hua_amd.models import build_detector
    monkeypatch.setenv('AOD_PRETRAINED_DIR', str(tmp_path))                # empty
    monkeypatch.setenv('TORCH_HOME', str(tmp_path / 'nothing'))
    mmcv_lite._pretrained_warned.clear()
    model = build_detector(_retina_cfg().model)
    with pytest.warns(RuntimeWarning, match=r"torchvision://resnet50.*RANDOM"):
        model.init_weights()
    assert bool(torch.isfinite(model.backbone.conv1.weight).all())


def test_a_file_of_foreign_keys_is_an_error_not_a_partial_load(tmp_path, monkeypatch):
    from aod_meh_hua_amd.models import build_detector
    torch.save({'encoder.w': torch.zeros(3)}, tmp_path / 'resnet50.pth')
    monkeypatch.setenv('AOD_PRETRAINED_DIR', str(tmp_path))
    model = build_detector(_retina_cfg().model)
    with pytest.raises(RuntimeError, match='none of its 1 keys'):
        model.init_weights()


def test_deprecated_pretrained_kwarg_becomes_init_cfg():
    from aod_meh_hua_amd.models.backbones.resnet import ResNet
    r = ResNet(50, pretrained='torchvision://resnet50')
    assert r.init_cfg == dict(type='Pretrained', checkpoint='torchvision://resnet50') and not r.zero_init_residual
    with pytest.raises(AssertionError):
        ResNet(50, pretrained='x', init_cfg=dict(type='Kaiming', layer='Conv2d'))

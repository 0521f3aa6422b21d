"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/aod_hip.h declares, the product
refuses CPU tensors, and the host-side mirror of the reference interface (registry / config / plugins / AL logic)
behaves like the reference's."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='module')
def lib():
    from aod_meh_hua_amd.build import build
    return ctypes.CDLL(build(verbose=False))


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, 'include', 'aod_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = set(re.findall(r'\b(aod_[a-z0-9_]+)\s*\(', hdr))
    assert len(names) >= 30
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    lib.aod_last_error.restype = ctypes.c_char_p
    assert lib.aod_version() >= 1 and lib.aod_last_error() is not None


def test_bad_arguments_are_rejected_without_a_gpu(lib):
    """Argument validation happens before any launch: error code -1 + message, no crash."""
    from aod_meh_hua_amd._C import ConvDesc
    d = ConvDesc()
    d.C, d.N, d.R, d.S, d.stride, d.pad, d.dil, d.nseg = 12, 64, 3, 3, 1, 1, 1, 1      # C not a multiple of 8
    one = ctypes.c_void_p(16)
    lib.aod_conv2d.restype = ctypes.c_int
    rc = lib.aod_conv2d(ctypes.byref(d), one, one, one, None, None, None, None, None, None, None, None)
    assert rc == -1
    lib.aod_last_error.restype = ctypes.c_char_p
    assert b'multiple of 8' in lib.aod_last_error()
    lib.aod_loss_partials_len.restype = ctypes.c_size_t
    assert lib.aod_loss_partials_len(ctypes.c_int64(1000)) == 48      # 3 per block of 64 rows (the 4-lanes-per-row form for > 24 classes)
    # the level-fused loss entry points: 1 .. 8 levels, and their workspace is the per-level sum (a level's last block is its own)
    rows = (ctypes.c_int64 * 3)(1000, 10, 0)
    lib.aod_loss_levels_partials_len.restype = ctypes.c_size_t
    assert lib.aod_loss_levels_partials_len(3, rows) == 48 + 3 + 0
    one = ctypes.c_void_p(16)
    for nlev in (0, 9):
        rc = lib.aod_edl_focal_l1_levels_fwd(one, one, one, None, None, None, nlev, rows, 20, ctypes.c_float(2.0), ctypes.c_float(0.25), one, one, one,
                                             None, 0, None, None, None)
        assert rc == -1 and b'1..8 levels' in lib.aod_last_error()
        rc = lib.aod_meh_loss_levels_bwd(one, one, one, nlev, rows, one, one, 0, 9, 9, None)
        assert rc == -1 and b'1..8 levels' in lib.aod_last_error()
    # (num_pos without a divisor buffer)
    rc = lib.aod_edl_focal_l1_levels_fwd(one, one, one, None, None, None, 3, rows, 20, ctypes.c_float(2.0), ctypes.c_float(0.25), one, one, one,
                                         one, 16, None, None, None)
    assert rc == -1 and b'divisor' in lib.aod_last_error()
    # the x3 halo kernel says no to anything but a narrow fp32-destination 3 x 3 / stride-1 forward conv of the reference-precision mode
    from aod_meh_hua_amd._C import ConvDesc
    dd = ConvDesc()
    assert lib.aod_halo_conv3x3_x3_applies(ctypes.byref(dd)) == 0


def test_product_refuses_cpu_tensors():
    from aod_meh_hua_amd import hipops as ho
    from aod_meh_hua_amd._C import AodHipError
    with pytest.raises(AodHipError):
        ho.add_relu(torch.zeros(8, dtype=torch.bfloat16), torch.zeros(8, dtype=torch.bfloat16))
    with pytest.raises(AodHipError):
        ho.edl_focal_l1_fwd(torch.zeros(4, 20), torch.zeros(4, dtype=torch.long), torch.ones(4))


def test_product_does_not_import_the_oracle():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import aod_meh_hua_amd.models, aod_meh_hua_amd.scoring, aod_meh_hua_amd.apis; "
            "assert not [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]") % ROOT
    subprocess.check_call([sys.executable, '-c', code])
    for dp, _, fs in os.walk(os.path.join(ROOT, 'aod_meh_hua_amd')):
        for f in fs:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_registry_and_config_build_the_reference_types():
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    assert cfg.X_L_0_size == 16551 // 20 and cfg.model.bbox_head.loss_cls.last_activation == 'relu'
    cfg.model.backbone.pop('init_cfg')
    model = build_detector(cfg.model)
    assert type(model).__name__ == 'SSL_L_RetinaNet' and type(model.bbox_head).__name__ == 'Lambda_L2Net'
    g = np.load(os.path.join(G, 'state_dict_spec.npz'))
    assert list(model.state_dict().keys()) == list(g['keys'])
    assert sum(p.numel() for p in model.parameters()) == int(g['n_params'])
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == int(g['n_trainable'])
    # frozen stem + layer1, BN kept in eval mode by train() (resnet.py:612-656)
    model.train()
    assert not model.backbone.conv1.weight.requires_grad and not model.backbone.layer1[0].conv1.weight.requires_grad
    assert model.backbone.layer2[0].bn1.weight.requires_grad and not model.backbone.layer2[0].bn1.training
    # init_cfg semantics: head Normal(0.01) + retina_cls bias_prob=0.01; FPN Xavier uniform
    model.init_weights()
    h = model.bbox_head
    assert abs(float(h.retina_cls.bias[0]) + np.log(99)) < 1e-5 and float(h.retina_reg.bias.abs().max()) == 0
    assert 0.008 < float(h.cls_convs[0].conv.weight.std()) < 0.012
    w = model.neck.lateral_convs[0].conv.weight
    assert float(w.abs().max()) <= np.sqrt(6.0 / (512 + 256)) + 1e-6
    # the driver touches these attributes (train_RetinaNet.py:157-176, train_Lambda.py:55-61)
    assert h.num_anchors == 9 and h.L_names == ['retina_L', 'L_convs'] and len(h.cls_convs) == 4
    with pytest.raises(KeyError):
        build_detector(dict(type='NoSuchDetector'))


def test_anchor_generator_matches_reference_golden():
    from aod_meh_hua_amd.core.anchor import AnchorGenerator, SSDAnchorGenerator
    g = np.load(os.path.join(G, 'anchors.npz'))
    ag = AnchorGenerator(octave_base_scale=4, scales_per_octave=3, ratios=[0.5, 1.0, 2.0], strides=[8, 16, 32, 64, 128])
    assert np.array_equal(np.stack([b.numpy() for b in ag.base_anchors]), g['base'])
    small = ag.grid_anchors([(16, 12), (8, 6), (4, 3), (2, 2), (1, 1)], 'cpu')
    assert np.array_equal(torch.cat(small).numpy(), g['grid_small'])
    f = ag.valid_flags([(16, 12), (8, 6), (4, 3), (2, 2), (1, 1)], (120, 96, 3), 'cpu')
    assert np.array_equal(torch.cat(f).numpy(), g['flags_small'])
    assert ag.grid_anchors([(16, 12), (8, 6), (4, 3), (2, 2), (1, 1)], 'cpu')[0] is small[0]      # cached
    ssd = SSDAnchorGenerator(strides=[8, 16, 32, 64, 100, 300], ratios=[[2], [2, 3], [2, 3], [2, 3], [2], [2]], basesize_ratio_range=(0.2, 0.9),
                             input_size=300, scale_major=False)
    assert ssd.base_sizes == [30, 60, 111, 162, 213, 264] and ssd.num_base_anchors == [4, 6, 6, 6, 4, 4]
    assert sum(a.shape[0] for a in ssd.grid_anchors([(38, 38), (19, 19), (10, 10), (5, 5), (3, 3), (1, 1)], 'cpu')) == 8732


def test_coder_and_iou_api_match_golden():
    from aod_meh_hua_amd.core.bbox import BboxOverlaps2D, DeltaXYWHBBoxCoder
    g = np.load(os.path.join(G, 'coder.npz'))
    c = DeltaXYWHBBoxCoder()
    rois, gts, d = (torch.from_numpy(g[k]) for k in ('rois', 'gts', 'deltas'))
    assert np.array_equal(c.encode(rois, gts).numpy(), g['enc'])
    assert np.array_equal(c.decode(rois, d, max_shape=(128, 160, 3)).numpy(), g['dec'])
    iou = BboxOverlaps2D()(torch.tensor([[0., 0., 10., 10., 0.9]]), torch.tensor([[0., 0., 10., 5.]]))
    assert float(iou) == 0.5


def test_selection_rule_matches_reference():
    from aod_meh_hua_amd.utils.active_datasets import update_X_L
    g = np.load(os.path.join(G, 'selection.npz'))
    np.random.seed(20)
    a, b = update_X_L(g['unc'].copy(), np.arange(400), g['X_L'].copy(), 20, zeroRate=0.15)
    assert np.array_equal(a, g['XL_zero']) and np.array_equal(b, g['XU_zero'])
    np.random.seed(20)
    a, b = update_X_L(torch.from_numpy(g['unc'].copy()), np.arange(400), g['X_L'].copy(), 20)
    assert np.array_equal(a, g['XL_plain']) and np.array_equal(b, g['XU_plain'])


def test_parse_losses_sums_every_loss_key():
    from aod_meh_hua_amd.models.detectors.SSL_Lambda import SSLBase_L_Detector

    class D(SSLBase_L_Detector):
        def extract_feat(self, x): ...
        def simple_test(self, *a, **k): ...
    losses = dict(loss_cls=[torch.tensor(1.0), torch.tensor(2.0)], loss_bbox=[torch.tensor(0.5)], loss_noR=[torch.tensor([1.0, 3.0])],
                  acc=torch.tensor(9.0))
    loss, lv = D()._parse_losses(losses, device='cpu')
    assert float(loss) == 1 + 2 + 0.5 + 2.0 and float(lv['acc']) == 9.0 and float(lv['loss_noR']) == 2.0


def test_agg_codes_and_functions():
    from aod_meh_hua_amd.scoring import extract_agg_codes
    from aod_meh_hua_amd.utils.functions import ExtractAggFunc, StartEnd
    assert extract_agg_codes('objectSum_scaleMax_classSum') == (0, 2, 0)
    assert extract_agg_codes('objectAvg_scaleSum_classMax') == (2, 0, 1)
    f = ExtractAggFunc('objectSum_scaleMax_classAvg')
    assert f['object'] is torch.sum and f['scale'] is torch.max and f['class'] is torch.mean
    assert StartEnd([torch.zeros(2, 1000, 20), torch.zeros(2, 576, 20), torch.zeros(2, 144, 20)], 1) == (1000, 1576)


def test_synthetic_dataset_and_loader_contract():
    from aod_meh_hua_amd.datasets import build_dataloader, build_dataset
    from aod_meh_hua_amd.mmcv_lite import scatter_kwargs
    ds = build_dataset(dict(type='RepeatDataset', times=2, dataset=dict(type='SyntheticVOCDataset', num_images=5, size=(64, 64))))
    assert len(ds) == 10 and len(ds.CLASSES) == 20
    a, b = ds[3], ds[8]
    assert torch.equal(a['img'], b['img'])                       # deterministic per index
    dl = build_dataloader(ds.dataset, 2, 0, dist=False, shuffle=False)
    batch = next(iter(dl))
    data = scatter_kwargs(batch, 'cpu')
    assert data['img'].shape == (2, 3, 64, 64) and len(data['img_metas']) == 2 and data['img_metas'][0]['pad_shape'] == (64, 64, 3)
    assert len(data['gt_bboxes']) == 2 and data['gt_bboxes'][0].shape[1] == 4 and data['gt_labels'][0].dtype == torch.int64
    with pytest.raises(FileNotFoundError):                         # the real VOC path is built (tests/test_voc_data.py); no data here
        build_dataset(dict(type='VOCDataset', ann_file='/nonexistent/ImageSets/Main/x.txt', img_prefix='/nonexistent/VOC2007/', pipeline=[]))


def test_ssd512_config_builds_the_seven_level_model():
    """configs/ssd/ssd512_voc.py (reference: configs/ssd/ssd512_voc.py): 7 pyramid levels, strides 8 ... 512, 24 564 anchors per image."""
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/ssd/ssd512_voc.py'))
    assert cfg.input_size == 512 and cfg.uncertainty_pool2 == 'objectSum_scaleAvg_classSum' and cfg.model.type == 'SSD_L_SingleStageDetector'
    cfg.model.backbone.pop('init_cfg', None)
    m = build_detector(cfg.model)
    ag = m.bbox_head.anchor_generator
    assert ag.num_levels == 7 and list(ag.num_base_anchors) == [4, 6, 6, 6, 6, 4, 4]
    sizes = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]
    assert sum(a.shape[0] for a in ag.grid_anchors(sizes, 'cpu')) == 24564
    assert ag.base_sizes == [20, 51, 133, 215, 296, 378, 460]          # int(512 * {4, 10, 26, 42, 58, 74, 90} / 100)
    assert len(m.neck.extra_layers) == 5 and m.neck.extra_layers[-1][1].conv.kernel_size == (4, 4)
    assert len(m.bbox_head.cls_convs) == 7


def test_wgrad_group_plan_is_host_logic_with_the_documented_properties(lib, bf16_mode):
    """aod_conv2d_wgrad_group_plan (host code of the C ABI, no launch): members of a group get FEWER pixel splits than alone, every member of a
    group runs the same number of 64-pixel steps per workgroup, the grid fits the chip's slots, mixed tile forms are refused (return 1), and a
    single member fills the slots too."""
    from aod_meh_hua_amd.hipops import Seg, make_desc, out_segs
    lib.aod_conv2d_wgrad_group_plan.restype = ctypes.c_int
    lib.aod_conv2d_wgrad_splits.restype = ctypes.c_int

    def desc(cin, cout, k, B, H, W):
        src = [Seg(B, H, W, 0)]
        return make_desc(cin, cout, k, k, 1, k // 2, 1, src, out_segs(src, k, k, 1, k // 2, 1), False, False, False)

    def plan(ds):
        n = len(ds)
        out = (ctypes.c_int32 * n)()
        rc = lib.aod_conv2d_wgrad_group_plan((ctypes.c_void_p * n)(*[ctypes.addressof(d) for d in ds]), n, out)
        return rc, list(out)

    # the three convs of a layer-2 identity bottleneck at 16 x 64 x 64 (128-tile form): 4 + 9 + 4 tiles
    blk = [desc(128, 512, 1, 16, 64, 64), desc(128, 128, 3, 16, 64, 64), desc(512, 128, 1, 16, 64, 64)]
    alone = [lib.aod_conv2d_wgrad_splits(ctypes.byref(d)) for d in blk]
    rc, grp = plan(blk)
    assert rc == 0 and all(g < a for g, a in zip(grp, alone)), (alone, grp)
    assert len(set(grp)) == 1                                   # same pixel count -> same number of steps -> same splits
    assert (4 + 9 + 4) * grp[0] <= 512 < (4 + 9 + 4) * (grp[0] + 2)
    # one member through the group planner: it fills the slots (the single launches keep their own cost model: aod_conv2d_wgrad_splits)
    rc, one = plan(blk[:1])
    assert rc == 0 and 4 * one[0] <= 512 < 4 * (one[0] + 2) and alone[0] <= one[0]
    # a head-tower conv (256 x 2304 over 87 296 pixels: the 256 x 256 tile) cannot share a grid with a 128-channel layer
    tower = desc(256, 256, 3, 16, 64, 64)
    rc, _ = plan([tower, blk[1]])
    assert rc == 1
    # four towers: 4 x 9 big tiles, the splits fill one round of the 256 CUs
    rc, tw = plan([tower] * 4)
    assert rc == 0 and 36 * tw[0] <= 256 < 36 * (tw[0] + 1)
    # argument errors
    rc, _ = plan([tower] * 5) if False else (lib.aod_conv2d_wgrad_group_plan(None, 1, None), None)
    assert rc == -1


def test_head_tail_rounding_is_monotone_and_idempotent():
    """X-layout rounding v -> bf16(v) + bf16(v - bf16(v)) (csrc/x3_ops.hip xstore / xload): the fused reference-precision stem
    (csrc/stem_x3.hip) takes the max-pool BEFORE this rounding where the three-launch path rounds first -- equal because the map is
    non-decreasing -- and re-rounding a rounded value changes nothing.  Dense sweeps of consecutive fp32 values across bf16 boundaries."""
    import numpy as np
    import torch

    def f(u):
        h = u.bfloat16().float()
        return h + (u - h).bfloat16().float()
    for base in (1.0, 3.14159, 0.007, 1234.5, -2.5):
        b = int(np.float32(base).view(np.uint32))
        bits = np.arange(b - (1 << 17), b + (1 << 17), dtype=np.uint32)
        u = torch.from_numpy(bits.view(np.float32).copy())
        u, _ = u.sort()
        v = f(u)
        assert bool((v[1:] >= v[:-1]).all())
        assert bool((f(v) == v).all())
        assert float(((v - u).abs() / u.abs()).max()) < 2 ** -16


def test_dense_concat_recognises_adjacent_pyramid_levels_only():
    """functional.dense_concat: the level-fused loss launches take the pyramid levels as ONE row range -- legal only when each level starts
    where the previous one ends in the same buffer."""
    import torch
    from aod_meh_hua_amd import functional as AF
    buf = torch.arange(40 * 6, dtype=torch.float32).view(40, 6)
    lv = [buf[0:16], buf[16:16], buf[16:28], buf[28:40]]                      # (an empty level in the middle)
    d = AF.dense_concat(lv)
    assert d is not None and d.shape == (40, 6) and d.data_ptr() == buf.data_ptr() and torch.equal(d, buf)
    d = AF.dense_concat([buf[4:16], buf[16:28]])                               # does not start at the buffer's first row
    assert d.shape == (24, 6) and torch.equal(d, buf[4:28])
    assert AF.dense_concat([buf[0:16], buf[20:28]]) is None                    # a gap
    assert AF.dense_concat([buf[16:28], buf[0:16]]) is None                    # wrong order
    assert AF.dense_concat([buf[0:16], buf[16:28].clone()]) is None            # another buffer
    assert AF.dense_concat([buf[0:16], buf[16:28].double()]) is None           # another dtype
    assert AF.dense_concat([buf[0:16, :3], buf[16:28, :3]]) is None            # not contiguous
    flat = torch.arange(30)
    assert torch.equal(AF.dense_concat([flat[0:10], flat[10:30]]), flat)


def test_parse_losses_sums_a_shared_term_matrix_once():
    """SSL_Lambda._parse_losses (SSL_Lambda.py:126-154) on functional.PackedLosses that are the rows of ONE [3, L] matrix (what the
    level-fused loss launch returns): log_vars = the row sums, loss = their sum, gradient of every entry = 1; a loss outside the group or a
    non-loss key falls back to the per-name sums."""
    import torch
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd.models.detectors.SSL_Lambda import SSLBase_L_Detector
    parse = SSLBase_L_Detector._parse_losses
    Q = torch.rand(3, 5, dtype=torch.float64, requires_grad=True)
    mk = lambda: dict(loss_cls=AF.PackedLosses(Q[0].unbind(0), Q[0], group=(Q, 0)), loss_bbox=AF.PackedLosses(Q[1].unbind(0), Q[1], group=(Q, 1)),
                      loss_noR=AF.PackedLosses([torch.zeros(7)] * 5, Q[2], group=(Q, 2)))
    loss, lv = parse(None, mk())
    assert list(lv) == ['loss_cls', 'loss_bbox', 'loss_noR']
    assert torch.allclose(loss, Q.sum()) and all(torch.allclose(lv[k], Q[i].sum()) for i, k in enumerate(lv))
    loss.backward()
    assert torch.equal(Q.grad, torch.ones_like(Q))
    assert not any(v.requires_grad for v in lv.values())
    # a fourth loss outside the matrix, and an 'acc' entry that is not a loss term
    Q.grad = None
    extra = torch.tensor(2.5, dtype=torch.float64, requires_grad=True)
    losses = mk()
    losses['loss_L'] = [extra * 1.0, extra * 2.0]
    losses['acc'] = torch.tensor([1.0, 3.0], dtype=torch.float64)
    loss, lv = parse(None, losses)
    assert torch.allclose(loss, Q.sum() + 3.0 * extra) and float(lv['acc']) == 2.0
    loss.backward()
    assert torch.equal(Q.grad, torch.ones_like(Q)) and float(extra.grad) == 3.0
    # ungrouped PackedLosses keep the per-name sum
    P = torch.rand(4, dtype=torch.float64, requires_grad=True)
    loss, lv = parse(None, dict(loss_L=AF.PackedLosses(P.unbind(0), P)))
    assert torch.allclose(loss, P.sum())

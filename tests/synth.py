"""Seeded synthetic inputs shared by tools/golden/make_golden.py (which feeds them to the
reference) and by the parity tests (which feed them to the oracle / the HIP path).
Pure data recipes: CPU torch.Generator streams, identical on every box with this image."""
import numpy as np
import torch


def gen(seed):
    return torch.Generator().manual_seed(seed)


def metas(B, H, W, scale=1.0):
    return [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), ori_shape=(H, W, 3),
                 scale_factor=np.full(4, scale, np.float32), flip=False, flip_direction=None,
                 filename=f'synth_{i}.jpg', ori_filename=f'synth_{i}.jpg') for i in range(B)]


def images(B, H, W, seed=20):
    return torch.randn(B, 3, H, W, generator=gen(seed))


def random_gts(B, H, W, seed=20, gmin=1, gmax=5, num_classes=20):
    """SURVEY 8d C1 recipe: G~U{gmin..gmax}, w,h ~ U(H/16, 3H/4), clipped inside the image."""
    g = gen(seed)
    boxes, labels = [], []
    for _ in range(B):
        G = int(torch.randint(gmin, gmax + 1, (1,), generator=g))
        wh = torch.rand(G, 2, generator=g) * torch.tensor([W * 0.6875, H * 0.6875]) + torch.tensor([W / 16., H / 16.])
        xy = torch.rand(G, 2, generator=g) * (torch.tensor([float(W), float(H)]) - wh)
        boxes.append(torch.cat([xy, xy + wh], dim=1))
        labels.append(torch.randint(0, num_classes, (G,), generator=g))
    return boxes, labels


def assign_cases(H=128, W=128):
    """Three images for the assigner: G=0; a gt whose best IoU < 0.5 (tiny box) next to
    a normal one; duplicate gts (tie rule: later gt wins) plus an anchor-aligned gt."""
    b0 = torch.zeros(0, 4)
    l0 = torch.zeros(0, dtype=torch.long)
    b1 = torch.tensor([[3., 5., 9., 12.], [20., 30., 90., 100.]])
    l1 = torch.tensor([4, 11])
    # an exact level-0 anchor (stride 8, cell (2,3), ratio 1 scale 4: 32x32 centred at (24,16)) twice + shifted
    b2 = torch.tensor([[8., 0., 40., 32.], [8., 0., 40., 32.], [40., 40., 104., 120.]])
    l2 = torch.tensor([2, 7, 19])
    return [b0, b1, b2], [l0, l1, l2]


def loss_inputs(N=1024, C=20, seed=21):
    g = gen(seed)
    logits = torch.randn(N, C, generator=g) * 2.0
    labels = torch.randint(0, C + 1, (N,), generator=g)          # C == background
    labels[torch.rand(N, generator=g) < 0.6] = C
    lw = (torch.rand(N, generator=g) > 0.1).float()
    bpred = torch.randn(N, 4, generator=g) * 0.5
    btgt = torch.randn(N, 4, generator=g) * 0.5
    bw = ((labels < C).float() * lw)[:, None].expand(N, 4).contiguous()
    lam = torch.rand(N, generator=g) * 0.3
    return dict(logits=logits, labels=labels, label_weights=lw, bbox_pred=bpred, bbox_targets=btgt,
                bbox_weights=bw, lam=lam, num_total_samples=max(int((labels < C).sum()), 1))


def planted_heads(B=2, H=128, W=128, C=20, A=9, seed=22, n_plant=6, plant_small=True):
    """Planted-logit head outputs (SURVEY 8c 'scoring'): cls = 0.5*N(0,1) with +8.0 on a few
    (anchor, class) 3x3 patches so that some anchors pass 0.3 / NMS finds objects;
    reg = 0.1*N(0,1); L = U(.01,.31).  Returns NCHW lists like the head does.  plant_small=False: nothing is planted on levels smaller
    than 4 x 4 (with n_plant=0 the image then has no confident anchor at all and scores exactly 0)."""
    g = gen(seed)
    cls, reg, Ls = [], [], []
    for s in (8, 16, 32, 64, 128):
        h, w = max(H // s, 1), max(W // s, 1)
        c = 0.5 * torch.randn(B, A * C, h, w, generator=g)
        for b in range(B):
            for _ in range(n_plant if h >= 4 else (1 if (h >= 2 and plant_small) else 0)):
                a = int(torch.randint(0, A, (1,), generator=g))
                k = int(torch.randint(0, C, (1,), generator=g))
                y = int(torch.randint(0, h, (1,), generator=g))
                x = int(torch.randint(0, w, (1,), generator=g))
                c[b, a * C + k, max(y - 1, 0):y + 2, max(x - 1, 0):x + 2] += 8.0
        cls.append(c)
        reg.append(0.1 * torch.randn(B, A * 4, h, w, generator=g))
        Ls.append(torch.rand(B, A, h, w, generator=g) * 0.3 + 0.01)
    return cls, reg, Ls


SSD_SIZES = (38, 19, 10, 5, 3, 1)
SSD_ANCHORS = (4, 6, 6, 6, 4, 4)


def planted_heads_ssd(B=2, C1=21, seed=23, n_plant=5, sizes=None, anchors=None):
    """SSD300 head outputs with planted foreground logits: cls = 0.5*N(0,1) over C1 = 21 logits (background last, +2.0 so most
    anchors are background) with +9.0 on a few (anchor, class) 3x3 patches; reg = 0.3*N(0,1); L = U(.01,.31)."""
    g = gen(seed)
    cls, reg, Ls = [], [], []
    for h, A in zip(sizes or SSD_SIZES, anchors or SSD_ANCHORS):
        c = 0.5 * torch.randn(B, A * C1, h, h, generator=g)
        c.view(B, A, C1, h, h)[:, :, C1 - 1] += 2.0
        for b in range(B):
            for _ in range(n_plant if h >= 5 else 1):
                a = int(torch.randint(0, A, (1,), generator=g))
                k = int(torch.randint(0, C1 - 1, (1,), generator=g))
                y = int(torch.randint(0, h, (1,), generator=g))
                x = int(torch.randint(0, h, (1,), generator=g))
                c[b, a * C1 + k, max(y - 1, 0):y + 2, max(x - 1, 0):x + 2] += 9.0
        cls.append(c)
        reg.append(0.3 * torch.randn(B, A * 4, h, h, generator=g))
        Ls.append(torch.rand(B, A, h, h, generator=g) * 0.3 + 0.01)
    return cls, reg, Ls


def detection_eval_case(seed=50, n_img=24, num_classes=20, with_ignore=True):
    """Seeded detections + annotations in eval_map's format (numpy): per image 0-4 gts, detections = jittered copies of most gts
    (some duplicated) + random false positives, a few ignored gts; scores in (0, 1)."""
    import numpy as np
    r = np.random.RandomState(seed)
    dets, anns = [], []
    for i in range(n_img):
        G = r.randint(0, 5)
        wh = r.uniform(20, 200, (G, 2))
        xy = r.uniform(0, 300, (G, 2))
        gtb = np.concatenate([xy, xy + wh], 1).astype(np.float32)
        gtl = r.randint(0, num_classes, G).astype(np.int64)
        ign = np.zeros(G, bool)
        if with_ignore and G and i % 5 == 0:
            ign[r.randint(0, G)] = True
        per_cls = [[] for _ in range(num_classes)]
        for b, l in zip(gtb, gtl):
            for _ in range(r.randint(0, 3)):
                j = b + r.normal(0, 8, 4).astype(np.float32)
                per_cls[l].append(np.concatenate([j, [r.uniform(0.05, 1.0)]]).astype(np.float32))
        for _ in range(r.randint(0, 6)):
            l = r.randint(0, num_classes)
            xy2, wh2 = r.uniform(0, 300, 2), r.uniform(20, 200, 2)
            per_cls[l].append(np.concatenate([xy2, xy2 + wh2, [r.uniform(0.05, 0.9)]]).astype(np.float32))
        dets.append([np.stack(c).astype(np.float32) if c else np.zeros((0, 5), np.float32) for c in per_cls])
        ann = dict(bboxes=gtb[~ign], labels=gtl[~ign])
        if with_ignore:
            ann.update(bboxes_ignore=gtb[ign], labels_ignore=gtl[ign])
        anns.append(ann)
    return dets, anns


# ---------------------------------------------------------------- tiny VOC tree (real data path: tests/test_voc_data.py, tools/golden/make_golden_data.py)
TINY_VOC = [('000001', 500, 375, [('dog', 0, (48, 240, 195, 371)), ('person', 0, (8, 12, 352, 498 - 200)), ('cat', 1, (100, 100, 200, 200))]),
            ('000002', 333, 500, [('car', 0, (10.6, 20, 300, 480))]),
            ('000003', 480, 360, [('unicorn', 0, (1, 1, 50, 50))]),                     # no VOC class -> filtered in train mode
            ('000004', 20, 300, [('bird', 0, (1, 1, 15, 100))]),                        # too small (min side < 32)
            ('000005', 400, 300, [('sofa', 0, (30, 40, 200, 220)), ('chair', 0, (5, 5, 40, 60))])]


def voc_xml(w, h, objs, with_size=True):
    o = ''.join(f'<object><name>{n}</name><difficult>{d}</difficult><bndbox><xmin>{b[0]}</xmin><ymin>{b[1]}</ymin><xmax>{b[2]}</xmax>'
                f'<ymax>{b[3]}</ymax></bndbox></object>' for n, d, b in objs)
    size = f'<size><width>{w}</width><height>{h}</height><depth>3</depth></size>' if with_size else ''
    return f'<annotation>{size}{o}</annotation>'


def write_tiny_voc(root):
    """VOC2007-shaped tree under `root` (a directory that ends in 'VOC2007'): five images (random pixels, seeded), annotations with a
    difficult object, a float coordinate, a non-VOC class, a too-small image and one XML without a <size> element.  Returns root + '/'."""
    import os

    from PIL import Image
    root = str(root)
    for d in ('JPEGImages', 'Annotations', 'ImageSets/Main'):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    rng = np.random.RandomState(0)
    for i, (iid, w, h, objs) in enumerate(TINY_VOC):
        Image.fromarray(rng.randint(0, 255, (h, w, 3), dtype=np.uint8)).save(os.path.join(root, 'JPEGImages', f'{iid}.jpg'), quality=95)
        with open(os.path.join(root, 'Annotations', f'{iid}.xml'), 'w') as f:
            f.write(voc_xml(w, h, objs, with_size=i != 4))
    with open(os.path.join(root, 'ImageSets/Main/trainval.txt'), 'w') as f:
        f.write('\n'.join(i[0] for i in TINY_VOC) + '\n')
    return root + '/'

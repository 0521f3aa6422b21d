"""Per-parameter gradient error of the reference-precision mode (bf16x3) against the CPU oracle run in fp64, with the oracle's own fp32 run
beside it.  For every trainable tensor of RetinaNet-R50 + MEH at B x H x W:

    e_x3   = |g_hip - g_f64| / |g_f64|        (the product's error)
    e_mask = |g_hip - g_f64m| / |g_f64m|      (g_f64m: the fp64 oracle evaluated with the HIP run's ReLU sign pattern, oracle.model.relu_masks:
                                               what is left of e_x3 once both backward passes walk the same piecewise-linear branch)
    e_f32  = |g_f32 - g_f64| / |g_f64|        (what an fp32 implementation with another summation order shows: the noise floor)
    norm   = |g_hip| / |g_f64| - 1
    flips  = ReLU elements whose sign differs between the HIP run and the fp64 oracle (per site, summed)

    gpurun -- 'python tools/dbg/x3_grad_table.py [B H [out.json]]'      (default 2 128 -> gpurun_out/x3_grad_table_2x128.json)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model as omodel          # noqa: E402   (debug tooling: the checker, never the product)
from tests import synth                      # noqa: E402


def oracle_grads(sd0, img, gtb, gtl, dtype, masks=None):
    if masks is not None:
        with omodel.relu_masks(masks):
            return oracle_grads(sd0, img, gtb, gtl, dtype)
    sd = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    for k, v in sd.items():
        if v.is_floating_point() and not any(s in k for s in ('running', 'backbone.conv1.', 'backbone.bn1.', 'layer1.')):
            v.requires_grad_(True)
    o = omodel.train_step(sd, img.to(dtype), gtb, gtl)          # (targets stay fp32 like the reference's: exact in either run)
    o['loss'].backward()
    g = {k: v.grad.detach().double() for k, v in sd.items() if v.is_floating_point() and v.grad is not None}
    oL = omodel.train_step_L(sd, o['feats'], o['loss_noR'], o['targets'])
    for v in sd.values():
        if v.is_floating_point():
            v.grad = None
    oL['loss'].backward()
    gL = {k: v.grad.detach().double() for k, v in sd.items() if v.is_floating_point() and v.grad is not None}
    return float(o['loss']), g, gL


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    H = W = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, 'gpurun_out', f'x3_grad_table_{B}x{H}.json')
    from aod_meh_hua_amd import functional as AF
    from aod_meh_hua_amd.mmcv_lite import Config
    from aod_meh_hua_amd.models import build_detector
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/_base_/Config_RetinaNet.py'))
    cfg.model.backbone.pop('init_cfg')
    model = build_detector(cfg.model)
    sd0 = omodel.seeded_state_dict()
    model.load_state_dict(sd0, strict=True)
    model = model.cuda().train()
    gtb, gtl = synth.random_gts(B, H, W, seed=24, gmin=1, gmax=3)
    img = synth.images(B, H, W)
    data = dict(img=img.cuda(), img_metas=synth.metas(B, H, W), gt_bboxes=[b.cuda() for b in gtb], gt_labels=[l.cuda() for l in gtl])
    AF.set_precision(os.environ.get('PREC', 'bf16x3'))
    from tests.maskcap import capture_relu_masks
    with capture_relu_masks(model) as masks:
        outp, head_out, feat_out, prev = model.train_step(data, Labeled=True, Pseudo=False)
    model.zero_grad()
    outp['loss'].backward()
    pd = dict(model.named_parameters())
    grads = {k: pd[k].grad.detach().double().cpu().clone() for k in pd if pd[k].grad is not None}
    with capture_relu_masks(model, masks):
        lossL = model.train_step_L(prev, head_out, feat_out)
    model.zero_grad()
    lossL['loss'].backward()
    gradsL = {k: pd[k].grad.detach().double().cpu().clone() for k in pd if pd[k].grad is not None}
    torch.cuda.synchronize()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    l64, g64, gL64 = oracle_grads(sd0, img, gtb, gtl, torch.float64)
    l32, g32, gL32 = oracle_grads(sd0, img, gtb, gtl, torch.float32)
    lm, gm64, gLm64 = oracle_grads(sd0, img, gtb, gtl, torch.float64, masks=masks)
    # sign disagreements per site: the fp64 oracle's own activations against the HIP masks
    seen = {}

    orig_relu = omodel._relu

    def spy(z, key):
        m = masks.get(key)
        if m is not None:
            seen[key] = (int(((z.detach() > 0) != m).sum()), m.numel())
        return orig_relu(z, key)
    omodel._relu = spy
    try:
        with torch.no_grad():
            sdd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd0.items()}
            feats = omodel.fpn(sdd, omodel.backbone(sdd, img.double()))
            omodel.head_forward(sdd, feats), omodel.head_forward_L(sdd, feats)
    finally:
        omodel._relu = orig_relu
    flips, sites = sum(a for a, _ in seen.values()), sum(b for _, b in seen.values())
    rows = []
    for tag, gh, gf, gd, gmk in (('main', grads, g32, g64, gm64), ('meh', gradsL, gL32, gL64, gLm64)):
        for k, ref in gd.items():
            if k not in gh:
                continue
            n = float(ref.norm())
            if n == 0:
                continue
            a, b = gh[k].flatten(), ref.flatten()
            rows.append(dict(name=k, step=tag, numel=ref.numel(), norm=n, e_x3=float((a - b).norm() / n),
                             e_mask=float((a - gmk[k].flatten()).norm() / gmk[k].norm()),
                             e_f32=float((gf[k].flatten() - b).norm() / n), norm_dev=float(a.norm() / n - 1),
                             cos=float((a @ b) / (a.norm() * b.norm()))))
    rows.sort(key=lambda r: -r['e_x3'])
    rec = dict(B=B, H=H, W=W, precision=AF.get_precision(), loss_hip=float(outp['loss']), loss_f64=l64, loss_f32=l32,
               relu_sites=len(seen), relu_elements=sites, relu_sign_flips=flips, loss_f64_masked=lm,
               worst_e_x3=rows[0]['e_x3'], worst_e_mask=max(r['e_mask'] for r in rows), median_e_x3=float(np.median([r['e_x3'] for r in rows])),
               median_e_mask=float(np.median([r['e_mask'] for r in rows])), worst_e_f32=max(r['e_f32'] for r in rows), worst_norm_dev=max(abs(r['norm_dev']) for r in rows),
               n_tensors=len(rows), n_over_5e4=sum(r['e_x3'] > 5e-4 for r in rows), rows=rows)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rec, open(out, 'w'), indent=1)
    print(f'{B}x{H}x{W} {rec["precision"]}: loss hip {rec["loss_hip"]:.8f} f64 {l64:.8f} f32 {l32:.8f}; tensors {len(rows)}, e_x3 > 5e-4: {rec["n_over_5e4"]}; '
          f'ReLU sign flips {flips} of {sites} elements at {len(seen)} sites; worst e_x3 {rec["worst_e_x3"]:.2e} -> with the HIP masks {rec["worst_e_mask"]:.2e} '
          f'(medians {rec["median_e_x3"]:.2e} -> {rec["median_e_mask"]:.2e}); fp32 floor {rec["worst_e_f32"]:.2e}')
    print(f'{"name":58s} {"numel":>9s} {"e_x3":>9s} {"e_mask":>9s} {"e_f32":>9s} {"norm_dev":>10s}')
    for r in rows[:40]:
        print(f'{r["step"] + ":" + r["name"]:58s} {r["numel"]:9d} {r["e_x3"]:9.2e} {r["e_mask"]:9.2e} {r["e_f32"]:9.2e} {r["norm_dev"]:10.2e}')


if __name__ == '__main__':
    main()

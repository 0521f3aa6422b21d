"""Golden for the evaluation row: run the REFERENCE's eval_map (mmdet/core/evaluation/mean_ap.py, with its
ceil-to-2-decimals quirk) on the seeded cases of tests/synth.detection_eval_case.   python tools/golden/make_golden_eval.py"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore')
import mmcv_shim  # noqa: E402

mmcv_shim.install()
from mmdet.core.evaluation import mean_ap as ref  # noqa: E402

from tests import synth  # noqa: E402

ref.print_map_summary = lambda *a, **k: None
out = {}
for name, seed, ign in (('a', 50, True), ('b', 51, False)):
    dets, anns = synth.detection_eval_case(seed=seed, with_ignore=ign)
    for ds, tag in (('voc07', 'voc07'), (tuple(str(i) for i in range(20)), 'area')):
        m, res = ref.eval_map(dets, anns, iou_thr=0.5, dataset=ds, logger='silent', nproc=2)
        out[f'{name}_{tag}_map'] = np.float64(m)
        out[f'{name}_{tag}_ap'] = np.array([r['ap'] for r in res], np.float64)
        out[f'{name}_{tag}_ngt'] = np.array([r['num_gts'] for r in res])
        out[f'{name}_{tag}_ndet'] = np.array([r['num_dets'] for r in res])
        out[f'{name}_{tag}_rec3'] = res[3]['recall']
        out[f'{name}_{tag}_prec3'] = res[3]['precision']
    m, res = ref.eval_map(dets, anns, scale_ranges=[(0, 64), (64, 1000)], iou_thr=0.4, dataset=None, logger='silent', nproc=2)
    out[f'{name}_scales_map'] = np.array(m, np.float64)
    out[f'{name}_scales_ap'] = np.stack([r['ap'] for r in res]).astype(np.float64)
out['ap_kat_area'] = np.float64(ref.average_precision(np.array([0.1, 0.4, 0.4, 0.8, 1.0]), np.array([1.0, 0.5, 0.67, 0.5, 0.4]), 'area'))
out['ap_kat_11'] = np.float64(ref.average_precision(np.array([0.1, 0.4, 0.4, 0.8, 1.0]), np.array([1.0, 0.5, 0.667, 0.5, 0.401]), '11points'))
path = os.path.join(ROOT, 'tests', 'golden', 'eval_map.npz')
np.savez_compressed(path, **out)
print('eval_map golden:', os.path.getsize(path), 'bytes;', {k: float(v) for k, v in out.items() if k.endswith('_map') and np.ndim(v) == 0})

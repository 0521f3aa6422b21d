"""Hot-path subset of mmdet/utils/functions.py (:364-367, 415-444, 467-483); the visualisation helpers of that
file (cv2 / PIL drawing) are out of scope."""
import os

import numpy as np
import torch


def EditCfg(cfg, new_dict):
    for key, val in new_dict.items():
        if key in cfg:
            cfg[key] = val


def ExtractAggFunc(type):
    """'objectSum_scaleMax_classSum' -> {'object': torch.sum, 'scale': torch.max, 'class': torch.sum}."""
    funcDict = {'Sum': torch.sum, 'Avg': torch.mean, 'Max': torch.max}
    output = {}
    for name in ['object', 'scale', 'class']:
        for splitType in type.split('_'):
            if name in splitType:
                output[name] = funcDict[splitType.replace(name, '')]
    return output


def StartEnd(mlvl, sIdx):
    start, end = 0, 0
    for si, slvl in enumerate(mlvl):
        end = end + slvl.size(1)
        if si == sIdx:
            return start, end
        start = end


def getMaxConf(mlvl_cls_scores, nCls):
    B = mlvl_cls_scores[0].size(0)
    output = torch.zeros(B, len(mlvl_cls_scores), device=mlvl_cls_scores[0].device)
    for sIdx, cls_scores in enumerate(mlvl_cls_scores):
        bar = cls_scores.permute([0, 2, 3, 1]).reshape(B, -1, nCls)
        output[:, sIdx] = bar.softmax(dim=-1).reshape(B, -1).max(dim=-1)[0]
    return output.max(dim=-1)[0].tolist(), output


def ResumeCycle(cfg, currentCycle, fromStartCycle):
    if currentCycle < fromStartCycle:
        return (False, False)
    return (np.load(cfg.work_dir + '/X_L_' + str(fromStartCycle) + '.npy'), np.load(cfg.work_dir + '/X_U_' + str(fromStartCycle) + '.npy'))


def DelJunkSave(work_dir):
    for f in os.listdir(work_dir):
        if f.endswith('.pth'):
            os.remove(os.path.join(work_dir, f))

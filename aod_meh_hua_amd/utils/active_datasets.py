"""Active-learning bookkeeping with the reference's function names (mmdet/utils/active_datasets.py:37-135):
initial split, per-cycle annotation lists, and the selection rule `update_X_L` (pure numpy host logic)."""
import numpy as np
import torch

from ..mmcv_lite import mkdir_or_exist


def load_ann_list(paths):
    return [np.loadtxt(path, dtype='str') for path in paths]


def get_X_L_0_prev(cfg):
    """:37-48 -- shuffle all indices with the global numpy RNG, first X_L_0_size labelled."""
    anns = load_ann_list(cfg.data.train.dataset.ann_file)
    X_all = np.arange(sum(len(a) for a in anns))
    np.random.shuffle(X_all)
    X_L = X_all[:cfg.X_L_0_size].copy()
    X_U = X_all[cfg.X_L_0_size:cfg.X_L_0_size * 2].copy()
    X_L.sort()
    X_U.sort()
    return X_L, X_U, X_all, anns


def _write_split(cfg, X, anns, cycle, tag, repeat):
    bounds = np.cumsum([0] + [len(a) for a in anns])
    paths = []
    years = ['07', '12'] + [str(i) for i in range(2, len(anns))]
    for i, ann in enumerate(anns):
        sel = X[(X >= bounds[i]) & (X < bounds[i + 1])] - bounds[i]
        folder = cfg.work_dir + '/cycle' + str(cycle)
        mkdir_or_exist(folder)
        path = folder + f'/trainval_{tag}_' + years[i] + '.txt'
        np.savetxt(path, np.atleast_1d(ann)[sel], fmt='%s')
        paths.append(path)
    cfg.data.train.dataset.ann_file = paths
    cfg.data.train.times = repeat
    return cfg


def create_X_L_file(cfg, X_L, anns, cycle):
    """:50-64."""
    return _write_split(cfg, X_L, anns, cycle, 'X_L', cfg.X_L_repeat)


def create_X_U_file(cfg, X_U, anns, cycle):
    return _write_split(cfg, X_U, anns, cycle, 'X_U', cfg.X_U_repeat)


def update_X_L(uncertainty, X_all, X_L, X_S_size, **kwargs):
    """:102-135.  int(X_S_size*zeroRate) zero-score images drawn WITH replacement (np.random.choice), the rest top
    scores by argsort; X_L_next = sorted concat (not uniqued); X_U_next = first len(X_L_next) of the shuffled remainder."""
    if torch.is_tensor(uncertainty):
        uncertainty = uncertainty.cpu().numpy()
    all_X_U = np.array(list(set(X_all) - set(X_L)))
    uncertainty_X_U = uncertainty[all_X_U]
    arg = uncertainty_X_U.argsort()
    if kwargs.get('zeroRate'):
        zeros = (uncertainty_X_U == 0).nonzero()[0]
        zeroSize = int(X_S_size * kwargs['zeroRate'])
        nonZeroSize = X_S_size - zeroSize
        if len(zeros) < zeroSize:
            zeroSize = len(zeros)
        if kwargs.get('useMaxConf', 'False') != 'False':
            maxConfArg = np.array(kwargs['maxconf'])[all_X_U].argsort()
            zeroIdx = maxConfArg[:zeroSize] if kwargs['useMaxConf'] == 'min' else maxConfArg[-zeroSize:]
        else:
            zeroIdx = np.random.choice(zeros, zeroSize)
        X_S = np.concatenate((all_X_U[zeroIdx], all_X_U[arg[-nonZeroSize:]]))
    else:
        X_S = all_X_U[arg[-X_S_size:]]
    X_L_next = np.concatenate((X_L, X_S))
    all_X_U_next = np.array(list(set(X_all) - set(X_L_next)))
    np.random.shuffle(all_X_U_next)
    X_U_next = all_X_U_next[:X_L_next.shape[0]]
    X_L_next.sort()
    X_U_next.sort()
    return X_L_next, X_U_next

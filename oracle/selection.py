"""Oracle: active-learning selection rule.  TEST INFRASTRUCTURE ONLY."""
import numpy as np


def update_X_L(uncertainty, X_all, X_L, X_S_size, zeroRate=None, rng=np.random):
    """mmdet/utils/active_datasets.py:102-135 (useMaxConf='False' branch).

    zero-score images: int(X_S_size*zeroRate) drawn WITH replacement; rest = top by argsort
    (numpy default quicksort order on ties); X_L_next sorted, not uniqued; X_U_next = first
    len(X_L_next) of a shuffled remainder, sorted."""
    uncertainty = np.asarray(uncertainty)
    all_X_U = np.array(list(set(X_all) - set(X_L)))
    unc_U = uncertainty[all_X_U]
    arg = unc_U.argsort()
    if zeroRate:
        zeros = (unc_U == 0).nonzero()[0]
        zeroSize = int(X_S_size * zeroRate)
        nonZeroSize = X_S_size - zeroSize
        if len(zeros) < zeroSize:
            zeroSize = len(zeros)
        zeroIdx = rng.choice(zeros, zeroSize)
        X_S = np.concatenate((all_X_U[zeroIdx], all_X_U[arg[-nonZeroSize:]]))
    else:
        X_S = all_X_U[arg[-X_S_size:]]
    X_L_next = np.concatenate((X_L, X_S))
    rest = np.array(list(set(X_all) - set(X_L_next)))
    rng.shuffle(rest)
    X_U_next = rest[:X_L_next.shape[0]]
    X_L_next.sort()
    X_U_next.sort()
    return X_L_next, X_U_next

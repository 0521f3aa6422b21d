"""mmdet/core/utils/misc.py:10-42 (multi_apply, unmap) and dist_utils.py:63-69 (reduce_mean)."""
from functools import partial

import torch
import torch.distributed as dist


def multi_apply(func, *args, **kwargs):
    pfunc = partial(func, **kwargs) if kwargs else func
    map_results = map(pfunc, *args)
    return tuple(map(list, zip(*map_results)))


def unmap(data, count, inds, fill=0):
    if data.dim() == 1:
        ret = data.new_full((count, ), fill)
        ret[inds.type(torch.bool)] = data
    else:
        ret = data.new_full((count, ) + data.size()[1:], fill)
        ret[inds.type(torch.bool), :] = data
    return ret


def reduce_mean(tensor):
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return tensor

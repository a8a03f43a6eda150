"""allaverage (reference linklink/dist_helper.py:33-36): tensor /= world; all-reduce SUM -- through linklink.allreduce,
i.e. the C-ABI RCCL wrapper for fp32 device tensors."""
import torch.distributed as dist

from . import allreduce


def allaverage(tensor):
    t = tensor.data
    if t.is_contiguous():
        t /= dist.get_world_size()
        allreduce(t)
    else:                          # a strided view (a column of the activation table): collectives want a dense buffer
        d = t.contiguous()
        d /= dist.get_world_size()
        allreduce(d)
        t.copy_(d)
    return tensor

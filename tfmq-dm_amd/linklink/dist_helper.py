"""allaverage (reference linklink/dist_helper.py:33-36): tensor /= world; all_reduce(SUM)."""
import torch.distributed as dist


def allaverage(tensor):
    tensor.data /= dist.get_world_size()
    dist.all_reduce(tensor.data)
    return tensor

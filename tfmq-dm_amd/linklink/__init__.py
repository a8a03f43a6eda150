"""torch.distributed aliases used by the calibration path (reference linklink/__init__.py:6-13).
On ROCm the "nccl" backend is RCCL (xGMI); "gloo" is used by the CPU multi-process tests."""
import torch.distributed as dist

allreduce = dist.all_reduce
allgather = dist.all_gather
broadcast = dist.broadcast
barrier = dist.barrier
synchronize = dist.barrier
init_process_group = dist.init_process_group
get_rank = dist.get_rank
get_world_size = dist.get_world_size

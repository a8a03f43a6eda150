"""The calibration path's collectives (reference linklink/__init__.py:6-13) on MI355X.

Same names as the reference module (`allreduce`, `allgather`, `broadcast`, `barrier`, `synchronize`,
`init_process_group`, `get_rank`, `get_world_size`).  Control plane = torch.distributed (rendezvous, barriers,
the gloo backend of the CPU multi-process tests).  Data plane on GPUs = the C ABI's RCCL wrappers
(include/tfmq_hip.h: tfmq_comm_init / tfmq_allreduce_sum_f32): `init_process_group` with the "nccl" backend
(= RCCL on ROCm) draws a rendezvous id on rank 0, hands it to every rank through the process group's store and
binds one communicator per GPU to the tfmq handle of that device; `allreduce` of a contiguous fp32 device tensor
is then ONE ncclAllReduce enqueued on the caller's current stream -- between the unit's backward GEMMs and the
fused AdaRound-backward + Adam kernel, with no host synchronisation (quant/reconstruction.py:72-75,193-195,298-300).
Everything else (CPU tensors, other dtypes, gloo) goes to torch.distributed unchanged."""
import ctypes as _C

import torch as _torch
import torch.distributed as dist

allgather = dist.all_gather
broadcast = dist.broadcast
barrier = dist.barrier
synchronize = dist.barrier
get_rank = dist.get_rank
get_world_size = dist.get_world_size

def comm_device():
    """Device index whose tfmq handle owns this process's RCCL communicator (None: no communicator).  The state lives on
    the Handle objects of tfmq_dm_amd._lib, so the module behaves the same imported as `linklink` (drop-in name, with
    tfmq-dm_amd/ on sys.path) or as `tfmq_dm_amd.linklink`."""
    from tfmq_dm_amd import _lib
    for dev, h in _lib._handles.items():
        if getattr(h, "comm_world", 0) > 0:
            return dev
    return None


def init_comm(device=None):
    """Create the C-ABI RCCL communicator of this process over the ranks of the default process group (collective).
    Called by init_process_group for the "nccl" backend; callable on its own after a gloo rendezvous."""
    from tfmq_dm_amd._lib import TfmqError, handle, load
    if comm_device() is not None:
        return
    dev = _torch.cuda.current_device() if device is None else int(device)
    lib = load()
    world, rank = dist.get_world_size(), dist.get_rank()
    ident = [None]
    if rank == 0:
        buf = (_C.c_uint8 * 128)()
        rc = lib.tfmq_comm_unique_id(buf)
        if rc != 0:
            raise TfmqError(f"tfmq_comm_unique_id failed ({rc}): librccl could not be loaded")
        ident[0] = bytes(buf)
    dist.broadcast_object_list(ident, src=0)
    buf = (_C.c_uint8 * 128).from_buffer_copy(ident[0])
    h = handle(dev)
    h.call("comm_init", buf, rank, world)
    h.comm_world = world


def destroy_comm():
    dev = comm_device()
    if dev is not None:
        from tfmq_dm_amd._lib import handle
        handle(dev).call("comm_destroy")
        handle(dev).comm_world = 0


def init_process_group(backend="nccl", init_method=None, world_size=-1, rank=-1, **kw):
    if not dist.is_initialized():      # a launcher (torchrun-style driver, bench.py) may have made the rendezvous already
        dist.init_process_group(backend=backend, init_method=init_method, world_size=world_size, rank=rank, **kw)
    if str(backend).lower() == "nccl" and _torch.cuda.is_available():
        init_comm()


def allreduce(tensor, *a, **kw):
    """SUM all-reduce in place (reference linklink.allreduce = dist.all_reduce)."""
    if not a and not kw and tensor.is_cuda and tensor.dtype == _torch.float32 and tensor.is_contiguous():
        dev = comm_device()
        if dev is not None and (tensor.device.index or 0) == dev:
            from tfmq_dm_amd._lib import handle
            handle(dev).call("allreduce_sum_f32", _C.c_void_p(tensor.data_ptr()), tensor.numel(),
                             _C.c_void_p(_torch.cuda.current_stream(dev).cuda_stream))
            return None
    return dist.all_reduce(tensor, *a, **kw)

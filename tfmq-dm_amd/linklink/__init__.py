"""The calibration path's collectives (reference linklink/__init__.py:6-13) on MI355X.

Same names as the reference module (`allreduce`, `allgather`, `broadcast`, `barrier`, `synchronize`,
`init_process_group`, `get_rank`, `get_world_size`).  Control plane = torch.distributed (rendezvous, barriers,
the gloo backend of the CPU multi-process tests).  Data plane on GPUs = the C ABI's RCCL wrappers
(include/tfmq_hip.h: tfmq_comm_init / tfmq_allreduce_sum_f32).  `init_process_group` with the "nccl" backend
(= RCCL on ROCm) only RECORDS that a communicator is wanted: the reference's entry point calls it before
`torch.cuda.set_device(gpu)` (quant/calibration.py:241-245), when every spawned rank still sits on device 0.  The
communicator is created by the first `allreduce` of a device tensor, on THAT tensor's device: rank 0 draws the
rendezvous id and hands it round through the process group's key-value store (no device collective is involved).
`allreduce` of a contiguous fp32 device tensor is then ONE ncclAllReduce enqueued on the caller's current stream --
between the unit's backward GEMMs and the fused AdaRound-backward + Adam kernel, with no host synchronisation
(quant/reconstruction.py:72-75,193-195,298-300).  Everything else (CPU tensors, other dtypes, gloo) goes to
torch.distributed unchanged."""
import ctypes as _C
import os as _os

# the host driver supports dmabuf IPC only: without it RCCL's intra-node transport fails in hipIpcGetMemHandle.  Effective when this module is
# imported before the process's first HIP call (the launchers -- bench.py, mp.spawn parents -- export it as well; children inherit it).
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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


_want_comm = False      # an "nccl" rendezvous was made: the first device all-reduce creates the communicator
_comm_epoch = 0         # communicators created so far (key of the rendezvous id in the store)


def _exchange_id(ident):
    """Rank 0's 128-byte rendezvous id to every rank through the default process group's store (host side only: the
    ranks need not have chosen their devices yet, and no NCCL communicator of torch's own is created for it)."""
    key = f"tfmq_comm_id_{_comm_epoch}"
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception:
        store = None
    if store is None:                      # no store (exotic init methods): fall back to an object broadcast
        box = [ident]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    if dist.get_rank() == 0:
        store.set(key, ident)
        return ident
    return bytes(store.get(key))           # blocks until rank 0 has set it


def _device_identity(dev):
    """A string that is equal for two ranks exactly when they sit on the same physical GPU (host name + the device's uuid / PCI address)."""
    import socket
    p = _torch.cuda.get_device_properties(dev)
    pci = tuple(getattr(p, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    uuid = getattr(p, "uuid", None)
    if uuid is None and all(v is None for v in pci):
        return ""            # this torch build exposes neither: no identity, no check (never a false refusal on a real 8-GPU node)
    if all(v is None for v in pci) and not str(uuid).strip("0-"):
        return ""            # no PCI address and an all-zero uuid (some ROCm builds): nothing that tells two GPUs apart
    return f"{socket.gethostname()}|{uuid}|{pci}"


def _device_hint(dev):
    """What the rank believes about its device besides the identity: the visible-device mask and the local index.  Two ranks of one host
    that see the SAME mask and chose DIFFERENT indices cannot share a GPU, whatever the driver reports as identity."""
    import os
    return "|".join([os.environ.get("HIP_VISIBLE_DEVICES", ""), os.environ.get("ROCR_VISIBLE_DEVICES", ""), os.environ.get("CUDA_VISIBLE_DEVICES", ""), str(int(dev))])


def duplicate_devices(identities):
    """[(rank_a, rank_b), ...] of ranks that named the same device (pure function: unit-tested without a GPU)."""
    seen, dup = {}, []
    for r, ident in enumerate(identities):
        if ident in seen:
            dup.append((seen[ident], r))
        else:
            seen[ident] = r
    return dup


def _contradicted(pair, hints):
    """An identity match between two ranks whose hints say otherwise (same visible-device mask, different local index): the identity the
    driver reports is not trustworthy on this box -- never refuse a real multi-GPU node on its account."""
    a, b = (hints[r].rsplit("|", 1) for r in pair)
    return a[0] == b[0] and a[1] != b[1]


def _refuse_duplicate_devices(mine, hint=""):
    """Every rank publishes its device identity through the rendezvous store and reads the others': two ranks on one GPU would sit in
    ncclCommInitRank until its bootstrap times out (or for ever), so the mismatch is refused HERE, on every rank at once, before any
    RCCL call.  TFMQ_COMM_ALLOW_SHARED_DEVICE=1 skips the check (never useful with RCCL; kept for experiments)."""
    import os
    from tfmq_dm_amd._lib import TfmqError
    world, rank = dist.get_world_size(), dist.get_rank()
    if world == 1 or os.environ.get("TFMQ_COMM_ALLOW_SHARED_DEVICE") == "1":
        return
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception:
        store = None
    mine_h = mine + "\t" + (hint or "|")
    if store is not None:
        store.set(f"tfmq_comm_dev_{_comm_epoch}_{rank}", mine_h.encode())
        # a rank that died before its store.set must not park the others for the store's default timeout (30 min): wait a bounded time,
        # then name the missing ranks
        import datetime
        keys = [f"tfmq_comm_dev_{_comm_epoch}_{r}" for r in range(world)]
        try:
            store.wait(keys, datetime.timedelta(seconds=float(os.environ.get("TFMQ_COMM_RENDEZVOUS_TIMEOUT_S", "120"))))
        except Exception as e:      # noqa: BLE001 -- a timeout of the store (the class differs between store kinds)
            missing = []
            for r, k in enumerate(keys):
                try:
                    store.wait([k], datetime.timedelta(milliseconds=10))
                except Exception:      # noqa: BLE001
                    missing.append(r)
            raise TfmqError(f"tfmq_comm_init: rank(s) {missing} never published their device identity ({type(e).__name__}); "
                            "did they fail before the first device all-reduce?") from e
        both = [bytes(store.get(k)).decode() for k in keys]
    else:
        both = [None] * world
        dist.all_gather_object(both, mine_h)
    ids, hints = [b.split("\t", 1)[0] for b in both], [b.split("\t", 1)[1] for b in both]
    dup = [] if any(not i for i in ids) else [p for p in duplicate_devices(ids) if not _contradicted(p, hints)]
    if dup:
        raise TfmqError(f"tfmq_comm_init: ranks {dup} share a GPU ({ids[dup[0][0]]}); RCCL needs one device per rank "
                        "(set the device before the first device all-reduce: quant/calibration.py:241-245)")


def init_comm(device=None):
    """Create the C-ABI RCCL communicator of this process over the ranks of the default process group (collective: every
    rank calls it, each with ITS device).  Called lazily by the first device all-reduce after an "nccl" rendezvous;
    callable on its own after a gloo rendezvous.  Refuses a rank / device mismatch the process group can see."""
    global _comm_epoch
    from tfmq_dm_amd._lib import TfmqError, handle, load
    if comm_device() is not None:
        return
    dev = _torch.cuda.current_device() if device is None else int(device)
    lib = load()
    world, rank = dist.get_world_size(), dist.get_rank()
    _refuse_duplicate_devices(_device_identity(dev), _device_hint(dev))
    ident = None
    if rank == 0:
        buf = (_C.c_uint8 * 128)()
        rc = lib.tfmq_comm_unique_id(buf)
        if rc != 0:
            raise TfmqError(f"tfmq_comm_unique_id failed ({rc}): librccl could not be loaded")
        ident = bytes(buf)
    ident = _exchange_id(ident)
    _comm_epoch += 1
    buf = (_C.c_uint8 * 128).from_buffer_copy(ident)
    h = handle(dev)
    h.call("comm_init", buf, rank, world)
    h.comm_world = world


def destroy_comm():
    global _want_comm
    dev = comm_device()
    if dev is not None:
        from tfmq_dm_amd._lib import handle
        handle(dev).call("comm_destroy")
        handle(dev).comm_world = 0
    _want_comm = False


def init_process_group(backend="nccl", init_method=None, world_size=-1, rank=-1, device=None, **kw):
    """reference linklink.init_process_group = dist.init_process_group.  `device` (optional, not in the reference) binds the
    RCCL communicator at once to that device instead of at the first device all-reduce."""
    global _want_comm
    if not dist.is_initialized():      # a launcher (torchrun-style driver, bench.py) may have made the rendezvous already
        dist.init_process_group(backend=backend, init_method=init_method, world_size=world_size, rank=rank, **kw)
    if str(backend).lower() == "nccl" and _torch.cuda.is_available():
        _want_comm = True
        if device is not None:
            init_comm(device)


def allreduce(tensor, *a, **kw):
    """SUM all-reduce in place (reference linklink.allreduce = dist.all_reduce)."""
    if not a and not kw and tensor.is_cuda and tensor.dtype == _torch.float32 and tensor.is_contiguous():
        dev = comm_device()
        if dev is None and _want_comm:     # first device all-reduce after an "nccl" rendezvous: bind to THIS tensor's device
            init_comm(tensor.device.index or 0)
            dev = comm_device()
        if dev is not None and (tensor.device.index or 0) == dev:
            from tfmq_dm_amd._lib import handle
            handle(dev).call("allreduce_sum_f32", _C.c_void_p(tensor.data_ptr()), tensor.numel(),
                             _C.c_void_p(_torch.cuda.current_stream(dev).cuda_stream))
            return None
    return dist.all_reduce(tensor, *a, **kw)

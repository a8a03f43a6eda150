"""Reconstruction data capture (reference quant/data_utill.py): the FP output of a unit and its
input under the already-quantised upstream (asymmetric reconstruction, A7).

The reference registers a forward hook on the unit and aborts the forward with an exception; here the
engine plan exposes every unit's input/output as taps of a forward pass and a `StopAt` tap dictionary ends the pass at
the unit, so a capture is two partial engine runs per calibration batch (FP for the target, weight-quantised for the
input)."""
from __future__ import annotations

import logging
import os
from typing import Tuple, Union

import torch

from tfmq_dm_amd import ops
from tfmq_dm_amd.engine import StopAt
from tfmq_dm_amd._lib import TfmqError

logger = logging.getLogger(__name__)


class StopForwardException(Exception):
    """Kept for API compatibility (the engine does not need to abort a forward)."""


def unit_name(model, unit) -> str:
    for n, m in model.model.named_modules():
        if m is unit:
            return n
    raise KeyError("unit is not a sub-module of the quantised model")


class HostRows:
    """Rows of a reconstruction cache kept in PINNED HOST memory -- the reference's `keep_gpu=False` (quant/calibration.py:62-67 switches it on
    for the widest SD units; quant/data_utill.py:39-46 keeps the cache on the CPU, quant/reconstruction.py:66,184 moves each mini-batch
    `.to(device)`).  The reconstruction units only ever `index_select(0, idx)` their caches, so this class answers that call: the selected rows
    (contiguous in the cache) travel as one asynchronous host -> device copy each on the current stream.  Used by save_inout when a cached
    tensor does not fit beside the device memory the capture and the iterations need (12 800 samples x 64 x 64 x 960 fp32 channels of the SD
    recipe's `output_blocks.9` input are 201 GB)."""

    def __init__(self, shape, dtype, device):
        need = torch.empty((), dtype=dtype).element_size() * int(torch.Size(shape).numel())
        room = host_room()
        if need > room:      # fail before the allocation, not in the host's OOM killer (or the sandbox's: see host_room)
            raise TfmqError(f"HostRows: {need / 2**30:.1f} GiB of pinned host memory wanted for a reconstruction cache, "
                            f"{room / 2**30:.1f} GiB allowed on this host (MemAvailable, cgroup limit, TFMQ_CACHE_HOST_MAX_GB)")
        self.buf = torch.empty(tuple(shape), dtype=dtype, pin_memory=True)
        self.device, self.dtype, self.shape = torch.device(device), dtype, torch.Size(shape)
        self.is_host_rows = True

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def __len__(self):
        return self.shape[0]

    def fill(self, start: int, rows: torch.Tensor):
        self.buf[start:start + rows.shape[0]].copy_(rows)              # device -> pinned host

    def index_select(self, dim: int, idx: torch.Tensor) -> torch.Tensor:
        assert dim == 0
        host = getattr(idx, "_host", None)       # reconstruction._run keeps the host copy of the indices it drew on the host
        ids = (host if host is not None else idx.cpu()).tolist()
        out = torch.empty((len(ids),) + tuple(self.shape[1:]), dtype=self.dtype, device=self.device)
        for j, i in enumerate(ids):
            out[j].copy_(self.buf[i], non_blocking=True)
        return out


class HalfRows:
    """LAST RESORT, opt-in (TFMQ_CACHE_F16=1): a reconstruction cache that fits neither beside the passes on the device in its own dtype nor
    in the host memory this process may pin is kept ON THE DEVICE AS fp16 and widened row by row at selection.  The reference has no such
    case (it needs the host RAM); here it happens for ONE tensor of the SD recipe -- the 960-channel input of `output_blocks.9.0` over 12 800
    samples, 201 GB in fp32 -- on a box whose sandbox dies under a pinned allocation of that size.  `inexact` counts the elements the
    narrowing changed (0 when the tapped tensor came out of the fp16 activation stream); the caller logs it."""

    def __init__(self, shape, dtype, device):
        self.buf = torch.empty(tuple(shape), dtype=torch.float16, device=device)
        self.device, self.dtype, self.shape = torch.device(device), dtype, torch.Size(shape)
        self.inexact = 0

    def size(self, d=None):
        return self.shape if d is None else self.shape[d]

    def __len__(self):
        return self.shape[0]

    def fill(self, start: int, rows: torch.Tensor):
        h = rows.to(torch.float16)
        self.inexact += int((h.to(rows.dtype) != rows).sum())
        self.buf[start:start + rows.shape[0]].copy_(h)

    def index_select(self, dim: int, idx: torch.Tensor) -> torch.Tensor:
        assert dim == 0
        return self.buf.index_select(0, idx.to(self.buf.device)).to(self.dtype)


def host_room() -> int:
    """Bytes of pinned host memory a new cache may take: MemAvailable minus 16 GiB, the cgroup's memory limit minus what it already uses,
    and TFMQ_CACHE_HOST_MAX_GB (default 64: round 6 lost a GPU box -- a microVM reporting 3 TB of RAM -- to a 201 GB pinned allocation)."""
    room = int(float(os.environ.get("TFMQ_CACHE_HOST_MAX_GB", "64")) * (1 << 30))
    avail = _host_available()
    if avail is not None:
        room = min(room, avail - (16 << 30))
    for lim, use in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            with open(lim) as f:
                v = f.read().strip()
            if v.isdigit() and int(v) < (1 << 60):
                with open(use) as f:
                    room = min(room, int(v) - int(f.read().strip()) - (4 << 30))
        except (OSError, ValueError):
            pass
    return max(room, 0)


def _host_available():
    """MemAvailable of /proc/meminfo in bytes (None where the file is missing)."""
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) * 1024
    except OSError:
        pass
    return None


def _device_room(dev) -> int:
    """Bytes a new cache tensor may take on `dev`: what the driver reports free plus what the caching allocator holds unused, minus a reserve
    for the capture forwards and the iterations' buffers (TFMQ_CACHE_RESERVE_GB, default 40)."""
    free, _ = torch.cuda.mem_get_info(dev)
    idle = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    return int(free + idle - float(os.environ.get("TFMQ_CACHE_RESERVE_GB", "40")) * (1 << 30))


def save_inout(model, layer, cali_data: Tuple[torch.Tensor], asym: bool = False, use_act: bool = False,
               batch_size: int = 128, keep_gpu: bool = True):
    """-> (cached_inputs: tuple of tensors, cached_output).  Unit inputs (reference DataSaverHook, :79-104):
    ResnetBlock / ResBlock (x NHWC, temb | emb); BasicTransformerBlock (tokens [N,T,C], context [N,L,D]);
    attention block / single layer (x,).  cali_data = (xs, ts) or (xs, ts, cs) for context-conditioned UNets.
    A cached tensor stays on the device (288 GB HBM) when it fits beside a reserve for the passes themselves; one that does not -- or every
    one, with `keep_gpu=False` and TFMQ_CACHE_HOST=1 -- lives in pinned host memory as `HostRows` (the reference's keep_gpu=False,
    calibration.py:62-67), from which the iterations fetch their mini-batches."""
    from .quant_block import QuantBasicTransformerBlock, QuantQKMatMul, QuantResBlock, QuantResnetBlock, QuantSMVMatMul
    name = unit_name(model, layer)
    dev = next(model.model.parameters()).device
    xs, ts = cali_data[0], cali_data[1]
    cs = cali_data[2] if len(cali_data) > 2 else None
    # Round 5: every cached tensor is allocated ONCE at its final size and filled batch by batch -- collecting the batches in a list and
    # torch.cat'ing them at the end needs twice the memory for a moment, which the 64 x 64 up-path units of the cin256 recipe (10 240
    # samples x 64 x 64 x 576 channels = 90 GiB of input) do not have even on 288 GB.
    n_total = int(xs.size(0))
    # (round 6: the previous unit's caches must be gone before this unit's are sized -- in the recipe-size SD job a 125 GiB target was judged not
    # to fit because the 187 GiB of the unit before it were still waiting for the collector)
    import gc
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()

    class _Rows:
        def __init__(self, host_ok=False):
            self.buf, self.n, self.host_ok = None, 0, host_ok      # host_ok: a tensor the units only index_select (unit input / target)

        def append(self, t, bsz):
            if self.buf is None and self.n == 0 and not getattr(self, "parts", None):
                if t.shape[0] % bsz == 0:      # rows per sample: 1, or heads for the [(b h), ...] tensors of the stand-alone matmul units
                    self.rows = n_total * (t.shape[0] // bsz)
                    shape = (self.rows,) + tuple(t.shape[1:])
                    nbytes = t.element_size() * self.rows * int(t[0].numel())
                    force = os.environ.get("TFMQ_CACHE_HOST")          # "1": every cache with keep_gpu=False on the host; "all": every cache (tests)
                    if t.is_cuda and self.host_ok and (force == "all" or (force == "1" and not keep_gpu) or nbytes > _device_room(t.device)):
                        if (nbytes > host_room() and force is None and os.environ.get("TFMQ_CACHE_F16") == "1" and t.dtype == torch.float32
                                and nbytes // 2 <= _device_room(t.device)):
                            self.buf = HalfRows(shape, t.dtype, t.device)
                            logger.warning(f"save_inout: {nbytes / 2**30:.1f} GiB of cache for '{name}' fit neither on the device nor in "
                                           f"{host_room() / 2**30:.0f} GiB of host memory: kept on the device as fp16 (TFMQ_CACHE_F16=1)")
                        else:
                            self.buf = HostRows(shape, t.dtype, t.device)
                            logger.info(f"save_inout: {nbytes / 2**30:.1f} GiB of cache for '{name}' kept in pinned host memory")
                    else:
                        self.buf = torch.empty(shape, dtype=t.dtype, device=t.device)
                else:                          # (a tensor that does not scale with the batch: collected and concatenated as before)
                    self.parts = []
            if self.buf is None:
                self.parts.append(t)
                return
            if isinstance(self.buf, (HostRows, HalfRows)):
                self.buf.fill(self.n, t)
            else:
                self.buf[self.n:self.n + t.shape[0]].copy_(t)
            self.n += t.shape[0]

        def __bool__(self):
            return self.buf is not None or bool(getattr(self, "parts", None))

        def tensor(self):
            if self.buf is None:
                return torch.cat(self.parts)
            if self.n != self.rows:       # (a tap whose leading dimension divides the capture batch without scaling with it)
                from tfmq_dm_amd._lib import TfmqError
                raise TfmqError(f"save_inout: cache of '{name}' filled {self.n} of {self.rows} rows -- a tensor that does not scale with the batch")
            return self.buf
    # (the delta-learning units -- use_act -- reshape their caches: those stay on the device)
    ins, outs, tembs, ctxs = _Rows(not use_act), _Rows(not use_act), _Rows(), _Rows()
    # `batch_size` is the reconstruction mini-batch (8 in the SD recipe); the capture forwards are per-sample
    # independent (tests/test_full_size_properties_gpu.py: batch 12 == 6 + 6 bit for bit), so they run at a batch that
    # fills the GPU instead of a launch-bound one
    batch_size = max(int(batch_size), int(os.environ.get("TFMQ_CAPTURE_BATCH", "32")))

    def fwd(x, t, c, taps):
        eng = model.engine(dev)
        if c is None:
            eng.forward(x, t, taps=taps)
        else:
            eng.forward(x, t, c, taps=taps)

    for i in range(0, xs.size(0), batch_size):
        x = ops.nchw_to_nhwc(xs[i:i + batch_size].to(dev).float().contiguous())
        t = ts[i:i + batch_size].to(dev).float().contiguous()
        c = None if cs is None else cs[i:i + batch_size].to(dev).float().contiguous()
        taps = StopAt(name)                             # nothing downstream of the unit is launched
        model.set_quant_state(False, False)             # target: FP model
        fwd(x, t, c, taps)
        if name not in taps:
            raise KeyError(f"save_inout: the engine exposes no tap for unit '{name}'")
        outs.append(taps[name][1], x.shape[0])
        if asym:                                        # input: upstream with the already-quantised weights
            taps = StopAt(name)
            model.set_quant_state(True, use_act)
            fwd(x, t, c, taps)
        tin = taps[name][0]
        if isinstance(layer, (QuantBasicTransformerBlock, QuantQKMatMul, QuantSMVMatMul)):      # two-input units: (tokens, context) / (q, k) / (weight, v)
            ins.append(tin[0], x.shape[0])
            ctxs.append(tin[1], x.shape[0])
            continue
        if isinstance(tin, tuple):       # (h, skip): concatenated input of an up-path block
            tin = torch.cat(tin, dim=-1)
        ins.append(tin, x.shape[0])
        if isinstance(layer, (QuantResnetBlock, QuantResBlock)):
            tembs.append(taps["__temb__"], x.shape[0])
    model.set_quant_state(False, False)
    layer.set_quant_state(True, use_act)
    cached_out = outs.tensor()
    cached_in = (ins.tensor(),) + ((tembs.tensor(),) if tembs else ()) + ((ctxs.tensor(),) if ctxs else ())
    for c_ in (cached_out,) + cached_in:
        if isinstance(c_, HalfRows):
            msg = f"save_inout: fp16 cache of '{name}' {tuple(c_.shape)}: {c_.inexact} of {c_.buf.numel()} elements changed by the narrowing"
            logger.warning(msg)
            print("[cali] " + msg, file=__import__("sys").stderr, flush=True)
    logger.info(f"input shapes: {[tuple(c.shape) for c in cached_in]} output shape: {tuple(cached_out.shape)}")
    return cached_in, cached_out


class GetLayerGrad:
    """dL/d(unit output) for a batch of calibration samples (reference quant/data_utill.py:191-256): FP forward, forward of the model
    "quantised till" the unit, loss = F.kl_div(F.log_softmax(out_q, 1), F.softmax(out_fp, 1), 'batchmean'), backward to the unit's
    output (of the FP forward: see __call__).  The reference hooks autograd; here the engine's exact-fp32 forward records the
    hand-written backward of every launch downstream of the unit on a tape (engine/fisher.py) and replays it.  Returns the gradient in the unit's device layout: NHWC for
    conv-type units, [N, T, C] for transformer blocks (the layout of save_inout's cached output, which the loss kernel weights)."""

    def __init__(self, model, layer, device=None, use_aq: bool = False) -> None:
        self.model, self.layer, self.use_aq = model, layer, use_aq
        self.device = device or next(model.model.parameters()).device

    def _quantize_model_till(self):
        """Modules are visited in definition order; every QuantLayer / quant block up to and including the unit is switched on
        (reference :219-231)."""
        from .quant_block import BaseQuantBlock
        from .quant_layer import QuantLayer
        self.model.set_quant_state(False, False)
        for _, module in self.model.named_modules():
            if isinstance(module, (QuantLayer, BaseQuantBlock)):
                module.set_quant_state(True, self.use_aq)
            if module is self.layer:
                break

    def __call__(self, xs: torch.Tensor, ts: torch.Tensor, cs: torch.Tensor = None) -> torch.Tensor:
        from tfmq_dm_amd.engine import ddim_unet as E
        from tfmq_dm_amd.engine.fisher import GradTape
        model, dev = self.model, self.device
        name = unit_name(model, self.layer)
        x = ops.nchw_to_nhwc(xs.to(dev).float().contiguous())
        t = ts.to(dev).float().contiguous()
        c = None if cs is None else cs.to(dev).float().contiguous()
        args = (x, t) if c is None else (x, t, c)
        old = (getattr(model, "_exact_fp", False), getattr(model, "_soft_targets", False))
        model._exact_fp, model._soft_targets = True, True
        try:
            model.set_quant_state(False, False)
            tape = GradTape(name)
            E.TAPE = tape
            try:
                out_fp = model.engine(dev).forward(*args, taps=tape)
            finally:
                E.TAPE = None
            if out_fp is None or tape.leaf is None:
                raise KeyError(f"GetLayerGrad: the engine exposes no tap for unit '{name}'")
            self._quantize_model_till()
            out_q = model.engine(dev).forward(*args)
            # WHICH gradient the reference caches: softmax(out_fp) is not detached, so loss.backward() runs through BOTH forwards, the
            # unit's backward hook fires once per forward and GradSaverHook keeps the last call -- autograd reaches the earlier (FP)
            # forward last.  What save_grad caches is therefore dL/d(unit output of the FP pass) through the TARGET branch of the KL
            # term (to first order the negative of the out_q branch; Fisher weights use |g| and g^2).  Reproduced as released.
            g_out, _ = ops.kl_softmax_grad(out_q.contiguous(), out_fp.contiguous(), wrt_target=True)
            grad = tape.backward(out_fp, g_out)
        finally:
            model._exact_fp, model._soft_targets = old
            model.set_quant_state(False, False)
            self.layer.set_quant_state(True, self.use_aq)
        return grad


def save_grad(model, layer, cali_data: Tuple[torch.Tensor], damping: float = 1., use_aq: bool = False, batch_size: int = 32,
              keep_gpu: bool = True) -> torch.Tensor:
    """Fisher weights of a unit for the whole calibration set: |dL/d(unit output)| + 1.0 (reference :54-73), on the device, in the
    unit's device layout (see GetLayerGrad)."""
    dev = next(model.model.parameters()).device
    get_grad = GetLayerGrad(model, layer, dev, use_aq)
    grads = []
    for i in range(0, cali_data[0].size(0), batch_size):
        grads.append(get_grad(*(_[i: i + batch_size] for _ in cali_data)))
    g = torch.cat(grads)
    return g.abs_().add_(1.0)

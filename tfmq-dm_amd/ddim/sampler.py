"""DDIM sampling loop of the pixel-space path.

`generalized_steps` keeps the reference's signature and return value
(ddim/functions/denoising.py:10-41) for drop-in use; `GraphDdimSampler` is the MI355X-native
form of the same loop: the whole step (UNet forward on the HIP engine, DDIM update, step counter)
is captured once into a hipGraph and replayed per step -- no per-step `load_state_dict`
(denoising.py:26-29), no host<->device copies of x (:23,32,38), no Python in the timed loop.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from .. import ops
from .._lib import TfmqError, handle


def linear_betas(beta_start: float = 1e-4, beta_end: float = 0.02, n: int = 1000) -> torch.Tensor:
    """get_beta_schedule('linear') -> float32 (ddim/runners/diffusion.py:46-49,83)."""
    return torch.from_numpy(np.linspace(beta_start, beta_end, n, dtype=np.float64)).float()


def step_sequence(skip_type: str, timesteps: int, num_timesteps: int = 1000) -> List[int]:
    """sample_image (ddim/runners/diffusion.py:437-447)."""
    if skip_type == "uniform":
        return list(range(0, num_timesteps, num_timesteps // timesteps))
    if skip_type == "quad":
        return [int(s) for s in list(np.linspace(0, np.sqrt(num_timesteps * 0.8), timesteps) ** 2)]
    raise NotImplementedError(skip_type)


def alpha_bar(betas: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """compute_alpha (ddim/functions/denoising.py:4-7) on the host: schedule constants."""
    b = torch.cat([torch.zeros(1), betas.cpu()], dim=0)
    return (1 - b).cumprod(dim=0).index_select(0, t.long() + 1)


def coef_table(seq: Sequence[int], betas: torch.Tensor, eta: float = 0.0) -> torch.Tensor:
    """Row k (k-th executed step): {sqrt(1-a_t), sqrt(a_t), sqrt(a_next), c1, c2, t, 0, 0}, every
    entry produced by the same fp32 tensor ops the reference applies per step (denoising.py:21-37)."""
    seq = list(seq)
    seq_next = [-1] + seq[:-1]
    rows = []
    for i, j in zip(reversed(seq), reversed(seq_next)):
        at = alpha_bar(betas, torch.tensor([i]))
        an = alpha_bar(betas, torch.tensor([j]))
        c1 = eta * ((1 - at / an) * (1 - an) / (1 - at)).sqrt()
        c2 = ((1 - an) - c1 ** 2).sqrt()
        rows.append(torch.cat([(1 - at).sqrt(), at.sqrt(), an.sqrt(), c1.reshape(1).float(), c2, torch.tensor([float(i)]),
                               torch.zeros(2)]))
    return torch.stack(rows).float().contiguous()


def generalized_steps(x, seq, model, b, **kwargs):
    """Drop-in for ddim/functions/denoising.py:10-41 on device tensors (NCHW).  `model(xt, t)`
    returns eps; `kwargs`: eta, untill_fake_t, tot/cali_ckpt (per-step act_k via model.load_state_dict)."""
    if not x.is_cuda:
        raise TfmqError("generalized_steps: tensors must live on the MI355X (no CPU fallback)")
    with torch.no_grad():
        n = x.size(0)
        seq = list(seq)
        coef = coef_table(seq, b, kwargs.get("eta", 0)).to(x.device)
        xs, x0_preds = [x], []
        t = xt = None
        for cnt, i in enumerate(reversed(seq)):
            t = torch.full((n,), float(i), device=x.device)
            xt = xs[-1]
            if "untill_fake_t" in kwargs and cnt == kwargs["untill_fake_t"] - 1:
                break
            if kwargs.get("tot") is not None:
                model.load_state_dict(kwargs["cali_ckpt"][f"act_{cnt}"], strict=False)
            et = model(xt, t)
            noise = torch.randn_like(xt) if kwargs.get("eta", 0) else None
            xn, x0 = ops.ddim_update(xt.contiguous(), et.contiguous(), coef[cnt:cnt + 1].contiguous(), noise=noise, want_x0=True)
            x0_preds.append(x0)
            xs.append(xn)
    return xs, x0_preds, xt, t


def check_fsc_rows(engine, n_steps: int, who: str) -> None:
    """A Finite-Set-Calibration table with G > 1 rows is indexed by the device step counter 0..n_steps-1
    (csrc/common.hpp load_qparam, no bound check on the device): a checkpoint calibrated for fewer timestep groups than
    the sampling run would read past the table.  The reference raises KeyError on the missing `act_k` there."""
    qt = getattr(engine, "qtable", None)
    if qt is not None and qt.shape[0] > 1 and qt.shape[0] < n_steps:
        raise TfmqError(f"{who}: the activation table holds {qt.shape[0]} timestep groups but the sampler runs {n_steps} steps "
                        "(calibrate with one group per sampling step, or install a matching table)")


STREAM_GUARD_REL_L2 = 0.25     # eps of the two streams further apart than this: the fp16 stream lost something (bin flips alone: 2 ... 3 %)


def streams_disagree(pairs) -> bool:
    """The guard's decision over [(eps with the fp16 stream, eps with the fp32 stream), ...] of the probed steps (pure function): an fp16
    result that is not finite, or further than STREAM_GUARD_REL_L2 from a finite fp32 result, at ANY probed step."""
    for e16, e32 in pairs:
        if not bool(torch.isfinite(e16).all()):
            return True
        if bool(torch.isfinite(e32).all()) and float((e16 - e32).norm() / e32.norm().clamp_min(1e-30)) > STREAM_GUARD_REL_L2:
            return True
    return False


def fp16_stream_overflowed(sampler, probe, probe_last=None) -> bool:
    """The fp16 activation stream stores the tensors that travel between blocks as fp16: a checkpoint whose
    residual stream exceeds 65504 somewhere turns into inf there -- and an inf that meets an activation quantizer is clamped to the top
    bin, i.e. the result can be finite and wrong -- where the reference's fp32 stream is fine.  After the FIRST sampling of a graph
    sampler the UNet is therefore evaluated eagerly with the fp16 and with the fp32 stream at BOTH ENDS of the trajectory: `probe()` = the
    first step on x_T, `probe_last()` = the last step (its time embedding, its row of the activation table) on the latents the sampling
    ended with -- activation ranges drift along the trajectory, which is why the table has a row per step.  Non-finite latents, or eps
    further apart than 25 % rel-L2 at either end, switch the engine to the fp32 stream, drop the captured graphs and tell the caller to
    sample again.  Later samplings are not checked (overflow is a property of the weights and the schedule, not of the noise).
    TFMQ_STREAM_GUARD=0 switches the check off."""
    if getattr(sampler, "_stream_checked", False) or os.environ.get("TFMQ_STREAM_GUARD", "1") == "0":
        return False
    sampler._stream_checked = True
    eng = sampler.eng
    sampler.stream.synchronize()
    finite = bool(torch.isfinite(sampler.x).all())
    if not getattr(eng, "stream_f16", False):
        if finite:
            return False
        # the fp32 stream is already on: what overflows is an fp16 OPERAND of an un-quantised conv / attention (they round their inputs
        # to fp16 for the matrix cores).  Loud, not silent; the exact-fp32 engine mode runs those layers in fp32.
        raise TfmqError("non-finite latents with the fp32 activation stream: an un-quantised layer's input exceeds the fp16 operand range "
                        "(or the model itself diverges); TFMQ_EXACT_FP=1 runs the un-quantised layers on exact-fp32 GEMMs")
    if finite:
        with torch.cuda.stream(sampler.stream), ops.use_arena(None):
            step_after = sampler.step.clone()
            pairs = []
            n_steps = int(getattr(sampler, "n_steps", 0) or sampler.coef.shape[0])
            for k, fn in ((0, probe), (n_steps - 1, probe_last)):
                if fn is None:
                    continue
                sampler.step.fill_(k)
                e16 = fn().float().clone()
                eng.stream_f16 = False
                try:
                    e32 = fn().float().clone()
                finally:
                    eng.stream_f16 = True
                pairs.append((e16, e32))
            sampler.step.copy_(step_after)          # (callers read the device step counter after a sampling)
            sampler.stream.synchronize()
        if not streams_disagree(pairs):
            return False
    import warnings
    warnings.warn("tfmq: the fp16 activation stream overflows on this checkpoint -- falling back to the fp32 stream (TFMQ_STREAM_F32=1 selects it up front)")
    eng.stream_f16 = False
    sampler.arena = ops.Arena()
    sampler.gid = None
    if hasattr(sampler, "gids"):
        sampler.gids = None
    return True


class GraphDdimSampler:
    """DDIM loop over a prepared DdimUNetEngine, one hipGraph replay per step."""

    def __init__(self, engine, seq: Sequence[int], betas: torch.Tensor, batch: int, eta: float = 0.0):
        if eta != 0.0:
            raise TfmqError("GraphDdimSampler: eta != 0 needs a device RNG stream (not wired yet); use generalized_steps")
        self.eng, self.seq, self.batch = engine, list(seq), batch
        self.dev = engine.dev
        self.n_steps = len(self.seq)
        self.coef = coef_table(self.seq, betas, eta).to(self.dev)
        if engine.step is None:
            raise TfmqError("GraphDdimSampler: engine.prepare() needs a device step counter")
        self.step = engine.step
        check_fsc_rows(engine, self.n_steps, "GraphDdimSampler")
        engine.build_tib_table([float(i) for i in reversed(self.seq)])
        cfg = engine.cfg
        self.x = torch.empty(batch, cfg["resolution"], cfg["resolution"], cfg.get("in_channels", 3), device=self.dev)
        self.stream = torch.cuda.Stream(self.dev)
        self.arena = ops.Arena()
        self.h = handle(self.x.device.index)      # (the tensor's device is concrete even when the engine was given a bare "cuda")
        self.gid = None

    def _step_body(self):
        eps = self.eng.forward(self.x, None)
        ops.ddim_update(self.x, eps, self.coef, self.step, out=self.x)
        ops.step_advance(self.step, 1)

    def capture(self):
        sp = C.c_void_p(self.stream.cuda_stream)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))      # inputs / set-up produced on the caller's stream
        with torch.cuda.stream(self.stream):
            self.step.zero_()
            # the warm-up pass allocates every intermediate once and times the tile variants of every conv / linear shape
            # (ops.set_conv_autotune); the captured pass replays the allocation log and pins the winners
            if not hasattr(self.eng, "tiles"):
                self.eng.tiles = {}
            self.tiles = self.eng.tiles          # one cache per engine: eager forwards and the captured graph agree
            ops.set_conv_autotune(self.tiles)
            try:
                with ops.use_arena(self.arena):
                    self._step_body()
                self.stream.synchronize()
                with ops.use_arena(self.arena):
                    self.h.call("graph_begin", sp)
                    self._step_body()
                    gid = C.c_int()
                    self.h.call("graph_end", sp, C.byref(gid))
            finally:
                ops.set_conv_autotune(None)
            self.gid = gid.value
            if os.environ.get("TFMQ_TUNE_REPORT"):
                import collections, sys
                print("[tfmq] tile selection:", dict(collections.Counter(ops.tile_name(v) for v in self.tiles.values())),
                      file=sys.stderr)
                if os.environ["TFMQ_TUNE_REPORT"] == "2":
                    for k, v in self.tiles.items():
                        print("   ", k, ops.tile_name(v), file=sys.stderr)
        return self

    def sample_nhwc(self, x_T: torch.Tensor, steps: Optional[int] = None) -> torch.Tensor:
        """x_T: [B,H,W,C] fp32 on the device -> x_0 (same layout, a view of the sampler's buffer)."""
        if self.gid is None:
            self.capture()
        sp = C.c_void_p(self.stream.cuda_stream)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))      # inputs / set-up produced on the caller's stream
        with torch.cuda.stream(self.stream):
            self.x.copy_(x_T, non_blocking=True)
            self.step.zero_()
            sync_every = int(os.environ.get("TFMQ_GRAPH_SYNC_EVERY", "16"))   # bounded host run-ahead (ldm/sampler.py)
            for i in range(self.n_steps if steps is None else steps):
                self.h.call("graph_launch", self.gid, sp)
                if sync_every and (i + 1) % sync_every == 0:
                    self.stream.synchronize()
        if fp16_stream_overflowed(self, lambda: self.eng.forward(x_T.float().contiguous(), None), lambda: self.eng.forward(self.x.float().clone(), None)):
            return self.sample_nhwc(x_T, steps)
        return self.x

    def sample(self, x_T_nchw: torch.Tensor) -> torch.Tensor:
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))      # inputs / set-up produced on the caller's stream
        with torch.cuda.stream(self.stream):
            xin = ops.nchw_to_nhwc(x_T_nchw.contiguous())
        out = self.sample_nhwc(xin)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))      # inputs / set-up produced on the caller's stream
        with torch.cuda.stream(self.stream):
            y = ops.nhwc_to_nchw(out)
        self.stream.synchronize()
        return y

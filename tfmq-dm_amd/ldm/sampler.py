"""Latent DDIM sampling with classifier-free guidance on the HIP engine.

Host side = schedule constants exactly as the reference builds them (make_beta_schedule /
register_schedule ldm/models/diffusion/ddpm.py:117-131, make_ddim_timesteps /
make_ddim_sampling_parameters ldm/modules/diffusionmodules/util.py:46-74); device side = one hipGraph
replay per step: duplicate the latent for the (uncond, cond) halves, UNet forward on the batch-2B plan with
the step's Finite-Set activation table (k = t_max - (t-1)//tot == step counter), CFG combine +
DDIM update in one kernel, step counter increment (DDIMSampler.ddim_sampling / p_sample_ddim,
ldm/models/diffusion/ddim.py:118-212)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

from .. import ops
from .._lib import TfmqError, handle
from ..ddim.sampler import check_fsc_rows, fp16_stream_overflowed


def alphas_cumprod_linear(linear_start: float = 0.00085, linear_end: float = 0.012, n: int = 1000) -> torch.Tensor:
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=torch.float64) ** 2).numpy()
    return torch.tensor(np.cumprod(1.0 - betas, axis=0), dtype=torch.float32)


def ddim_timesteps(S: int, n: int = 1000) -> np.ndarray:
    return np.asarray(list(range(0, n, n // S))) + 1


def ddim_coef_table(alphas_cumprod: torch.Tensor, S: int, eta: float = 0.0) -> torch.Tensor:
    """Row i (i-th executed step) = {sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sigma, sqrt(1-a_prev-sigma^2), t, 0, 0}
    with the fp32 tensor arithmetic of p_sample_ddim."""
    ts = ddim_timesteps(S, alphas_cumprod.shape[0])
    a = alphas_cumprod[ts]
    a_prev = torch.tensor([float(alphas_cumprod[0])] + alphas_cumprod[ts[:-1]].tolist(), dtype=torch.float32)
    sig = eta * torch.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))
    rows = []
    n = len(ts)          # S' = ceil(1000 / (1000 // S)) executed steps: S when S divides the schedule, else more
    for i, step in enumerate(np.flip(ts)):
        idx = n - i - 1
        a_t, ap, sg = a[idx].reshape(1), a_prev[idx].reshape(1), sig[idx].reshape(1).float()
        rows.append(torch.cat([torch.sqrt(1.0 - a[idx]).reshape(1), a_t.sqrt(), ap.sqrt(), sg, (1.0 - ap - sg ** 2).sqrt(),
                               torch.tensor([float(step)]), torch.zeros(2)]))
    return torch.stack(rows).float().contiguous()


class GraphLatentDdimSampler:
    def __init__(self, engine, S: int, batch: int, latent_shape, context_shape, scale: float = 7.5,
                 alphas_cumprod: Optional[torch.Tensor] = None, eta: float = 0.0):
        self.eng, self.S, self.batch, self.scale = engine, S, batch, float(scale)
        self.eta = float(eta)
        self.dev = engine.dev
        ac = alphas_cumprod if alphas_cumprod is not None else alphas_cumprod_linear()
        self.coef = ddim_coef_table(ac, S, eta).to(self.dev)
        if engine.step is None:
            raise TfmqError("GraphLatentDdimSampler: engine.prepare() needs a device step counter")
        self.step = engine.step
        check_fsc_rows(engine, self.coef.shape[0], type(self).__name__)
        engine.build_tib_table([float(t) for t in np.flip(ddim_timesteps(S, ac.shape[0]))])
        Cc, H, W = latent_shape
        self.x = torch.empty(batch, H, W, Cc, device=self.dev)
        # eta > 0 (p_sample_ddim's sigma_t * noise term, ddim.py:173-212): the captured update reads this buffer; every replay is preceded by a
        # fresh draw into it on the sampler's stream, in the host loop's order (one torch.randn of the NCHW shape per step)
        self.noise = torch.empty(batch, H, W, Cc, device=self.dev) if self.eta > 0.0 else None
        self.nchw = (batch, Cc, H, W)
        # context_shape None: unconditional LDM (CelebA-HQ / LSUN configs, sample_diffusion_ldm.py): one UNet call on the
        # batch itself per step, plain DDIM update -- no guidance pair
        self.uncond = context_shape is None
        self.x2 = None if self.uncond else torch.empty(2 * batch, H, W, Cc, device=self.dev)
        self.ctx2 = None if self.uncond else torch.empty((2 * batch,) + tuple(context_shape), device=self.dev)
        self.pair_prefix = (not self.uncond) and os.environ.get("TFMQ_PAIR_PREFIX", "1") != "0" and hasattr(engine, "_dup")
        self.stream = torch.cuda.Stream(self.dev)
        self.arena = ops.Arena()
        self.h = handle(self.x.device.index)      # (the tensor's device is concrete even when the engine was given a bare "cuda")
        self.gid = None

    def _step_body(self):
        B = self.batch
        if self.uncond:
            eps = self.eng.forward(self.x, None, None)
            ops.ddim_update(self.x, eps, self.coef, self.step, self.noise, out=self.x)
            ops.step_advance(self.step, 1)
            return
        eps2 = self._eps_pair(self.x)
        ops.ddim_update_cfg(self.x, eps2[:B], eps2[B:], self.scale, self.coef, self.step, noise=self.noise, out=self.x)
        ops.step_advance(self.step, 1)

    def _eps_pair(self, xin):
        """eps of the guidance pair cat([x] * 2) under cat([uc, c]) (ddim.py:180-186).  The engine computes what the two members share
        -- everything in front of the first cross attention -- once (`pair_prefix`, TFMQ_PAIR_PREFIX=0: the materialised 2B batch)."""
        B = self.batch
        if self.pair_prefix:
            return self.eng.forward(xin, None, self.ctx2, pair_prefix=True)
        self.x2[:B].copy_(xin)
        self.x2[B:].copy_(xin)
        return self.eng.forward(self.x2, None, self.ctx2)

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

    def sample_nhwc(self, x_T: torch.Tensor, cond: Optional[torch.Tensor] = None, uncond: Optional[torch.Tensor] = None,
                    steps: Optional[int] = None):
        """x_T [B,H,W,C]; cond/uncond [B,L,D] (CLIP / class embeddings: glue, computed elsewhere; None for an
        unconditional sampler)."""
        if self.gid is None:
            self.capture()
        sp = C.c_void_p(self.stream.cuda_stream)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))      # inputs / set-up produced on the caller's stream
        with torch.cuda.stream(self.stream):
            self.x.copy_(x_T, non_blocking=True)
            if not self.uncond:
                self.ctx2[:self.batch].copy_(uncond, non_blocking=True)   # c_in = cat[uc, c]  (ddim.py:183)
                self.ctx2[self.batch:].copy_(cond, non_blocking=True)
            self.step.zero_()
            # at most 8 step graphs (~10 k kernel dispatches) are enqueued ahead of the GPU: an unbounded run-ahead of
            # the host buys nothing and overflows rocprofv3's dispatch records at large batches
            sync_every = int(os.environ.get("TFMQ_GRAPH_SYNC_EVERY", "8"))
            for i in range(self.coef.shape[0] if steps is None else steps):
                if self.noise is not None:
                    self.noise.copy_(ops.nchw_to_nhwc(torch.randn(self.nchw, device=self.dev)))
                self.h.call("graph_launch", self.gid, sp)
                if sync_every and (i + 1) % sync_every == 0:
                    self.stream.synchronize()
        if fp16_stream_overflowed(self, lambda: (self.eng.forward(x_T.float().contiguous(), None, None) if self.uncond
                                                 else self._eps_pair(x_T.float().contiguous())),
                                  lambda: (self.eng.forward(self.x.float().clone(), None, None) if self.uncond else self._eps_pair(self.x.float().clone()))):
            return self.sample_nhwc(x_T, cond, uncond, steps)
        return self.x


class GraphLatentPlmsSampler(GraphLatentDdimSampler):
    """PLMS (pseudo linear multi-step, Adams-Bashforth orders 1-4: PLMSSampler.plms_sampling / p_sample_plms,
    ldm/models/diffusion/plms.py:119-242 -- the sampler of the README's Stable Diffusion recipe) with classifier-free
    guidance, as hipGraph replays.  Four captured graphs share one activation arena:

      first : e_t at step 0; x' = DDIM update; e_next at step 1 (pseudo improved Euler, the one extra UNet call);
              e' = (e_t + e_next)/2; x <- update(x, e') with row 0 of the schedule.  The step counter ends at 1.
      2, 3  : e_t; e' = (3 e_t - h1)/2, (23 e_t - 16 h1 + 5 h2)/12;  x <- update(x, e'); rotate the history; step += 1
      n     : e' = (55 e_t - 59 h1 + 37 h2 - 9 h3)/24; ...

    The kernels (cfg_combine, plms_combine, ddim_update) and their order are those of the drop-in PLMSSampler
    (ldm/ddim.py), so the two agree bit for bit; the Finite-Set activation group of every UNet call is the device step
    counter, which equals DiffusionWrapper's k = t_max - (t-1)//tot for t and for t_next alike."""

    def __init__(self, engine, S: int, batch: int, latent_shape, context_shape, scale: float = 7.5,
                 alphas_cumprod: Optional[torch.Tensor] = None):
        super().__init__(engine, S, batch, latent_shape, context_shape, scale, alphas_cumprod, 0.0)
        if self.coef.shape[0] < 2:
            raise TfmqError("GraphLatentPlmsSampler needs at least two steps")
        mk = lambda: torch.empty_like(self.x)
        self.e_t, self.e_next, self.e_prime, self.xprev = mk(), mk(), mk(), mk()
        self.hist = [mk(), mk(), mk()]          # previous model outputs, newest first
        self.gids = None

    def _eps(self, xin, dst):
        B = self.batch
        with ops.use_arena(self.arena):           # every UNet call replays the same allocation log
            eps2 = self._eps_pair(xin)
            ops.cfg_combine(eps2[:B], eps2[B:], self.scale, out=dst)

    def _rotate(self):
        h = self.hist
        h[2].copy_(h[1])
        h[1].copy_(h[0])
        h[0].copy_(self.e_t)

    def _body(self, kind: int):
        h = self.hist
        if kind == 1:
            self._eps(self.x, self.e_t)
            ops.ddim_update(self.x, self.e_t, self.coef[0:1], None, None, out=self.xprev)
            ops.step_advance(self.step, 1)
            self._eps(self.xprev, self.e_next)
            ops.plms_combine(1, self.e_t, self.e_next, out=self.e_prime)
            ops.ddim_update(self.x, self.e_prime, self.coef[0:1], None, None, out=self.x)
            h[0].copy_(self.e_t)
            return
        self._eps(self.x, self.e_t)
        if kind == 2:
            ops.plms_combine(2, self.e_t, h[0], out=self.e_prime)
        elif kind == 3:
            ops.plms_combine(3, self.e_t, h[0], h[1], out=self.e_prime)
        else:
            ops.plms_combine(4, self.e_t, h[0], h[1], h[2], out=self.e_prime)
        ops.ddim_update(self.x, self.e_prime, self.coef, self.step, None, out=self.x)
        self._rotate()
        ops.step_advance(self.step, 1)

    def capture(self):
        sp = C.c_void_p(self.stream.cuda_stream)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))      # inputs / set-up produced on the caller's stream
        with torch.cuda.stream(self.stream):
            self.step.zero_()
            if not hasattr(self.eng, "tiles"):
                self.eng.tiles = {}
            self.tiles = self.eng.tiles
            ops.set_conv_autotune(self.tiles)
            try:
                self._body(1)                     # eager: records the arena (first UNet call) and measures the tiles
                self.stream.synchronize()
                gids = []
                for kind in (1, 2, 3, 4):
                    self.h.call("graph_begin", sp)
                    self._body(kind)
                    gid = C.c_int()
                    self.h.call("graph_end", sp, C.byref(gid))
                    gids.append(gid.value)
            finally:
                ops.set_conv_autotune(None)
            self.gids = gids
            self.gid = gids[0]
        return self

    def sample_nhwc(self, x_T: torch.Tensor, cond: torch.Tensor, uncond: torch.Tensor, steps: Optional[int] = None):
        """steps: run only the first `steps` iterations (untill_fake_t - 1 of the drop-in)."""
        if self.gids is None:
            self.capture()
        sp = C.c_void_p(self.stream.cuda_stream)
        n = self.coef.shape[0] if steps is None else int(steps)
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))      # inputs / set-up produced on the caller's stream
        with torch.cuda.stream(self.stream):
            self.x.copy_(x_T, non_blocking=True)
            self.ctx2[:self.batch].copy_(uncond, non_blocking=True)
            self.ctx2[self.batch:].copy_(cond, non_blocking=True)
            self.step.zero_()
            sync_every = int(os.environ.get("TFMQ_GRAPH_SYNC_EVERY", "8"))
            for i in range(n):
                self.h.call("graph_launch", self.gids[min(i, 3)], sp)
                if sync_every and (i + 1) % sync_every == 0:
                    self.stream.synchronize()
        if fp16_stream_overflowed(self, lambda: (self.eng.forward(x_T.float().contiguous(), None, None) if self.uncond
                                                 else self._eps_pair(x_T.float().contiguous())),
                                  lambda: (self.eng.forward(self.x.float().clone(), None, None) if self.uncond else self._eps_pair(self.x.float().clone()))):
            return self.sample_nhwc(x_T, cond, uncond, steps)
        return self.x

"""Drop-in `DDIMSampler` / `PLMSSampler` (reference ldm/models/diffusion/ddim.py:14-212, plms.py:12-242): same
constructor, `make_schedule`, `sample(...)` keywords (incl. the `untill_fake_t` early stop the calibration-set
generators rely on) and return values.  The loop stays on the host like the reference's; every arithmetic step is a HIP
kernel (CFG combine, Adams-Bashforth combine, DDIM update) and the UNet call is whatever `model.apply_model` lowers to
(QuantModel -> engine plan).  The throughput path is `ldm/sampler.py: GraphLatentDdimSampler` (one hipGraph per step)."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .. import ops
from .._lib import TfmqError
from .sampler import ddim_coef_table, ddim_timesteps


class DDIMSampler:
    def __init__(self, model, schedule: str = "linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def make_schedule(self, ddim_num_steps: int, ddim_discretize: str = "uniform", ddim_eta: float = 0.0, verbose: bool = True):
        if ddim_discretize != "uniform":
            raise TfmqError("make_schedule: only the 'uniform' discretisation is used by the drivers")
        ac = self.model.alphas_cumprod.detach().float().cpu()
        self.ddim_timesteps = ddim_timesteps(ddim_num_steps, self.ddpm_num_timesteps)
        self.ddim_eta = float(ddim_eta)
        # row i = i-th executed step: {sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sigma, sqrt(1-a_prev-sigma^2), t, 0, 0}
        self._coef = ddim_coef_table(ac, ddim_num_steps, ddim_eta).to(self.model.alphas_cumprod.device)

    # -------------------------------------------------------------------------------------------- model output
    def _eps(self, x, t, c, scale, uc):
        if uc is None or scale == 1.0:
            return self.model.apply_model(x, t, c).contiguous()
        e2 = self.model.apply_model(torch.cat([x] * 2), torch.cat([t] * 2), torch.cat([uc, c]))
        b = x.shape[0]
        return ops.cfg_combine(e2[:b].contiguous(), e2[b:].contiguous(), scale)

    # -------------------------------------------------------------------------------------------- graph fast path
    def _graph_sample(self, plms: bool, S: int, cond, shape, x_T, scale: float, uc, untill_fake_t, eta: float = 0.0):
        """Opt-in (`sample(..., _graph=True)`, used by the calibration-set generators, which need neither callbacks nor
        intermediates): the same recurrence as the host loop below, replayed as captured step graphs
        (ldm/sampler.py) on the engine `model.apply_model` lowers to.  Returns None when the call does not qualify
        (guidance-free conditional sampling, PLMS with Finite-Set wrapper attributes set, model not lowered to an engine).  eta > 0
        (DDIM only): the noise of every step is drawn on the sampler's stream right before its replay, in the host loop's order.  DDIM with the drivers' Finite-Set attributes (wrapper.tot / t_max / ckpt): the group of step i is
        k_i = t_max - (t_i - 1) // tot (ddpm.py:1402-1405); the table installed for the replay is [act_{k_0}, act_{k_1}, ...], one
        row per executed step, indexed by the graph's device step counter."""
        from .sampler import GraphLatentDdimSampler, GraphLatentPlmsSampler
        wrapper = getattr(self.model, "model", None)
        qnn = getattr(wrapper, "diffusion_model", None)
        if qnn is None or not hasattr(qnn, "engine"):
            return None
        if hasattr(wrapper, "tot"):
            if plms or not hasattr(qnn, "set_act_table"):
                return None
            rows = tuple(int(wrapper.t_max - (int(t) - 1) // wrapper.tot) for t in np.flip(self.ddim_timesteps))
            mark = ("steps", id(wrapper.ckpt), rows)
            if getattr(wrapper, "_table_of", None) != mark:
                qnn.set_act_table(wrapper.ckpt, rows=rows)
                wrapper._table_of = mark               # (the eager wrapper.forward re-installs the group table when it runs next)
                qnn.__dict__.pop("_graph_samplers", None)
        dev = self.model.betas.device
        if dev.type != "cuda":
            return None
        guided = uc is not None and scale != 1.0
        if cond is not None and (not guided or not torch.is_tensor(cond) or cond.dim() != 3 or tuple(uc.shape) != tuple(cond.shape)):
            return None
        if cond is None and plms:
            return None
        eng = qnn.engine(dev)
        if eng.step is None:
            return None
        b, Cc, H, W = shape
        ctx_shape = None if cond is None else tuple(cond.shape[1:])
        key = (id(eng), plms, int(S), b, (Cc, H, W), ctx_shape, float(scale), float(eta))
        cache = qnn.__dict__.setdefault("_graph_samplers", {})
        smp = cache.get(key)
        if smp is None:
            cache.clear()                      # one captured plan (and its arena) at a time
            ac = self.model.alphas_cumprod.detach().float().cpu()
            if plms:
                smp = GraphLatentPlmsSampler(eng, S, b, (Cc, H, W), ctx_shape, scale=scale, alphas_cumprod=ac)
            else:
                smp = GraphLatentDdimSampler(eng, S, b, (Cc, H, W), ctx_shape, scale=scale, alphas_cumprod=ac, eta=eta)
            smp.capture()
            cache[key] = smp
        img = (torch.randn(shape, device=dev) if x_T is None else x_T.to(dev)).float().contiguous()
        total = smp.coef.shape[0]
        n = total if untill_fake_t == float("inf") else max(0, min(total, int(untill_fake_t) - 1))
        out = smp.sample_nhwc(ops.nchw_to_nhwc(img), cond, uc, steps=n)
        smp.stream.synchronize()
        res = ops.nhwc_to_nchw(out).clone()
        eng.step.zero_()
        return res, {"x_inter": [img, res], "pred_x0": [img]}

    def _check(self, mask, x0, quantize_x0, score_corrector, temperature, noise_dropout):
        if mask is not None or x0 is not None or quantize_x0 or score_corrector is not None or noise_dropout != 0.0 \
                or temperature != 1.0:
            raise TfmqError("sampler: inpainting masks, x0 quantisation, score correctors, temperature and noise dropout "
                            "are not used by the TFMQ-DM drivers")

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0.0, mask=None, x0=None, temperature=1.0, noise_dropout=0.0, score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.0,
               unconditional_conditioning=None, untill_fake_t: Optional[int] = None, **kwargs):
        self._check(mask, x0, quantize_x0, score_corrector, temperature, noise_dropout)
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        if not untill_fake_t:
            untill_fake_t = float("inf")
        Cc, H, W = shape
        if kwargs.get("_graph") and callback is None and img_callback is None:
            r = self._graph_sample(False, S, conditioning, (batch_size, Cc, H, W), x_T, float(unconditional_guidance_scale),
                                   unconditional_conditioning, untill_fake_t, float(eta))
            if r is not None:
                return r
        return self.ddim_sampling(conditioning, (batch_size, Cc, H, W), x_T=x_T, callback=callback, img_callback=img_callback,
                                  log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, untill_fake_t=untill_fake_t)

    @torch.no_grad()
    def ddim_sampling(self, cond, shape, x_T=None, callback=None, img_callback=None, log_every_t=100,
                      unconditional_guidance_scale=1.0, unconditional_conditioning=None, untill_fake_t=float("inf"), **kw):
        dev = self.model.betas.device
        if dev.type != "cuda":
            raise TfmqError("DDIMSampler: the model is not on an MI355X device (no CPU fallback)")
        b = shape[0]
        img = (torch.randn(shape, device=dev) if x_T is None else x_T.to(dev)).float().contiguous()
        total = self.ddim_timesteps.shape[0]
        inter = {"x_inter": [img], "pred_x0": [img]}
        for i, step in enumerate(np.flip(self.ddim_timesteps)):
            if i == untill_fake_t - 1:
                break
            index = total - i - 1
            ts = torch.full((b,), int(step), device=dev, dtype=torch.long)
            e_t = self._eps(img, ts, cond, float(unconditional_guidance_scale), unconditional_conditioning)
            noise = torch.randn(shape, device=dev) if self.ddim_eta > 0.0 else None
            img, pred_x0 = ops.ddim_update(img, e_t, self._coef[i:i + 1], None, noise, want_x0=True)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total - 1:
                inter["x_inter"].append(img)
                inter["pred_x0"].append(pred_x0)
        return img, inter


class PLMSSampler(DDIMSampler):
    """Pseudo linear multi-step sampler (Adams-Bashforth orders 1-4; reference plms.py).  eta must be 0."""

    def make_schedule(self, ddim_num_steps: int, ddim_discretize: str = "uniform", ddim_eta: float = 0.0, verbose: bool = True):
        if ddim_eta != 0:
            raise ValueError("ddim_eta must be 0 for PLMS")
        super().make_schedule(ddim_num_steps, ddim_discretize, 0.0, verbose)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0.0, mask=None, x0=None, temperature=1.0, noise_dropout=0.0, score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.0,
               unconditional_conditioning=None, untill_fake_t: Optional[int] = None, **kwargs):
        self._check(mask, x0, quantize_x0, score_corrector, temperature, noise_dropout)
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        if not untill_fake_t:
            untill_fake_t = float("inf")
        Cc, H, W = shape
        if kwargs.get("_graph") and callback is None and img_callback is None:
            r = self._graph_sample(True, S, conditioning, (batch_size, Cc, H, W), x_T, float(unconditional_guidance_scale),
                                   unconditional_conditioning, untill_fake_t)
            if r is not None:
                return r
        return self.plms_sampling(conditioning, (batch_size, Cc, H, W), x_T=x_T, callback=callback, img_callback=img_callback,
                                  log_every_t=log_every_t, unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, untill_fake_t=untill_fake_t)

    @torch.no_grad()
    def plms_sampling(self, cond, shape, x_T=None, callback=None, img_callback=None, log_every_t=100,
                      unconditional_guidance_scale=1.0, unconditional_conditioning=None, untill_fake_t=float("inf"), **kw):
        dev = self.model.betas.device
        if dev.type != "cuda":
            raise TfmqError("PLMSSampler: the model is not on an MI355X device (no CPU fallback)")
        b = shape[0]
        img = (torch.randn(shape, device=dev) if x_T is None else x_T.to(dev)).float().contiguous()
        time_range = np.flip(self.ddim_timesteps)
        total = time_range.shape[0]
        inter = {"x_inter": [img], "pred_x0": [img]}
        scale, uc = float(unconditional_guidance_scale), unconditional_conditioning
        old_eps = []
        for i, step in enumerate(time_range):
            if i == untill_fake_t - 1:
                break
            index = total - i - 1
            ts = torch.full((b,), int(step), device=dev, dtype=torch.long)
            ts_next = torch.full((b,), int(time_range[min(i + 1, total - 1)]), device=dev, dtype=torch.long)
            coef = self._coef[i:i + 1]
            e_t = self._eps(img, ts, cond, scale, uc)
            if len(old_eps) == 0:       # pseudo improved Euler: one extra model call at t_next
                x_prev = ops.ddim_update(img, e_t, coef, None, None)
                e_next = self._eps(x_prev, ts_next, cond, scale, uc)
                e_prime = ops.plms_combine(1, e_t, e_next)
            elif len(old_eps) == 1:
                e_prime = ops.plms_combine(2, e_t, old_eps[-1])
            elif len(old_eps) == 2:
                e_prime = ops.plms_combine(3, e_t, old_eps[-1], old_eps[-2])
            else:
                e_prime = ops.plms_combine(4, e_t, old_eps[-1], old_eps[-2], old_eps[-3])
            img, pred_x0 = ops.ddim_update(img, e_prime, coef, None, None, want_x0=True)
            old_eps.append(e_t)
            if len(old_eps) >= 4:
                old_eps.pop(0)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total - 1:
                inter["x_inter"].append(img)
                inter["pred_x0"].append(pred_x0)
        return img, inter

"""Drop-in `DPMSolverSampler` (reference ldm/models/diffusion/dpm_solver/sampler.py + the part of dpm_solver.py its
`sample()` exercises): DPM-Solver++ with data prediction, multistep order 2, uniform time steps, lower-order final steps,
classifier-free guidance, discrete VP noise schedule, and the `untill_fake_t` early stop of the calibration-set
generators.  Scalars (lambda, alpha, sigma, h) are computed on the host in fp32 torch arithmetic like the reference's
0-d / [B] tensors; the per-element updates are HIP kernels."""
from __future__ import annotations

from typing import Optional

import torch

from .. import ops
from .._lib import TfmqError


class NoiseScheduleVP:
    """'discrete' schedule of the reference (dpm_solver.py:7-175): log alpha_t at t_i = (i+1)/N, piecewise-linear in t."""

    def __init__(self, schedule: str = "discrete", betas=None, alphas_cumprod=None):
        if schedule != "discrete":
            raise TfmqError("NoiseScheduleVP: only the discrete schedule is used by the samplers")
        if betas is not None:
            log_alphas = 0.5 * torch.log(1 - betas.float().cpu()).cumsum(dim=0)
        else:
            log_alphas = 0.5 * torch.log(alphas_cumprod.float().cpu())
        self.schedule = schedule
        self.total_N = len(log_alphas)
        self.T = 1.0
        self.t_array = torch.linspace(0.0, 1.0, self.total_N + 1)[1:]
        self.log_alpha_array = log_alphas

    def marginal_log_mean_coeff(self, t: torch.Tensor) -> torch.Tensor:
        xp, yp = self.t_array, self.log_alpha_array
        t = t.reshape(-1).float()
        idx = torch.searchsorted(xp, t).clamp(1, self.total_N - 1)
        x0, x1, y0, y1 = xp[idx - 1], xp[idx], yp[idx - 1], yp[idx]
        return y0 + (t - x0) * (y1 - y0) / (x1 - x0)

    def marginal_alpha(self, t):
        return torch.exp(self.marginal_log_mean_coeff(t))

    def marginal_std(self, t):
        return torch.sqrt(1.0 - torch.exp(2.0 * self.marginal_log_mean_coeff(t)))

    def marginal_lambda(self, t):
        lm = self.marginal_log_mean_coeff(t)
        return lm - 0.5 * torch.log(1.0 - torch.exp(2.0 * lm))


class DPMSolverSampler:
    def __init__(self, model, **kwargs):
        self.model = model
        self.alphas_cumprod = model.alphas_cumprod.detach().float()

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0.0, mask=None, x0=None, temperature=1.0, noise_dropout=0.0, score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.0,
               unconditional_conditioning=None, untill_fake_t: Optional[int] = None, **kwargs):
        if mask is not None or x0 is not None or quantize_x0 or score_corrector is not None:
            raise TfmqError("DPMSolverSampler: masks / x0 quantisation / score correctors are not used by the drivers")
        if not untill_fake_t:
            untill_fake_t = float("inf")
        dev = self.model.betas.device
        if dev.type != "cuda":
            raise TfmqError("DPMSolverSampler: the model is not on an MI355X device (no CPU fallback)")
        Cc, H, W = shape
        x = (torch.randn((batch_size, Cc, H, W), device=dev) if x_T is None else x_T.to(dev)).float().contiguous()
        ns = NoiseScheduleVP("discrete", alphas_cumprod=self.alphas_cumprod)
        scale, uc, cond = float(unconditional_guidance_scale), unconditional_conditioning, conditioning
        b = batch_size

        def data_prediction(xx, tc: torch.Tensor):
            """x0 prediction at continuous time tc (0-d fp32): CFG noise prediction -> (x - sigma eps) / alpha."""
            t_in = ((tc - 1.0 / ns.total_N) * 1000.0).to(dev).expand(b).contiguous()
            if scale == 1.0 or uc is None:
                eps = self.model.apply_model(xx, t_in, cond).contiguous()
            else:
                e2 = self.model.apply_model(torch.cat([xx] * 2), torch.cat([t_in] * 2), torch.cat([uc, cond]))
                eps = ops.cfg_combine(e2[:b].contiguous(), e2[b:].contiguous(), scale)
            return ops.dpm_x0(xx, eps, float(ns.marginal_std(tc)), float(ns.marginal_alpha(tc)))

        def update(xx, models, times, t, order):
            """multistep_dpm_solver_update with predict_x0 (orders 1, 2)."""
            s = times[-1]
            lam_s, lam_t = ns.marginal_lambda(s), ns.marginal_lambda(t)
            sig_s, sig_t = ns.marginal_std(s), ns.marginal_std(t)
            alpha_t = torch.exp(ns.marginal_log_mean_coeff(t))
            h = lam_t - lam_s
            c_x = sig_t / sig_s
            c_m = alpha_t * (torch.exp(-h) - 1.0)
            if order == 1:
                return ops.dpm_update(1, xx, models[-1], None, float(c_x), float(c_m))
            lam_p = ns.marginal_lambda(times[-2])
            r0 = (lam_s - lam_p) / h
            return ops.dpm_update(2, xx, models[-1], models[-2], float(c_x), float(c_m), float(0.5 * c_m), float(1.0 / r0))

        order = 2
        steps = int(S)
        if steps < order:
            raise TfmqError("DPMSolverSampler: steps must be >= 2")
        timesteps = torch.linspace(ns.T, 1.0 / ns.total_N, steps + 1)          # skip_type 'time_uniform'
        vec_t = timesteps[0]
        models, times = [data_prediction(x, vec_t)], [vec_t]
        stop = untill_fake_t == 1
        for init_order in range(1, order):          # lower-order start
            if stop:
                break
            vec_t = timesteps[init_order]
            x = update(x, models, times, vec_t, init_order)
            if init_order >= untill_fake_t - 1:
                stop = True
                break
            models.append(data_prediction(x, vec_t))
            times.append(vec_t)
        for step in range(order, steps + 1):
            if stop:
                break
            vec_t = timesteps[step]
            step_order = min(order, steps + 1 - step) if steps < 15 else order      # lower_order_final
            x = update(x, models, times, vec_t, step_order)
            models[0], times[0] = models[1], times[1]
            times[-1] = vec_t
            if step >= untill_fake_t - 1:
                break
            if step < steps:
                models[-1] = data_prediction(x, vec_t)
        return x, vec_t.to(dev).expand(b)

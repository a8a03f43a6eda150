"""Driver flows of the reference's latent-diffusion scripts, as a thin runner in the style of ddim/runner.py:

    flow "uncond"  sample_diffusion_ldm.py:445-565 + make_convolutional_sample :110-150   (LDM-4 CelebA-HQ, BASELINE configs[2])
    flow "class"   latent_imagenet_diffusion.py:190-341                                   (LDM ImageNet-256, configs[4])
    flow "text"    txt2img.py:381-598                                                     (Stable Diffusion v1-4, configs[3])

What the three scripts share, and what this class restates:

  quantize()      opt.ptq: wq / aq parameter dictionaries (MSE scalers when calibrating, MINMAX when loading), q_mode
                  [NORMAL, QDIFF], asym, running_stat -- then EITHER `load_cali_model` from opt.cali_ckpt and, with opt.use_aq,
                  the Finite-Set hooks on the DiffusionWrapper (tot = 1000 // groups, t_max = groups - 1, ckpt), OR calibration-set
                  generation by the flow's generator, `cali_model` / `mp.spawn(cali_model_multi)` and the checkpoint at
                  opt.cali_save_path (the reference exits there; quantize() returns "calibrated").
  sample_batch()  the sampler call, timed exactly as the reference times it -- t0 / t1 around the sampler, BEFORE the first-stage
                  decode (sample_diffusion_ldm.py:127-150) -- and the 'throughput' entry = batch / (t1 - t0).
  run()           batches until n_samples.

Glue stays outside: config / checkpoint loading, EMA scopes, the text / class encoders (the caller passes
conditioning tensors or an `encode` callable), image writing.  The first-stage decode runs when a decoder is attached to the model
(ldm/autoencoder.py: FirstStageDecoder)."""
from __future__ import annotations

import logging
import time
from typing import Callable, List, Optional, Sequence

import torch

from .._lib import TfmqError

logger = logging.getLogger(__name__)

FLOWS = ("uncond", "class", "text")


def _get(opt, name, default=None):
    return getattr(opt, name, default)


class LatentRunner:
    def __init__(self, model, opt, flow: str, device="cuda:0"):
        """model: ldm.ddpm.LatentDiffusion (FP UNet inside model.model.diffusion_model); opt: the script's argparse namespace
        (ptq, cali, use_aq, wq, aq, cali_ckpt, cali_save_path, softmax_a_bit, custom_steps | ddim_steps, eta | ddim_eta, plms, dpm,
        interval_length, multi_gpu, scale, ...)."""
        if flow not in FLOWS:
            raise TfmqError(f"LatentRunner: flow must be one of {FLOWS}")
        self.model, self.opt, self.flow = model, opt, flow
        self.device = torch.device(device)
        self.sampler = None

    # ------------------------------------------------------------------ helpers
    @property
    def steps(self) -> int:
        return int(_get(self.opt, "custom_steps") or _get(self.opt, "ddim_steps"))

    @property
    def eta(self) -> float:
        e = _get(self.opt, "eta")
        return float(e if e is not None else (_get(self.opt, "ddim_eta", 0.0) or 0.0))

    def latent_shape(self) -> List[int]:
        o = self.opt
        if self.flow == "text" and _get(o, "C") is not None:
            return [int(o.C), int(o.H) // int(o.f), int(o.W) // int(o.f)]
        unet = self._fp_unet()
        c = int(_get(o, "channels") or getattr(unet, "in_channels"))
        s = int(_get(o, "image_size") or getattr(unet, "image_size"))
        return [c, s, s]

    def _fp_unet(self):
        dm = self.model.model.diffusion_model
        return dm.model if hasattr(dm, "set_act_table") else dm

    def make_sampler(self):
        """--dpm / --plms / DDIM (sample_diffusion_ldm.py:131-144, txt2img.py:357-364)."""
        from .ddim import DDIMSampler, PLMSSampler
        if _get(self.opt, "dpm", False):
            from .dpm_solver import DPMSolverSampler
            self.sampler = DPMSolverSampler(self.model)
        elif _get(self.opt, "plms", False):
            self.sampler = PLMSSampler(self.model)
        else:
            self.sampler = DDIMSampler(self.model)
        return self.sampler

    # ------------------------------------------------------------------ the opt.ptq section
    def quant_params(self):
        from tfmq_dm_amd.quant.quant_layer import QMODE, Scaler
        o = self.opt
        o.q_mode = [QMODE.NORMAL.value, QMODE.QDIFF.value]
        o.asym = True
        o.running_stat = True
        scaler = Scaler.MSE if _get(o, "cali", False) else Scaler.MINMAX
        wq = {"bits": o.wq, "channel_wise": True, "scaler": scaler}
        aq = {"bits": o.aq, "channel_wise": False, "scaler": scaler, "leaf_param": bool(_get(o, "use_aq", False))}
        return wq, aq

    def _init_data(self, context: Optional[torch.Tensor]):
        """The one-sample tuple load_cali_model initialises the weight quantizers on (sample_diffusion_ldm.py:476,
        latent_imagenet_diffusion.py:215-216, txt2img.py:408)."""
        shape = self.latent_shape()
        init = (torch.randn(1, *shape), torch.randint(0, 1000, (1,)))
        if self.flow == "uncond":
            return init
        if context is None:
            if self.flow == "text":
                context = torch.randn(1, 77, 768)
            else:
                raise TfmqError("LatentRunner.quantize: the class-conditional flow needs the null-class embedding (init_context=...)")
        return init + (context[:1].detach().cpu().float(),)

    def quantize(self, init_context: Optional[torch.Tensor] = None, prompts: Optional[Sequence[str]] = None):
        """-> None (opt.ptq off), the QuantModel now installed in the wrapper (load path), or "calibrated" (opt.cali: the
        checkpoint was written; the reference's scripts exit at this point)."""
        from tfmq_dm_amd.quant.calibration import cali_model, cali_model_multi, load_cali_model
        from tfmq_dm_amd.quant.quant_model import QuantModel
        from tfmq_dm_amd.quant.reconstruction_util import RLOSS
        o = self.opt
        wq, aq = self.quant_params()
        if not _get(o, "ptq", False):
            return None
        wrapper = self.model.model
        unet = wrapper.diffusion_model
        kw = dict(softmax_a_bit=_get(o, "softmax_a_bit", 8), aq_mode=o.q_mode)
        if not _get(o, "cali", False):
            setattr(unet, "split", True)
            qnn = QuantModel(model=unet, wq_params=wq, aq_params=aq, cali=False, **kw).to(self.device).eval()
            load_cali_model(qnn, self._init_data(init_context), use_aq=bool(_get(o, "use_aq", False)), path=o.cali_ckpt)
            wrapper.diffusion_model = qnn
            if _get(o, "use_aq", False):
                ck = torch.load(o.cali_ckpt, map_location="cpu")
                tot = len(list(ck.keys())) - 1
                wrapper.tot, wrapper.t_max, wrapper.ckpt, wrapper.iter = 1000 // tot, tot - 1, ck, 0
            return qnn
        # ---- calibration
        logger.info("Generating calibration data...")
        from tfmq_dm_amd.quant import data_generate as DG
        shape = self.latent_shape()
        if self.flow == "uncond":
            n = int(_get(o, "cali_batch", 256))                    # the reference hard-codes 256 samples per step
            cali = DG.generate_cali_data_ldm(model=self.model, T=self.steps, c=1, batch_size=n, shape=shape,
                                             vanilla=_get(o, "vanilla_sample", False), dpm=_get(o, "dpm", False),
                                             plms=_get(o, "plms", False), eta=self.eta)
            tmp = [[cali[0][i * n:(i + 1) * n], cali[1][i * n:(i + 1) * n]] for i in range(0, self.steps, int(o.interval_length))]
            w_cali = [torch.cat([x[0] for x in tmp]), torch.cat([x[1] for x in tmp])]
            interval, bs = n, self.CALI_RECIPE["uncond"][1]
        elif self.flow == "class":
            cali = DG.generate_cali_data_ldm_imagenet(model=self.model, T=self.steps, c=1, batch_size=int(_get(o, "cali_batch", 8)),
                                                      shape=shape, eta=self.eta, scale=float(o.scale))
            w_cali, interval, bs = cali, int(_get(o, "cali_interval", self.CALI_RECIPE["class"][0])), self.CALI_RECIPE["class"][1]
        else:
            if self.sampler is None:
                self.make_sampler()
            if not prompts:
                raise TfmqError("LatentRunner.quantize: the text-guided calibration needs prompts")
            cali = DG.generate_cali_text_guided_data(self.model, self.sampler, T=self.steps, c=1, batch_size=1, prompts=tuple(prompts),
                                                     shape=shape)
            w_cali, interval, bs = cali, int(_get(o, "cali_interval", self.CALI_RECIPE["text"][0])), self.CALI_RECIPE["text"][1]       # txt2img.py:473-486 (32 only in its mp.spawn kwargs)
        logger.info("Calibration data generated.")
        torch.cuda.empty_cache()
        setattr(unet, "split", True)
        iters = int(_get(o, "cali_iters", 20000))
        if _get(o, "multi_gpu", False):
            import torch.multiprocessing as mp
            ngpus = int(_get(o, "ngpus_per_node") or torch.cuda.device_count())
            kwargs = dict(iters=iters, batch_size=32, w=0.01, asym=o.asym, warmup=0.2, opt_mode=RLOSS.MSE, wq_params=wq, aq_params=aq,
                          multi_gpu=ngpus > 1, **kw)
            mp.spawn(cali_model_multi, args=(o.dist_backend, o.world_size, o.dist_url, o.rank, ngpus, wrapper,
                                             bool(_get(o, "use_aq", False)), o.cali_save_path, w_cali, cali, interval, o.running_stat,
                                             kwargs), nprocs=ngpus)
        else:
            qnn = QuantModel(model=unet, wq_params=wq, aq_params=aq, **kw).to(self.device).eval()
            cali_model(qnn=qnn, use_aq=bool(_get(o, "use_aq", False)), path=o.cali_save_path, running_stat=o.running_stat,
                       interval=interval, w_cali_data=w_cali, a_cali_data=cali, iters=iters, batch_size=bs, w=0.01, asym=o.asym,
                       warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
        return "calibrated"

    #: (interval, single-GPU AdaRound mini-batch) of cali_model per driver flow: sample_diffusion_ldm.py:534-546 (256, 32),
    #: latent_imagenet_diffusion.py:275-287 (512, 8), txt2img.py:473-486 (256, 8); the multi-GPU kwargs use 32 in all three
    CALI_RECIPE = {"uncond": (256, 32), "class": (512, 8), "text": (256, 8)}

    # ------------------------------------------------------------------ sampling
    @torch.no_grad()
    def sample_batch(self, batch_size: int, cond: Optional[torch.Tensor] = None, uc: Optional[torch.Tensor] = None,
                     x_T: Optional[torch.Tensor] = None, decode: bool = True) -> dict:
        """One sampler call + the reference's log: {'sample' (decoded when a first stage is attached, else the latents),
        'latents', 'time', 'throughput'}."""
        if self.sampler is None:
            self.make_sampler()
        shape = self.latent_shape()
        kw = dict(S=self.steps, batch_size=batch_size, shape=shape, verbose=False, eta=self.eta)
        if x_T is not None:
            kw["x_T"] = x_T
        if self.flow != "uncond":
            if cond is None:
                raise TfmqError(f"LatentRunner.sample_batch: flow {self.flow!r} needs conditioning")
            scale = float(_get(self.opt, "scale", 1.0))
            kw.update(conditioning=cond, unconditional_guidance_scale=scale, unconditional_conditioning=uc if scale != 1.0 else None)
        torch.cuda.synchronize(self.device)
        t0 = time.time()
        latents, _ = self.sampler.sample(**kw)
        torch.cuda.synchronize(self.device)
        t1 = time.time()
        x = latents
        if decode and hasattr(self.model, "first_stage_model"):
            x = self.model.decode_first_stage(latents)
        log = {"sample": x, "latents": latents, "time": t1 - t0, "throughput": latents.shape[0] / (t1 - t0)}
        logger.info(f'Throughput for this batch: {log["throughput"]}')
        return log

    def run(self, n_samples: int, batch_size: int, encode: Optional[Callable[[int], tuple]] = None) -> dict:
        """Batches until n_samples (sample_diffusion_ldm.py run(), txt2img.py's prompt loop).  encode(i) -> (cond, uc) for batch i."""
        logs, done, i = [], 0, 0
        while done < n_samples:
            c, u = encode(i) if encode is not None else (None, None)
            logs.append(self.sample_batch(batch_size, c, u))
            done += batch_size
            i += 1
        return {"samples": torch.cat([l["sample"] for l in logs])[:n_samples], "time": sum(l["time"] for l in logs),
                "throughput": done / sum(l["time"] for l in logs), "batches": logs}

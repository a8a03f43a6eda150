"""`Diffusion` runner of the pixel-space path (reference ddim/runners/diffusion.py: __init__ :72-107,
sample :203-324, sample_fid :326-364, sample_image :429-476) without the training / dataset /
checkpoint-download glue.  Image files are not written here (torchvision is glue); `sample_fid` returns
the uint8 array the reference dumps to NPZ."""
from __future__ import annotations

import logging
import math
from typing import Optional

import numpy as np
import torch

from .._lib import TfmqError
from .sampler import GraphDdimSampler, generalized_steps, linear_betas, step_sequence

logger = logging.getLogger(__name__)


def get_beta_schedule(beta_schedule, *, beta_start, beta_end, num_diffusion_timesteps) -> np.ndarray:
    """ddim/runners/diffusion.py:38-68 (float64 tables; the runner casts to float32)."""
    def sigmoid(x):
        return 1 / (np.exp(-x) + 1)
    n = num_diffusion_timesteps
    if beta_schedule == "quad":
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=np.float64) ** 2
    elif beta_schedule == "linear":
        betas = np.linspace(beta_start, beta_end, n, dtype=np.float64)
    elif beta_schedule == "const":
        betas = beta_end * np.ones(n, dtype=np.float64)
    elif beta_schedule == "jsd":
        betas = 1.0 / np.linspace(n, 1, n, dtype=np.float64)
    elif beta_schedule == "sigmoid":
        betas = sigmoid(np.linspace(-6, 6, n)) * (beta_end - beta_start) + beta_start
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (n,)
    return betas


def inverse_data_transform(x: torch.Tensor) -> torch.Tensor:
    """ddim/datasets/__init__.py:206-215 for rescaled data: clamp((x+1)/2, 0, 1)."""
    return torch.clamp((x + 1.0) / 2.0, 0.0, 1.0)


class Diffusion:
    def __init__(self, args, config, device: Optional[torch.device] = None):
        self.args, self.config = args, config
        self.device = device or (torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
        config.device = self.device
        d = config.diffusion
        self.betas = torch.from_numpy(get_beta_schedule(d.beta_schedule, beta_start=d.beta_start, beta_end=d.beta_end,
                                                        num_diffusion_timesteps=d.num_diffusion_timesteps)).float().to(self.device)
        self.num_timesteps = self.betas.shape[0]

    def _seq(self):
        if getattr(self.args, "sample_type", "generalized") != "generalized":
            raise NotImplementedError("ddpm_noisy sampling is not a BASELINE config")
        return step_sequence(self.args.skip_type, self.args.timesteps, self.num_timesteps)

    def sample(self, model=None):
        """Driver flow of the reference's `Diffusion.sample` (ddim/runners/diffusion.py:203-324): FP model -> (--ptq)
        QuantModel -> either `load_cali_model` from --cali_ckpt (plus the per-step activation tables when --use_aq) or
        calibration-set generation + `cali_model` (--cali; the reference exits after saving, this returns) -> sample_fid.
        `model`: the FP DDPM UNet with its weights loaded (checkpoint download / EMA restore are glue, done by the caller);
        None builds `Model(config)` and loads `args.ckpt` when given.  Returns (model used for sampling, uint8 images | None)."""
        from .models import Model
        from tfmq_dm_amd.quant.calibration import cali_model, load_cali_model
        from tfmq_dm_amd.quant.data_generate import generate_cali_data_ddim
        from tfmq_dm_amd.quant.quant_layer import Scaler
        from tfmq_dm_amd.quant.quant_model import QuantModel
        from tfmq_dm_amd.quant.reconstruction_util import RLOSS
        args, cfg = self.args, self.config
        if model is None:
            model = Model(cfg)
            ck = getattr(args, "ckpt", None)
            if ck is None:
                raise TfmqError("Diffusion.sample: pass the FP model or args.ckpt (checkpoint download is glue outside this package)")
            model.load_state_dict(torch.load(ck, map_location="cpu"))
        model.to(self.device).eval()
        tot = cali_ckpt = t_max = None
        if getattr(args, "ptq", False):
            cali = bool(getattr(args, "cali", False))
            use_aq = bool(getattr(args, "use_aq", False))
            scaler = Scaler.MSE if cali else Scaler.MINMAX
            wq_params = {"bits": args.wq, "channel_wise": True, "scaler": scaler}
            aq_params = {"bits": args.aq, "channel_wise": False, "scaler": scaler, "leaf_param": use_aq}
            kw = dict(softmax_a_bit=getattr(args, "softmax_a_bit", 8), aq_mode=getattr(args, "q_mode", [2]))
            if not cali:
                qnn = QuantModel(model=model, wq_params=wq_params, aq_params=aq_params, cali=False, **kw).to(self.device).eval()
                init = (torch.randn(1, cfg.data.channels, cfg.data.image_size, cfg.data.image_size), torch.randint(0, 1000, (1,)))
                load_cali_model(qnn, init, use_aq=use_aq, path=args.cali_ckpt)
                model = qnn
                if use_aq:
                    cali_ckpt = torch.load(args.cali_ckpt, map_location="cpu")
                    tot = 1000 - (len(list(cali_ckpt.keys())) - 1)
                    t_max = len(list(cali_ckpt.keys())) - 2
            else:
                logger.info("Generating calibration data...")
                n = getattr(args, "cali_batch", 256)          # the reference hard-codes 256 samples per timestep
                shape = (cfg.data.channels, cfg.data.image_size, cfg.data.image_size)
                cali_data = generate_cali_data_ddim(runnr=self, model=model, T=args.timesteps, c=1, batch_size=n, shape=shape)
                tmp = [[cali_data[0][i * n:(i + 1) * n], cali_data[1][i * n:(i + 1) * n]]
                       for i in range(0, args.timesteps, args.interval_length)]
                w_cali_data = [torch.cat([x[0] for x in tmp], dim=0), torch.cat([x[1] for x in tmp], dim=0)]
                logger.info("Calibration data generated.")
                qnn = QuantModel(model=model, wq_params=wq_params, aq_params=aq_params, **kw).to(self.device).eval()
                cali_model(qnn=qnn, use_aq=use_aq, path=args.cali_save_path, running_stat=getattr(args, "running_stat", False),
                           interval=n, w_cali_data=w_cali_data, a_cali_data=cali_data, iters=getattr(args, "cali_iters", 20000),
                           batch_size=32, w=0.01, asym=getattr(args, "asym", True), warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
                return qnn, None
        if getattr(args, "fid", True):
            imgs = self.sample_fid(model, n_images=args.max_images, batch_size=cfg.sampling.batch_size, cali_ckpt=cali_ckpt,
                                   seed=getattr(args, "seed", None))
            return model, imgs
        raise NotImplementedError("Sample procedeure not defined")

    def sample_image(self, x, model, last=True, untill_fake_t=114514, tot=None, cali_ckpt=None, t_max=None):
        xs, x0_preds, x_t, t_t = generalized_steps(x, self._seq(), model, self.betas, eta=self.args.eta,
                                                   untill_fake_t=untill_fake_t, tot=tot, cali_ckpt=cali_ckpt, t_max=t_max)
        out = (xs, x0_preds)
        if last:
            out = out[0][-1]
        return out, x_t, t_t

    def sample_fid(self, model, n_images: int, batch_size: int, cali_ckpt=None, use_graph: bool = True, seed: Optional[int] = None):
        """-> uint8 [n, H, W, 3].  With a QuantModel + calibration checkpoint the whole FSC table is installed
        on the device and each batch is sampled by hipGraph replay."""
        cfg = self.config
        shape = (batch_size, cfg.data.channels, cfg.data.image_size, cfg.data.image_size)
        g = torch.Generator(device="cpu")
        if seed is not None:
            g.manual_seed(seed)
        sampler = None
        if use_graph and hasattr(model, "set_act_table"):
            if cali_ckpt is not None:
                model.set_act_table(cali_ckpt)
            eng = model.engine(self.device)
            if eng.step is None:
                eng.step = torch.zeros(1, dtype=torch.int32, device=self.device)
            sampler = GraphDdimSampler(eng, self._seq(), self.betas.cpu(), batch_size, eta=self.args.eta)
        res = []
        for _ in range(math.ceil(n_images / batch_size)):
            x = torch.randn(shape, generator=g).to(self.device)
            if sampler is not None:
                x0 = sampler.sample(x)
            else:
                x0 = self.sample_image(x, model, cali_ckpt=cali_ckpt, tot=None if cali_ckpt is None else 1)[0]
            img = inverse_data_transform(x0).permute(0, 2, 3, 1).cpu().numpy() * 255.0
            res.append(img.round().astype(np.uint8))
        return np.concatenate(res, axis=0)[:n_images]

"""Golden vectors for the latent-diffusion (Stable-Diffusion-style) UNet path, produced by the reference.

    python tests/golden/gen_golden_ldm.py

Tiny SpatialTransformer UNet (model_channels 32, mult [1,2], 2 heads, context 64): FP / w4 / w4a8 eps,
quantizer tables, and a 4-step DDIM trajectory with classifier-free guidance run by the reference's own
DDIMSampler over a minimal stand-in for LatentDiffusion (schedule buffers + apply_model)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refharness as H  # noqa: E402

torch = H.install()
from gen_golden import quant_tables, save, sd_arrays  # noqa: E402
from quant.quant_layer import QMODE, Scaler  # noqa: E402
from quant.quant_model import QuantModel  # noqa: E402

UNET_KW = dict(image_size=8, in_channels=4, model_channels=32, out_channels=4, num_res_blocks=1,
               attention_resolutions=[1, 2], channel_mult=[1, 2], num_heads=2, use_spatial_transformer=True,
               transformer_depth=1, context_dim=64, legacy=False)


def build():
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    torch.manual_seed(21)
    m = UNetModel(**UNET_KW).eval()
    H.rerandomize_zero_params(m, seed=9)
    return m


class FakeLDM:
    """What DDIMSampler needs from LatentDiffusion (ddpm.py:117-169 register_schedule, :891-900 apply_model)."""

    def __init__(self, unet, linear_start=0.00085, linear_end=0.012, n=1000):
        from ldm.modules.diffusionmodules.util import make_beta_schedule
        betas = make_beta_schedule("linear", n, linear_start=linear_start, linear_end=linear_end)
        alphas_cumprod = np.cumprod(1.0 - betas, axis=0)
        alphas_cumprod_prev = np.append(1.0, alphas_cumprod[:-1])
        self.num_timesteps = n
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = torch.tensor(alphas_cumprod, dtype=torch.float32)
        self.alphas_cumprod_prev = torch.tensor(alphas_cumprod_prev, dtype=torch.float32)
        self.device = torch.device("cpu")
        self.unet = unet

    def apply_model(self, x, t, c):
        return self.unet(x, t, c)


def main():
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.diffusionmodules.util import make_ddim_sampling_parameters, make_ddim_timesteps, timestep_embedding
    m = build()
    out = sd_arrays(m)
    g = torch.Generator().manual_seed(1111)
    x = torch.randn(2, 4, 8, 8, generator=g)
    t = torch.tensor([981, 21])
    ctx = torch.randn(2, 5, 64, generator=g)
    out.update(x=x, t=t, ctx=ctx)
    with torch.no_grad():
        out["eps_fp"] = m(x, t, ctx)
    out["temb_32"] = timestep_embedding(torch.tensor([0, 1, 21, 981]), 32)
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    qnn = QuantModel(m, wq, aq, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
    from quant.quant_layer import QuantLayer
    out["quant_layer_names"] = np.array([n for n, mod in qnn.model.named_modules() if isinstance(mod, QuantLayer)])
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        _ = qnn(x, t, ctx)
    qnn.disable_out_quantization()
    with torch.no_grad():
        out["eps_w4"] = qnn(x, t, ctx)
        out["tib_w4"] = torch.cat(qnn.tib(x, t), dim=1)
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        _ = qnn(x, t, ctx)
        out["eps_w4a8"] = qnn(x, t, ctx)
    out.update(quant_tables(qnn))
    # schedules
    ldm = FakeLDM(qnn)
    out["alphas_cumprod"] = ldm.alphas_cumprod
    for S in (4, 20, 50):
        ts = make_ddim_timesteps("uniform", S, 1000, verbose=False)
        sig, al, alp = make_ddim_sampling_parameters(ldm.alphas_cumprod.cpu(), ts, 0.0, verbose=False)
        out[f"ddim_ts_{S}"] = ts
        out[f"ddim_alphas_{S}"] = np.asarray(al, dtype=np.float64)
        out[f"ddim_alphas_prev_{S}"] = np.asarray(alp, dtype=np.float64)
    # 4-step DDIM with CFG 7.5 through the reference sampler (w4a8, fixed act table)
    sampler = DDIMSampler(ldm)
    x_T = torch.randn(2, 4, 8, 8, generator=g)
    uc = torch.randn(2, 5, 64, generator=g)
    out["traj_xT"], out["traj_uc"] = x_T, uc
    inter = []
    samples, _ = sampler.sample(S=4, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False,
                                unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0, x_T=x_T,
                                img_callback=lambda px0, i: inter.append(px0.clone()), log_every_t=1)
    out["traj_w4a8_final"] = samples
    out["traj_w4a8_predx0"] = torch.stack(inter)
    qnn.set_quant_state(False, False)
    samples, _ = sampler.sample(S=4, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False,
                                unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0, x_T=x_T)
    out["traj_fp_final"] = samples
    # PLMS (Adams-Bashforth 1-4, one extra model call on the first step), 6 steps, CFG 7.5: FP and w4a8, plus the
    # `untill_fake_t` early stop used by the calibration-set generators
    from ldm.models.diffusion.plms import PLMSSampler
    psampler = PLMSSampler(ldm)
    inter = []
    samples, _ = psampler.sample(S=6, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False,
                                 unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0, x_T=x_T,
                                 img_callback=lambda px0, i: inter.append(px0.clone()), log_every_t=1)
    out["plms_fp_final"] = samples
    out["plms_fp_predx0"] = torch.stack(inter)
    samples, _ = psampler.sample(S=6, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False,
                                 unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0, x_T=x_T,
                                 untill_fake_t=4)
    out["plms_fp_until4"] = samples
    samples, _ = sampler.sample(S=4, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False,
                                unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0, x_T=x_T,
                                untill_fake_t=3)
    out["traj_fp_until3"] = samples
    # DPM-Solver++ (multistep order 2, data prediction, CFG) through the reference's DPMSolverSampler, FP model
    for name in ("ldm.models.diffusion.dpm_solver", "ldm.models.diffusion.dpm_solver.sampler"):
        sys.modules.pop(name, None)       # the harness stubs these (annotation-only imports elsewhere); use the real ones
    from ldm.models.diffusion.dpm_solver.sampler import DPMSolverSampler
    dsampler = DPMSolverSampler(ldm)
    samples, vt = dsampler.sample(S=6, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False,
                                  unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0, x_T=x_T)
    out["dpm_fp_final"], out["dpm_fp_vect"] = samples, vt
    samples, vt = dsampler.sample(S=6, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False,
                                  unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0, x_T=x_T,
                                  untill_fake_t=3)
    out["dpm_fp_until3"], out["dpm_fp_until3_vect"] = samples, vt
    qnn.set_quant_state(True, True)
    samples, _ = psampler.sample(S=6, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False,
                                 unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0, x_T=x_T)
    out["plms_w4a8_final"] = samples
    save("f11_ldm_tiny", **out)


def f12():
    """Tiny end-to-end LDM calibration by the reference's own cali_model (weight init -> TIAR on the LDM TIB ->
    layer / ResBlock / BasicTransformerBlock reconstruction -> Finite-Set activation calibration), plus the
    load_cali_model round trip.  SD-style UNet as above."""
    import tempfile
    import quant.reconstruction as _rec
    from quant.calibration import cali_model, load_cali_model
    from quant.reconstruction_util import RLOSS
    # torch 2.10 CPU segfaults in the backward of a block whose cached input is channels-last-strided (the output of
    # the stride-2 `op` conv).  Same values, contiguous memory: harness-side workaround, the reference is untouched.
    _orig_save_inout = _rec.save_inout

    def _contig_save_inout(*a, **k):
        ci, co = _orig_save_inout(*a, **k)
        return tuple(c.contiguous() for c in ci), (co.contiguous() if torch.is_tensor(co) else co)
    _rec.save_inout = _contig_save_inout
    m = build()
    out = sd_arrays(m)
    g = torch.Generator().manual_seed(1212)
    G, I = 3, 16
    xs = torch.randn(G * I, 4, 8, 8, generator=g)
    ts = torch.cat([torch.full((I,), float(t)) for t in (901, 501, 101)])
    cs = torch.randn(G * I, 5, 64, generator=g)
    out["cali_x"], out["cali_t"], out["cali_c"] = xs, ts, cs
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    qnn = QuantModel(m, wq, aq, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
    qnn.set_grad_ckpt(False)
    torch.manual_seed(5)
    np.random.seed(5)
    path = os.path.join(tempfile.mkdtemp(), "c.pth")
    cali_model(qnn, (xs, ts, cs), (xs, ts, cs), use_aq=True, path=path, running_stat=True, interval=I,
               iters=10, batch_size=8, w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
    ck = torch.load(path, map_location="cpu")
    keys = sorted(ck["weight"].keys())
    out["weight_keys"] = np.array(keys)
    for k in keys:
        out["ck/weight/" + k] = ck["weight"][k]
    for gi in range(G):
        ak = sorted(ck[f"act_{gi}"].keys())
        if gi == 0:
            out["act_keys"] = np.array(ak)
        out[f"ck/act_{gi}/delta"] = torch.stack([ck[f"act_{gi}"][k].reshape(()) for k in ak if k.endswith("delta")])
        out[f"ck/act_{gi}/zp"] = torch.stack([ck[f"act_{gi}"][k].reshape(()) for k in ak if k.endswith("zero_point")])
    m2 = build()
    qnn2 = QuantModel(m2, wq, aq, cali=False, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
    init = (torch.randn(1, 4, 8, 8, generator=g), torch.randint(0, 1000, (1,), generator=g), torch.randn(1, 5, 64, generator=g))
    load_cali_model(qnn2, init, use_aq=True, path=path)
    qnn2.load_state_dict(ck["act_1"], strict=False)
    xe = torch.randn(2, 4, 8, 8, generator=g)
    te = torch.tensor([501.0, 501.0])
    ce = torch.randn(2, 5, 64, generator=g)
    with torch.no_grad():
        out["reload_x"], out["reload_t"], out["reload_c"] = xe, te, ce
        out["reload_eps_act1"] = qnn2(xe, te, ce)
    save("f12_ldm_cali_tiny", **out)


ATTN_UNET_KW = dict(image_size=8, in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1,
                    attention_resolutions=[1, 2], channel_mult=[1, 2], num_head_channels=16)


def f13():
    """Unconditional LDM family (CelebA-HQ / LSUN configs): UNetModel with plain AttentionBlocks (Conv1d qkv / proj_out,
    QKVAttentionLegacy head layout), no context.  FP / w4 / w4a8 eps, quantizer tables, the tree-rewrite layer list
    (QKMatMul / SMVMatMul seams become Quant*MatMul under leaf_param), a 4-step DDIM trajectory (eta 0)."""
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from ldm.models.diffusion.ddim import DDIMSampler
    from quant.quant_layer import QuantLayer
    from quant.quant_block import BaseQuantBlock
    torch.manual_seed(31)
    m = UNetModel(**ATTN_UNET_KW).eval()
    H.rerandomize_zero_params(m, seed=10)
    out = sd_arrays(m)
    g = torch.Generator().manual_seed(1313)
    x = torch.randn(2, 3, 8, 8, generator=g)
    t = torch.tensor([981, 21])
    out.update(x=x, t=t)
    with torch.no_grad():
        out["eps_fp"] = m(x, t)
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    qnn = QuantModel(m, wq, aq, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
    out["quant_layer_names"] = np.array([n for n, mod in qnn.model.named_modules() if isinstance(mod, QuantLayer)])
    out["quant_block_names"] = np.array([f"{n}:{type(mod).__name__}" for n, mod in qnn.model.named_modules()
                                         if isinstance(mod, BaseQuantBlock)])
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        _ = qnn(x, t)
    qnn.disable_out_quantization()
    with torch.no_grad():
        out["eps_w4"] = qnn(x, t)
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        _ = qnn(x, t)
        out["eps_w4a8"] = qnn(x, t)
    out.update(quant_tables(qnn))
    ldm = FakeLDM(qnn, linear_start=0.0015, linear_end=0.0195)
    ldm.apply_model = lambda xx, tt, cc: qnn(xx, tt)
    sampler = DDIMSampler(ldm)
    x_T = torch.randn(2, 3, 8, 8, generator=g)
    out["traj_xT"] = x_T
    samples, _ = sampler.sample(S=4, batch_size=2, shape=[3, 8, 8], verbose=False, eta=0.0, x_T=x_T)
    out["traj_w4a8_final"] = samples
    qnn.set_quant_state(False, False)
    samples, _ = sampler.sample(S=4, batch_size=2, shape=[3, 8, 8], verbose=False, eta=0.0, x_T=x_T)
    out["traj_fp_final"] = samples
    out["alphas_cumprod"] = ldm.alphas_cumprod
    save("f13_ldm_attnblock_tiny", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["f11", "f12", "f13"]
    if "f13" in which:
        f13()
    if "f11" in which:
        main()
    if "f12" in which:
        f12()

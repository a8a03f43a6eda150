"""Round-3 golden vectors, produced by importing the reference (build container only):

    python tests/golden/gen_golden_r03.py [f19]

  f19  the LDM-side DRIVER FLOWS in miniature (sample_diffusion_ldm.py:445-565, latent_imagenet_diffusion.py:190-341,
       txt2img.py:381-598): the reference's cali_model checkpoint of the tiny UNets of F12 (text-guided family:
       SpatialTransformer, 5 context tokens), F15 (class-conditional: one context token) and F16 (unconditional
       AttentionBlock LDM) is loaded the way the drivers do it (QuantModel(cali=False) -> load_cali_model -> per-call
       Finite-Set group k = t_max - (t-1)//tot through DiffusionWrapper.forward's load_state_dict, ddpm.py:1402-1405),
       then the reference's DDIMSampler / PLMSSampler samples a batch.  Recorded: the (timestep, group) of every UNet call
       and the final latents.  pytorch_lightning is absent here, so LatentDiffusion / DiffusionWrapper cannot be imported;
       their two relevant pieces (register_schedule's buffers, the four lines of DiffusionWrapper.forward quoted above)
       are driven from this harness -- everything numerical (QuantModel, load_cali_model, samplers, UNet) is the reference."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refharness as H  # noqa: E402

torch = H.install()
from gen_golden import save  # noqa: E402
from gen_golden_ldm import ATTN_UNET_KW, FakeLDM, build as build_sd  # noqa: E402
from gen_golden_r02 import AQ, MODE, WQ, _contig_save_inout  # noqa: E402
import gen_golden_r02 as R2  # noqa: E402
R2.Scaler = __import__('quant.quant_layer', fromlist=['Scaler']).Scaler
from quant.quant_model import QuantModel  # noqa: E402


class RefWrapper:
    """DiffusionWrapper.forward (ddpm.py:1402-1424) for conditioning_key None / 'crossattn' with the drivers' attributes
    tot / t_max / ckpt; records the (t, k) of every call."""

    def __init__(self, qnn, ckpt, groups):
        self.diffusion_model, self.ckpt = qnn, ckpt
        self.tot, self.t_max = 1000 // groups, groups - 1          # txt2img.py:413-417
        self.calls = []

    def __call__(self, x, t, c=None):
        k = int(self.t_max - (t[0].item() - 1) // self.tot)
        self.calls.append((float(t[0].item()), k))
        self.diffusion_model.load_state_dict(self.ckpt[f"act_{k}"], strict=False)
        with torch.no_grad():
            return self.diffusion_model(x, t) if c is None else self.diffusion_model(x, t, context=c)


def _cali(build, data, interval):
    import tempfile
    from quant.calibration import cali_model
    from quant.reconstruction_util import RLOSS
    qnn = QuantModel(build(), dict(WQ), dict(AQ), aq_mode=MODE).eval()
    if hasattr(qnn, "set_grad_ckpt"):
        qnn.set_grad_ckpt(False)
    torch.manual_seed(5)
    np.random.seed(5)
    path = os.path.join(tempfile.mkdtemp(), "c.pth")
    cali_model(qnn, data, data, use_aq=True, path=path, running_stat=True, interval=interval, iters=10, batch_size=8, w=0.01,
               asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
    return path


def _flow(out, tag, build, path, init, shape, cond, uc, scale, x_T, steps, plms, schedule):
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.models.diffusion.plms import PLMSSampler
    from quant.calibration import load_cali_model
    qnn = QuantModel(build(), dict(WQ), dict(AQ), cali=False, aq_mode=MODE).eval()
    load_cali_model(qnn, init, use_aq=True, path=path)
    ck = torch.load(path, map_location="cpu")
    groups = len(ck) - 1
    wrap = RefWrapper(qnn, ck, groups)
    ldm = FakeLDM(None, **schedule)
    ldm.apply_model = lambda x, t, c: wrap(x, t, c)
    sampler = (PLMSSampler if plms else DDIMSampler)(ldm)
    kw = dict(S=steps, batch_size=x_T.shape[0], shape=shape, verbose=False, eta=0.0, x_T=x_T)
    if cond is not None:
        kw.update(conditioning=cond, unconditional_guidance_scale=scale, unconditional_conditioning=uc)
    final, _ = sampler.sample(**kw)
    out[f"{tag}/final"] = final
    out[f"{tag}/calls"] = np.array(wrap.calls, dtype=np.float64)
    out[f"{tag}/groups"] = np.array([groups, wrap.tot, wrap.t_max])


def f19():
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    _contig_save_inout()
    out = {}
    # ---- text-guided family (F12's model and calibration set)
    f12 = np.load(os.path.join(HERE, "f12_ldm_cali_tiny.npz"), allow_pickle=False)
    data = tuple(torch.from_numpy(f12[k]) for k in ("cali_x", "cali_t", "cali_c"))
    path = _cali(build_sd, data, 16)
    ck = torch.load(path, map_location="cpu")
    ak = sorted(ck["act_1"].keys())
    d1 = torch.stack([ck["act_1"][k].reshape(()) for k in ak if k.endswith("delta")])
    assert np.array_equal(d1.numpy(), f12["ck/act_1/delta"]), "regenerated checkpoint differs from fixture F12"
    g = torch.Generator().manual_seed(1919)
    init = (torch.randn(1, 4, 8, 8, generator=g), torch.randint(0, 1000, (1,), generator=g), torch.randn(1, 5, 64, generator=g))
    x_T = torch.randn(2, 4, 8, 8, generator=g)
    c, uc = torch.randn(2, 5, 64, generator=g), torch.randn(1, 5, 64, generator=g).repeat(2, 1, 1)
    out.update({"text/x_T": x_T, "text/c": c, "text/uc": uc})
    sched = dict(linear_start=0.00085, linear_end=0.012)
    _flow(out, "text/ddim", build_sd, path, init, [4, 8, 8], c, uc, 7.5, x_T, 6, False, sched)
    _flow(out, "text/plms", build_sd, path, init, [4, 8, 8], c, uc, 7.5, x_T, 6, True, sched)
    # ---- class-conditional family (F15's model; its calibration set is regenerated as gen_golden_r02.f15 draws it)
    f15 = np.load(os.path.join(HERE, "f15_cin_tiny.npz"), allow_pickle=False)

    def build_cin():
        torch.manual_seed(51)
        m = UNetModel(**R2.CIN_KW).eval()
        H.rerandomize_zero_params(m, seed=15)
        return m
    data = tuple(torch.from_numpy(f15[k]) for k in ("cali_x", "cali_t", "cali_c"))
    path = _cali(build_cin, data, 16)
    ck = torch.load(path, map_location="cpu")
    d1 = torch.stack([ck["act_1"][k].reshape(()) for k in sorted(ck["act_1"].keys()) if k.endswith("delta")])
    assert np.array_equal(d1.numpy(), f15["ck/act_1/delta"]), "regenerated checkpoint differs from fixture F15"
    init = (torch.randn(1, 3, 8, 8, generator=g), torch.randint(0, 1000, (1,), generator=g), torch.randn(1, 1, 64, generator=g))
    x_T = torch.randn(2, 3, 8, 8, generator=g)
    c, uc = torch.randn(2, 1, 64, generator=g), torch.randn(1, 1, 64, generator=g).repeat(2, 1, 1)
    out.update({"class/x_T": x_T, "class/c": c, "class/uc": uc})
    _flow(out, "class/ddim", build_cin, path, init, [3, 8, 8], c, uc, 3.0, x_T, 5, False, dict(linear_start=0.0015, linear_end=0.0195))
    # ---- unconditional AttentionBlock LDM (F16's model)
    f13 = np.load(os.path.join(HERE, "f13_ldm_attnblock_tiny.npz"), allow_pickle=False)
    f16 = np.load(os.path.join(HERE, "f16_attnblock_cali_tiny.npz"), allow_pickle=False)

    def build_attn():
        m = UNetModel(**ATTN_UNET_KW).eval()
        m.load_state_dict({k[3:]: torch.from_numpy(f13[k]) for k in f13.files if k.startswith("sd/")})
        return m
    data = tuple(torch.from_numpy(f16[k]) for k in ("cali_x", "cali_t"))
    path = _cali(build_attn, data, 16)
    ck = torch.load(path, map_location="cpu")
    d1 = torch.stack([ck["act_1"][k].reshape(()) for k in sorted(ck["act_1"].keys()) if k.endswith("delta")])
    assert np.array_equal(d1.numpy(), f16["ck/act_1/delta"]), "regenerated checkpoint differs from fixture F16"
    init = (torch.randn(1, 3, 8, 8, generator=g), torch.randint(0, 1000, (1,), generator=g))
    x_T = torch.randn(3, 3, 8, 8, generator=g)
    out["uncond/x_T"] = x_T
    _flow(out, "uncond/ddim", build_attn, path, init, [3, 8, 8], None, None, 1.0, x_T, 8, False, dict(linear_start=0.0015, linear_end=0.0195))
    save("f19_ldm_flows", **out)


def f20():
    """The README's other recipe half, --wq 8 (README.md:86-125): the tiny DDPM UNet of F7 and the tiny SD-style UNet of F11 under
    8-bit channel-wise weights (MSE scaler) + 8-bit activations: weight-only and w8a8 eps and every quantizer's (delta, zero point)."""
    from gen_golden import quant_tables, tiny_model
    from quant.quant_layer import Scaler
    wq = {"bits": 8, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    out = {}
    # ---- DDPM UNet (weights of F7)
    f7 = np.load(os.path.join(HERE, "f7_ddim_tiny.npz"), allow_pickle=False)
    cfg, m = tiny_model()
    m.load_state_dict({k[3:]: torch.from_numpy(f7[k]) for k in f7.files if k.startswith("sd/")})
    x, ts = torch.from_numpy(f7["x"]), torch.from_numpy(f7["t"])
    qnn = QuantModel(m, dict(wq), dict(aq), aq_mode=MODE).eval()
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        _ = qnn(x, ts)
    qnn.disable_out_quantization()
    with torch.no_grad():
        out["ddim/eps_w8"] = qnn(x, ts)
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        _ = qnn(x, ts)
        out["ddim/eps_w8a8"] = qnn(x, ts)
    out.update({"ddim/" + k: v for k, v in quant_tables(qnn).items()})
    # ---- SD-style UNet (weights of F11)
    f11 = np.load(os.path.join(HERE, "f11_ldm_tiny.npz"), allow_pickle=False)
    m = build_sd()
    m.load_state_dict({k[3:]: torch.from_numpy(f11[k]) for k in f11.files if k.startswith("sd/")})
    x, t, ctx = torch.from_numpy(f11["x"]), torch.from_numpy(f11["t"]), torch.from_numpy(f11["ctx"])
    qnn = QuantModel(m, dict(wq), dict(aq), aq_mode=MODE).eval()
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        _ = qnn(x, t, ctx)
    qnn.disable_out_quantization()
    with torch.no_grad():
        out["ldm/eps_w8"] = qnn(x, t, ctx)
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        _ = qnn(x, t, ctx)
        out["ldm/eps_w8a8"] = qnn(x, t, ctx)
    out.update({"ldm/" + k: v for k, v in quant_tables(qnn).items()})
    save("f20_w8a8", **out)


def _force_attention_quant(qnn):
    """Switch ON what no driver of the reference switches on: the activation quantizers of
    the attention matmuls.  Returns {qualified name: UniformAffineQuantizer} of every attention quantizer and hooks that record
    the tensor each one sees."""
    from quant.quant_block import QuantAttnBlock, QuantBasicTransformerBlock, QuantQKMatMul, QuantSMVMatMul
    quantizers = {}
    for n, mod in qnn.model.named_modules():
        if isinstance(mod, QuantAttnBlock):
            mod.use_aq = True
            for s_ in ("q", "k", "v", "w"):
                quantizers[f"{n}.aqtizer_{s_}"] = getattr(mod, f"aqtizer_{s_}")
        elif isinstance(mod, QuantBasicTransformerBlock):
            for an in ("attn1", "attn2"):
                a = getattr(mod, an)
                a.use_aq = True
                for s_ in ("q", "k", "v", "w"):
                    quantizers[f"{n}.{an}.aqtizer_{s_}"] = getattr(a, f"aqtizer_{s_}")
        elif isinstance(mod, QuantQKMatMul):
            mod.use_aq = True
            quantizers[f"{n}.aqtizer_q"], quantizers[f"{n}.aqtizer_k"] = mod.aqtizer_q, mod.aqtizer_k
        elif isinstance(mod, QuantSMVMatMul):
            mod.use_aq = True
            quantizers[f"{n}.aqtizer_v"], quantizers[f"{n}.aqtizer_w"] = mod.aqtizer_v, mod.aqtizer_w
    seen = {}
    hooks = [q.register_forward_pre_hook((lambda nm: (lambda m, a: seen.__setitem__(nm, a[0].detach().clone())))(name))
             for name, q in quantizers.items()]
    return quantizers, seen, hooks


def _attn_quant_arrays(out, tag, quantizers, seen):
    names = sorted(quantizers)
    out[f"{tag}/attn_q_names"] = np.array(names)
    for n in names:
        q = quantizers[n]
        d, z = float(q.delta), float(q.zero_point)
        out[f"{tag}/attn_q/{n}/delta"], out[f"{tag}/attn_q/{n}/zp"] = torch.tensor(d), torch.tensor(z)
        out[f"{tag}/attn_q/{n}/level"] = torch.tensor(q.level)
        x = seen[n]
        bins = torch.clamp(torch.round(x / q.delta.detach()) + q.zero_point, 0, q.level - 1)
        out[f"{tag}/attn_q/{n}/bins"] = bins.to(torch.uint8 if q.level <= 256 else torch.int32)
        out[f"{tag}/attn_q/{n}/shape"] = np.array(list(x.shape))


def f21():
    """SURVEY section 8f-3 with the switch forced ON: 8-bit quantisers on q, k, v and the (always-zero) softmax of every attention
    matmul -- QuantAttnBlock (DDPM UNet of F7), cross_attn_forward of QuantBasicTransformerBlock (SD-style UNet of F11),
    QuantQKMatMul / QuantSMVMatMul (AttentionBlock UNet of F13) -- lazily initialised (MSE) on the first forward with the flag set,
    on top of the ordinary w4a8 state.  Recorded: eps, every attention quantizer's (delta, zero point, level) and the BINS of the
    tensor it saw in the recorded forward."""
    from gen_golden import quant_tables, tiny_model
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    wq = {"bits": 4, "channel_wise": True, "scaler": R2.Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": R2.Scaler.MSE, "leaf_param": True}
    out = {}

    def run(tag, m, args):
        qnn = QuantModel(m, dict(wq), dict(aq), aq_mode=MODE).eval()
        qnn.set_quant_state(True, False)
        with torch.no_grad():
            _ = qnn(*args)
        qnn.disable_out_quantization()
        qnn.set_quant_state(True, True)
        with torch.no_grad():
            _ = qnn(*args)                       # ordinary w4a8 state: every layer quantizer initialised
        quantizers, seen, hooks = _force_attention_quant(qnn)
        with torch.no_grad():
            _ = qnn(*args)                       # lazy init of the attention quantizers
            seen.clear()
            out[f"{tag}/eps_w4a8_attnq"] = qnn(*args)
        for h in hooks:
            h.remove()
        out.update({f"{tag}/" + k: v for k, v in quant_tables(qnn).items()})
        _attn_quant_arrays(out, tag, quantizers, seen)

    f7 = np.load(os.path.join(HERE, "f7_ddim_tiny.npz"), allow_pickle=False)
    cfg, m = tiny_model()
    m.load_state_dict({k[3:]: torch.from_numpy(f7[k]) for k in f7.files if k.startswith("sd/")})
    run("ddim", m, (torch.from_numpy(f7["x"]), torch.from_numpy(f7["t"])))
    f11 = np.load(os.path.join(HERE, "f11_ldm_tiny.npz"), allow_pickle=False)
    m = build_sd()
    m.load_state_dict({k[3:]: torch.from_numpy(f11[k]) for k in f11.files if k.startswith("sd/")})
    run("ldm", m, (torch.from_numpy(f11["x"]), torch.from_numpy(f11["t"]), torch.from_numpy(f11["ctx"])))
    f13 = np.load(os.path.join(HERE, "f13_ldm_attnblock_tiny.npz"), allow_pickle=False)
    m = UNetModel(**ATTN_UNET_KW).eval()
    m.load_state_dict({k[3:]: torch.from_numpy(f13[k]) for k in f13.files if k.startswith("sd/")})
    run("attnblock", m, (torch.from_numpy(f13["x"]), torch.from_numpy(f13["t"])))
    save("f21_attention_quant", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["f19", "f20", "f21"]
    if "f19" in which:
        f19()
    if "f20" in which:
        f20()
    if "f21" in which:
        f21()

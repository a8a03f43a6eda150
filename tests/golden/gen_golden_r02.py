"""Round-2 golden vectors, produced by importing the reference (build container only):

    python tests/golden/gen_golden_r02.py [f8b f15 f16 f17 f18]

  f8b  the tiny DDPM calibration of F8 run for 400 Adam iterations per unit (warm-up 0.2): reconstruction / total loss of
       every unit at counts {1, 80, 200, 400} (through the warm-up boundary and most of the b: 20 -> 2 decay) and the
       final AdaRound masks.  Pins the optimisation itself, not only its first 10 steps.
  f15  BASELINE configs[4] in miniature: class-conditional LDM UNet in the cin256-v2 style (SpatialTransformer, ONE
       attention head, ONE context token, linear betas 0.0015..0.0195): FP / w4 / w4a8 eps, quantizer tables, 4-step
       DDIM with classifier-free guidance 3.0 by the reference's DDIMSampler, and the reference's own cali_model run.
  f16  BASELINE configs[2] in miniature: the reference's cali_model (TIAR + ResBlock / layer reconstruction + Finite-Set
       activation calibration) on the AttentionBlock UNet of F13 (unconditional LDM-4 family).
  f17  the calibration-set generators (quant/data_generate.py) and the pixel-space runner's sample_fid
       (ddim/runners/diffusion.py) run by the reference on tiny FP models; every torch.randn draw is recorded so the
       HIP path can be fed the same noise.
  f18  the histogram scalers kl / hist of quant/quant_layer.py on five input distributions."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refharness as H  # noqa: E402

torch = H.install()
from gen_golden import quant_tables, save, sd_arrays, tiny_model  # noqa: E402
from gen_golden_ldm import ATTN_UNET_KW, FakeLDM, UNET_KW  # noqa: E402
from quant.quant_layer import QMODE, Scaler  # noqa: E402
from quant.quant_model import QuantModel  # noqa: E402

WQ = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
AQ = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
MODE = [QMODE.NORMAL.value, QMODE.QDIFF.value]


def _contig_save_inout():
    """torch 2.10 CPU segfaults in the backward of a block whose cached input is channels-last-strided (the output of
    a stride-2 conv).  Same values, contiguous memory: harness-side workaround, the reference is untouched."""
    import quant.reconstruction as _rec
    if getattr(_rec, "_tfmq_patched", False):
        return
    orig = _rec.save_inout

    def patched(*a, **k):
        ci, co = orig(*a, **k)
        return tuple(c.contiguous() for c in ci), (co.contiguous() if torch.is_tensor(co) else co)
    _rec.save_inout = patched
    _rec._tfmq_patched = True


def _ckpt_arrays(out, ck, G):
    keys = sorted(ck["weight"].keys())
    out["weight_keys"] = np.array(keys)
    for k in keys:
        if k.endswith((".w", ".b", ".weight", ".bias")):
            continue          # unchanged copies of the model state (stored once under sd/): the key list pins the schema
        out["ck/weight/" + k] = ck["weight"][k]
    for gi in range(G):
        ak = sorted(ck[f"act_{gi}"].keys())
        if gi == 0:
            out["act_keys"] = np.array(ak)
        out[f"ck/act_{gi}/delta"] = torch.stack([ck[f"act_{gi}"][k].reshape(()) for k in ak if k.endswith("delta")])
        out[f"ck/act_{gi}/zp"] = torch.stack([ck[f"act_{gi}"][k].reshape(()) for k in ak if k.endswith("zero_point")])


# ---------------------------------------------------------------------------- F8b
def f8b():
    import tempfile
    import quant.reconstruction_util as RU
    from quant.calibration import cali_model
    from quant.reconstruction_util import RLOSS
    f8 = np.load(os.path.join(HERE, "f8_cali_tiny.npz"), allow_pickle=False)
    cfg, m = tiny_model(seed=13)
    m.load_state_dict({k[3:]: torch.from_numpy(f8[k]) for k in f8.files if k.startswith("sd/")})
    xs, ts = torch.from_numpy(f8["cali_x"]), torch.from_numpy(f8["cali_t"])
    ITERS, COUNTS = 400, (1, 80, 200, 400)
    rows = []            # (unit index, count, rec, total)
    state = {"unit": -1, "acc": 0.0}
    lp = RU.lp_loss

    def lp_rec(*a, **k):
        v = lp(*a, **k)
        state["acc"] += float(v)
        return v
    RU.lp_loss = lp_rec
    for cls in (RU.LossFunc, RU.LossFuncTimeEmbedding):
        init, call = cls.__init__, cls.__call__

        def mk(init, call):
            def new_init(self, *a, **k):
                init(self, *a, **k)
                state["unit"] += 1
                self._unit = state["unit"]

            def new_call(self, *a, **k):
                state["acc"] = 0.0
                tot = call(self, *a, **k)
                if self.count in COUNTS:
                    rows.append((self._unit, self.count, state["acc"], float(tot)))
                return tot
            return new_init, new_call
        cls.__init__, cls.__call__ = mk(init, call)
    qnn = QuantModel(m, dict(WQ), dict(AQ), aq_mode=MODE).eval()
    torch.manual_seed(5)
    np.random.seed(5)
    path = os.path.join(tempfile.mkdtemp(), "c.pth")
    cali_model(qnn, (xs, ts), (xs, ts), use_aq=True, path=path, running_stat=True, interval=16,
               iters=ITERS, batch_size=8, w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
    ck = torch.load(path, map_location="cpu")
    out = {"iters": ITERS, "counts": np.array(COUNTS), "loss_rows": np.array(rows, dtype=np.float64)}
    akeys = sorted(k for k in ck["weight"] if k.endswith("alpha"))
    out["alpha_keys"] = np.array(akeys)
    masks = torch.cat([(ck["weight"][k] >= 0).reshape(-1) for k in akeys]).numpy()
    out["alpha_sizes"] = np.array([ck["weight"][k].numel() for k in akeys])
    out["masks_packed"] = np.packbits(masks)
    # how far the optimisation moved the rounding away from nearest: the fraction of weights whose learned mask differs
    # from the initial one (alpha_init >= 0  <=>  frac(w/delta) >= 0.5)
    out["n_units"] = state["unit"] + 1
    for gi in range(3):
        ak = sorted(ck[f"act_{gi}"].keys())
        out[f"ck/act_{gi}/delta"] = torch.stack([ck[f"act_{gi}"][k].reshape(()) for k in ak if k.endswith("delta")])
    save("f8b_cali_curve", **out)


# ---------------------------------------------------------------------------- F15
CIN_KW = dict(image_size=8, in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1,
              attention_resolutions=[1, 2], channel_mult=[1, 2], num_heads=1, use_spatial_transformer=True,
              transformer_depth=1, context_dim=64, legacy=False)


def f15():
    import tempfile
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from ldm.models.diffusion.ddim import DDIMSampler
    from quant.calibration import cali_model, load_cali_model
    from quant.quant_layer import QuantLayer
    from quant.reconstruction_util import RLOSS
    _contig_save_inout()

    def build():
        torch.manual_seed(51)
        m = UNetModel(**CIN_KW).eval()
        H.rerandomize_zero_params(m, seed=15)
        return m
    m = build()
    out = sd_arrays(m)
    g = torch.Generator().manual_seed(1515)
    x = torch.randn(2, 3, 8, 8, generator=g)
    t = torch.tensor([951, 51])
    ctx = torch.randn(2, 1, 64, generator=g)              # ONE context token: the class embedding
    out.update(x=x, t=t, ctx=ctx)
    with torch.no_grad():
        out["eps_fp"] = m(x, t, ctx)
    qnn = QuantModel(m, dict(WQ), dict(AQ), aq_mode=MODE).eval()
    out["quant_layer_names"] = np.array([n for n, mod in qnn.model.named_modules() if isinstance(mod, QuantLayer)])
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        _ = qnn(x, t, ctx)
    qnn.disable_out_quantization()
    with torch.no_grad():
        out["eps_w4"] = qnn(x, t, ctx)
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        _ = qnn(x, t, ctx)
        out["eps_w4a8"] = qnn(x, t, ctx)
    out.update(quant_tables(qnn))
    ldm = FakeLDM(qnn, linear_start=0.0015, linear_end=0.0195)
    out["alphas_cumprod"] = ldm.alphas_cumprod
    sampler = DDIMSampler(ldm)
    x_T = torch.randn(2, 3, 8, 8, generator=g)
    uc = torch.randn(1, 1, 64, generator=g).repeat(2, 1, 1)        # the "null class" embedding, same for every sample
    out["traj_xT"], out["traj_uc"] = x_T, uc
    kw = dict(S=4, conditioning=ctx, batch_size=2, shape=[3, 8, 8], verbose=False, unconditional_guidance_scale=3.0,
              unconditional_conditioning=uc, eta=0.0, x_T=x_T)
    out["traj_w4a8_final"], _ = sampler.sample(**kw)
    qnn.set_quant_state(False, False)
    out["traj_fp_final"], _ = sampler.sample(**kw)
    # ---- the reference's own calibration of this model
    m = build()
    G, I = 3, 16
    xs = torch.randn(G * I, 3, 8, 8, generator=g)
    ts = torch.cat([torch.full((I,), float(tv)) for tv in (901, 501, 101)])
    cs = torch.randn(G * I, 1, 64, generator=g)
    out["cali_x"], out["cali_t"], out["cali_c"] = xs, ts, cs
    qnn = QuantModel(m, dict(WQ), dict(AQ), aq_mode=MODE).eval()
    qnn.set_grad_ckpt(False)
    torch.manual_seed(5)
    np.random.seed(5)
    path = os.path.join(tempfile.mkdtemp(), "c.pth")
    cali_model(qnn, (xs, ts, cs), (xs, ts, cs), use_aq=True, path=path, running_stat=True, interval=I,
               iters=10, batch_size=8, w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
    ck = torch.load(path, map_location="cpu")
    _ckpt_arrays(out, ck, G)
    qnn2 = QuantModel(build(), dict(WQ), dict(AQ), cali=False, aq_mode=MODE).eval()
    init = (torch.randn(1, 3, 8, 8, generator=g), torch.randint(0, 1000, (1,), generator=g), torch.randn(1, 1, 64, generator=g))
    load_cali_model(qnn2, init, use_aq=True, path=path)
    qnn2.load_state_dict(ck["act_1"], strict=False)
    xe, te, ce = torch.randn(2, 3, 8, 8, generator=g), torch.tensor([501.0, 501.0]), torch.randn(2, 1, 64, generator=g)
    with torch.no_grad():
        out["reload_x"], out["reload_t"], out["reload_c"] = xe, te, ce
        out["reload_eps_act1"] = qnn2(xe, te, ce)
    save("f15_cin_tiny", **out)


# ---------------------------------------------------------------------------- F16
def f16():
    import tempfile
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from quant.calibration import cali_model, load_cali_model
    from quant.reconstruction_util import RLOSS
    _contig_save_inout()
    f13 = np.load(os.path.join(HERE, "f13_ldm_attnblock_tiny.npz"), allow_pickle=False)

    def build():
        m = UNetModel(**ATTN_UNET_KW).eval()
        m.load_state_dict({k[3:]: torch.from_numpy(f13[k]) for k in f13.files if k.startswith("sd/")})
        return m
    out = {}
    g = torch.Generator().manual_seed(1616)
    G, I = 3, 16
    xs = torch.randn(G * I, 3, 8, 8, generator=g)
    ts = torch.cat([torch.full((I,), float(tv)) for tv in (901, 501, 101)])
    out["cali_x"], out["cali_t"] = xs, ts
    qnn = QuantModel(build(), dict(WQ), dict(AQ), aq_mode=MODE).eval()
    torch.manual_seed(5)
    np.random.seed(5)
    path = os.path.join(tempfile.mkdtemp(), "c.pth")
    cali_model(qnn, (xs, ts), (xs, ts), use_aq=True, path=path, running_stat=True, interval=I,
               iters=10, batch_size=8, w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
    ck = torch.load(path, map_location="cpu")
    _ckpt_arrays(out, ck, G)
    qnn2 = QuantModel(build(), dict(WQ), dict(AQ), cali=False, aq_mode=MODE).eval()
    init = (torch.randn(1, 3, 8, 8, generator=g), torch.randint(0, 1000, (1,), generator=g))
    load_cali_model(qnn2, init, use_aq=True, path=path)
    qnn2.load_state_dict(ck["act_1"], strict=False)
    xe, te = torch.randn(2, 3, 8, 8, generator=g), torch.tensor([501.0, 501.0])
    with torch.no_grad():
        out["reload_x"], out["reload_t"] = xe, te
        out["reload_eps_act1"] = qnn2(xe, te)
    save("f16_attnblock_cali_tiny", **out)


# ---------------------------------------------------------------------------- F17
class RandnTape:
    """Records every torch.randn / randn_like draw (in call order) while active."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self._randn, self._like = torch.randn, torch.randn_like

        def randn(*a, **k):
            v = self._randn(*a, **k)
            self.draws.append(v.detach().clone())
            return v

        def randn_like(x, **k):
            v = self._like(x, **k)
            self.draws.append(v.detach().clone())
            return v
        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._randn, self._like
        return False

    def arrays(self, prefix):
        out = {f"{prefix}/n": len(self.draws)}
        for i, d in enumerate(self.draws):
            out[f"{prefix}/{i}"] = d
        return out


class CondLDM(FakeLDM):
    """FakeLDM + the conditioning interface the generators call (class embedder / text encoder stand-ins: a fixed
    random table, glue in the real drivers)."""
    cond_stage_key = "class_label"

    def __init__(self, unet, dim, tokens, seed, **kw):
        super().__init__(unet, **kw)
        gen = torch.Generator().manual_seed(seed)
        self.table = torch.randn(1001, tokens, dim, generator=gen)
        self.vocab = {}

    def eval(self):
        return self

    def ema_scope(self):
        from contextlib import nullcontext
        return nullcontext()

    def get_learned_conditioning(self, c):
        if isinstance(c, dict):
            return self.table[c[self.cond_stage_key].long()]
        idx = [0 if s == "" else 1 + (sum(ord(ch) for ch in s) % 999) for s in c]
        return self.table[torch.tensor(idx)]


def f17():
    import quant.data_generate as DG
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    out = {}
    # (a) unconditional LDM (AttentionBlock UNet of F13), DDIM and PLMS
    f13 = np.load(os.path.join(HERE, "f13_ldm_attnblock_tiny.npz"), allow_pickle=False)
    m = UNetModel(**ATTN_UNET_KW).eval()
    m.load_state_dict({k[3:]: torch.from_numpy(f13[k]) for k in f13.files if k.startswith("sd/")})
    ldm = FakeLDM(m, linear_start=0.0015, linear_end=0.0195)
    ldm.apply_model = lambda xx, tt, cc: m(xx, tt)
    for tag, kw in (("ldm_ddim", {}), ("ldm_plms", dict(plms=True))):
        torch.manual_seed(70)
        with RandnTape() as tape, torch.no_grad():
            xt, tt = DG.generate_cali_data_ldm(ldm, T=4, c=1, batch_size=2, shape=[3, 8, 8], **kw)
        out[f"{tag}/x"], out[f"{tag}/t"] = xt, tt
        out.update(tape.arrays(f"{tag}/randn"))
    # (b) class-conditional LDM (cin256 style, F15 model), CFG 3.0
    f15 = np.load(os.path.join(HERE, "f15_cin_tiny.npz"), allow_pickle=False)
    mc = UNetModel(**CIN_KW).eval()
    mc.load_state_dict({k[3:]: torch.from_numpy(f15[k]) for k in f15.files if k.startswith("sd/")})
    cldm = CondLDM(mc, 64, 1, seed=71, linear_start=0.0015, linear_end=0.0195)
    out["imagenet/table"] = cldm.table
    torch.manual_seed(72)
    with RandnTape() as tape, torch.no_grad():
        xt, tt, ct = DG.generate_cali_data_ldm_imagenet(cldm, T=2, c=1, batch_size=2, shape=[3, 8, 8], eta=0.0, scale=3.0)
    out["imagenet/x"], out["imagenet/t"], out["imagenet/c"] = xt, tt, ct
    out.update(tape.arrays("imagenet/randn"))
    # (c) text-guided (SD style, F11 model), CFG 7.5, PLMS sampler as txt2img.py --plms
    from ldm.models.diffusion.plms import PLMSSampler
    f11 = np.load(os.path.join(HERE, "f11_ldm_tiny.npz"), allow_pickle=False)
    mt = UNetModel(**UNET_KW).eval()
    mt.load_state_dict({k[3:]: torch.from_numpy(f11[k]) for k in f11.files if k.startswith("sd/")})
    tldm = CondLDM(mt, 64, 5, seed=73)
    out["text/table"] = tldm.table
    from contextlib import nullcontext
    torch.manual_seed(74)
    with RandnTape() as tape:
        xt, tt, ct = DG.generate_cali_text_guided_data(tldm, PLMSSampler(tldm), T=4, c=2, batch_size=2, prompts=("a cat", "two dogs"),
                                                       shape=[4, 8, 8], precision_scope=lambda dev: nullcontext())
    out["text/x"], out["text/t"], out["text/c"] = xt, tt, ct
    out.update(tape.arrays("text/randn"))
    # (d) pixel-space: generate_cali_data_ddim through the reference's Diffusion runner, and sample_fid's image batch
    # dataset / image-writing glue the image lacks (torchvision, lmdb): empty stand-in modules, import-only
    for name in ("torchvision", "torchvision.utils", "torchvision.transforms", "torchvision.transforms.functional",
                 "torchvision.datasets", "torchvision.datasets.utils", "lmdb"):
        mod = sys.modules.setdefault(name, types.ModuleType(name))
        mod.__path__ = []
    sys.modules["torchvision.datasets"].CIFAR10 = object
    sys.modules["torchvision.datasets.utils"].verify_str_arg = sys.modules["torchvision.datasets.utils"].iterable_to_str = None
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    saved = []
    sys.modules["torchvision.utils"].save_image = lambda img, path, **k: saved.append(img.detach().clone())
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    try:
        from ddim.runners.diffusion import Diffusion
    except Exception as e:       # noqa: BLE001 -- dataset glue the image lacks: restate the 3 lines sample_image needs
        print("reference runner not importable here (%s); using a minimal stand-in over generalized_steps" % e)
        Diffusion = None
    f7 = np.load(os.path.join(HERE, "f7_ddim_tiny.npz"), allow_pickle=False)
    cfg, md = tiny_model()
    md.load_state_dict({k[3:]: torch.from_numpy(f7[k]) for k in f7.files if k.startswith("sd/")})
    import argparse
    args = argparse.Namespace(sample_type="generalized", skip_type="quad", timesteps=6, eta=0.0, image_folder="/tmp/_tfmq_f17",
                              numpy_folder="/tmp/_tfmq_f17_np", max_images=4, fid=True)
    cfg.model.var_type = "fixedlarge"
    cfg.sampling = argparse.Namespace(batch_size=4)
    cfg.data.rescaled = True
    cfg.data.logit_transform = False
    cfg.data.uniform_dequantization = False
    cfg.data.gaussian_dequantization = False
    out["runner_import"] = Diffusion is not None
    if Diffusion is not None:
        r = Diffusion(args, cfg, device=torch.device("cpu"))
        torch.manual_seed(75)
        with RandnTape() as tape, torch.no_grad():
            xt, tt = DG.generate_cali_data_ddim(r, md, T=6, c=2, batch_size=2, shape=(3, 16, 16))
        out["ddim/x"], out["ddim/t"] = xt, tt
        out.update(tape.arrays("ddim/randn"))
        for d in (args.image_folder, args.numpy_folder):
            os.makedirs(d, exist_ok=True)
            for f in os.listdir(d):
                os.remove(os.path.join(d, f))
        torch.manual_seed(76)
        with RandnTape() as tape, torch.no_grad():
            r.sample_fid(md)
        npz = [f for f in os.listdir(args.numpy_folder) if f.endswith("-samples.npz")]
        out["fid/uint8"] = np.load(os.path.join(args.numpy_folder, npz[0]))["arr_0"]       # the array the reference dumps
        out.update(tape.arrays("fid/randn"))
    save("f17_cali_generators", **out)


def f18():
    """The two histogram scalers (quant/quant_layer.py:67-133): kl -- the clip ratio in linspace(0.5, 1, 50) whose
    clipped-data histogram is closest (KL divergence, level bins) to the raw histogram, then MINMAX of the clipped tensor; hist --
    the smallest symmetric clip that keeps 99.96 % of the mass of the (0, max|x|) histogram, then MINMAX.  Inputs of the shapes a
    layer sees (normal, post-SiLU, heavy-tailed, one-sided) and (delta, zero_point) by the reference's own functions."""
    from quant.quant_layer import hist, kl
    g = torch.Generator().manual_seed(18)
    xs = {
        "normal": torch.randn(40000, generator=g) * 1.7 + 0.3,
        "silu": torch.nn.functional.silu(torch.randn(8, 64, 14, 14, generator=g) * 2.0),
        "heavy": torch.randn(30000, generator=g) * torch.exp(torch.randn(30000, generator=g)),
        "positive": torch.rand(5000, generator=g) ** 3 * 9.0,
        "small": torch.randn(700, generator=g) * 0.05 - 0.02,
    }
    out = {"names": np.array(sorted(xs))}
    for n, x in xs.items():
        out[f"x/{n}"] = x.numpy()
        for level in (256, 16):
            for az in (False, True):
                if az and float(x.min()) < 0 and n != "silu":
                    continue            # always_zero is used on non-negative (softmax) inputs; silu's small negative part is a stress case
                for fn in (kl, hist):
                    d, z = fn(x, False, level, az)
                    out[f"{fn.__name__}/{n}/{level}/{int(az)}"] = np.array([float(d), float(z)], dtype=np.float64)
    save("f18_hist_scalers", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["f8b", "f15", "f16", "f17", "f18"]
    for name in ("f8b", "f15", "f16", "f17", "f18"):
        if name in which:
            globals()[name]()

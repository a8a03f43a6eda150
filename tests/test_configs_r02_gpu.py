"""BASELINE configs[4] / [2] / [1] in miniature and the calibration-side generators, against fixtures the reference
produced (tests/golden/gen_golden_r02.py):

  F15  class-conditional LDM UNet in the cin256-v2 style (ONE attention head, ONE context token): engine eps FP / w4 /
       w4a8, classifier-free-guidance (3.0) DDIM by hipGraph replay, `cali_model` end to end + `load_cali_model`.
  F16  `cali_model` (TIAR + ResBlock / layer reconstruction + Finite-Set calibration) on the AttentionBlock UNet (F13).
  F8b  400 Adam iterations per unit on the tiny DDPM UNet: reconstruction loss through the warm-up boundary and the
       b: 20 -> 2 decay, final AdaRound masks.
  F17  generate_cali_data_ddim / _ldm / _ldm_imagenet / generate_cali_text_guided_data and the runner's sample_fid, fed
       the reference's own noise draws."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))  # drop-in: `import quant.*`

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CIN_KW = dict(image_size=8, in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1,
              attention_resolutions=[1, 2], channel_mult=[1, 2], num_heads=1, use_spatial_transformer=True,
              transformer_depth=1, context_dim=64, legacy=False)
ATTN_UNET_KW = dict(image_size=8, in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1,
                    attention_resolutions=[1, 2], channel_mult=[1, 2], num_head_channels=16)
UNET_KW = dict(image_size=8, in_channels=4, model_channels=32, out_channels=4, num_res_blocks=1,
               attention_resolutions=[1, 2], channel_mult=[1, 2], num_heads=2, use_spatial_transformer=True,
               transformer_depth=1, context_dim=64, legacy=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(y):
    return y.permute(0, 3, 1, 2).contiguous().cpu()


def rel_l2(a, b):
    return float((a - b).norm() / b.norm())


def sd_of(g):
    return {k[3:]: T(g[k]) for k in g.files if k.startswith("sd/")}


def unet(kw, sd=None, device=None):
    from tfmq_dm_amd.ldm.unet import UNetModel
    m = UNetModel(**kw)
    if sd is not None:
        m.load_state_dict(sd)
    return m.to(device) if device is not None else m


def qnn_of(m, cali=True):
    from quant.quant_layer import QMODE, Scaler
    from quant.quant_model import QuantModel
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    return QuantModel(m, wq, aq, cali=cali, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()


def layerq(g, with_act):
    from tfmq_dm_amd.engine import LayerQ
    act_names = sorted(k[3:-6] for k in g.files if k.startswith("aq/") and k.endswith("/delta"))
    qid = {n: i for i, n in enumerate(act_names)}
    wq = {}
    for k in g.files:
        if k.startswith("wq/") and k.endswith("/delta"):
            n = k[3:-6]
            wq[n] = LayerQ(T(g[k]), T(g[f"wq/{n}/zp"]), None, qid.get(n) if with_act else None)
    qtable = torch.tensor([[[float(g[f"aq/{n}/delta"]), float(g[f"aq/{n}/zp"])] for n in act_names]])
    return wq, qtable


# ------------------------------------------------------------------------------------------------ F15: configs[4]
def test_cin_style_engine_and_cfg_ddim(golden):
    from tfmq_dm_amd.engine import LdmUNetEngine
    from tfmq_dm_amd.ldm.sampler import GraphLatentDdimSampler, alphas_cumprod_linear
    g = golden("f15_cin_tiny")
    cfg = dict(model_channels=32, num_heads=1, in_channels=3)
    x, t, ctx = T(g["x"]), T(g["t"]).float(), T(g["ctx"])
    eng = LdmUNetEngine(sd_of(g), cfg, DEV)
    eng.prepare()
    eps = nchw(eng.forward(nhwc(x), t.to(DEV), ctx.to(DEV)))
    ref = T(g["eps_fp"])
    assert float((eps - ref).abs().max() / ref.abs().max()) <= 1e-2
    wq, _ = layerq(g, False)
    eng.prepare(wq)
    eps = nchw(eng.forward(nhwc(x), t.to(DEV), ctx.to(DEV)))
    ref = T(g["eps_w4"])
    assert float((eps - ref).abs().max() / ref.abs().max()) <= 1e-2
    wq, qtable = layerq(g, True)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    eng = LdmUNetEngine(sd_of(g), cfg, DEV)
    eng.prepare(wq, qtable.repeat(4, 1, 1).contiguous().to(DEV), step)
    eps = nchw(eng.forward(nhwc(x), t.to(DEV), ctx.to(DEV)))
    r = rel_l2(eps, T(g["eps_w4a8"]))
    print("cin-style w4a8 eps rel-L2:", r)
    assert r <= 3.5e-2
    eng.stream_f16 = False
    eps32 = nchw(eng.forward(nhwc(x), t.to(DEV), ctx.to(DEV)))
    eng.stream_f16 = True
    assert torch.equal(nchw(eng.forward(nhwc(x), t.to(DEV), ctx.to(DEV), taps={})), eps32)     # fused == un-fused data path
    assert rel_l2(eps, eps32) <= 3e-2                                                          # fp16 vs fp32 activation stream
    # ONE context token: the block adds to_out(to_v(context)) to every token (softmax over one key is exactly 1).  The whole cross
    # attention -- norm2, to_q, to_k, the attention kernel on fp16 operands -- gives the same eps up to the fp16 rounding of v there
    os.environ["TFMQ_SINGLE_CTX_TOKEN"] = "0"
    try:
        eps_full = nchw(eng.forward(nhwc(x), t.to(DEV), ctx.to(DEV)))
    finally:
        del os.environ["TFMQ_SINGLE_CTX_TOKEN"]
    print("single-token shortcut vs whole cross attention, eps rel-L2:", rel_l2(eps, eps_full))
    assert rel_l2(eps, eps_full) <= 2e-2 and not torch.equal(eps, eps_full)
    assert rel_l2(eps_full, T(g["eps_w4a8"])) <= 3.5e-2
    ac = alphas_cumprod_linear(0.0015, 0.0195)
    assert np.array_equal(ac.numpy(), g["alphas_cumprod"])
    sampler = GraphLatentDdimSampler(eng, 4, 2, (3, 8, 8), (1, 64), scale=3.0, alphas_cumprod=ac).capture()
    out = sampler.sample_nhwc(nhwc(T(g["traj_xT"])), ctx.to(DEV), T(g["traj_uc"]).to(DEV))
    sampler.stream.synchronize()
    rr = rel_l2(nchw(out), T(g["traj_w4a8_final"]))
    print("cin-style CFG-3.0 DDIM-4 latent rel-L2:", rr)
    assert rr <= 6e-2 and int(step.item()) == 4


def _check_ckpt(md, g, sd, n_groups=3, mask_bar=0.99):
    ref_keys = set(str(k) for k in g["weight_keys"])
    assert set(md["weight"].keys()) == ref_keys, sorted(ref_keys ^ set(md["weight"].keys()))[:8]
    assert [k for k in md if k.startswith("act_")] == [f"act_{i}" for i in range(n_groups)]
    assert sorted(md["act_0"].keys()) == [str(k) for k in g["act_keys"]]
    n_exact = n_tot = 0
    mask_agree, n_alpha, worst = 0.0, 0, 0.0
    for k in ref_keys:
        mine = md["weight"][k].float()
        if k.endswith((".w", ".b", ".weight", ".bias")):
            name = k[len("model."):]
            src = sd.get(name if not k.endswith((".w", ".b")) else name[:-1] + ("weight" if k.endswith(".w") else "bias"))
            assert src is not None and torch.equal(mine.reshape(src.shape), src), k      # unchanged copies of the model state
            continue
        ref = T(g["ck/weight/" + k])
        mine = mine.reshape(ref.shape)
        if k.endswith("wqtizer.delta") or k.endswith("wqtizer.zero_point"):
            n_tot += ref.numel()
            n_exact += int((mine == ref).sum())
        elif k.endswith("alpha"):
            mask_agree += float(((mine >= 0) == (ref >= 0)).float().sum())
            n_alpha += ref.numel()
            worst = max(worst, float((mine - ref).abs().max()))
    assert n_exact / n_tot >= 0.98, n_exact / n_tot
    assert mask_agree / n_alpha >= mask_bar, mask_agree / n_alpha
    assert worst <= 5e-2, worst
    for gi in range(n_groups):
        act = md[f"act_{gi}"]
        keys = sorted(act.keys())
        d = torch.stack([act[k].reshape(()) for k in keys if k.endswith("delta")])
        z = torch.stack([act[k].reshape(()) for k in keys if k.endswith("zero_point")])
        rd, rz = T(g[f"ck/act_{gi}/delta"]), T(g[f"ck/act_{gi}/zp"])
        rel = ((d - rd).abs() / rd).numpy()
        assert np.median(rel) <= 5e-3 and rel.max() <= 0.15, (gi, np.median(rel), rel.max())
        assert float((z - rz).abs().max()) <= 4


def test_cin_style_cali_model_matches_reference_run(golden):
    from quant.calibration import cali_model, load_cali_model
    from quant.reconstruction_util import RLOSS
    g = golden("f15_cin_tiny")
    sd = sd_of(g)
    qnn = qnn_of(unet(CIN_KW, sd, DEV)).to(DEV)
    xs, ts, cs = T(g["cali_x"]), T(g["cali_t"]), T(g["cali_c"])
    torch.manual_seed(5)
    np.random.seed(5)
    path = os.path.join(tempfile.mkdtemp(), "c.pth")
    md = cali_model(qnn, (xs, ts, cs), (xs, ts, cs), use_aq=True, path=path, running_stat=True, interval=16, iters=10,
                    batch_size=8, w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
    _check_ckpt(md, g, sd)
    qnn2 = qnn_of(unet(CIN_KW, sd, DEV), cali=False).to(DEV)
    load_cali_model(qnn2, (torch.randn(1, 3, 8, 8), torch.randint(0, 1000, (1,)).float(), torch.randn(1, 1, 64)), use_aq=True, path=path)
    ck = torch.load(path, map_location="cpu")
    qnn2.load_state_dict(ck["act_1"], strict=False)
    eps = qnn2(T(g["reload_x"]).to(DEV), T(g["reload_t"]).to(DEV), T(g["reload_c"]).to(DEV)).cpu()
    rel = rel_l2(eps, T(g["reload_eps_act1"]))
    print("cin-style reload eps rel-L2 vs reference:", rel)
    assert rel <= 5e-2


# ------------------------------------------------------------------------------------------------ F16: configs[2]
def test_attnblock_unet_cali_model_matches_reference_run(golden):
    from quant.calibration import cali_model, load_cali_model
    from quant.reconstruction_util import RLOSS
    g, g13 = golden("f16_attnblock_cali_tiny"), golden("f13_ldm_attnblock_tiny")
    sd = sd_of(g13)
    qnn = qnn_of(unet(ATTN_UNET_KW, sd, DEV)).to(DEV)
    xs, ts = T(g["cali_x"]), T(g["cali_t"])
    torch.manual_seed(5)
    np.random.seed(5)
    path = os.path.join(tempfile.mkdtemp(), "c.pth")
    md = cali_model(qnn, (xs, ts), (xs, ts), use_aq=True, path=path, running_stat=True, interval=16, iters=10,
                    batch_size=8, w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
    _check_ckpt(md, g, sd)
    qnn2 = qnn_of(unet(ATTN_UNET_KW, sd, DEV), cali=False).to(DEV)
    load_cali_model(qnn2, (torch.randn(1, 3, 8, 8), torch.randint(0, 1000, (1,)).float()), use_aq=True, path=path)
    ck = torch.load(path, map_location="cpu")
    qnn2.load_state_dict(ck["act_1"], strict=False)
    eps = qnn2(T(g["reload_x"]).to(DEV), T(g["reload_t"]).to(DEV)).cpu()
    rel = rel_l2(eps, T(g["reload_eps_act1"]))
    print("AttentionBlock-UNet reload eps rel-L2 vs reference:", rel)
    assert rel <= 5e-2


# ------------------------------------------------------------------------------------------------ F8b: the optimisation
def test_reconstruction_loss_curve_400_iterations(golden):
    import tfmq_dm_amd.ddim.models as M
    import quant.reconstruction as REC
    from quant.calibration import cali_model
    from quant.reconstruction_util import RLOSS
    g, g8 = golden("f8b_cali_curve"), golden("f8_cali_tiny")
    m = M.Model(M.make_config(ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), image_size=16, dropout=0.0))
    m.load_state_dict(sd_of(g8))
    qnn = qnn_of(m.to(DEV).eval()).to(DEV)
    xs, ts = T(g8["cali_x"]), T(g8["cali_t"])
    counts = tuple(int(c) for c in g["counts"])
    trace = {"counts": counts, "rows": [], "unit": 0}
    REC.LOSS_TRACE = trace
    try:
        torch.manual_seed(5)
        np.random.seed(5)
        md = cali_model(qnn, (xs, ts), (xs, ts), use_aq=True, path=None, running_stat=True, interval=16, iters=int(g["iters"]),
                        batch_size=8, w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
    finally:
        REC.LOSS_TRACE = None
    ref = g["loss_rows"]                       # (unit, count, rec, total) of the reference, same walk order
    mine = np.array(trace["rows"])             # (unit, count, rec, round)
    assert trace["unit"] == int(g["n_units"]) and mine.shape == ref.shape
    assert np.array_equal(mine[:, :2], ref[:, :2])
    rec_r, rec_m = ref[:, 2], mine[:, 2]
    rnd_r, rnd_m = ref[:, 3] - ref[:, 2], mine[:, 3]
    for i in range(len(ref)):
        # reconstruction loss of the same mini-batch (same host RNG stream) within 5 %: before, at and after the point
        # where the rounding regulariser switches on, and at the end of the temperature decay
        tol = 0.05 * abs(rec_r[i]) + 1e-7
        assert abs(rec_m[i] - rec_r[i]) <= tol, (ref[i, :2], rec_m[i], rec_r[i])
        if ref[i, 1] >= 80:
            assert abs(rnd_m[i] - rnd_r[i]) <= 0.02 * abs(rnd_r[i]), (ref[i, :2], rnd_m[i], rnd_r[i])
        else:
            assert rnd_m[i] == 0.0
    # final AdaRound masks
    akeys = [str(k) for k in g["alpha_keys"]]
    sizes = [int(s) for s in g["alpha_sizes"]]
    ref_mask = np.unpackbits(g["masks_packed"])[:sum(sizes)].astype(bool)
    my_mask = torch.cat([(md["weight"][k] >= 0).reshape(-1) for k in akeys]).numpy()
    agree = float((my_mask == ref_mask).mean())
    print("AdaRound masks after 400 iterations: agreement with the reference", agree)
    assert agree >= 0.99
    for gi in range(3):
        keys = sorted(md[f"act_{gi}"].keys())
        d = torch.stack([md[f"act_{gi}"][k].reshape(()) for k in keys if k.endswith("delta")])
        rd = T(g[f"ck/act_{gi}/delta"])
        rel = ((d - rd).abs() / rd).numpy()
        assert np.median(rel) <= 1e-2, (gi, np.median(rel))


# ------------------------------------------------------------------------------------------------ F17: generators
class Replay:
    """torch.randn / randn_like hand out the reference's recorded draws (the sampler start tensors x_T; the sigma = 0
    noise draws of the reference's eta = 0 steps are never requested by this implementation)."""

    def __init__(self, draws):
        self.q = list(draws)

    def __enter__(self):
        self._randn = torch.randn

        def randn(*size, **kw):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            v = self.q.pop(0)
            assert tuple(v.shape) == tuple(size), (v.shape, size)
            dev = kw.get("device")
            return v.to(dev) if dev is not None else v.clone()
        torch.randn = randn
        return self

    def __exit__(self, *exc):
        torch.randn = self._randn
        return False


def tape(g, tag, positions):
    return [T(g[f"{tag}/randn/{i}"]) for i in positions]


class CondLDM:
    """LatentDiffusion of this package + the conditioning stand-in of the fixture (a fixed table)."""

    def __new__(cls, qnn, table, **kw):
        from tfmq_dm_amd.ldm.ddpm import LatentDiffusion

        class _M(LatentDiffusion):
            cond_stage_key = "class_label"

            def get_learned_conditioning(self, c):
                if isinstance(c, dict):
                    return self.table[c[self.cond_stage_key].long().cpu()].to(DEV)
                idx = [0 if s == "" else 1 + (sum(ord(ch) for ch in s) % 999) for s in c]
                return self.table[torch.tensor(idx)].to(DEV)
        m = _M(qnn, **kw).to(DEV)
        m.table = table
        return m


def test_generate_cali_data_ldm_vs_reference(golden):
    from quant.data_generate import generate_cali_data_ldm
    from tfmq_dm_amd.ldm.ddpm import LatentDiffusion
    g, g13 = golden("f17_cali_generators"), golden("f13_ldm_attnblock_tiny")
    q = qnn_of(unet(ATTN_UNET_KW, sd_of(g13), DEV)).to(DEV)
    q.set_quant_state(False, False)
    m = LatentDiffusion(q, linear_start=0.0015, linear_end=0.0195, conditioning_key=None).to(DEV)
    # reference draws per sample: x_T + one (sigma = 0) noise per executed step; PLMS draws twice on its first step
    for tag, kw, per in (("ldm_ddim", {}, [0, 1, 2, 3]), ("ldm_plms", dict(plms=True), [0, 2, 3, 4])):
        pos, p = [], 0
        for n in per:
            pos.append(p)
            p += 1 + n
        assert p == int(g[f"{tag}/randn/n"])
        with Replay(tape(g, tag, pos)) as rp:
            x, t = generate_cali_data_ldm(m, T=4, c=1, batch_size=2, shape=[3, 8, 8], **kw)
        assert not rp.q
        assert np.array_equal(t.cpu().numpy(), g[f"{tag}/t"])
        r = rel_l2(x.cpu(), T(g[f"{tag}/x"]))
        print(tag, "x_t rel-L2 vs reference:", r)
        assert r <= 2e-2


def test_generate_cali_data_ldm_imagenet_vs_reference(golden):
    from quant.data_generate import generate_cali_data_ldm_imagenet
    g, g15 = golden("f17_cali_generators"), golden("f15_cin_tiny")
    q = qnn_of(unet(CIN_KW, sd_of(g15), DEV)).to(DEV)
    q.set_quant_state(False, False)
    m = CondLDM(q, T(g["imagenet/table"]), linear_start=0.0015, linear_end=0.0195)
    pos = list(range(32)) + [32 + 2 * i for i in range(32)]       # i = 1: x_T only; i = 2: x_T + one noise draw each
    with Replay(tape(g, "imagenet", pos)) as rp:
        x, t, c = generate_cali_data_ldm_imagenet(m, T=2, c=1, batch_size=2, shape=[3, 8, 8], eta=0.0, scale=3.0)
    assert not rp.q
    assert np.array_equal(t.cpu().numpy(), g["imagenet/t"])
    assert torch.equal(c.cpu(), T(g["imagenet/c"]))               # (x_t, t, c) and (x_t, t, uc) interleaved, class order
    r = rel_l2(x.cpu(), T(g["imagenet/x"]))
    print("imagenet generator x_t rel-L2 vs reference:", r)
    assert r <= 2e-2


def test_generate_cali_text_guided_data_vs_reference(golden):
    from quant.data_generate import generate_cali_text_guided_data
    from tfmq_dm_amd.ldm.ddim import PLMSSampler
    g, g11 = golden("f17_cali_generators"), golden("f11_ldm_tiny")
    q = qnn_of(unet(UNET_KW, sd_of(g11), DEV)).to(DEV)
    q.set_quant_state(False, False)
    m = CondLDM(q, T(g["text/table"]))
    # t = 2: two prompts x (x_T + 2 noise draws); t = 4: two prompts x (x_T + 4)
    pos = [0, 3, 6, 11]
    assert int(g["text/randn/n"]) == 16
    with Replay(tape(g, "text", pos)) as rp:
        x, t, c = generate_cali_text_guided_data(m, PLMSSampler(m), T=4, c=2, batch_size=2, prompts=("a cat", "two dogs"),
                                                 shape=[4, 8, 8], precision_scope=None)
    assert not rp.q
    assert np.array_equal(t.cpu().numpy(), g["text/t"])
    assert torch.equal(c.cpu(), T(g["text/c"]))
    r = rel_l2(x.cpu(), T(g["text/x"]))
    print("text-guided generator x_t rel-L2 vs reference:", r)
    assert r <= 3e-2


def _runner(g7):
    import argparse
    import tfmq_dm_amd.ddim.models as M
    from tfmq_dm_amd.ddim.runner import Diffusion
    cfg = M.make_config(ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), image_size=16, dropout=0.0)
    cfg.sampling = argparse.Namespace(batch_size=4)
    m = M.Model(cfg)
    m.load_state_dict(sd_of(g7))
    args = argparse.Namespace(sample_type="generalized", skip_type="quad", timesteps=6, eta=0.0, max_images=4, fid=True, ptq=False)
    return Diffusion(args, cfg, device=torch.device(DEV)), m.to(DEV).eval()


def test_generate_cali_data_ddim_and_sample_fid_vs_reference(golden):
    from quant.data_generate import generate_cali_data_ddim
    g, g7 = golden("f17_cali_generators"), golden("f7_ddim_tiny")
    assert bool(g["runner_import"])
    r, m = _runner(g7)
    # per i in (2, 4, 6): the start tensor, then i - 1 (unused, eta = 0) noise draws of the reference's loop
    with Replay(tape(g, "ddim", [0, 2, 6])) as rp:
        x, t = generate_cali_data_ddim(r, m, T=6, c=2, batch_size=2, shape=(3, 16, 16))
    assert not rp.q
    assert np.array_equal(t.cpu().numpy(), g["ddim/t"])
    rr = rel_l2(x.cpu(), T(g["ddim/x"]))
    print("generate_cali_data_ddim x_t rel-L2 vs reference:", rr)
    assert rr <= 1e-2
    # Diffusion.sample() -> sample_fid: the uint8 array the reference dumps to NPZ (ddim/runners/diffusion.py:326-364)
    for use_graph in (False, True):
        with Replay(tape(g, "fid", [0])):
            imgs = r.sample_fid(m, n_images=4, batch_size=4, use_graph=use_graph)
        ref = g["fid/uint8"].astype(np.int32)
        d = np.abs(imgs.astype(np.int32) - ref)
        print("sample_fid uint8 |diff| mean / max:", d.mean(), d.max())
        assert imgs.dtype == np.uint8 and imgs.shape == ref.shape
        assert d.mean() <= 0.5 and d.max() <= 6
    with Replay(tape(g, "fid", [0])):
        model, imgs2 = r.sample(m)
    assert model is m and np.array_equal(imgs2, imgs)


def test_unsupported_bit_widths_fail_loudly(golden):
    """Bit widths the device path does not implement must be refused, not packed into another grid silently (the bit
    width used to be dropped): 8-bit weights run since round 3 (tests/test_w8a8_gpu.py), 4-bit ACTIVATIONS do not."""
    import tfmq_dm_amd.ddim.models as M
    from quant.quant_layer import QMODE, Scaler
    from quant.quant_model import QuantModel
    from tfmq_dm_amd._lib import TfmqError
    g8 = golden("f8_cali_tiny")
    m = M.Model(M.make_config(ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), image_size=16, dropout=0.0))
    m.load_state_dict(sd_of(g8))
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MINMAX}
    aq = {"bits": 4, "channel_wise": False, "scaler": Scaler.MINMAX, "leaf_param": True}
    q = QuantModel(m.to(DEV).eval(), wq, aq, cali=False, aq_mode=[QMODE.NORMAL.value]).to(DEV).eval()
    q.set_quant_state(True, True)
    with pytest.raises(TfmqError, match="8-bit activations"):
        q(torch.randn(2, 3, 16, 16, device=DEV), torch.tensor([10.0, 500.0], device=DEV))


@pytest.mark.parametrize("B,T_,C,half", [(3, 50, 64, True), (2, 1024, 384, True), (4, 7, 960, False)])
def test_row_broadcast_add(B, T_, C, half):
    import tfmq_dm_amd.ops as ops
    gen = torch.Generator().manual_seed(T_ + C)
    x = torch.randn(B, T_, C, generator=gen).to(DEV)
    r = torch.randn(B, C, generator=gen).to(DEV)
    if half:
        x = x.half()
    y = ops.row_broadcast_add(x, r)
    want = (x.float() + r[:, None, :])
    want = want.half() if half else want
    assert y.dtype == x.dtype and torch.equal(y, want)

"""Generate golden vectors by running the *reference* (imported read-only from /root/reference).

Runs only in the build container:   python tests/golden/gen_golden.py [f1 f2 ...]
Outputs small .npz files next to this script; they are committed and are what pins the
oracle (oracle/tfmq_oracle.py) and, through it, the HIP path.  Every file is stamped with
the torch version and the seeds used.  This script contains no reference source; it only
calls the reference's public functions/classes.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refharness as H  # noqa: E402

torch = H.install()
import torch.nn as nn  # noqa: E402
from quant.quant_layer import (QMODE, QuantLayer, Scaler, UniformAffineQuantizer, lp_loss, minmax, mse)  # noqa: E402
from quant.adaptive_rounding import AdaRoundQuantizer, RMODE  # noqa: E402
from quant.quant_model import QuantModel  # noqa: E402
from quant.reconstruction_util import LossFunc, RLOSS  # noqa: E402

STAMP = dict(torch_version=torch.__version__)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    out["_torch_version"] = np.array(torch.__version__)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def t2f(x):
    return float(x) if not torch.is_tensor(x) else float(x.item())


# ---------------------------------------------------------------------------- F1
def f1():
    g = torch.Generator().manual_seed(101)
    acts = torch.randn(4, 8, 6, 6, generator=g) * 1.7 + 0.3
    pos = torch.rand(3, 5, 7, generator=g)  # softmax-like, always_zero
    wts = torch.randn(16, 8, 3, 3, generator=g) * 0.02
    wts[3] = wts[3].abs() + 0.01  # same-sign channel -> zp outside [0,15] under mse
    lin = torch.randn(12, 20, generator=g) * 0.05
    out = dict(acts=acts, pos=pos, wts=wts, lin=lin)
    # per-tensor 8 bit
    for nm, fn in (("minmax", minmax), ("mse", mse)):
        d, z = fn(acts, False, 256, False)
        out[f"acts_{nm}_delta"], out[f"acts_{nm}_zp"] = t2f(d), t2f(z)
        q = UniformAffineQuantizer(bits=8, channel_wise=False, scaler=fn)
        dq = q(acts)
        idx = torch.clamp(torch.round(acts / q.delta) + q.zero_point, 0, 255)
        out[f"acts_{nm}_idx"] = idx.to(torch.uint8)
        out[f"acts_{nm}_dq"] = dq
    # mse candidate losses for the act tensor (argmin-tie tolerance)
    x_min, x_max = acts.min().item(), acts.max().item()
    losses, deltas, zps = [], [], []
    for i in range(80):
        nmin, nmax = x_min * (1.0 - i * 0.01), x_max * (1.0 - i * 0.01)
        nd = torch.tensor(float(nmax - nmin) / 255)
        nz = torch.round(-nmin / nd)
        xq = torch.clamp(torch.round(acts / nd) + nz, 0, 255)
        losses.append(float(lp_loss(nd * (xq - nz), acts, p=2.4, reduction=__import__("quant.quant_layer").quant_layer.REDUCTION.ALL)))
        deltas.append(float(nd))
        zps.append(float(nz))
    out["acts_mse_cand_loss"], out["acts_mse_cand_delta"], out["acts_mse_cand_zp"] = losses, deltas, zps
    # always_zero (softmax quantizer, quant_block.py:469-472)
    for nm, fn in (("minmax", minmax), ("mse", mse)):
        d, z = fn(pos, False, 256, True)
        out[f"pos_{nm}_delta"], out[f"pos_{nm}_zp"] = t2f(d), t2f(z)
    # per-channel 4 bit
    for nm, fn in (("minmax", minmax), ("mse", mse)):
        for tn, tt in (("wts", wts), ("lin", lin)):
            q = UniformAffineQuantizer(bits=4, channel_wise=True, scaler=fn)
            dq = q(tt)
            out[f"{tn}_{nm}_delta"] = q.delta
            out[f"{tn}_{nm}_zp"] = q.zero_point
            out[f"{tn}_{nm}_idx"] = torch.clamp(torch.round(tt / q.delta) + q.zero_point, 0, 15).to(torch.uint8)
            out[f"{tn}_{nm}_dq"] = dq
    save("f1_quantizer", **out)


# ---------------------------------------------------------------------------- F2
def f2():
    g = torch.Generator().manual_seed(202)
    q = UniformAffineQuantizer(bits=8, channel_wise=False, scaler=mse, leaf_param=True)
    xs = [torch.randn(16, 6, 5, 5, generator=g) * (1.0 + 0.3 * i) + 0.1 * i for i in range(6)]
    out = {"x": torch.stack(xs)}
    y0 = q(xs[0])
    out["init_delta"], out["init_zp"] = t2f(q.delta), t2f(q.zero_point)
    q.running_stat = True
    ds, zs, mins, maxs, ys = [], [], [], [], []
    for x in xs[1:]:
        y = q(x)
        ds.append(t2f(q.delta)); zs.append(t2f(q.zero_point))
        mins.append(t2f(q.x_min)); maxs.append(t2f(q.x_max))
        ys.append(y.detach())
    out.update(delta=ds, zp=zs, x_min=mins, x_max=maxs, y_last=ys[-1], y0=y0.detach())
    save("f2_momentum", **out)


# ---------------------------------------------------------------------------- F3
def f3():
    g = torch.Generator().manual_seed(303)
    out = {}
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}

    def run(tag, layer, x):
        with torch.no_grad():
            layer.weight.copy_(torch.randn(layer.weight.shape, generator=g) * 0.05)
            if layer.bias is not None:
                layer.bias.copy_(torch.randn(layer.bias.shape, generator=g) * 0.1)
        ql = QuantLayer(layer, dict(wq), dict(aq))
        ql.set_quant_state(True, True)
        with torch.no_grad():
            y = ql(x)
            y_fp = layer(x)
        out[f"{tag}_x"] = x
        out[f"{tag}_w"] = layer.weight
        if layer.bias is not None:
            out[f"{tag}_b"] = layer.bias
        out[f"{tag}_y"] = y
        out[f"{tag}_yfp"] = y_fp
        out[f"{tag}_wdelta"] = ql.wqtizer.delta
        out[f"{tag}_wzp"] = ql.wqtizer.zero_point
        out[f"{tag}_adelta"] = t2f(ql.aqtizer.delta)
        out[f"{tag}_azp"] = t2f(ql.aqtizer.zero_point)

    run("lin2d", nn.Linear(64, 24), torch.randn(5, 64, generator=g) * 1.3 + 0.2)
    run("lin3d", nn.Linear(64, 40, bias=False), torch.randn(2, 9, 64, generator=g) * 0.8)
    run("conv3", nn.Conv2d(64, 48, 3, padding=1), torch.randn(2, 64, 8, 8, generator=g) * 1.7 + 0.3)
    run("conv1", nn.Conv2d(64, 32, 1), torch.randn(2, 64, 6, 6, generator=g) * 1.1 - 0.4)
    save("f3_quantlayer", **out)


# ---------------------------------------------------------------------------- F4
def f4():
    g = torch.Generator().manual_seed(404)
    out = {}
    layer = nn.Conv2d(16, 24, 3, padding=1)
    with torch.no_grad():
        layer.weight.copy_(torch.randn(layer.weight.shape, generator=g) * 0.05)
        layer.bias.copy_(torch.randn(layer.bias.shape, generator=g) * 0.1)
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": False}
    ql = QuantLayer(layer, wq, aq)
    ql.set_quant_state(True, False)
    x = torch.randn(8, 16, 6, 6, generator=g)
    with torch.no_grad():
        y_fp = layer(x)
        _ = ql(x)  # init weight quantizer
    out.update(w=layer.weight, b=layer.bias, x=x, y_fp=y_fp, wdelta=ql.wqtizer.delta, wzp=ql.wqtizer.zero_point)
    ql.wqtizer = AdaRoundQuantizer(uaqtizer=ql.wqtizer, rmode=RMODE.LEARNED_HARD_SIGMOID, w=ql.original_w.data)
    ada = ql.wqtizer
    out["alpha0"] = ada.alpha.detach().clone()
    out["soft0"] = ada.get_soft_tgt().detach()
    with torch.no_grad():
        ada.soft_tgt = False
        out["w_hard0"] = ada(ql.w)
        ada.soft_tgt = True
        out["w_soft0"] = ada(ql.w)
    iters = 20
    lf = LossFunc(o=ql, round_loss=RLOSS.RELAXATION, w=0.01, max_count=iters, rec_loss=RLOSS.MSE,
                  b_range=(20, 2), decay_start=0.0, warmup=0.2, p=2.0)
    opt = torch.optim.Adam([ada.alpha])
    tot, grads, alphas, bs = [], [], [], []
    for it in range(iters):
        opt.zero_grad()
        yq = ql(x)
        err = lf(yq, y_fp)
        err.backward()
        tot.append(float(err))
        if it in (0, 3, 4, 11, 19):
            grads.append(ada.alpha.grad.detach().clone())
        opt.step()
        if it in (0, 9, 19):
            alphas.append(ada.alpha.detach().clone())
    out["loss"] = tot
    out["grad_iters"] = [0, 3, 4, 11, 19]
    out["grads"] = torch.stack(grads)
    out["alpha_iters"] = [0, 9, 19]
    out["alphas"] = torch.stack(alphas)
    ada.soft_tgt = False
    with torch.no_grad():
        out["w_hard_final"] = ada(ql.w)
        out["mask_final"] = (ada.alpha >= 0).to(torch.uint8)
    save("f4_adaround", **out)


# ---------------------------------------------------------------------------- DDIM tiny model helpers
def tiny_model(seed=11, **kw):
    from ddim.models.diffusion import Model
    cfg = H.ddim_config(**kw)
    torch.manual_seed(seed)
    m = Model(cfg).eval()
    H.rerandomize_zero_params(m)
    return cfg, m


def sd_arrays(m, prefix="sd/"):
    return {prefix + k: v.detach().clone() for k, v in m.state_dict().items()}


def quant_tables(qnn):
    """Collect (delta, zp, alpha) of every initialised quantizer, keyed by module name."""
    out = {}
    for name, mod in qnn.model.named_modules():
        if isinstance(mod, QuantLayer):
            wqz = mod.wqtizer
            if getattr(wqz, "delta", None) is not None and mod.use_wq:
                out[f"wq/{name}/delta"] = wqz.delta.detach().clone()
                zp = wqz.zero_point
                out[f"wq/{name}/zp"] = zp.detach().clone() if torch.is_tensor(zp) else torch.tensor(float(zp))
                if isinstance(wqz, AdaRoundQuantizer):
                    out[f"wq/{name}/alpha"] = wqz.alpha.detach().clone()
            aqz = mod.aqtizer
            if aqz.delta is not None and mod.use_aq and not mod.disable_aq:
                out[f"aq/{name}/delta"] = torch.tensor(t2f(aqz.delta))
                out[f"aq/{name}/zp"] = torch.tensor(t2f(aqz.zero_point))
    return out


# ---------------------------------------------------------------------------- F5/F6/F7 (DDIM)
def f7():
    from ddim.functions.denoising import generalized_steps
    cfg, m = tiny_model()
    out = sd_arrays(m)
    out["cfg_ch"], out["cfg_ch_mult"], out["cfg_nres"], out["cfg_attn"], out["cfg_res"] = 32, [1, 2], 1, [8], 16
    g = torch.Generator().manual_seed(707)
    x = torch.randn(4, 3, 16, 16, generator=g)
    ts = torch.tensor([3.0, 250.0, 640.0, 999.0])
    with torch.no_grad():
        out["x"], out["t"] = x, ts
        out["eps_fp"] = m(x, ts)
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    qnn = QuantModel(m, wq, aq, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
    # block taps (F6) with hooks on the quant blocks, FP and w4
    from quant.quant_block import BaseQuantBlock
    taps = {}

    def mk(name):
        def hook(mod, inp, outp):
            taps[name] = outp.detach().clone()
        return hook
    hs = [mod.register_forward_hook(mk(n)) for n, mod in qnn.model.named_modules() if isinstance(mod, BaseQuantBlock)]
    qnn.set_quant_state(False, False)
    with torch.no_grad():
        _ = qnn(x, ts)
    for k, v in taps.items():
        out[f"tap_fp/{k}"] = v
    # TIB (F5)
    with torch.no_grad():
        tib = qnn.tib(x, ts)
    out["tib_fp"] = torch.cat(tib, dim=1)
    qnn.set_quant_state(True, False)
    with torch.no_grad():
        _ = qnn(x, ts)  # weight quantizer init (mse)
    qnn.disable_out_quantization()
    taps.clear()
    with torch.no_grad():
        out["eps_w4"] = qnn(x, ts)
        tibq = qnn.tib(x, ts)
    out["tib_w4"] = torch.cat(tibq, dim=1)
    for k, v in taps.items():
        out[f"tap_w4/{k}"] = v
    for h in hs:
        h.remove()
    # activation quantizers: init on this batch (mse), then w4a8 eps + bin-index traces
    qnn.set_quant_state(True, True)
    with torch.no_grad():
        _ = qnn(x, ts)
        out["eps_w4a8"] = qnn(x, ts)
        tibqa = qnn.tib(x, ts)
    out["tib_w4a8"] = torch.cat(tibqa, dim=1)
    out.update(quant_tables(qnn))
    # 10-step DDIM trajectories (eta=0): FP and w4a8 with a fixed act table for all steps
    betas = torch.from_numpy(np.linspace(1e-4, 0.02, 1000, dtype=np.float64)).float()
    seq = [int(s) for s in list(np.linspace(0, np.sqrt(1000 * 0.8), 10) ** 2)]
    out["seq"] = seq
    x0 = torch.randn(2, 3, 16, 16, generator=g)
    out["traj_x0"] = x0
    qnn.set_quant_state(False, False)
    xs, _, _, _ = generalized_steps(x0, seq, qnn, betas, eta=0.0)
    out["traj_fp"] = torch.stack(xs)
    qnn.set_quant_state(True, True)
    xs, _, _, _ = generalized_steps(x0, seq, qnn, betas, eta=0.0)
    out["traj_w4a8"] = torch.stack(xs)
    # early stop semantics (untill_fake_t) used by generate_cali_data_ddim
    qnn.set_quant_state(False, False)
    xs, _, xt, tt = generalized_steps(x0, seq, qnn, betas, eta=0.0, untill_fake_t=4)
    out["until4_xt"], out["until4_t"] = xt, tt
    save("f7_ddim_tiny", **out)


# ---------------------------------------------------------------------------- F8
def f8():
    """Tiny end-to-end calibration run (recipe of SURVEY Appendix B-6)."""
    import tempfile
    from quant.calibration import cali_model, load_cali_model
    cfg, m = tiny_model(seed=13)
    out = sd_arrays(m)
    g = torch.Generator().manual_seed(808)
    G, I = 3, 16
    xs = torch.randn(G * I, 3, 16, 16, generator=g)
    ts = torch.cat([torch.full((I,), float(t)) for t in (900, 500, 100)])
    out["cali_x"], out["cali_t"] = xs, ts
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    qnn = QuantModel(m, wq, aq, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
    torch.manual_seed(5)
    np.random.seed(5)
    path = os.path.join(tempfile.mkdtemp(), "c.pth")
    cali_model(qnn, (xs, ts), (xs, ts), use_aq=True, path=path, running_stat=True, interval=I,
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
    # reload into a fresh model + sample one eps with act_1 (load_cali_model round trip)
    cfg2, m2 = tiny_model(seed=13)
    qnn2 = QuantModel(m2, wq, aq, cali=False, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
    init = (torch.randn(1, 3, 16, 16, generator=g), torch.randint(0, 1000, (1,), generator=g))
    load_cali_model(qnn2, init, use_aq=True, path=path)
    qnn2.load_state_dict(ck["act_1"], strict=False)
    xe = torch.randn(2, 3, 16, 16, generator=g)
    te = torch.tensor([500.0, 500.0])
    with torch.no_grad():
        out["reload_x"], out["reload_t"] = xe, te
        out["reload_eps_act1"] = qnn2(xe, te)
    save("f8_cali_tiny", **out)


# ---------------------------------------------------------------------------- F9 / F10
def f9():
    out = {}
    for T in (10, 20, 50, 100):
        out[f"quad_{T}"] = [int(s) for s in list(np.linspace(0, np.sqrt(1000 * 0.8), T) ** 2)]
        out[f"uniform_{T}"] = list(range(0, 1000, 1000 // T))
    betas = torch.from_numpy(np.linspace(1e-4, 0.02, 1000, dtype=np.float64)).float()
    out["betas"] = betas
    from ddim.functions.denoising import compute_alpha
    t = torch.arange(-1, 1000)
    out["alpha_bar"] = compute_alpha(betas, t).reshape(-1)
    from ddim.models.diffusion import get_timestep_embedding
    out["temb_t"] = torch.tensor([0.0, 1.0, 17.0, 500.0, 999.0])
    out["temb_128"] = get_timestep_embedding(out["temb_t"], 128)
    out["temb_32"] = get_timestep_embedding(out["temb_t"], 32)
    save("f9_schedules", **out)


def f10():
    out = {}
    for I, W in ((256, 8), (512, 8), (256, 4), (16, 2)):
        n = I * 3
        data = torch.arange(n)
        for gpu in range(W):
            d = []
            for j in range(n // I):
                d.append(data[j * I + gpu * I // W: j * I + (gpu + 1) * I // W])
            out[f"I{I}_W{W}_r{gpu}"] = torch.cat(d)
    save("f10_shards", **out)


ALL = dict(f1=f1, f2=f2, f3=f3, f4=f4, f7=f7, f8=f8, f9=f9, f10=f10)

if __name__ == "__main__":
    which = sys.argv[1:] or list(ALL)
    for w in which:
        print("==", w)
        ALL[w]()

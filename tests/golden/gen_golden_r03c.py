"""Round-3 golden vectors, part c, produced by importing the reference (build container only):

    python tests/golden/gen_golden_r03c.py

  f23  FISHER-weighted reconstruction (`opt_mode=RLOSS.FISHER_DIAG / FISHER_FULL`; reference quant/reconstruction.py:58-61,177-180,
       quant/data_utill.py:54-73,191-256 `save_grad` / `GetLayerGrad`, quant/reconstruction_util.py:53-59).  No driver passes a Fisher
       `opt_mode`, so the reference's layer_ / block_reconstruction are called directly with it.  State: the tiny DDPM UNet of F8 (and the
       tiny SD-style UNet of F12) with the fixture's own checkpoint loaded the drivers' way (QuantModel(cali=False) -> load_cali_model),
       i.e. every unit hard-rounded; then per unit, each from that same state and with the host RNG re-seeded,
           layer_ / block_reconstruction(unit, opt_mode=FISHER_*, asym=True, iters=20, batch_size=16, w=0.01, warmup=0.2, use_aq=False).
       Recorded per unit: GetLayerGrad's raw dL/d(unit output) of the first calibration batch (before save_grad's abs() + 1), the cached
       Fisher weights, the total / reconstruction loss of every iteration and the units' final AdaRound alphas.
       Besides: LossFunc's two Fisher formulas on random pred / tgt / grad tensors with their autograd gradients (kernel check)."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refharness as H  # noqa: E402

torch = H.install()
from gen_golden import save, tiny_model  # noqa: E402
from quant.quant_layer import QMODE, QuantLayer, Scaler  # noqa: E402
from quant.quant_model import QuantModel  # noqa: E402

ITERS, BS = 20, 16
UNITS = (("block", "down.1.block.0", "FISHER_DIAG"), ("block", "mid.attn_1", "FISHER_DIAG"), ("layer", "up.1.upsample.conv", "FISHER_FULL"),
         ("block", "up.0.block.1", "FISHER_FULL"))
LDM_UNITS = (("block", "input_blocks.1.0", "FISHER_DIAG"), ("block", "input_blocks.1.1.transformer_blocks.0", "FISHER_DIAG"),
             ("layer", "input_blocks.1.1.proj_out", "FISHER_DIAG"), ("block", "output_blocks.1.0", "FISHER_FULL"))


def loss_formulas(out):
    from quant.reconstruction_util import LossFunc, RLOSS
    g = torch.Generator().manual_seed(5)
    for tag, shape in (("a", (6, 8, 5, 5)), ("b", (4, 24, 16))):
        pred = torch.randn(*shape, generator=g).requires_grad_(True)
        tgt, fg = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g).abs() + 1.0
        out[f"loss/{tag}/pred"], out[f"loss/{tag}/tgt"], out[f"loss/{tag}/fg"] = pred.detach(), tgt, fg
        for mode in ("FISHER_DIAG", "FISHER_FULL"):
            if mode == "FISHER_FULL" and len(shape) != 4:
                continue
            lf = LossFunc(o=None, round_loss=RLOSS.NONE, w=0.0, rec_loss=RLOSS[mode], max_count=10, b_range=(20, 2), decay_start=0.0, warmup=0.0, p=2.0)
            if pred.grad is not None:
                pred.grad = None
            val = lf(pred, tgt, fg)
            val.backward()
            out[f"loss/{tag}/{mode}/value"], out[f"loss/{tag}/{mode}/grad"] = val.detach(), pred.grad.clone()


def family(out, pre, fixture, build, units, xshape, cshape):
    import quant.reconstruction as REC
    import quant.data_utill as DU
    from quant.calibration import load_cali_model
    from quant.reconstruction_util import RLOSS, LossFunc
    from quant.adaptive_rounding import AdaRoundQuantizer
    f8 = np.load(os.path.join(HERE, fixture), allow_pickle=False)
    ck = {"weight": {str(k): torch.from_numpy(f8["ck/weight/" + str(k)]) for k in f8["weight_keys"]}}
    akeys = [str(k) for k in f8["act_keys"]]
    dk, zk = [k for k in akeys if k.endswith("delta")], [k for k in akeys if k.endswith("zero_point")]
    for gi in range(3):
        d, z = torch.from_numpy(f8[f"ck/act_{gi}/delta"]), torch.from_numpy(f8[f"ck/act_{gi}/zp"])
        ck[f"act_{gi}"] = {**{k: d[i].clone() for i, k in enumerate(dk)}, **{k: z[i].clone() for i, k in enumerate(zk)}}
    path = os.path.join(tempfile.mkdtemp(), "c.pth")
    torch.save(ck, path)
    data = (torch.from_numpy(f8["cali_x"]), torch.from_numpy(f8["cali_t"])) + ((torch.from_numpy(f8["cali_c"]),) if cshape else ())
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    g = torch.Generator().manual_seed(99)
    init = (torch.randn(1, *xshape, generator=g), torch.randint(0, 1000, (1,), generator=g)) + ((torch.randn(1, *cshape, generator=g),) if cshape else ())
    out[pre + "init_x"], out[pre + "init_t"] = init[0], init[1]
    if cshape:
        out[pre + "init_c"] = init[2]
    orig_call, orig_getgrad = LossFunc.__call__, DU.GetLayerGrad.__call__
    orig_save_inout = REC.save_inout

    def _contig(*a, **k):       # torch 2.10 CPU: contiguous cached tensors (gen_golden_ldm.f12's harness-side workaround)
        ci, co = orig_save_inout(*a, **k)
        return tuple(c.contiguous() for c in ci), (co.contiguous() if torch.is_tensor(co) else co)
    REC.save_inout = _contig
    for kind, name, mode in units:
        m = build()
        qnn = QuantModel(m, wq, aq, cali=False, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
        if hasattr(qnn, "set_grad_ckpt"):
            qnn.set_grad_ckpt(False)
        load_cali_model(qnn, init, use_aq=False, path=path)
        unit = dict(qnn.model.named_modules())[name]
        losses, raw = [], []

        def rec_call(self, pred, tgt, grad=None, _l=losses):
            r = orig_call(self, pred, tgt, grad)
            _l.append(float(r.detach()))
            return r

        def rec_grad(self, *a, _r=raw, **k):
            gg = orig_getgrad(self, *a, **k)
            _r.append(gg.detach().clone())
            return gg
        LossFunc.__call__, DU.GetLayerGrad.__call__ = rec_call, rec_grad
        torch.manual_seed(78)
        np.random.seed(78)
        try:
            kw = dict(cali_data=data, batch_size=BS, iters=ITERS, w=0.01, opt_mode=RLOSS[mode], asym=True, warmup=0.2, use_aq=False, multi_gpu=False)
            if kind == "layer":
                REC.layer_reconstruction(qnn, unit, **kw)
            else:
                REC.block_reconstruction(qnn, unit, **kw)
        finally:
            LossFunc.__call__, DU.GetLayerGrad.__call__ = orig_call, orig_getgrad
        assert len(losses) == ITERS and raw
        rawc = torch.cat(raw)
        out[f"{pre}{name}/mode"] = np.array(mode)
        out[f"{pre}{name}/raw_grad"] = rawc[:BS].clone()          # the first calibration batch (fixture size)
        out[f"{pre}{name}/loss"] = np.array(losses, dtype=np.float64)
        layers = [(n, mod) for n, mod in unit.named_modules() if isinstance(mod, QuantLayer)] if kind == "block" else [("", unit)]
        names = []
        for n, mod in layers:
            if isinstance(mod.wqtizer, AdaRoundQuantizer) and not mod.quant_emb:
                full = (name + "." + n).rstrip(".")
                names.append(full)
                out[f"{pre}{name}/alpha/{full}"] = mod.wqtizer.alpha.detach().clone()
        out[f"{pre}{name}/alpha_names"] = np.array(names)
        print(pre + name, mode, "raw grad", tuple(rawc.shape), "max |g|", float(rawc.abs().max()), "loss", losses[0], losses[3], losses[4], losses[-1])
    REC.save_inout = orig_save_inout


def f23():
    out = {"iters": np.array(ITERS), "batch_size": np.array(BS), "units": np.array([u for _, u, _ in UNITS]), "kinds": np.array([k for k, _, _ in UNITS]),
           "ldm/units": np.array([u for _, u, _ in LDM_UNITS]), "ldm/kinds": np.array([k for k, _, _ in LDM_UNITS])}
    loss_formulas(out)
    family(out, "", "f8_cali_tiny.npz", lambda: tiny_model(seed=13)[1], UNITS, (3, 16, 16), None)
    from gen_golden_ldm import build as build_sd
    family(out, "ldm/", "f12_ldm_cali_tiny.npz", build_sd, LDM_UNITS, (4, 8, 8), (5, 64))
    save("f23_fisher", **out)


if __name__ == "__main__":
    f23()

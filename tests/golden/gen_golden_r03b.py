"""Round-3 golden vectors, part b, produced by importing the reference (build container only):

    python tests/golden/gen_golden_r03b.py

  f22  DELTA-LEARNING reconstruction (`use_aq=True`; reference quant/reconstruction.py:36-48 layer, :135-166 block): no driver requests it
       (cali_model never forwards `use_aq` to the reconstruction calls), so the reference's functions are called directly.  State: the tiny
       DDPM UNet of F8 with F8's own checkpoint loaded the drivers' way (QuantModel(cali=False) -> load_cali_model -> act_1), i.e. hard
       AdaRound weights + initialised activation quantizers.  Then, each from that same state and with the host RNG re-seeded:
         layer_reconstruction(up.1.upsample.conv), block_reconstruction(down.1.block.0  [ResnetBlock with nin_shortcut]),
         block_reconstruction(down.1.attn.0 [AttnBlock: the first one, so that its captured input is not yet downstream of a rounding tie]) with use_aq=True, asym=True, iters=30, batch_size=48 (= the whole set), lr=5e-4
       (a larger lr than the default 4e-5 so that 30 iterations move the deltas measurably; full-set batches because Adam on a scalar is sign-driven and mini-batch noise makes the trajectory a random walk that no two summation orders share).  Recorded per unit: the names of the trained
       deltas, their values before / after every optimiser step, and the reconstruction loss of every iteration.
       The same for the tiny SD-style UNet of F12 (context-conditioned): block_reconstruction(input_blocks.1.0 [ResBlock]) and
       block_reconstruction(input_blocks.1.1.transformer_blocks.0 [BasicTransformerBlock: ten deltas]) -> keys prefixed "ldm/"."""
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

ITERS, LR, BS = 30, 5e-4, 48      # the whole calibration set per iteration: a deterministic gradient, smooth trajectories
UNITS = (("layer", "up.1.upsample.conv"), ("block", "down.1.block.0"), ("block", "down.1.attn.0"))
LDM_UNITS = (("block", "input_blocks.1.0"), ("block", "input_blocks.1.1.transformer_blocks.0"))


def f22():
    out = {"iters": np.array(ITERS), "lr": np.array(LR), "batch_size": np.array(BS), "units": np.array([u for _, u in UNITS]),
           "ldm/units": np.array([u for _, u in LDM_UNITS])}
    _family(out, "", "f8_cali_tiny.npz", lambda: tiny_model(seed=13)[1], UNITS, 2222, (3, 16, 16), None)
    from gen_golden_ldm import build as build_sd
    import quant.reconstruction as _rec
    _orig = _rec.save_inout

    def _contig(*a, **k):       # torch 2.10 CPU: contiguous cached tensors (gen_golden_ldm.f12's harness-side workaround)
        ci, co = _orig(*a, **k)
        return tuple(c.contiguous() for c in ci), (co.contiguous() if torch.is_tensor(co) else co)
    _rec.save_inout = _contig
    _family(out, "ldm/", "f12_ldm_cali_tiny.npz", build_sd, LDM_UNITS, 2223, (4, 8, 8), (5, 64))
    save("f22_delta_learning", **out)


def _family(out, pre, fixture, build, units, seed, xshape, cshape):
    import quant.reconstruction as REC
    from quant.calibration import load_cali_model
    from quant.reconstruction_util import RLOSS
    f8 = np.load(os.path.join(HERE, fixture), allow_pickle=False)
    # the reference's checkpoint of F8, rebuilt as the dict torch.save wrote
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
    g = torch.Generator().manual_seed(seed)
    init = (torch.randn(1, *xshape, generator=g), torch.randint(0, 1000, (1,), generator=g)) + ((torch.randn(1, *cshape, generator=g),) if cshape else ())
    out[pre + "init_x"], out[pre + "init_t"] = init[0], init[1]
    if cshape:
        out[pre + "init_c"] = init[2]
    # the losses of chosen iterations: wrap LossFunc.__call__
    from quant.reconstruction_util import LossFunc
    orig_call = LossFunc.__call__
    for kind, name in units:
        m = build()
        qnn = QuantModel(m, wq, aq, cali=False, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
        if hasattr(qnn, "set_grad_ckpt"):
            qnn.set_grad_ckpt(False)
        load_cali_model(qnn, init, use_aq=True, path=path)
        qnn.load_state_dict(ck["act_1"], strict=False)
        unit = dict(qnn.model.named_modules())[name]
        layers = [(n, mod) for n, mod in unit.named_modules() if isinstance(mod, QuantLayer)] if kind == "block" else [("", unit)]
        trained = [(n, mod) for n, mod in layers if not mod.quant_emb and mod.aqtizer.delta is not None and not mod.disable_aq]
        names = [(name + "." + n).rstrip(".") for n, _ in trained]
        before = torch.stack([mod.aqtizer.delta.detach().reshape(()).clone() for _, mod in trained])
        losses = []

        def rec_call(self, pred, tgt, grad=None, _l=losses):
            r = orig_call(self, pred, tgt, grad)
            _l.append(float(r.detach()))
            return r
        LossFunc.__call__ = rec_call
        traj = []
        orig_step = torch.optim.Adam.step

        def rec_step(self, *a, _t=traj, **k):
            r = orig_step(self, *a, **k)
            _t.append(torch.stack([p_.detach().reshape(()).clone() for p_ in self.param_groups[0]["params"]]))
            return r
        torch.optim.Adam.step = rec_step
        torch.manual_seed(77)
        np.random.seed(77)
        try:
            kw = dict(cali_data=data, batch_size=BS, iters=ITERS, w=0.01, opt_mode=RLOSS.MSE, asym=True, warmup=0.2, use_aq=True, lr=LR,
                      multi_gpu=False)
            if kind == "layer":
                REC.layer_reconstruction(qnn, unit, **kw)
            else:
                REC.block_reconstruction(qnn, unit, **kw)
        finally:
            LossFunc.__call__ = orig_call
            torch.optim.Adam.step = orig_step
        after = torch.stack([mod.aqtizer.delta.detach().reshape(()).clone() for _, mod in trained])
        assert len(losses) == ITERS and not torch.equal(before, after)
        out[f"{pre}{name}/names"] = np.array(names)
        out[f"{pre}{name}/before"], out[f"{pre}{name}/after"] = before, after
        out[f"{pre}{name}/loss"] = np.array(losses, dtype=np.float64)
        out[f"{pre}{name}/trajectory"] = torch.stack(traj)          # [iters, n deltas]: the deltas after every optimiser step
        assert torch.equal(traj[-1], after)
        print(pre + name, names, "\n  before", before.tolist(), "\n  after ", after.tolist(), "\n  loss", losses[0], losses[9], losses[19], losses[-1])


if __name__ == "__main__":
    f22()

"""Round-4 golden vectors, produced by importing the reference (build container only):

    python tests/golden/gen_golden_r04.py

  f25  DELTA-LEARNING reconstruction with the ATTENTION-MATMUL quantizers live (second half; reference
       quant/reconstruction.py:135-166: the `A` lists -- [aqtizer_q, aqtizer_k, aqtizer_v (, aqtizer_w)] of QuantAttnBlock, of attn1 / attn2 of a
       QuantBasicTransformerBlock, of QuantQKMatMul and QuantSMVMatMul -- join the optimiser's parameters).  No driver switches those
       quantizers on, so F21's procedure is repeated first: `use_aq = True` set by hand on the attention blocks, one forward over the
       calibration set for their lazy (MSE) initialisation.  State and recipe otherwise as F22 (tests/golden/gen_golden_r03b.py): the tiny DDPM
       UNet of F8 / the tiny SD-style UNet of F12 with their own checkpoints loaded the drivers' way, block_reconstruction(use_aq=True,
       asym=True, iters=30, batch_size=48 = the whole set, lr=5e-4), host RNG re-seeded per unit.  Units:
         down.1.attn.0                                 QuantAttnBlock: 4 layer deltas + q, k, v, w
         ldm/input_blocks.1.1.transformer_blocks.0     QuantBasicTransformerBlock: 10 layer deltas + 2 x (q, k, v) + 2 x w
       Recorded per unit: the names of the trained deltas in the optimiser's order, every attention quantizer's (delta, zero point, level)
       BEFORE the reconstruction, the deltas before / after every optimiser step, the reconstruction loss of every iteration."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refharness as H  # noqa: E402

torch = H.install()
from gen_golden import save, tiny_model  # noqa: E402
from gen_golden_r03 import _force_attention_quant  # noqa: E402
from quant.quant_layer import QMODE, QuantLayer, Scaler  # noqa: E402
from quant.quant_model import QuantModel  # noqa: E402

ITERS, LR, BS = 30, 5e-4, 48


def f25():
    out = {"iters": np.array(ITERS), "lr": np.array(LR), "batch_size": np.array(BS)}
    _family(out, "", "f8_cali_tiny.npz", lambda: tiny_model(seed=13)[1], ("down.1.attn.0",), 2225, (3, 16, 16), None)
    from gen_golden_ldm import build as build_sd
    import quant.reconstruction as _rec
    _orig = _rec.save_inout

    def _contig(*a, **k):       # torch 2.10 CPU: contiguous cached tensors (gen_golden_ldm.f12's harness-side workaround)
        ci, co = _orig(*a, **k)
        return tuple(c.contiguous() for c in ci), (co.contiguous() if torch.is_tensor(co) else co)
    _rec.save_inout = _contig
    _family(out, "ldm/", "f12_ldm_cali_tiny.npz", build_sd, ("input_blocks.1.1.transformer_blocks.0",), 2226, (4, 8, 8), (5, 64))
    save("f25_delta_learning_attention", **out)


def _wb_of(m):
    """{<layer>.w / <layer>.b: tensor} of an FP model's conv / linear layers under the names QuantLayer gives them"""
    d = {}
    for n, mod in m.named_modules():
        if isinstance(mod, (torch.nn.Conv2d, torch.nn.Linear)):
            d[n + ".w"] = mod.weight
            if mod.bias is not None:
                d[n + ".b"] = mod.bias
    return d


def _family(out, pre, fixture, build, units, seed, xshape, cshape):
    import quant.reconstruction as REC
    from quant.calibration import load_cali_model
    from quant.quant_block import QuantAttnBlock, QuantBasicTransformerBlock, QuantQKMatMul, QuantSMVMatMul
    from quant.reconstruction_util import LossFunc, RLOSS
    f8 = np.load(os.path.join(HERE, fixture), allow_pickle=False)
    ck = {"weight": {str(k): torch.from_numpy(f8["ck/weight/" + str(k)]) for k in f8["weight_keys"] if "ck/weight/" + str(k) in f8.files}}
    if len(ck["weight"]) < len(f8["weight_keys"]):           # F16 keeps the quantizer entries only: .w / .b are the model's own parameters
        ck["weight"].update({"model." + k_: v_.detach().clone() for k_, v_ in _wb_of(build()).items() if "model." + k_ in set(map(str, f8["weight_keys"]))})
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
    orig_call = LossFunc.__call__
    for name in units:
        m = build()
        qnn = QuantModel(m, wq, aq, cali=False, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
        if hasattr(qnn, "set_grad_ckpt"):
            qnn.set_grad_ckpt(False)
        load_cali_model(qnn, init, use_aq=True, path=path)
        qnn.load_state_dict(ck["act_1"], strict=False)
        quantizers, seen, hooks = _force_attention_quant(qnn)
        for h_ in hooks:
            h_.remove()
        with torch.no_grad():
            _ = qnn(*[d_[:BS] for d_ in data])               # lazy (MSE) initialisation of every attention quantizer on the calibration set
        unit = dict(qnn.model.named_modules())[name]
        layers = [(n, mod) for n, mod in unit.named_modules() if isinstance(mod, QuantLayer)]
        trained = [((name + "." + n).rstrip("."), mod.aqtizer) for n, mod in layers if not mod.quant_emb and mod.aqtizer.delta is not None and not mod.disable_aq]
        if isinstance(unit, QuantBasicTransformerBlock):
            A = [("attn1.aqtizer_q", unit.attn1.aqtizer_q), ("attn1.aqtizer_k", unit.attn1.aqtizer_k), ("attn1.aqtizer_v", unit.attn1.aqtizer_v),
                 ("attn2.aqtizer_q", unit.attn2.aqtizer_q), ("attn2.aqtizer_k", unit.attn2.aqtizer_k), ("attn2.aqtizer_v", unit.attn2.aqtizer_v)]
            if unit.attn1.aqtizer_w.level != (2 ** 16):
                A.append(("attn1.aqtizer_w", unit.attn1.aqtizer_w))
            if unit.attn2.aqtizer_w.level != (2 ** 16):
                A.append(("attn2.aqtizer_w", unit.attn2.aqtizer_w))
        elif isinstance(unit, QuantQKMatMul):                # reference quant/reconstruction.py:155-156
            A = [("aqtizer_q", unit.aqtizer_q), ("aqtizer_k", unit.aqtizer_k)]
        elif isinstance(unit, QuantSMVMatMul):               # :157-160
            A = [("aqtizer_v", unit.aqtizer_v)]
            if unit.aqtizer_w.level != (2 ** 16):
                A.append(("aqtizer_w", unit.aqtizer_w))
        else:
            assert isinstance(unit, QuantAttnBlock)
            A = [("aqtizer_q", unit.aqtizer_q), ("aqtizer_k", unit.aqtizer_k), ("aqtizer_v", unit.aqtizer_v)]
            if unit.aqtizer_w.level != (2 ** 16):
                A.append(("aqtizer_w", unit.aqtizer_w))
        for n, q in A:
            assert q.delta is not None, n
            out[f"{pre}{name}/attn_q/{n}/delta"], out[f"{pre}{name}/attn_q/{n}/zp"] = q.delta.detach().reshape(()).clone(), torch.tensor(float(q.zero_point))
            out[f"{pre}{name}/attn_q/{n}/level"] = torch.tensor(q.level)
        trained += [(name + "." + n, q) for n, q in A]
        names = [n for n, _ in trained]
        before = torch.stack([q.delta.detach().reshape(()).clone() for _, q in trained])
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
            REC.block_reconstruction(qnn, unit, cali_data=data, batch_size=BS, iters=ITERS, w=0.01, opt_mode=RLOSS.MSE, asym=True, warmup=0.2,
                                     use_aq=True, lr=LR, multi_gpu=False)
        finally:
            LossFunc.__call__ = orig_call
            torch.optim.Adam.step = orig_step
        after = torch.stack([q.delta.detach().reshape(()).clone() for _, q in trained])
        assert len(losses) == ITERS and len(traj) == ITERS and traj[0].numel() == len(trained), (len(losses), len(traj), traj[0].numel(), len(trained))
        assert torch.equal(traj[-1], after) and not torch.equal(before, after)
        out[f"{pre}{name}/names"] = np.array(names)
        out[f"{pre}{name}/attn_names"] = np.array([n for n, _ in A])
        out[f"{pre}{name}/before"], out[f"{pre}{name}/after"] = before, after
        out[f"{pre}{name}/loss"] = np.array(losses, dtype=np.float64)
        out[f"{pre}{name}/trajectory"] = torch.stack(traj)
        print(pre + name, names, "\n  before", before.tolist(), "\n  after ", after.tolist(), "\n  loss", losses[0], losses[9], losses[19], losses[-1])


def f26():
    """f26  the same for the stand-alone matmul modules of the LDM AttentionBlock (QKVAttentionLegacy's seams): block_reconstruction(use_aq=True)
    called on a QuantQKMatMul (A = [aqtizer_q, aqtizer_k]) and on a QuantSMVMatMul (A = [aqtizer_v, aqtizer_w]) of the tiny AttentionBlock UNet of
    F13 / F16 -- reference quant/reconstruction.py:155-160.  Reachable only by a direct call (recon_model stops at the enclosing block)."""
    from gen_golden_r03 import ATTN_UNET_KW
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    f13 = np.load(os.path.join(HERE, "f13_ldm_attnblock_tiny.npz"), allow_pickle=False)

    def build_attn():
        m = UNetModel(**ATTN_UNET_KW).eval()
        m.load_state_dict({k[3:]: torch.from_numpy(f13[k]) for k in f13.files if k.startswith("sd/")})
        return m
    out = {"iters": np.array(ITERS), "lr": np.array(LR), "batch_size": np.array(BS)}
    _family(out, "attnblock/", "f16_attnblock_cali_tiny.npz", build_attn,
            ("middle_block.1.attention.qkv_matmul", "middle_block.1.attention.smv_matmul", "input_blocks.1.1.attention.qkv_matmul"), 2227, (3, 8, 8), None)
    save("f26_delta_learning_qk_smv", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["f25", "f26"]
    if "f25" in which:
        f25()
    if "f26" in which:
        f26()

"""Operand precision of the reconstruction GEMMs (tfmq_set_gemm_precision; TFMQ_RECON_GEMM = f32 | bf16x3 | f16).

SURVEY section 7-1 asked for split-bf16 / fp16 operand MFMA in the AdaRound iterations "if the loss curve keeps parity"; fixture F8b (the
reference's own 400-iteration block reconstruction: losses at four counts of every unit, final masks) is the yardstick.  The bars are
those of tests/test_configs_r02_gpu.py::test_reconstruction_loss_curve_400_iterations (reconstruction loss within 5 %, rounding loss
within 2 %, >= 99 % identical final masks): bf16x3 must meet them; fp16 is measured and reported (its product error 2^-11 is of the size
of the reconstruction residuals themselves)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("M,N,K,ta,tb", [(512, 320, 2880, False, True), (2880, 320, 4096, True, False), (300, 96, 640, False, False)])
def test_gemm_precision_modes_vs_float64(M, N, K, ta, tb):
    import tfmq_dm_amd.ops as ops
    gen = torch.Generator().manual_seed(M + K)
    A = torch.randn((K, M) if ta else (M, K), generator=gen)
    B = torch.randn((N, K) if tb else (K, N), generator=gen)
    ref = ((A.t() if ta else A).double() @ (B.t() if tb else B).double())
    scale = float(ref.abs().max())
    Ad, Bd = A.to(DEV), B.to(DEV)
    err = {}
    for mode in ("f32", "bf16x3", "f16"):
        with ops.gemm_precision(mode):
            C = ops.gemm(Ad, Bd, trans_a=ta, trans_b=tb)
            C2 = ops.gemm(Ad, Bd, trans_a=ta, trans_b=tb)
        assert torch.equal(C, C2)                                   # deterministic in every mode
        err[mode] = float((C.cpu().double() - ref).abs().max()) / scale
    exact = ops.gemm(Ad, Bd, trans_a=ta, trans_b=tb)               # the context manager restored exact fp32
    assert float((exact.cpu().double() - ref).abs().max()) / scale == err["f32"]
    print(f"{M}x{N}x{K}: max-normalised error  f32 {err['f32']:.2e}  bf16x3 {err['bf16x3']:.2e}  f16 {err['f16']:.2e}")
    assert err["f32"] <= 2e-6
    assert err["bf16x3"] <= 2e-5 and err["bf16x3"] < 0.05 * err["f16"]       # 2^-16 per product, random signs over K
    assert err["f16"] <= 2e-3


def test_precision_is_a_property_of_the_handle():
    """ops.gemm_precision selects a second handle of the device (precision set once, at creation); the base handle is never toggled: a
    launch through it from INSIDE the context is the exact-fp32 product, bit for bit."""
    import ctypes as C
    import tfmq_dm_amd.ops as ops
    from tfmq_dm_amd import _lib
    gen = torch.Generator().manual_seed(11)
    A, B = torch.randn(512, 2880, generator=gen).to(DEV), torch.randn(2880, 320, generator=gen).to(DEV)
    exact = ops.gemm(A, B)
    base = _lib.handle(0)
    assert ops.handle(0) is base
    with ops.gemm_precision("bf16x3"):
        hp = ops.handle(0)
        assert hp is not base and hp is _lib.handle(0, 1) and hp.gemm_precision == 1
        split = ops.gemm(A, B)
        token = ops._gemm_prec.set(0)                  # what another thread / context sees: the base handle, untouched
        try:
            assert ops.handle(0) is base
            again = ops.gemm(A, B)
        finally:
            ops._gemm_prec.reset(token)
        with ops.gemm_precision("f16"):
            assert ops.handle(0) is _lib.handle(0, 2)
        assert ops.handle(0) is hp
    assert ops.handle(0) is base
    assert torch.equal(again, exact) and not torch.equal(split, exact)
    assert torch.equal(ops.gemm(A, B), exact)


def _loss_curve(golden, mode, monkeypatch):
    import tfmq_dm_amd.ddim.models as M
    import quant.reconstruction as REC
    from quant.calibration import cali_model
    from quant.quant_layer import QMODE, Scaler
    from quant.quant_model import QuantModel
    from quant.reconstruction_util import RLOSS
    monkeypatch.setenv("TFMQ_RECON_GEMM", mode)
    g, g8 = golden("f8b_cali_curve"), golden("f8_cali_tiny")
    m = M.Model(M.make_config(ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(8,), image_size=16, dropout=0.0))
    m.load_state_dict({k[3:]: T(g8[k]) for k in g8.files if k.startswith("sd/")})
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    qnn = QuantModel(m.to(DEV).eval(), wq, aq, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval().to(DEV)
    xs, ts = T(g8["cali_x"]), T(g8["cali_t"])
    trace = {"counts": tuple(int(c) for c in g["counts"]), "rows": [], "unit": 0}
    REC.LOSS_TRACE = trace
    try:
        torch.manual_seed(5)
        np.random.seed(5)
        md = cali_model(qnn, (xs, ts), (xs, ts), use_aq=True, path=None, running_stat=True, interval=16, iters=int(g["iters"]),
                        batch_size=8, w=0.01, asym=True, warmup=0.2, opt_mode=RLOSS.MSE, multi_gpu=False)
    finally:
        REC.LOSS_TRACE = None
    ref, mine = g["loss_rows"], np.array(trace["rows"])
    assert mine.shape == ref.shape and np.array_equal(mine[:, :2], ref[:, :2])
    rec_dev = float(np.max(np.abs(mine[:, 2] - ref[:, 2]) / (np.abs(ref[:, 2]) + 1e-7)))
    on = ref[:, 1] >= 80
    rnd_ref = ref[on, 3] - ref[on, 2]
    rnd_dev = float(np.max(np.abs(mine[on, 3] - rnd_ref) / np.abs(rnd_ref)))
    akeys, sizes = [str(k) for k in g["alpha_keys"]], [int(s) for s in g["alpha_sizes"]]
    ref_mask = np.unpackbits(g["masks_packed"])[:sum(sizes)].astype(bool)
    my_mask = torch.cat([(md["weight"][k] >= 0).reshape(-1) for k in akeys]).numpy()
    return rec_dev, rnd_dev, float((my_mask == ref_mask).mean())


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "f16"])
def test_reconstruction_loss_curve_under_reduced_operand_precision(golden, monkeypatch, mode):
    rec_dev, rnd_dev, agree = _loss_curve(golden, mode, monkeypatch)
    print(f"[TFMQ_RECON_GEMM={mode}] vs the reference's 400-iteration curve (F8b): worst reconstruction-loss deviation {rec_dev:.3%}, "
          f"worst rounding-loss deviation {rnd_dev:.3%}, final AdaRound masks identical {agree:.4%}")
    if mode in ("f32", "bf16x3"):       # the default (bf16x3) is held to the bars of exact fp32 products
        assert rec_dev <= 0.05 and rnd_dev <= 0.02 and agree >= 0.99
    else:
        # fp16 operands: recorded, not required (DESIGN.md section 4 quotes the printed numbers); it must still converge to a sane state
        assert agree >= 0.90 and np.isfinite(rec_dev)

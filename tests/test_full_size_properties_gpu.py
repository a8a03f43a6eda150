"""Size-independent properties at BASELINE.json's full sizes (Stable Diffusion v1 UNet, 859.5 M parameters, 64x64x4 latents,
77x768 context, w4a8 with a Finite-Set table) -- the oracle cannot run these sizes in seconds, so the checks are structural:

  * batch independence: eps of sample i does not depend on what else is in the batch (every kernel reduces per output
    element / per image / per (batch, head) in a fixed order) -- bit-exact;
  * run-to-run determinism -- bit-exact;
  * the fused forward (int8 / fp16 epilogue modes, fused GEGLU) equals the un-fused forward that exposes every unit's
    tensors -- bit-exact;
  * the Finite-Set table is honoured: a different step row gives a different eps, the same row the same eps;
  * quantizer idempotence at size: quantising the de-quantised tensor returns the same bins.
"""
import argparse
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def sd():
    import bench
    args = argparse.Namespace(batch=1, ddim_steps=2)
    run, fwd, cpu, info = bench.setup_sd(args, torch.device(DEV), 0, lambda *a: None)
    eng = fwd.__closure__ and None
    return run, fwd, info


def _engine_of(fwd):
    for c in fwd.__closure__:
        v = c.cell_contents
        if hasattr(v, "forward") and hasattr(v, "qtable"):
            return v
    raise RuntimeError("engine not found")


def test_sd_full_size_properties(sd):
    run, fwd, info = sd
    eng = _engine_of(fwd)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 64, 64, 4, generator=g).to(DEV)
    ctx = torch.randn(2, 77, 768, generator=g).to(DEV)
    t = torch.tensor([981.0, 981.0], device=DEV)
    with torch.cuda.stream(info["stream"]):
        info["step"].zero_()
        e2 = eng.forward(x, t, ctx).clone()
        e2b = eng.forward(x, t, ctx).clone()
        ea = eng.forward(x[:1].contiguous(), t[:1], ctx[:1].contiguous()).clone()
        eb = eng.forward(x[1:].contiguous(), t[1:], ctx[1:].contiguous()).clone()
        e_taps = eng.forward(x, t, ctx, taps={}).clone()
        eng.stream_f16 = False                  # the fp32 activation stream (what the tap-exposing forward runs on)
        e_f32 = eng.forward(x, t, ctx).clone()
        eng.stream_f16 = True
        info["step"].fill_(1)
        e_step1 = eng.forward(x, t, ctx).clone()
        info["step"].zero_()
        info["stream"].synchronize()
    assert torch.isfinite(e2).all()
    assert torch.equal(e2, e2b)                                   # deterministic
    assert torch.equal(e2[:1], ea) and torch.equal(e2[1:], eb)    # batch independent
    assert torch.equal(e_f32, e_taps)                             # fused == un-fused (same fp32 stream)
    # the fp16 activation stream (tensors between blocks stored as fp16; statistics from the fp32 values): measured, not
    # estimated -- deviation of eps from the fp32-stream forward at full SD size
    dev16 = float((e2 - e_f32).norm() / e_f32.norm())
    print("SD v1 full size: eps rel-L2 fp16 stream vs fp32 stream:", dev16)
    assert dev16 <= 4e-2       # measured 3.1e-2: the size of the engine's own deviation from the CPU oracle (bin flips of the
                               # 8-bit activation quantizers compound the same way for any rounding perturbation)
    assert not torch.equal(e2, e_step1)                           # the step's activation table is used


def test_sd_batch_independent_across_tile_choices(sd):
    """The tile shape of every conv / linear is measured per shape (hence per batch size) -- the GroupNorm statistics
    of the conv epilogue are summed in one canonical order, so a UNet-batch-12 forward still equals two batch-6
    forwards bit for bit although the two sizes run different tile kernels."""
    run, fwd, info = sd
    eng = _engine_of(fwd)
    g = torch.Generator().manual_seed(10)
    x = torch.randn(12, 64, 64, 4, generator=g).to(DEV)
    ctx = torch.randn(12, 77, 768, generator=g).to(DEV)
    t = torch.full((12,), 501.0, device=DEV)
    with torch.cuda.stream(info["stream"]):
        info["step"].zero_()
        e = eng.forward(x, t, ctx).clone()
        a = eng.forward(x[:6].contiguous(), t[:6], ctx[:6].contiguous()).clone()
        b = eng.forward(x[6:].contiguous(), t[6:], ctx[6:].contiguous()).clone()
        info["stream"].synchronize()
    assert len({v for k, v in eng.tiles.items() if k[1] == 12}) > 1      # several tile shapes in use
    assert torch.equal(e[:6], a) and torch.equal(e[6:], b)


def test_sd_guidance_pair_prefix_is_bit_identical_at_full_size(sd):
    """The full SD v1 UNet on a guidance batch: 6 latents under cat([uc, c]) -- `pair_prefix` (the pair's shared prefix computed once, at
    batch 6, copied where the members part) against the materialised 12-item batch the reference feeds its UNet.  Bit for bit, although
    the two forms run different tile kernels on the prefix's shapes."""
    run, fwd, info = sd
    eng = _engine_of(fwd)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(6, 64, 64, 4, generator=g).to(DEV)
    ctx = torch.randn(12, 77, 768, generator=g).to(DEV)
    t = torch.full((12,), 301.0, device=DEV)
    with torch.cuda.stream(info["stream"]):
        info["step"].zero_()
        full = eng.forward(torch.cat([x, x]).contiguous(), t, ctx).clone()
        pair = eng.forward(x, t, ctx, pair_prefix=True).clone()
        info["stream"].synchronize()
    assert torch.isfinite(full).all() and torch.equal(pair, full)
    assert not torch.equal(full[:6], full[6:])


def test_sd_fused_context_kv_projection_is_bit_identical(sd, monkeypatch):
    """Cross attention at 640 / 1280 channels: to_k | to_v as ONE quantise pass + ONE GEMM (k as fp16 rows, v transposed) when their
    quantizers agree at every step, against the two separate projections (TFMQ_FUSED_KV=0)."""
    run, fwd, info = sd
    eng = _engine_of(fwd)
    assert any(f.kind == "w4a8" for f in eng.fused_kv.values())          # the synthetic tables do give to_k / to_v equal rows
    g = torch.Generator().manual_seed(14)
    x = torch.randn(4, 64, 64, 4, generator=g).to(DEV)
    ctx = torch.randn(4, 77, 768, generator=g).to(DEV)
    outs = []
    with torch.cuda.stream(info["stream"]):
        for flag in ("1", "0"):
            monkeypatch.setenv("TFMQ_FUSED_KV", flag)
            info["step"].zero_()
            outs.append(eng.forward(x, None, ctx).clone())
        info["stream"].synchronize()
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


def test_sd_row_chains_and_fused_ff_are_bit_identical_at_full_size(sd, monkeypatch):
    """Round 4: at the 64 x 64 level the transformer blocks run as three token-per-lane launches around the attentions -- tfmq_row_chain
    [norm + proj_in + norm1 + q|k|v], tfmq_row_chain [attn1.to_out + residual + norm2 + attn2.to_q], tfmq_ff_fused [norm3 + GEGLU + ff.net.2
    + residual] -- against the separate launches (TFMQ_ROW_CHAIN=0, TFMQ_FF_FUSED=0): eps bit for bit, plain batch and guidance pair."""
    run, fwd, info = sd
    eng = _engine_of(fwd)
    g = torch.Generator().manual_seed(15)
    x = torch.randn(3, 64, 64, 4, generator=g).to(DEV)
    ctx = torch.randn(6, 77, 768, generator=g).to(DEV)
    outs = {}
    monkeypatch.setenv("TFMQ_CHAIN_MIN_TOKENS", "0")       # (the engine's size policy keeps the chains for >= 32768 tokens: ops.chain_tokens_ok)
    with torch.cuda.stream(info["stream"]):
        for chain, ff, ffc in (("1", "1", "1"), ("0", "0", "1"), ("1", "0", "1"), ("0", "1", "1"), ("1", "1", "0")):
            monkeypatch.setenv("TFMQ_ROW_CHAIN", chain)
            monkeypatch.setenv("TFMQ_FF_FUSED", ff)
            monkeypatch.setenv("TFMQ_FF_CHAIN", ffc)        # attn2.to_out in front of / proj_out behind the feed-forward, in its launch
            info["step"].zero_()
            outs[(chain, ff, "plain", ffc)] = eng.forward(x, None, ctx[:3].contiguous()).clone()
            outs[(chain, ff, "pair", ffc)] = eng.forward(x, None, ctx, pair_prefix=True).clone()
        info["stream"].synchronize()
    ref_plain, ref_pair = outs[("0", "0", "plain", "1")], outs[("0", "0", "pair", "1")]
    assert torch.isfinite(ref_plain).all() and torch.isfinite(ref_pair).all()
    for k, v in outs.items():
        assert torch.equal(v, ref_plain if k[2] == "plain" else ref_pair), k


def test_sd_full_size_eps_vs_oracle(sd):
    """The full SD v1 UNet (859.5 M, w4a8, synthetic Finite-Set table) against the CPU oracle -- the reference's
    fake-quant forward restated on torch-CPU, pinned to the reference at tiny sizes by F11-F13 -- on one CFG pair:
    the same bar as the tiny fixtures (w4a8 eps rel-L2 <= 3e-2 there; 5e-2 allowed here, measured 3.1e-2: bin flips
    of the fp16 layers).  About 20 s of host time."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tfmq_oracle as O
    from tfmq_dm_amd.ldm.sampler import ddim_timesteps
    run, fwd, info = sd
    st = info["oracle_state"]
    eng, sdw, wq, act_names, cfg = st["eng"], st["sd"], st["wq"], st["act_names"], st["cfg"]
    g = torch.Generator().manual_seed(123)
    x = torch.randn(2, 4, 64, 64, generator=g)
    ctx = torch.randn(2, 77, 768, generator=g)
    t = torch.full((2,), float(np.flip(ddim_timesteps(2))[0]))
    with torch.cuda.stream(info["stream"]):
        info["step"].zero_()
        e = eng.forward(x.permute(0, 2, 3, 1).contiguous().to(DEV), t.to(DEV), ctx.to(DEV)).permute(0, 3, 1, 2).clone()
        info["stream"].synchronize()
    sdc = {k: v.cpu() for k, v in sdw.items()}
    wqc = {n: {"delta": q.delta.cpu().reshape((-1,) + (1,) * (sdc[n + ".weight"].dim() - 1)),
               "zp": q.zp.cpu().reshape((-1,) + (1,) * (sdc[n + ".weight"].dim() - 1)), "alpha": None} for n, q in wq.items()}
    qt = eng.qtable.cpu()
    aq = {n: (qt[0, j, 0], qt[0, j, 1]) for j, n in enumerate(act_names)}
    with torch.no_grad():
        ref = O.ldm_unet_forward(sdc, dict(cfg), x, t.long(), ctx, O.QuantSpec(wq=wqc, aq=aq))
    rel = float((e.cpu() - ref).norm() / ref.norm())
    print("full SD v1 w4a8 eps rel-L2 vs oracle:", rel)
    assert rel <= 5e-2


def test_quantizer_idempotent_at_size():
    import tfmq_dm_amd.ops as ops
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(16, 64, 64, 320, generator=g) * 3).to(DEV)
    qt = torch.tensor([[0.0471, 113.0]], device=DEV)
    sel = ops.qsel(qt)
    q1 = ops.quantize_act(x, sel)
    deq = (q1.float() + 128.0 - 113.0) * 0.0471
    assert torch.equal(ops.quantize_act(deq, sel), q1)


@pytest.mark.parametrize("preset", ["cin256", "celeba"])
def test_ldm_full_size_presets(preset):
    """BASELINE.json configs[4] / configs[2] at their full sizes -- the cin256-v2 class-conditional UNet (one attention head per
    level: head dims 384 / 576 / 960, cross attention over ONE class token, CFG) and the unconditional LDM-4 CelebA-HQ UNet
    (224 ... 896 channels: Cin % 64 != 0 layers, plain AttentionBlocks with 32-channel heads), w4a8 with a Finite-Set table:
    run-to-run determinism and batch independence bit for bit, fused forward == tap-exposing forward on the fp32 stream,
    the step's table row is honoured, and one sample / CFG pair against the CPU oracle (same bar as the SD test)."""
    import numpy as np
    import bench
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tfmq_oracle as O
    from tfmq_dm_amd.ldm.sampler import ddim_timesteps
    args = argparse.Namespace(batch=1, ddim_steps=2)
    run, fwd, cpu, info = bench.setup_sd(args, torch.device(DEV), 0, lambda *a: None, preset=preset)
    st = info["oracle_state"]
    eng, sdw, wq, act_names, cfg = st["eng"], st["sd"], st["wq"], st["act_names"], st["cfg"]
    P = bench.LDM_PRESETS[preset]
    C, H, W = P["latent"]
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, C, H, W, generator=g)
    ctx = None if P["ctx"] is None else torch.randn(2, P["ctx"][0], P["ctx"][1], generator=g)
    t = torch.full((2,), float(np.flip(ddim_timesteps(2))[0]))
    xd, td = x.permute(0, 2, 3, 1).contiguous().to(DEV), t.to(DEV)
    cd = None if ctx is None else ctx.to(DEV)

    def f(xx, tt, cc, **kw):
        return eng.forward(xx, tt, cc, **kw).clone()
    with torch.cuda.stream(info["stream"]):
        info["step"].zero_()
        e2, e2b = f(xd, td, cd), f(xd, td, cd)
        ea = f(xd[:1].contiguous(), td[:1], None if cd is None else cd[:1].contiguous())
        eb = f(xd[1:].contiguous(), td[1:], None if cd is None else cd[1:].contiguous())
        e_taps = f(xd, td, cd, taps={})
        eng.stream_f16 = False
        e_f32 = f(xd, td, cd)
        eng.stream_f16 = True
        info["step"].fill_(1)
        e_step1 = f(xd, td, cd)
        info["step"].zero_()
        info["stream"].synchronize()
    assert torch.isfinite(e2).all() and torch.equal(e2, e2b)
    assert torch.equal(e2[:1], ea) and torch.equal(e2[1:], eb)
    assert torch.equal(e_f32, e_taps)
    assert not torch.equal(e2, e_step1)
    sdc = {k: v.cpu() for k, v in sdw.items()}
    wqc = {n: {"delta": q.delta.cpu().reshape((-1,) + (1,) * (sdc[n + ".weight"].dim() - 1)),
               "zp": q.zp.cpu().reshape((-1,) + (1,) * (sdc[n + ".weight"].dim() - 1)), "alpha": None} for n, q in wq.items()}
    qt = eng.qtable.cpu()
    aq = {n: (qt[0, j, 0], qt[0, j, 1]) for j, n in enumerate(act_names)}
    with torch.no_grad():
        ref = O.ldm_unet_forward(sdc, dict(cfg), x, t.long(), ctx, O.QuantSpec(wq=wqc, aq=aq))
    rel = float((e2.permute(0, 3, 1, 2).cpu() - ref).norm() / ref.norm())
    print(f"full-size {preset} w4a8 eps rel-L2 vs oracle:", rel)
    assert rel <= 5e-2

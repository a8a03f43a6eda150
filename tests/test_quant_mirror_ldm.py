"""quant/ mirror on the Stable-Diffusion-style UNet (SpatialTransformer UNetModel): tree rewrite, layer order and flags
against what the reference's QuantModel produced (fixture F11), and -- on the GPU -- QuantModel.forward lowered to the
LDM engine against the reference's fake-quantised eps (F11) and its end-to-end calibration (F12)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))  # drop-in: `import quant.*`

UNET_KW = dict(image_size=8, in_channels=4, model_channels=32, out_channels=4, num_res_blocks=1,
               attention_resolutions=[1, 2], channel_mult=[1, 2], num_heads=2, use_spatial_transformer=True,
               transformer_depth=1, context_dim=64, legacy=False)
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.asarray(a))


def tiny_qnn(g=None, cali=True, device=None):
    from tfmq_dm_amd.ldm.unet import UNetModel
    from quant.quant_layer import QMODE, Scaler
    from quant.quant_model import QuantModel
    m = UNetModel(**UNET_KW)
    if g is not None:
        m.load_state_dict({k[3:]: T(g[k]) for k in g.files if k.startswith("sd/")})
    if device is not None:
        m = m.to(device)
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    return QuantModel(m, wq, aq, cali=cali, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()


def test_ldm_tree_rewrite_matches_reference(golden):
    g = golden("f11_ldm_tiny")
    q = tiny_qnn(g)
    names = [n for n, _ in q.named_quant_layers()]
    assert names == [str(n) for n in g["quant_layer_names"]]            # same layers, same module order
    assert not any(("skip_connection" in n or n.endswith(".op")) for n in names)     # quant_model.py:57-58
    from quant.quant_block import QuantBasicTransformerBlock, QuantResBlock, QuantTemporalInformationBlock
    kinds = [type(m).__name__ for m in q.model.modules()]
    n_res = sum(n.endswith(".emb_layers.1") for n in names)
    n_tb = sum(n.endswith(".attn1.to_q") for n in names)
    assert kinds.count("QuantResBlock") == n_res == 8 and kinds.count("QuantBasicTransformerBlock") == n_tb == 7
    assert isinstance(q.tib, QuantTemporalInformationBlock) and len(q.tib.emb_layers) == n_res
    assert all(seq[1].quant_emb for seq in q.tib.emb_layers)
    # first / last layer policy (quant_model.py:103-120): time_embed.0, conv_in, out.2 FP; time_embed.2 and the first
    # ResBlock conv weight-only
    q.set_quant_state(True, True)
    q.disable_out_quantization()
    q.set_quant_state(True, True)
    layers = dict(q.named_quant_layers())
    assert names[0] == "time_embed.0" and names[2] == "input_blocks.0.0" and names[-1] == "out.2"
    assert not layers[names[0]].use_wq and not layers[names[2]].use_wq and not layers[names[-1]].use_wq
    assert layers[names[1]].use_wq and layers[names[1]].disable_aq and layers[names[3]].disable_aq
    ref_aq = {k[3:-6] for k in g.files if k.startswith("aq/") and k.endswith("/delta")}
    assert set(q.act_layer_names()) == ref_aq


def test_ldm_state_dict_schema_matches_reference_checkpoint(golden):
    g = golden("f12_ldm_cali_tiny")
    ref_keys = set(str(k) for k in g["weight_keys"])
    q = tiny_qnn(cali=False)
    mine = set(q.state_dict().keys())
    assert mine <= ref_keys
    assert all("wqtizer" in k for k in ref_keys - mine)


@pytest.mark.gpu
def test_ldm_quantmodel_forward_vs_reference(golden):
    """QuantModel(UNetModel) lowered to the HIP engine: weight quantizers initialised on the device (MSE search), then
    the reference's own forward results: w4 (weight-only) and the TIB outputs."""
    g = golden("f11_ldm_tiny")
    q = tiny_qnn(g, device=DEV)
    x, t, ctx = T(g["x"]).to(DEV), T(g["t"]).float().to(DEV), T(g["ctx"]).to(DEV)
    q.set_quant_state(False, False)
    eps = q(x, t, ctx).cpu()
    ref = T(g["eps_fp"])
    assert float((eps - ref).abs().max() / ref.abs().max()) <= 1e-2
    q.set_quant_state(True, False)
    q(x, t, ctx)                              # initialises every weight quantizer
    q.disable_out_quantization()
    q.invalidate()
    eps = q(x, t, ctx).cpu()
    ref = T(g["eps_w4"])
    assert float((eps - ref).norm() / ref.norm()) <= 3e-2
    tib = torch.cat(q.tib(x, t), dim=1).cpu()
    np.testing.assert_allclose(tib.numpy(), g["tib_w4"], rtol=0, atol=5e-4 * float(np.abs(g["tib_w4"]).max()))
    # weight scales found on the device == the reference's (MSE search, per channel)
    layers = dict(q.named_quant_layers())
    same = tot = 0
    for k in g.files:
        if k.startswith("wq/") and k.endswith("/delta"):
            n = k[3:-6]
            d, z, _ = layers[n].weight_quant_state()
            same += int(np.isclose(d.reshape(-1).cpu().numpy(), np.asarray(g[k]).reshape(-1), rtol=1e-5).sum())
            tot += d.numel()
    assert same / tot >= 0.98


@pytest.mark.gpu
def test_ldm_blocks_run_in_isolation(golden):
    """QuantResBlock / QuantBasicTransformerBlock forward on their own (what block reconstruction and save_inout call)
    equals the same block inside the whole-model plan (engine taps)."""
    import tfmq_dm_amd.ops as ops
    g = golden("f11_ldm_tiny")
    q = tiny_qnn(g, device=DEV)
    x, t, ctx = T(g["x"]).to(DEV), T(g["t"]).float().to(DEV), T(g["ctx"]).to(DEV)
    q.set_quant_state(True, False)
    q(x, t, ctx)
    q.disable_out_quantization()
    q.invalidate()
    taps = {}
    eng = q.engine(DEV)
    eng.forward(ops.nchw_to_nhwc(x), t, ctx, taps=taps)
    res = q.model.input_blocks[3][0]
    hin, hout = taps["input_blocks.3.0"]
    y = res(ops.nhwc_to_nchw(hin), taps["__temb__"])
    # (the isolated block computes its GroupNorm statistics in the GroupNorm kernel, the engine in the producing conv's
    # epilogue: two fp32 summation orders)
    assert float((ops.nchw_to_nhwc(y) - hout).abs().max() / hout.abs().max()) <= 3e-5
    tb = q.model.input_blocks[1][1].transformer_blocks[0]
    tin, tout = taps["input_blocks.1.1.transformer_blocks.0"]
    y = tb(tin[0], tin[1])
    assert float((y - tout).abs().max() / tout.abs().max()) <= 1e-5

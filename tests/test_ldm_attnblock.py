"""Unconditional latent-diffusion UNets (CelebA-HQ / LSUN configs): UNetModel with plain AttentionBlocks, against the
reference (fixture F13): tree rewrite (the QKMatMul / SMVMatMul seams become Quant*MatMul blocks, Conv1d projections stay
un-quantised), FP / w4 / w4a8 eps through QuantModel -> engine, the unconditional DDIM sampler and calibration-set
generator."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))

KW = dict(image_size=8, in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=[1, 2],
          channel_mult=[1, 2], num_head_channels=16, use_spatial_transformer=False)
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel_l2(a, b):
    return float((a - b).norm() / b.norm())


def qnn_of(g, device=None, cali=True):
    from tfmq_dm_amd.ldm.unet import UNetModel
    from quant.quant_layer import QMODE, Scaler
    from quant.quant_model import QuantModel
    m = UNetModel(**KW)
    m.load_state_dict({k[3:]: T(g[k]) for k in g.files if k.startswith("sd/")})
    if device is not None:
        m = m.to(device)
    wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MSE}
    aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MSE, "leaf_param": True}
    return QuantModel(m, wq, aq, cali=cali, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()


def test_attnblock_tree_rewrite_matches_reference(golden):
    from quant.quant_block import BaseQuantBlock
    g = golden("f13_ldm_attnblock_tiny")
    q = qnn_of(g)
    assert [n for n, _ in q.named_quant_layers()] == [str(n) for n in g["quant_layer_names"]]
    blocks = [f"{n}:{type(m).__name__}" for n, m in q.model.named_modules() if isinstance(m, BaseQuantBlock)]
    assert blocks == [str(b) for b in g["quant_block_names"]]
    assert not any(".qkv" in n or ".proj_out" in n for n, _ in q.named_quant_layers())    # Conv1d: not in QMAP


@pytest.mark.gpu
def test_attnblock_unet_eps_vs_reference(golden):
    g = golden("f13_ldm_attnblock_tiny")
    q = qnn_of(g, DEV)
    x, t = T(g["x"]).to(DEV), T(g["t"]).float().to(DEV)
    q.set_quant_state(False, False)
    eps = q(x, t).cpu()
    ref = T(g["eps_fp"])
    assert float((eps - ref).abs().max() / ref.abs().max()) <= 1e-2
    q.set_quant_state(True, False)
    q(x, t)
    q.disable_out_quantization()
    q.invalidate()
    assert rel_l2(q(x, t).cpu(), T(g["eps_w4"])) <= 3e-2
    # w4a8 with the reference's activation parameters
    q.set_quant_state(True, True)
    layers = dict(q.named_quant_layers())
    for n in q.act_layer_names():
        aqz = layers[n].aqtizer
        aqz.delta = torch.nn.Parameter(T(g[f"aq/{n}/delta"]).reshape(()).to(DEV))
        aqz.zero_point = torch.nn.Parameter(T(g[f"aq/{n}/zp"]).reshape(()).to(DEV))
        aqz.init = True
    q.invalidate()
    assert rel_l2(q(x, t).cpu(), T(g["eps_w4a8"])) <= 4e-2
    # the re-hosted block runs on its own exactly as inside the plan
    import tfmq_dm_amd.ops as ops
    q.set_quant_state(False, False)
    taps = {}
    q.engine(DEV).forward(ops.nchw_to_nhwc(x), t, None, taps=taps)
    hin, hout = taps["input_blocks.1.1"]
    from quant.quant_block import QuantAttentionBlock
    y = QuantAttentionBlock(q.model.input_blocks[1][1], {"bits": 8, "channel_wise": False, "scaler": None, "leaf_param": False})(ops.nhwc_to_nchw(hin))
    assert float((ops.nchw_to_nhwc(y) - hout).abs().max() / hout.abs().max()) <= 1e-5


@pytest.mark.gpu
def test_unconditional_sampler_and_calibration_set(golden):
    from tfmq_dm_amd.ldm.ddpm import LatentDiffusion
    from tfmq_dm_amd.ldm.ddim import DDIMSampler
    from quant.data_generate import generate_cali_data_ldm
    g = golden("f13_ldm_attnblock_tiny")
    q = qnn_of(g, DEV)
    q.set_quant_state(False, False)
    m = LatentDiffusion(q, conditioning_key=None, linear_start=0.0015, linear_end=0.0195).to(DEV)
    assert np.array_equal(m.alphas_cumprod.cpu().numpy(), g["alphas_cumprod"])
    out, _ = DDIMSampler(m).sample(S=4, batch_size=2, shape=[3, 8, 8], verbose=False, eta=0.0, x_T=T(g["traj_xT"]).to(DEV))
    assert rel_l2(out.cpu(), T(g["traj_fp_final"])) <= 2e-2
    for plms in (False, True):
        xs, ts = generate_cali_data_ldm(m, T=4, c=2, batch_size=3, shape=[3, 8, 8], plms=plms)
        assert xs.shape == (6, 3, 8, 8) and ts.tolist() == [501] * 3 + [1] * 3 and torch.isfinite(xs).all()


@pytest.mark.gpu
def test_unconditional_graph_sampler_matches_dropin(golden):
    """GraphLatentDdimSampler without a context (unconditional LDM: one UNet call per step, plain DDIM update) against the
    drop-in DDIMSampler on the same weights (and through it against the reference trajectory of F13)."""
    from tfmq_dm_amd.ldm.ddpm import LatentDiffusion
    from tfmq_dm_amd.ldm.ddim import DDIMSampler
    from tfmq_dm_amd.ldm.sampler import GraphLatentDdimSampler
    g = golden("f13_ldm_attnblock_tiny")
    q = qnn_of(g, DEV)
    q.set_quant_state(False, False)
    m = LatentDiffusion(q, conditioning_key=None, linear_start=0.0015, linear_end=0.0195).to(DEV)
    x_T = T(g["traj_xT"]).to(DEV)
    ref, _ = DDIMSampler(m).sample(S=4, batch_size=2, shape=[3, 8, 8], verbose=False, eta=0.0, x_T=x_T)
    eng = q.engine(DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    eng.prepare(None, None, step)
    smp = GraphLatentDdimSampler(eng, 4, 2, (3, 8, 8), None, alphas_cumprod=m.alphas_cumprod.cpu()).capture()
    out = smp.sample_nhwc(x_T.permute(0, 2, 3, 1).contiguous())
    smp.stream.synchronize()
    out = out.permute(0, 3, 1, 2).clone()
    assert int(step.item()) == 4
    assert float((out - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    assert rel_l2(out.cpu(), T(g["traj_fp_final"])) <= 2e-2

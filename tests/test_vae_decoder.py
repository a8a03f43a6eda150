"""First-stage decoder: oracle and HIP engine against fixture F14, produced by the reference's own
Decoder (tests/golden/gen_golden_vae.py; ldm/modules/diffusionmodules/model.py:462-570, autoencoder.py:329-332,
ddpm.py:706-708)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import tfmq_oracle as O

CFG = dict(ch_mult=(1, 2), num_res_blocks=1, resolution=16, attn_resolutions=[])
DEV = "cuda:0"


@pytest.fixture(scope="module")
def f14():
    d = np.load(os.path.join(HERE, "golden", "f14_vae_decoder_tiny.npz"))
    sd = {k[3:]: torch.tensor(d[k]) for k in d.files if k.startswith("sd/")}
    return sd, torch.tensor(d["z"]), float(d["scale_factor"]), torch.tensor(d["pre_end"]), torch.tensor(d["img"])


def test_oracle_matches_reference_decoder(f14):
    sd, z, sf, pre, img = f14
    assert torch.equal(O.vae_decoder_forward(sd, CFG, z, sf, pre_end=True), pre)      # same ATen calls: bit-exact
    assert torch.equal(O.vae_decoder_forward(sd, CFG, z, sf), img)


@pytest.mark.gpu
def test_engine_matches_reference_decoder(f14):
    from tfmq_dm_amd.engine.vae_decoder import VaeDecoderEngine
    sd, z, sf, pre, img = f14
    eng = VaeDecoderEngine(sd, CFG, DEV)
    zz = z.permute(0, 2, 3, 1).contiguous().to(DEV)
    got_pre = eng.forward(zz, scale_factor=sf, pre_end=True).permute(0, 3, 1, 2).cpu()
    got = eng.forward(zz, scale_factor=sf).permute(0, 3, 1, 2).cpu()
    # un-quantised convs run on f16 operands (fp32 accumulation): 2^-11 relative per operand, ~10 layers deep
    assert (got_pre - pre).abs().max() <= 4e-3 * float(pre.abs().max())
    assert (got - img).abs().max() <= 4e-3 * float(img.abs().max())
    # batch independence / determinism (bit-exact)
    one = eng.forward(zz[1:2].contiguous(), scale_factor=sf).permute(0, 3, 1, 2).cpu()
    assert torch.equal(one, got[1:2])


@pytest.mark.gpu
def test_first_stage_dropin_and_decode_first_stage(f14):
    from tfmq_dm_amd.ldm.autoencoder import FirstStageDecoder
    from tfmq_dm_amd.ldm.ddpm import LatentDiffusion
    sd, z, sf, pre, img = f14
    fs = FirstStageDecoder(sd, CFG, DEV)
    ldm = LatentDiffusion(torch.nn.Identity(), conditioning_key=None)
    ldm.first_stage_model, ldm.scale_factor = fs, sf
    got = ldm.decode_first_stage(z.to(DEV)).cpu()
    assert got.shape == img.shape
    assert (got - img).abs().max() <= 4e-3 * float(img.abs().max())
    # decode() alone takes already-scaled latents (first_stage_model.decode(1/scale_factor * z))
    got2 = fs.decode((1.0 / sf * z).to(DEV)).cpu()
    assert (got2 - img).abs().max() <= 4e-3 * float(img.abs().max())


@pytest.mark.gpu
def test_sd_size_decoder_properties():
    """SD v1 first stage (ch 128, mult 1-2-4-4, 2 res blocks, 64x64x4 latents -> 512x512x3) with seeded random weights:
    finite, deterministic, batch independent; the exact-fp32 wide-head attention (512 channels, 4096 tokens) runs."""
    from tfmq_dm_amd.engine.vae_decoder import VaeDecoderEngine
    cfg = dict(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, resolution=256, attn_resolutions=[], z_channels=4, out_ch=3)
    g = torch.Generator().manual_seed(2)
    sd = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = torch.randn(co, ci, k, k, generator=g) * (1.0 / (ci * k * k) ** 0.5)
        sd[name + ".bias"] = torch.randn(co, generator=g) * 0.02

    def norm(name, c):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.05 * torch.randn(c, generator=g)

    def res(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", co, ci, 3); norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".nin_shortcut", co, ci, 1)

    conv("post_quant_conv", 4, 4, 1)
    ch, mult = 128, (1, 2, 4, 4)
    bi = ch * mult[-1]
    conv("decoder.conv_in", bi, 4, 3)
    res("decoder.mid.block_1", bi, bi)
    norm("decoder.mid.attn_1.norm", bi)
    for s in ("q", "k", "v", "proj_out"):
        conv("decoder.mid.attn_1." + s, bi, bi, 1)
    res("decoder.mid.block_2", bi, bi)
    for i in reversed(range(4)):
        bo = ch * mult[i]
        for j in range(3):
            res(f"decoder.up.{i}.block.{j}", bi, bo)
            bi = bo
        if i != 0:
            conv(f"decoder.up.{i}.upsample.conv", bi, bi, 3)
    norm("decoder.norm_out", bi)
    conv("decoder.conv_out", 3, bi, 3)
    eng = VaeDecoderEngine(sd, cfg, DEV)
    z = (torch.randn(2, 64, 64, 4, generator=g) * 5.0).to(DEV)
    a = eng.forward(z, scale_factor=0.18215).clone()
    b = eng.forward(z, scale_factor=0.18215).clone()
    c = eng.forward(z[:1].contiguous(), scale_factor=0.18215).clone()
    torch.cuda.synchronize()
    assert a.shape == (2, 512, 512, 3) and bool(torch.isfinite(a).all())
    assert torch.equal(a, b) and torch.equal(a[:1], c)

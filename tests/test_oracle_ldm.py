"""Pins the oracle's latent-diffusion (SD-style) UNet, TIB and CFG-DDIM sampler to the reference (fixture F11)."""
import numpy as np
import pytest
import torch

import tfmq_oracle as O

CFG = dict(model_channels=32, num_heads=2)


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def f11(golden):
    g = golden("f11_ldm_tiny")
    sd = {k[3:]: T(g[k]) for k in g.files if k.startswith("sd/")}
    return g, sd


def spec(g, with_act):
    wq, aq = {}, {}
    for k in g.files:
        if k.startswith("wq/") and k.endswith("/delta"):
            n = k[3:-6]
            wq[n] = {"delta": T(g[k]), "zp": T(g[f"wq/{n}/zp"]), "alpha": None}
        if with_act and k.startswith("aq/") and k.endswith("/delta"):
            n = k[3:-6]
            aq[n] = (T(g[k]), T(g[f"aq/{n}/zp"]))
    return O.QuantSpec(wq=wq, aq=aq)


def test_ldm_forward_fp_w4_w4a8(f11):
    g, sd = f11
    x, t, ctx = T(g["x"]), T(g["t"]), T(g["ctx"])
    assert np.array_equal(O.timestep_embedding_ldm(torch.tensor([0, 1, 21, 981]), 32).numpy(), g["temb_32"])
    with torch.no_grad():
        assert np.array_equal(O.ldm_unet_forward(sd, CFG, x, t, ctx).numpy(), g["eps_fp"])
        assert np.array_equal(O.ldm_unet_forward(sd, CFG, x, t, ctx, spec(g, False)).numpy(), g["eps_w4"])
        assert np.array_equal(torch.cat(O.ldm_tib_forward(sd, CFG, t, spec(g, False)), 1).numpy(), g["tib_w4"])
        assert np.array_equal(O.ldm_unet_forward(sd, CFG, x, t, ctx, spec(g, True)).numpy(), g["eps_w4a8"])


def test_ldm_schedule_and_cfg_ddim(f11):
    g, sd = f11
    ac = O.ldm_alphas_cumprod()
    assert np.array_equal(ac.numpy(), g["alphas_cumprod"])
    for S in (4, 20, 50):
        ts, a, ap = O.ldm_ddim_schedule(ac, S)
        assert np.array_equal(ts, g[f"ddim_ts_{S}"])
        assert np.array_equal(a.numpy().astype(np.float64), g[f"ddim_alphas_{S}"])
        assert np.array_equal(ap.numpy().astype(np.float64), g[f"ddim_alphas_prev_{S}"].astype(np.float32).astype(np.float64))
    x_T, uc, ctx = T(g["traj_xT"]), T(g["traj_uc"]), T(g["ctx"])
    qs = spec(g, True)
    with torch.no_grad():
        img, pred = O.ldm_ddim_sample(x_T, lambda x, t, c, k: O.ldm_unet_forward(sd, CFG, x, t, c, qs), ac, 4, ctx, uc, 7.5)
        assert np.array_equal(img.numpy(), g["traj_w4a8_final"])
        assert np.array_equal(torch.stack(pred).numpy(), g["traj_w4a8_predx0"])
        img, _ = O.ldm_ddim_sample(x_T, lambda x, t, c, k: O.ldm_unet_forward(sd, CFG, x, t, c), ac, 4, ctx, uc, 7.5)
        assert np.array_equal(img.numpy(), g["traj_fp_final"])


def test_ldm_attention_block_unet_fp_w4_w4a8(golden):
    """The oracle's plain AttentionBlock (unconditional LDM configs: CelebA-HQ, LSUN) pinned to the reference: fixture F13 holds the
    reference UNetModel's eps (FP) and the reference QuantModel's w4 / w4a8 eps on the same inputs -- bit for bit."""
    g = golden("f13_ldm_attnblock_tiny")
    sd = {k[3:]: T(g[k]) for k in g.files if k.startswith("sd/")}
    cfg = dict(model_channels=32, num_heads=-1, num_head_channels=16)
    x, t = T(g["x"]), T(g["t"])
    with torch.no_grad():
        assert np.array_equal(O.ldm_unet_forward(sd, cfg, x, t, None).numpy(), g["eps_fp"])
        assert np.array_equal(O.ldm_unet_forward(sd, cfg, x, t, None, spec(g, False)).numpy(), g["eps_w4"])
        assert np.array_equal(O.ldm_unet_forward(sd, cfg, x, t, None, spec(g, True)).numpy(), g["eps_w4a8"])

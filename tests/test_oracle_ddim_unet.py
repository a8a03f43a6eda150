"""Pins the oracle's DDIM-UNet / TIB / sampler restatement to the reference (fixture F5-F7)."""
import numpy as np
import pytest
import torch

import tfmq_oracle as O

CFG = dict(ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[8], resolution=16)


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def f7(golden):
    g = golden("f7_ddim_tiny")
    sd = {k[3:]: T(g[k]) for k in g.files if k.startswith("sd/")}
    return g, sd


def spec_from_fixture(g, with_act):
    wq, aq = {}, {}
    for k in g.files:
        if k.startswith("wq/") and k.endswith("/delta"):
            n = k[3:-6]
            wq[n] = {"delta": T(g[k]), "zp": T(g[f"wq/{n}/zp"]), "alpha": None}
        if with_act and k.startswith("aq/") and k.endswith("/delta"):
            n = k[3:-6]
            aq[n] = (T(g[k]), T(g[f"aq/{n}/zp"]))
    return O.QuantSpec(wq=wq, aq=aq)


def test_layer_name_order_matches_reference(f7):
    g, sd = f7
    names = O.ddim_quant_layer_names(CFG)
    assert len(names) == 45 and names[3] == "down.0.block.0.conv1" and names[-1] == "conv_out"
    wq_names = {k[3:-6] for k in g.files if k.startswith("wq/") and k.endswith("/delta")}
    assert wq_names == set(names) - {names[0], names[2], names[-1]}
    aq_names = {k[3:-6] for k in g.files if k.startswith("aq/") and k.endswith("/delta")}
    assert aq_names == set(O.ddim_act_layer_names(CFG))


def test_fp_forward_and_taps(f7):
    g, sd = f7
    x, t = T(g["x"]), T(g["t"])
    taps = {}
    with torch.no_grad():
        eps = O.ddim_unet_forward(sd, CFG, x, t, taps=taps)
    assert np.array_equal(eps.numpy(), g["eps_fp"])
    for k in g.files:
        if k.startswith("tap_fp/"):
            assert np.array_equal(taps[k[7:]][1].numpy(), g[k]), k
    tib = torch.cat(O.ddim_tib_forward(sd, CFG, t), dim=1)
    assert np.array_equal(tib.numpy(), g["tib_fp"])


def test_weight_init_and_w4_forward(f7):
    g, sd = f7
    x, t = T(g["x"]), T(g["t"])
    qs = O.ddim_init_quant_spec(sd, CFG, "mse")
    for n, q in qs.wq.items():
        assert np.array_equal(q["delta"].numpy(), g[f"wq/{n}/delta"]), n
        assert np.array_equal(q["zp"].numpy(), g[f"wq/{n}/zp"]), n
    taps = {}
    with torch.no_grad():
        eps = O.ddim_unet_forward(sd, CFG, x, t, qs, taps=taps)
    assert np.array_equal(eps.numpy(), g["eps_w4"])
    for k in g.files:
        if k.startswith("tap_w4/"):
            assert np.array_equal(taps[k[7:]][1].numpy(), g[k]), k
    assert np.array_equal(torch.cat(O.ddim_tib_forward(sd, CFG, t, qs), dim=1).numpy(), g["tib_w4"])


def test_w4a8_forward(f7):
    g, sd = f7
    x, t = T(g["x"]), T(g["t"])
    qs = spec_from_fixture(g, with_act=True)
    with torch.no_grad():
        eps = O.ddim_unet_forward(sd, CFG, x, t, qs)
    assert np.array_equal(eps.numpy(), g["eps_w4a8"])
    assert np.array_equal(torch.cat(O.ddim_tib_forward(sd, CFG, t, qs), dim=1).numpy(), g["tib_w4a8"])


def test_trajectories(f7):
    g, sd = f7
    betas = O.linear_betas()
    seq = [int(s) for s in g["seq"]]
    assert seq == O.ddim_seq("quad", 10)
    x0 = T(g["traj_x0"])
    with torch.no_grad():
        xs, _, _, _ = O.generalized_steps(x0, seq, lambda x, t, c: O.ddim_unet_forward(sd, CFG, x, t), betas)
        assert np.array_equal(torch.stack(xs).numpy(), g["traj_fp"])
        qs = spec_from_fixture(g, with_act=True)
        xs, _, _, _ = O.generalized_steps(x0, seq, lambda x, t, c: O.ddim_unet_forward(sd, CFG, x, t, qs), betas)
        assert np.array_equal(torch.stack(xs).numpy(), g["traj_w4a8"])
        _, _, xt, tt = O.generalized_steps(x0, seq, lambda x, t, c: O.ddim_unet_forward(sd, CFG, x, t), betas, until=4)
        assert np.array_equal(xt.numpy(), g["until4_xt"]) and np.array_equal(tt.numpy(), g["until4_t"])

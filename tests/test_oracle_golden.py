"""Pins oracle/tfmq_oracle.py to the golden vectors produced by the reference
(tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import torch

import tfmq_oracle as O


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_f1_per_tensor_scalers_bit_exact(golden):
    g = golden("f1_quantizer")
    acts = T(g["acts"])
    for nm, fn in (("minmax", O.minmax), ("mse", O.mse)):
        d, z = fn(acts, 256)
        assert float(d) == float(g[f"acts_{nm}_delta"]), nm
        assert float(z) == float(g[f"acts_{nm}_zp"]), nm
        idx = O.quant_index(acts, d, z, 256)
        assert np.array_equal(idx.numpy().astype(np.uint8), g[f"acts_{nm}_idx"])
        assert np.array_equal(O.fake_quant(acts, d, z, 256).numpy(), g[f"acts_{nm}_dq"])
    pos = T(g["pos"])
    for nm, fn in (("minmax", O.minmax), ("mse", O.mse)):
        d, z = fn(pos, 256, always_zero=True)
        assert float(d) == float(g[f"pos_{nm}_delta"]) and float(z) == float(g[f"pos_{nm}_zp"]) == 0.0


def test_f1_mse_candidates_and_losses(golden):
    g = golden("f1_quantizer")
    acts = T(g["acts"])
    d, z = O.mse_candidates(acts.min().item(), acts.max().item(), 256)
    assert np.array_equal(d.numpy(), g["acts_mse_cand_delta"].astype(np.float32))
    assert np.array_equal(z.numpy(), g["acts_mse_cand_zp"].astype(np.float32))
    losses = O.mse_losses(acts, d, z, 256)
    assert np.array_equal(losses.numpy(), g["acts_mse_cand_loss"].astype(np.float32))


def test_f1_channelwise_bit_exact(golden):
    g = golden("f1_quantizer")
    for tn in ("wts", "lin"):
        w = T(g[tn])
        for nm in ("minmax", "mse"):
            d, z = O.init_channelwise(w, 16, nm)
            assert np.array_equal(d.numpy(), g[f"{tn}_{nm}_delta"]), (tn, nm)
            assert np.array_equal(z.numpy(), g[f"{tn}_{nm}_zp"]), (tn, nm)
            assert np.array_equal(O.quant_index(w, d, z, 16).numpy().astype(np.uint8), g[f"{tn}_{nm}_idx"])
            assert np.array_equal(O.fake_quant(w, d, z, 16).numpy(), g[f"{tn}_{nm}_dq"])
    # the same-sign channel really has a zero-point outside [0, 15] (SURVEY §7-3 caveat)
    assert g["wts_mse_zp"].reshape(-1)[3] < 0


def test_f2_momentum_sequence(golden):
    g = golden("f2_momentum")
    xs = T(g["x"])
    d0, z0 = O.mse(xs[0], 256)
    assert float(d0) == float(g["init_delta"]) and float(z0) == float(g["init_zp"])
    assert np.array_equal(O.fake_quant(xs[0], d0, z0, 256).numpy(), g["y0"])
    st = O.MomentumState(xs[0])
    for i in range(1, xs.shape[0]):
        d, z = st.update(xs[i], 256)
        assert float(d) == float(g["delta"][i - 1]) and float(z) == float(g["zp"][i - 1]), i
        assert float(st.x_min) == float(g["x_min"][i - 1]) and float(st.x_max) == float(g["x_max"][i - 1])
    assert np.array_equal(O.fake_quant(xs[-1], d, z, 256).numpy(), g["y_last"])


def test_f3_quantlayer(golden):
    import torch.nn.functional as F
    g = golden("f3_quantlayer")
    for tag in ("lin2d", "lin3d", "conv3", "conv1"):
        x, w = T(g[f"{tag}_x"]), T(g[f"{tag}_w"])
        b = T(g[f"{tag}_b"]) if f"{tag}_b" in g.files else None
        wd, wz = O.init_channelwise(w, 16, "mse")
        assert np.array_equal(wd.numpy(), g[f"{tag}_wdelta"]) and np.array_equal(wz.numpy(), g[f"{tag}_wzp"])
        ad, az = O.mse(x, 256)
        assert float(ad) == float(g[f"{tag}_adelta"]) and float(az) == float(g[f"{tag}_azp"])
        xq, wq = O.fake_quant(x, ad, az, 256), O.fake_quant(w, wd, wz, 16)
        if tag.startswith("lin"):
            y = F.linear(xq, wq, b)
        else:
            y = F.conv2d(xq, wq, b, padding=1 if tag == "conv3" else 0)
        assert np.array_equal(y.numpy(), g[f"{tag}_y"]), tag


def test_f4_adaround(golden):
    import torch.nn.functional as F
    g = golden("f4_adaround")
    w, b, x, y_fp = T(g["w"]), T(g["b"]), T(g["x"]), T(g["y_fp"])
    wd, wz = O.init_channelwise(w, 16, "mse")
    assert np.array_equal(wd.numpy(), g["wdelta"]) and np.array_equal(wz.numpy(), g["wzp"])
    alpha = O.adaround_init_alpha(w, wd)
    assert np.array_equal(alpha.numpy(), g["alpha0"])
    assert np.array_equal(O.adaround_soft_tgt(alpha).numpy(), g["soft0"])
    assert np.array_equal(O.adaround_forward(w, alpha, wd, wz, 16, soft=False).numpy(), g["w_hard0"])
    assert np.array_equal(O.adaround_forward(w, alpha, wd, wz, 16, soft=True).numpy(), g["w_soft0"])
    # 20 Adam iterations of layer reconstruction (reconstruction.py:63-78 without the randperm:
    # the fixture uses the full batch every iteration)
    alpha = alpha.clone().requires_grad_(True)
    opt = torch.optim.Adam([alpha])
    iters = 20
    gi, ai = list(g["grad_iters"]), list(g["alpha_iters"])
    for it in range(iters):
        opt.zero_grad()
        yq = F.conv2d(x, O.adaround_forward(w, alpha, wd, wz, 16, soft=True), b, padding=1)
        tot, rec, rl, bb = O.recon_loss(yq, y_fp, [alpha], it + 1, iters)
        tot.backward()
        assert abs(float(tot) - float(g["loss"][it])) <= 1e-6 * abs(float(g["loss"][it])), it
        if it in gi:
            np.testing.assert_allclose(alpha.grad.numpy(), g["grads"][gi.index(it)], rtol=1e-5, atol=1e-9)
        opt.step()
        if it in ai:
            np.testing.assert_allclose(alpha.detach().numpy(), g["alphas"][ai.index(it)], rtol=1e-5, atol=1e-7)
    assert np.array_equal((alpha.detach() >= 0).numpy().astype(np.uint8), g["mask_final"])
    assert np.array_equal(O.adaround_forward(w, alpha.detach(), wd, wz, 16, soft=False).numpy(), g["w_hard_final"])


def test_f9_schedules(golden):
    g = golden("f9_schedules")
    for Tn in (10, 20, 50, 100):
        assert O.ddim_seq("quad", Tn) == list(g[f"quad_{Tn}"])
        assert O.ddim_seq("uniform", Tn) == list(g[f"uniform_{Tn}"])
    betas = O.linear_betas()
    assert np.array_equal(betas.numpy(), g["betas"])
    ab = O.compute_alpha(betas, torch.arange(-1, 1000)).reshape(-1)
    assert np.array_equal(ab.numpy(), g["alpha_bar"])
    t = T(g["temb_t"])
    assert np.array_equal(O.timestep_embedding_ddim(t, 128).numpy(), g["temb_128"])
    assert np.array_equal(O.timestep_embedding_ddim(t, 32).numpy(), g["temb_32"])


def test_f10_shards(golden):
    g = golden("f10_shards")
    for I, W in ((256, 8), (512, 8), (256, 4), (16, 2)):
        for r in range(W):
            assert O.shard_indices(I * 3, I, W, r) == list(g[f"I{I}_W{W}_r{r}"])


def test_histogram_scalers_kl_and_hist(golden):
    """Scaler.KL / Scaler.HIST restated in the oracle against the reference's own functions on five input
    distributions, 256 and 16 levels, with and without always_zero (fixture F18): delta and zero point bit for bit."""
    import tfmq_oracle as O
    g = golden("f18_hist_scalers")
    n = 0
    for k in g.files:
        if not (k.startswith("kl/") or k.startswith("hist/")):
            continue
        fn, name, level, az = k.split("/")
        x = torch.from_numpy(g[f"x/{name}"])
        d, z = (O.kl_scaler if fn == "kl" else O.hist_scaler)(x, int(level), bool(int(az)))
        assert float(d) == float(np.float32(g[k][0])) and float(z) == g[k][1], (k, float(d), float(z), g[k])
        n += 1
    assert n == 28

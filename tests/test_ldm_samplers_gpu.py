"""Drop-in latent samplers (DDIMSampler / PLMSSampler over LatentDiffusion.apply_model -> QuantModel -> HIP engine) against
trajectories produced by the reference's own samplers on the tiny SD-style UNet (fixture F11), incl. classifier-free
guidance, the `untill_fake_t` early stop, and the DiffusionWrapper's Finite-Set activation-group selection."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

from test_quant_mirror_ldm import T, tiny_qnn  # noqa: E402


def rel_l2(a, b):
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def ldm(golden):
    from tfmq_dm_amd.ldm.ddpm import LatentDiffusion
    g = golden("f11_ldm_tiny")
    q = tiny_qnn(g, device=DEV)
    q.set_quant_state(False, False)
    m = LatentDiffusion(q).to(DEV)
    assert np.array_equal(m.alphas_cumprod.cpu().numpy(), g["alphas_cumprod"])      # register_schedule, bit-exact
    return g, q, m


def test_ddim_dropin_fp(ldm):
    from tfmq_dm_amd.ldm.ddim import DDIMSampler
    g, q, m = ldm
    ctx, uc, x_T = T(g["ctx"]).to(DEV), T(g["traj_uc"]).to(DEV), T(g["traj_xT"]).to(DEV)
    s = DDIMSampler(m)
    inter = []
    out, im = s.sample(S=4, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False, unconditional_guidance_scale=7.5,
                       unconditional_conditioning=uc, eta=0.0, x_T=x_T, img_callback=lambda p, i: inter.append(p.clone()),
                       log_every_t=1)
    assert np.array_equal(s.ddim_timesteps, g["ddim_ts_4"])
    assert rel_l2(out.cpu(), T(g["traj_fp_final"])) <= 2e-2          # f16 MFMA layers, CFG 7.5 amplifies eps error
    assert len(inter) == 4 and len(im["x_inter"]) == 5
    out3, _ = s.sample(S=4, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False, unconditional_guidance_scale=7.5,
                       unconditional_conditioning=uc, eta=0.0, x_T=x_T, untill_fake_t=3)
    assert rel_l2(out3.cpu(), T(g["traj_fp_until3"])) <= 2e-2


def test_plms_dropin_fp(ldm):
    from tfmq_dm_amd.ldm.ddim import PLMSSampler
    g, q, m = ldm
    ctx, uc, x_T = T(g["ctx"]).to(DEV), T(g["traj_uc"]).to(DEV), T(g["traj_xT"]).to(DEV)
    s = PLMSSampler(m)
    inter = []
    out, _ = s.sample(S=6, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False, unconditional_guidance_scale=7.5,
                      unconditional_conditioning=uc, eta=0.0, x_T=x_T, img_callback=lambda p, i: inter.append(p.clone()),
                      log_every_t=1)
    ref_px0 = T(g["plms_fp_predx0"])
    assert len(inter) == ref_px0.shape[0] == 7                       # 1000 // 6 = 166 -> 7 executed steps
    assert rel_l2(inter[0].cpu(), ref_px0[0]) <= 1e-2                # first step: pseudo improved Euler (2 model calls)
    assert rel_l2(out.cpu(), T(g["plms_fp_final"])) <= 2e-2
    out4, _ = s.sample(S=6, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False, unconditional_guidance_scale=7.5,
                       unconditional_conditioning=uc, eta=0.0, x_T=x_T, untill_fake_t=4)
    assert rel_l2(out4.cpu(), T(g["plms_fp_until4"])) <= 2e-2
    with pytest.raises(ValueError):
        s.sample(S=6, conditioning=ctx, batch_size=2, shape=[4, 8, 8], eta=0.5, x_T=x_T)


def test_plms_combine_kernels_bit_exact():
    import tfmq_dm_amd.ops as ops
    gen = torch.Generator().manual_seed(1)
    e = [torch.randn(3, 4, 8, 8, generator=gen) for _ in range(4)]
    d = [t.to(DEV) for t in e]
    assert torch.equal(ops.plms_combine(1, d[0], d[1]).cpu(), (e[0] + e[1]) / 2)
    assert torch.equal(ops.plms_combine(2, d[0], d[1]).cpu(), (3 * e[0] - e[1]) / 2)
    assert torch.equal(ops.plms_combine(3, d[0], d[1], d[2]).cpu(), (23 * e[0] - 16 * e[1] + 5 * e[2]) / 12)
    assert torch.equal(ops.plms_combine(4, d[0], d[1], d[2], d[3]).cpu(), (55 * e[0] - 59 * e[1] + 37 * e[2] - 9 * e[3]) / 24)
    assert torch.equal(ops.cfg_combine(d[0], d[1], 7.5).cpu(), e[0] + 7.5 * (e[1] - e[0]))


def test_wrapper_selects_activation_group(golden):
    """DiffusionWrapper.{tot,t_max,ckpt} (txt2img.py): k = t_max - (t-1)//tot picks the act_k group; the device-table
    path gives the same eps as the reference's load_state_dict path."""
    from tfmq_dm_amd.ldm.ddpm import LatentDiffusion
    from quant.calibration import load_cali_model
    import tempfile
    g = golden("f12_ldm_cali_tiny")
    ck = {"weight": {str(k): T(g["ck/weight/" + str(k)]) for k in g["weight_keys"]}}
    akeys = [str(k) for k in g["act_keys"]]
    for gi in range(3):
        d, z = T(g[f"ck/act_{gi}/delta"]), T(g[f"ck/act_{gi}/zp"])
        dk = [k for k in akeys if k.endswith("delta")]
        zk = [k for k in akeys if k.endswith("zero_point")]
        ck[f"act_{gi}"] = {**{k: d[i] for i, k in enumerate(dk)}, **{k: z[i] for i, k in enumerate(zk)}}
    path = os.path.join(tempfile.mkdtemp(), "c.pth")
    torch.save(ck, path)
    q = tiny_qnn(g, cali=False, device=DEV)
    init = (torch.randn(1, 4, 8, 8), torch.randint(0, 1000, (1,)).float(), torch.randn(1, 5, 64))
    load_cali_model(q, init, use_aq=True, path=path)
    m = LatentDiffusion(q).to(DEV)
    xe, ce = T(g["reload_x"]).to(DEV), T(g["reload_c"]).to(DEV)
    te = torch.full((2,), 501, device=DEV, dtype=torch.long)
    # three groups over 1000 steps: tot = 334, t_max = 2 -> t = 501 selects act_1
    m.model.tot, m.model.t_max, m.model.ckpt = 334, 2, ck
    eps = m.apply_model(xe, te, ce).cpu()
    assert int(q._act_step.item()) == 1
    ref = T(g["reload_eps_act1"])
    assert rel_l2(eps, ref) <= 5e-2


def test_text_guided_calibration_set(ldm):
    """generate_cali_text_guided_data (quant/data_generate.py:13-49): per c-th step and prompt, CFG-7.5 sampling from
    fresh noise until that step; both the conditional and the unconditional context enter the set; timesteps follow
    real_time = (T - t) * 1000 // T + 1."""
    from quant.data_generate import generate_cali_text_guided_data
    from tfmq_dm_amd.ldm.ddim import DDIMSampler, PLMSSampler
    g, q, m = ldm
    gen = torch.Generator().manual_seed(7)
    table = {"": torch.randn(5, 64, generator=gen), "a": torch.randn(5, 64, generator=gen), "b": torch.randn(5, 64, generator=gen)}
    m.get_learned_conditioning = lambda prompts: torch.stack([table[p] for p in prompts]).to(DEV)
    T_, c_, bs = 4, 2, 2
    for cls in (DDIMSampler, PLMSSampler):
        torch.manual_seed(11)
        xs, ts, cs = generate_cali_text_guided_data(m, cls(m), T_, c_, bs, ("a", "b"), [4, 8, 8])
        n_t = T_ // c_
        assert xs.shape == (n_t * 2 * 2 * bs, 4, 8, 8) and ts.shape == (n_t * 2 * 2 * bs,) and cs.shape == (n_t * 2 * 2 * bs, 5, 64)
        assert ts.tolist() == [(T_ - t) * 1000 // T_ + 1 for t in (2, 4) for _ in range(2 * 2 * bs)]
        assert torch.equal(cs[:bs].cpu(), table["a"].expand(bs, 5, 64)) and torch.equal(cs[bs:2 * bs].cpu(), table[""].expand(bs, 5, 64))
        assert torch.equal(xs[:bs], xs[bs:2 * bs])                     # the same latent paired with c and with uc
        # first entry == sampling until step 2 from the same noise
        torch.manual_seed(11)
        s = cls(m)
        ref, _ = s.sample(S=T_, conditioning=m.get_learned_conditioning(bs * ["a"]), batch_size=bs, shape=[4, 8, 8], verbose=False,
                          unconditional_guidance_scale=7.5, unconditional_conditioning=m.get_learned_conditioning(bs * [""]),
                          untill_fake_t=2)
        # (the generator batches the prompts of a step into one captured-graph sampling: same noise order, same recurrence)
        assert float((ref - xs[:bs]).abs().max()) <= 1e-5 * float(ref.abs().max())
        assert torch.isfinite(xs).all()


def test_dpm_solver_dropin_fp(ldm):
    """DPMSolverSampler (DPM-Solver++ multistep order 2, data prediction, CFG 7.5) vs the reference's sampler on the FP
    model, full trajectory and the `untill_fake_t` early stop; returned time labels are the reference's continuous t."""
    from tfmq_dm_amd.ldm.dpm_solver import DPMSolverSampler, NoiseScheduleVP
    g, q, m = ldm
    ctx, uc, x_T = T(g["ctx"]).to(DEV), T(g["traj_uc"]).to(DEV), T(g["traj_xT"]).to(DEV)
    s = DPMSolverSampler(m)
    out, vt = s.sample(S=6, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False, unconditional_guidance_scale=7.5,
                       unconditional_conditioning=uc, eta=0.0, x_T=x_T)
    assert rel_l2(out.cpu(), T(g["dpm_fp_final"])) <= 2e-2
    np.testing.assert_allclose(vt.cpu().numpy(), g["dpm_fp_vect"], rtol=1e-6)
    out3, vt3 = s.sample(S=6, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False, unconditional_guidance_scale=7.5,
                         unconditional_conditioning=uc, eta=0.0, x_T=x_T, untill_fake_t=3)
    assert rel_l2(out3.cpu(), T(g["dpm_fp_until3"])) <= 2e-2
    np.testing.assert_allclose(vt3.cpu().numpy(), g["dpm_fp_until3_vect"], rtol=1e-6)
    # schedule: alpha^2 + sigma^2 = 1 and lambda = log(alpha / sigma), at the knots alpha = sqrt(alphas_cumprod)
    ns = NoiseScheduleVP("discrete", alphas_cumprod=m.alphas_cumprod)
    t = torch.tensor([0.001, 0.5, 1.0])
    a, sg = ns.marginal_alpha(t), ns.marginal_std(t)
    assert torch.allclose(a * a + sg * sg, torch.ones(3), atol=1e-6)
    ac = m.alphas_cumprod.cpu()
    assert torch.allclose(a, torch.stack([ac[0], ac[499], ac[999]]).sqrt(), rtol=1e-5)


def test_dpm_kernels_bit_exact():
    import tfmq_dm_amd.ops as ops
    gen = torch.Generator().manual_seed(3)
    x, m0, m1 = (torch.randn(2, 4, 8, 8, generator=gen) for _ in range(3))
    sg, al, cx, cm, ir = 0.73, 0.41, 0.9, -0.37, 1.7
    f = lambda v: torch.tensor(v, dtype=torch.float32)
    assert torch.equal(ops.dpm_x0(x.to(DEV), m0.to(DEV), sg, al).cpu(), (x - f(sg) * m0) / f(al))
    assert torch.equal(ops.dpm_update(1, x.to(DEV), m0.to(DEV), None, cx, cm).cpu(), f(cx) * x - f(cm) * m0)
    cd = float(0.5 * f(cm))
    want = f(cx) * x - f(cm) * m0 - f(cd) * (f(ir) * (m0 - m1))
    assert torch.equal(ops.dpm_update(2, x.to(DEV), m0.to(DEV), m1.to(DEV), cx, cm, cd, ir).cpu(), want)


def test_graph_fast_path_of_the_dropin_samplers(ldm):
    """`sample(..., _graph=True)` (what the calibration-set generators pass): captured step graphs instead of the host
    loop, same recurrence and kernels -- DDIM and PLMS, full length and `untill_fake_t`, CFG 7.5."""
    from tfmq_dm_amd.ldm.ddim import DDIMSampler, PLMSSampler
    g, q, m = ldm
    ctx, uc, x_T = T(g["ctx"]).to(DEV), T(g["traj_uc"]).to(DEV), T(g["traj_xT"]).to(DEV)
    for cls, S in ((DDIMSampler, 4), (PLMSSampler, 6)):
        for until in (None, 3):
            kw = dict(S=S, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False, unconditional_guidance_scale=7.5,
                      unconditional_conditioning=uc, eta=0.0, x_T=x_T, untill_fake_t=until)
            ref, _ = cls(m).sample(**kw)
            fast, inter = cls(m).sample(_graph=True, **kw)
            assert hasattr(q, "_graph_samplers") and len(q._graph_samplers) == 1          # the fast path was taken
            assert float((fast - ref).abs().max()) <= 1e-5 * float(ref.abs().max()), (cls.__name__, until)
            assert len(inter["x_inter"]) == 2
    # eta > 0 (ddim.py:173-212: sigma_t * noise): the step graphs read a noise buffer refilled before every replay in the host loop's order
    # of draws -- same generator state, same trajectory
    kw = dict(S=4, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False, unconditional_guidance_scale=7.5,
              unconditional_conditioning=uc, eta=0.7, x_T=x_T)
    torch.manual_seed(5)
    ref, _ = DDIMSampler(m).sample(**kw)
    torch.manual_seed(5)
    fast, _ = DDIMSampler(m).sample(_graph=True, **kw)
    det, _ = DDIMSampler(m).sample(_graph=True, **{**kw, "eta": 0.0})
    assert len(q._graph_samplers) == 1 and next(iter(q._graph_samplers.values())).eta == 0.0
    assert float((fast - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert float((fast - det).abs().max()) > 1e-2 * float(ref.abs().max())              # (the noise term is there)
    # a call with a callback keeps the host loop
    seen = []
    DDIMSampler(m).sample(S=4, conditioning=ctx, batch_size=2, shape=[4, 8, 8], verbose=False, unconditional_guidance_scale=7.5,
                          unconditional_conditioning=uc, x_T=x_T, img_callback=lambda p, i: seen.append(i), _graph=True)
    assert seen == [0, 1, 2, 3]


def test_class_conditional_calibration_set(ldm):
    """generate_cali_data_ldm_imagenet (quant/data_generate.py:115-154): 32 class labels x every c-th step, CFG with the
    learned 'unconditional' class 1000; classes share sampler calls, order and pairing as in the reference's loops."""
    from quant.data_generate import generate_cali_data_ldm_imagenet
    from tfmq_dm_amd.ldm.ddim import DDIMSampler
    g, q, m = ldm
    gen = torch.Generator().manual_seed(9)
    emb = torch.randn(1001, 5, 64, generator=gen)
    m.cond_stage_key = "class_label"
    m.get_learned_conditioning = lambda d: emb[d["class_label"].cpu()].to(DEV)
    T_, c_, bs = 4, 2, 2
    torch.manual_seed(21)
    xs, ts, cs = generate_cali_data_ldm_imagenet(m, T_, c_, bs, [4, 8, 8], eta=0.0, scale=3.0, max_batch=8)
    n_cls, n_t = 32, T_ // c_
    assert xs.shape == (n_t * n_cls * 2 * bs, 4, 8, 8) and cs.shape == (n_t * n_cls * 2 * bs, 5, 64)
    assert ts.tolist() == [(T_ - t) * 1000 // T_ + 1 for t in (2, 4) for _ in range(n_cls * 2 * bs)]
    classes = [i for i in range(0, 1000, 1000 // 31)]
    assert torch.equal(cs[:bs].cpu(), emb[classes[0]].expand(bs, 5, 64)) and torch.equal(cs[bs:2 * bs].cpu(), emb[1000].expand(bs, 5, 64))
    assert torch.equal(cs[2 * bs:3 * bs].cpu(), emb[classes[1]].expand(bs, 5, 64))
    assert torch.equal(xs[:bs], xs[bs:2 * bs]) and torch.isfinite(xs).all()
    # first entry == one sampling of class 0 until step 2 from the same noise
    torch.manual_seed(21)
    ref, _ = DDIMSampler(m).sample(S=T_, conditioning=emb[[classes[0]] * bs].to(DEV), batch_size=bs, shape=[4, 8, 8], verbose=False,
                                   unconditional_guidance_scale=3.0, unconditional_conditioning=emb[[1000] * bs].to(DEV),
                                   untill_fake_t=2)
    assert float((ref - xs[:bs]).abs().max()) <= 1e-5 * float(ref.abs().max())

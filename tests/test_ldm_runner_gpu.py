"""The LDM-side driver flows IN SEQUENCE (ldm/runner.py: LatentRunner) against fixture F19, which the reference produced by running the
same sequence on the same tiny models: reference checkpoint -> QuantModel(cali=False) -> load_cali_model -> per-call Finite-Set group
through DiffusionWrapper.forward -> DDIMSampler / PLMSSampler (sample_diffusion_ldm.py:445-565, latent_imagenet_diffusion.py:190-341,
txt2img.py:381-598).  Checked: the (timestep, activation group) of every UNet call EXACTLY, the final latents at the trajectory
bar of the sampler tests, the reference's timing region / throughput entry, and the calibrate -> checkpoint -> reload -> sample round
trip of the runner itself (incl. the graph-replay path with a per-step table built from the checkpoint's groups)."""
import argparse
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CIN_KW = dict(image_size=8, in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=[1, 2],
              channel_mult=[1, 2], num_heads=1, use_spatial_transformer=True, transformer_depth=1, context_dim=64, legacy=False)
ATTN_UNET_KW = dict(image_size=8, in_channels=3, model_channels=32, out_channels=3, num_res_blocks=1, attention_resolutions=[1, 2],
                    channel_mult=[1, 2], num_head_channels=16)
UNET_KW = dict(image_size=8, in_channels=4, model_channels=32, out_channels=4, num_res_blocks=1, attention_resolutions=[1, 2],
               channel_mult=[1, 2], num_heads=2, use_spatial_transformer=True, transformer_depth=1, context_dim=64, legacy=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def rel_l2(a, b):
    return float((a - b).norm() / b.norm())


def sd_of(g):
    return {k[3:]: T(g[k]) for k in g.files if k.startswith("sd/")}


def ref_ckpt(g):
    """The reference's checkpoint of a fixture: {'weight': {...}, 'act_0': {...}, ...} (w / b copies of the model state are not
    stored in F15 / F16; load_cali_model loads with strict=False and the model already holds them)."""
    ck = {"weight": {str(k): T(g["ck/weight/" + str(k)]) for k in g["weight_keys"] if ("ck/weight/" + str(k)) in g.files}}
    akeys = [str(k) for k in g["act_keys"]]
    dk = [k for k in akeys if k.endswith("delta")]
    zk = [k for k in akeys if k.endswith("zero_point")]
    gi = 0
    while f"ck/act_{gi}/delta" in g.files:
        d, z = T(g[f"ck/act_{gi}/delta"]), T(g[f"ck/act_{gi}/zp"])
        ck[f"act_{gi}"] = {**{k: d[i] for i, k in enumerate(dk)}, **{k: z[i] for i, k in enumerate(zk)}}
        gi += 1
    return ck


FAMILIES = {
    # flow: (fixture with the checkpoint, fixture with the weights, UNet kwargs, schedule, conditioning key, guidance scale, tag)
    "text": ("f12_ldm_cali_tiny", "f12_ldm_cali_tiny", UNET_KW, dict(linear_start=0.00085, linear_end=0.012), "crossattn", 7.5),
    "class": ("f15_cin_tiny", "f15_cin_tiny", CIN_KW, dict(linear_start=0.0015, linear_end=0.0195), "crossattn", 3.0),
    "uncond": ("f16_attnblock_cali_tiny", "f13_ldm_attnblock_tiny", ATTN_UNET_KW, dict(linear_start=0.0015, linear_end=0.0195), None, 1.0),
}


def build(golden, flow):
    from tfmq_dm_amd.ldm.ddpm import LatentDiffusion
    from tfmq_dm_amd.ldm.unet import UNetModel
    fck, fsd, kw, sched, key, scale = FAMILIES[flow]
    m = UNetModel(**kw)
    m.load_state_dict(sd_of(golden(fsd)))
    return LatentDiffusion(m.to(DEV), conditioning_key=key, **sched).to(DEV), ref_ckpt(golden(fck)), scale


def opts(**kw):
    base = dict(ptq=True, cali=False, use_aq=True, wq=4, aq=8, softmax_a_bit=8, eta=0.0, plms=False, dpm=False, multi_gpu=False)
    base.update(kw)
    return argparse.Namespace(**base)


@pytest.mark.parametrize("flow,tag,steps,plms", [("text", "text/ddim", 6, False), ("text", "text/plms", 6, True),
                                                 ("class", "class/ddim", 5, False), ("uncond", "uncond/ddim", 8, False)])
def test_flow_load_checkpoint_and_sample_matches_reference_sequence(golden, flow, tag, steps, plms):
    from tfmq_dm_amd.ldm.runner import LatentRunner
    g = golden("f19_ldm_flows")
    model, ck, scale = build(golden, flow)
    path = os.path.join(tempfile.mkdtemp(), "ref.pth")
    torch.save(ck, path)
    top = tag.split("/")[0]
    x_T = T(g[f"{top}/x_T"]).to(DEV)
    cond = T(g[f"{top}/c"]).to(DEV) if flow != "uncond" else None
    uc = T(g[f"{top}/uc"]).to(DEV) if flow != "uncond" else None
    r = LatentRunner(model, opts(cali_ckpt=path, custom_steps=steps, plms=plms, scale=scale), flow, DEV)
    qnn = r.quantize(init_context=cond)
    w = model.model
    groups, tot, t_max = (int(v) for v in g[f"{tag}/groups"])
    assert (w.tot, w.t_max) == (tot, t_max) and w.ckpt is not None and w.diffusion_model is qnn      # txt2img.py:413-417
    calls = []
    w.register_forward_pre_hook(lambda mod, args: calls.append(float(args[1][0].item())))
    ks = []
    sel = qnn.select_act_group
    qnn.select_act_group = lambda k: (ks.append(int(k)), sel(k))[1]
    log = r.sample_batch(x_T.shape[0], cond, uc, x_T=x_T)
    ref_calls = g[f"{tag}/calls"]
    assert calls == [float(t) for t in ref_calls[:, 0]]             # the UNet was called at exactly the reference's timesteps ...
    assert ks == [int(k) for k in ref_calls[:, 1]]                  # ... under exactly the reference's activation groups
    rr = rel_l2(log["latents"].cpu(), T(g[f"{tag}/final"]))
    print(f"[{tag}] final latents rel-L2 vs the reference's flow: {rr:.3e}; throughput {log['throughput']:.1f} samples/s")
    assert torch.isfinite(log["latents"]).all() and rr <= 1e-1      # w4a8 bin flips compound over the steps (x guidance scale)
    assert abs(log["throughput"] - x_T.shape[0] / log["time"]) < 1e-9 and log["sample"] is log["latents"]    # no first stage attached
    if not plms:
        # graph replay with the SAME checkpoint: the table rows of the captured steps are the groups above, so it equals the host loop
        qnn.select_act_group = sel
        r2 = LatentRunner(model, opts(cali_ckpt=path, custom_steps=steps, scale=scale), flow, DEV)
        r2.make_sampler()
        kw = dict(S=steps, batch_size=x_T.shape[0], shape=r2.latent_shape(), verbose=False, eta=0.0, x_T=x_T, _graph=True)
        if flow != "uncond":
            kw.update(conditioning=cond, unconditional_guidance_scale=scale, unconditional_conditioning=uc)
        n_before = len(calls)
        fast, _ = r2.sampler.sample(**kw)
        assert len(calls) == n_before                                # no host-side UNet call: every step was a graph replay
        assert torch.equal(fast, log["latents"])
        again = r.sample_batch(x_T.shape[0], cond, uc, x_T=x_T)      # and the host loop re-installs the group table afterwards
        assert torch.equal(again["latents"], log["latents"])


class _FakeText:
    """Stand-in for the frozen text encoder (glue): a deterministic embedding per prompt."""

    def __init__(self, tokens, dim):
        self.tokens, self.dim = tokens, dim

    def __call__(self, prompts):
        out = []
        for p in prompts:
            gen = torch.Generator().manual_seed(sum(map(ord, p)) + 1)
            out.append(torch.randn(self.tokens, self.dim, generator=gen))
        return torch.stack(out).to(DEV)


@pytest.mark.parametrize("flow", ["uncond", "text"])
def test_flow_calibrate_checkpoint_reload_sample(golden, flow):
    """The other branch of the drivers: generate the calibration set with the flow's generator, cali_model, checkpoint; then a fresh
    runner loads it and samples.  Short recipe (4 steps -> 4 Finite-Set groups, 4 AdaRound iterations per unit)."""
    from tfmq_dm_amd.ldm.runner import LatentRunner
    model, _, scale = build(golden, flow)
    path = os.path.join(tempfile.mkdtemp(), "mine.pth")
    kw = dict(cali=True, cali_save_path=path, custom_steps=4, interval_length=2, cali_batch=16, cali_iters=4, scale=scale, cali_interval=16)
    if flow == "text":
        model.get_learned_conditioning = _FakeText(5, 64)
        kw.update(C=4, H=64, W=64, f=8)
    torch.manual_seed(3)
    np.random.seed(3)
    r = LatentRunner(model, opts(**kw), flow, DEV)
    prompts = ["a photo of a cat", "a red cube", "two dogs", "the sea", "a tree", "city at night", "a bowl", "clouds"] if flow == "text" else None
    assert r.quantize(prompts=prompts) == "calibrated"
    ck = torch.load(path, map_location="cpu")
    n_groups = len(ck) - 1
    assert "weight" in ck and n_groups == 4 and any(k.endswith("alpha") for k in ck["weight"])
    model2, _, _ = build(golden, flow)
    kw2 = dict(cali_ckpt=path, custom_steps=4, scale=scale)
    if flow == "text":
        kw2.update(C=4, H=64, W=64, f=8)
    r2 = LatentRunner(model2, opts(**kw2), flow, DEV)
    r2.quantize(init_context=None if flow == "uncond" else torch.randn(1, 5, 64))
    assert (model2.model.tot, model2.model.t_max) == (1000 // n_groups, n_groups - 1)
    enc = None
    if flow == "text":
        te = _FakeText(5, 64)
        enc = lambda i: (te(["a photo of a cat", "the sea"]), te(["", ""]))
    out = r2.run(n_samples=4, batch_size=2, encode=enc)
    assert out["samples"].shape[0] == 4 and torch.isfinite(out["samples"]).all()
    assert abs(out["throughput"] - 4 / out["time"]) < 1e-9 and len(out["batches"]) == 2

"""CPU-only: the (interval, AdaRound mini-batch) pair LatentRunner.quantize hands to cali_model per driver flow equals the reference
scripts' single-GPU calls -- sample_diffusion_ldm.py:534-546 (interval 256, batch 32), latent_imagenet_diffusion.py:275-287 (512, 8),
txt2img.py:473-486 (256, 8; 32 appears only in its mp.spawn kwargs).  ADVICE round 3: the text flow passed 32."""
import types

import pytest
import torch


@pytest.mark.parametrize("flow,want", [("uncond", (256, 32)), ("class", (512, 8)), ("text", (256, 8))])
def test_cali_model_recipe_per_flow(monkeypatch, flow, want):
    from tfmq_dm_amd.ldm.runner import LatentRunner
    import tfmq_dm_amd.quant.calibration as CAL
    import tfmq_dm_amd.quant.data_generate as DG
    import tfmq_dm_amd.quant.quant_model as QM
    assert LatentRunner.CALI_RECIPE[flow] == want
    seen = {}

    class FakeQnn:
        def __init__(self, **kw):
            pass

        def to(self, *_):
            return self

        def eval(self):
            return self

    def fake_cali_model(**kw):
        seen.update(kw)

    xs, ts, cs = torch.zeros(4, 4, 8, 8), torch.zeros(4, dtype=torch.long), torch.zeros(4, 77, 16)
    monkeypatch.setattr(CAL, "cali_model", fake_cali_model)
    monkeypatch.setattr(QM, "QuantModel", FakeQnn)
    monkeypatch.setattr(DG, "generate_cali_data_ldm", lambda **kw: (torch.zeros(2 * 256, 4, 8, 8), torch.zeros(2 * 256, dtype=torch.long)))
    monkeypatch.setattr(DG, "generate_cali_data_ldm_imagenet", lambda **kw: (xs, ts, cs))
    monkeypatch.setattr(DG, "generate_cali_text_guided_data", lambda *a, **kw: (xs, ts, cs))
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    unet = types.SimpleNamespace(in_channels=4, image_size=8)
    model = types.SimpleNamespace(model=types.SimpleNamespace(diffusion_model=unet))
    opt = types.SimpleNamespace(ptq=True, cali=True, wq=4, aq=8, use_aq=True, custom_steps=2, eta=0.0, interval_length=1, scale=3.0,
                                cali_save_path="/tmp/none.pth", multi_gpu=False, C=4, H=64, W=64, f=8)
    r = LatentRunner(model, opt, flow, device="cpu")
    r.sampler = object()
    assert r.quantize(prompts=["a"]) == "calibrated"
    assert (seen["interval"], seen["batch_size"]) == want
    assert seen["iters"] == 20000 and seen["w"] == 0.01 and seen["warmup"] == 0.2 and seen["multi_gpu"] is False

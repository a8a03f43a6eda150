"""FP sampling of the calibration set at SD size: host-loop drop-in sampler vs its captured-graph path (s per 50-step
CFG sampling of `B` images)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
import torch
from tfmq_dm_amd.ldm.unet import UNetModel, SD_V1_UNET
from tfmq_dm_amd.ldm.ddpm import LatentDiffusion
from tfmq_dm_amd.ldm.ddim import PLMSSampler
from quant.quant_layer import QMODE, Scaler
from quant.quant_model import QuantModel
DEV = "cuda:0"
torch.manual_seed(1234)
m = UNetModel(**SD_V1_UNET)
g = torch.Generator().manual_seed(7)
with torch.no_grad():
    for p in m.parameters():
        if p.numel() and float(p.abs().max()) == 0.0:
            p.copy_(torch.randn(p.shape, generator=g) * 0.02)
m = m.to(DEV)
wq = {"bits": 4, "channel_wise": True, "scaler": Scaler.MINMAX}
aq = {"bits": 8, "channel_wise": False, "scaler": Scaler.MINMAX, "leaf_param": True}
qnn = QuantModel(m, wq, aq, cali=True, aq_mode=[QMODE.NORMAL.value, QMODE.QDIFF.value]).eval()
qnn.set_quant_state(False, False)
ld = LatentDiffusion(qnn, conditioning_key="crossattn").to(DEV)
B = int(os.environ.get("B", "8"))
c = torch.randn(B, 77, 768, generator=g).to(DEV); uc = torch.randn(B, 77, 768, generator=g).to(DEV)
for fast in (False, True, True):
    torch.cuda.synchronize(); t0 = time.time()
    out, _ = PLMSSampler(ld).sample(S=50, conditioning=c, batch_size=B, shape=[4, 64, 64], verbose=False, unconditional_guidance_scale=7.5,
                                    unconditional_conditioning=uc, untill_fake_t=26, _graph=fast)
    torch.cuda.synchronize()
    print(f"batch {B}, 25 PLMS steps, graph={fast}: {time.time()-t0:.2f}s  finite={bool(torch.isfinite(out).all())}", flush=True)

# ---- generate_cali_text_guided_data: 16 prompts x batch 1, steps t = 25 and 50: one sampler call per prompt vs batched
from quant.data_generate import generate_cali_text_guided_data
table = {}
def glc(prompts):
    return torch.stack([table.setdefault(p, torch.randn(77, 768, generator=g)) for p in prompts]).to(DEV)
ld.get_learned_conditioning = glc
prompts = tuple(f"p{i}" for i in range(16))
for mb in (1, 64, 64):
    torch.manual_seed(3); torch.cuda.synchronize(); t0 = time.time()
    xs, ts, cs = generate_cali_text_guided_data(ld, PLMSSampler(ld), 50, 25, 1, prompts, [4, 64, 64], max_batch=mb)
    torch.cuda.synchronize()
    print(f"16 prompts, t in (25, 50), max_batch={mb}: {time.time()-t0:.2f}s  {tuple(xs.shape)}", flush=True)

"""The full w4a8 calibration recipes of the two LDM drivers BASELINE.json names beside Stable Diffusion, end to end on one MI355X, measured:

  FLOW=celeba  configs[2]: LDM-4 CelebA-HQ 256 (unconditional, 274 M): sample_diffusion_ldm.py -c 200 -e 0.0 --cali --use_aq --interval_length 10
               (:512-541: a-set 200 steps x 256 samples from the FP DDIM sampler, w-set every 10th step, cali_model at mini-batch 32)
  FLOW=cin256  configs[4]: LDM ImageNet-256 class-conditional (cin256-v2, 400 M): latent_imagenet_diffusion.py --ddim_steps 20 --scale 3.0 --cali
               (:253-282: 32 classes x 8 samples x (cond, uncond) = 512 per step x 20 steps, cali_model at mini-batch 8, interval 512)

through ldm/runner.py: LatentRunner.quantize (the drivers' flow): calibration-set generation, weight-scale search (MSE), TIAR + every block /
layer reconstruction unit at ITERS (default 20 000) Adam iterations, Finite-Set activation calibration, checkpoint.  Random-init weights (no
checkpoints offline); the class embedder of cin256 is a fixed random table (glue outside the package).  Writes a JSON report with the phase
split.  T / CALI_BATCH / ITERS / ONLY (comma-separated unit prefixes) cut the recipe for estimates."""
import argparse, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
import numpy as np, torch
import tfmq_dm_amd.ldm.unet as U
from tfmq_dm_amd.ddim.models import random_init
from tfmq_dm_amd.ldm.ddpm import LatentDiffusion
from tfmq_dm_amd.ldm.runner import LatentRunner
import tfmq_dm_amd.quant.calibration as CAL
import tfmq_dm_amd.quant.data_generate as DG

FLOW = os.environ.get("FLOW", "celeba")
ITERS = int(os.environ.get("ITERS", "20000"))
dev = torch.device("cuda", 0)
P = {"celeba": dict(unet="CELEBAHQ_LDM_VQ4_UNET", flow="uncond", key=None, T=200, cali_batch=256, interval_length=10, scale=1.0,
                    name="LDM-4 CelebA-HQ 256 unconditional UNet"),
     "cin256": dict(unet="CIN256_V2_UNET", flow="class", key="crossattn", T=20, cali_batch=8, interval_length=1, scale=3.0,
                    name="LDM ImageNet-256 class-conditional UNet (cin256-v2)")}[FLOW]
T = int(os.environ.get("T", P["T"]))
NB = int(os.environ.get("CALI_BATCH", P["cali_batch"]))
out_path = os.environ.get("OUT", os.path.join(ROOT, "gpurun_out", "r05", f"{FLOW}_calibration_full.json"))
torch.manual_seed(40)
unet = random_init(U.UNetModel(**getattr(U, P["unet"])), 40)
n_params = sum(p.numel() for p in unet.parameters()) / 1e6
model = LatentDiffusion(unet.to(dev), conditioning_key=P["key"], linear_start=0.0015, linear_end=0.0195).to(dev).eval()
if FLOW == "cin256":
    table = torch.randn(1001, 1, 512, generator=torch.Generator().manual_seed(9)).to(dev)       # ClassEmbedder stand-in (1000 = the null class)
    model.cond_stage_key = "class_label"
    model.get_learned_conditioning = lambda batch: table[batch["class_label"].long().to(dev)]
ck = os.path.join(tempfile.mkdtemp(), f"{FLOW}_w4a8.pth")
opt = argparse.Namespace(ptq=True, cali=True, use_aq=True, wq=4, aq=8, softmax_a_bit=8, eta=0.0, plms=False, dpm=False, multi_gpu=False,
                         custom_steps=T, ddim_steps=T, interval_length=P["interval_length"], cali_batch=NB, cali_iters=ITERS, scale=P["scale"],
                         cali_save_path=ck, cali_interval=(NB * 32 * 2 if FLOW == "cin256" else NB))
phases, calls = {}, {}


def timed(mod, name, key):
    f = getattr(mod, name)

    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.time(); r = f(*a, **k); torch.cuda.synchronize()
        phases[key] = round(phases.get(key, 0.0) + time.time() - t0, 2); calls[key] = calls.get(key, 0) + 1
        return r
    setattr(mod, name, g)


timed(DG, "generate_cali_data_ldm", "calibration_set_generation_s")
timed(DG, "generate_cali_data_ldm_imagenet", "calibration_set_generation_s")
for n, k in (("tib_reconstruction", "tib_reconstruction_s"), ("block_reconstruction", "block_reconstruction_incl_capture_s"),
             ("layer_reconstruction", "layer_reconstruction_incl_capture_s"), ("_calibrate_activations", "finite_set_activation_calibration_s")):
    timed(CAL, n, k)
if os.environ.get("ONLY"):
    CAL.ONLY_UNITS = tuple(p for p in os.environ["ONLY"].split(",") if p)
torch.manual_seed(1234); np.random.seed(1234)
torch.cuda.synchronize()
t_all = time.time()
r = LatentRunner(model, opt, P["flow"], dev)
assert r.quantize(init_context=(table[1000:1001] if FLOW == "cin256" else None)) == "calibrated"
torch.cuda.synchronize()
total = time.time() - t_all
ckpt = torch.load(ck, map_location="cpu")
n_units = sum(calls.get(k, 0) for k in ("tib_reconstruction_s", "block_reconstruction_incl_capture_s", "layer_reconstruction_incl_capture_s"))
rec_s = sum(phases.get(k, 0.0) for k in ("tib_reconstruction_s", "block_reconstruction_incl_capture_s", "layer_reconstruction_incl_capture_s"))
finite = all(bool(torch.isfinite(v).all()) for v in ckpt["weight"].values() if torch.is_tensor(v) and v.is_floating_point())
rep = {"recipe": (f"{P['name']} ({n_params:.1f} M, random init), w4a8, DDIM-{T}, {NB}{' x 32 classes x (cond, uncond)' if FLOW == 'cin256' else ''} samples per step, "
                  f"interval_length {P['interval_length']}, {ITERS} AdaRound iterations per unit at mini-batch {LatentRunner.CALI_RECIPE[P['flow']][1]}, running_stat, 1 GPU"
                  + (f"; reconstruction restricted to the units under {os.environ['ONLY']}" if os.environ.get("ONLY") else "")),
       "wall_clock_s": round(total, 1), "phases_s": phases, "reconstruction_units": n_units, "iterations_per_unit": ITERS,
       "adaround_iterations_per_s": round(n_units * ITERS / max(rec_s, 1e-9), 1), "finite": finite,
       "adaround_tensors": sum(1 for k in ckpt["weight"] if k.endswith("alpha")), "act_groups": len([k for k in ckpt if k.startswith("act_")]),
       "checkpoint_MB": round(os.path.getsize(ck) / 1e6, 1), "recon_gemm": os.environ.get("TFMQ_RECON_GEMM", "bf16x3"),
       "weights": "random init (no checkpoints offline)", "device": torch.cuda.get_device_name(0)}
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(rep, open(out_path, "w"), indent=1)
print(json.dumps(rep))

"""The full CIFAR-10 w4a8 calibration recipe of the reference's README (sample_diffusion_ddim.py --timesteps 100 --skip_type quad
--wq 4 --aq 8 --cali --use_aq --interval_length 5 --running_stat), end to end through Diffusion.sample() on one MI355X, measured:
calibration-set generation (100 timesteps x 256 samples, FP sampler), weight-scale search, TIAR + 22 ResnetBlock + 6 AttnBlock + 4
layer reconstructions at 20 000 Adam iterations each (mini-batch 32), Finite-Set activation calibration (100 groups), checkpoint.
Random-init weights (no checkpoints offline).  Writes a JSON report (wall-clock per phase, final losses per unit)."""
import argparse, json, logging, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
import numpy as np, torch
import tfmq_dm_amd.ddim.models as M
from tfmq_dm_amd.ddim.runner import Diffusion
import tfmq_dm_amd.quant.reconstruction as REC
import tfmq_dm_amd.quant.calibration as CAL

ITERS = int(os.environ.get("ITERS", "20000"))
T = int(os.environ.get("T", "100"))
NB = int(os.environ.get("CALI_BATCH", "256"))
out_path = os.environ.get("OUT", os.path.join(ROOT, "gpurun_out", "r02", "cifar_calibration_full.json"))
dev = torch.device("cuda", 0)
cfg = M.make_config()
cfg.sampling = argparse.Namespace(batch_size=256)
model = M.random_init(M.Model(cfg)).to(dev).eval()
ck = os.path.join(tempfile.mkdtemp(), "cifar_w4a8.pth")
args = argparse.Namespace(sample_type="generalized", skip_type="quad", timesteps=T, eta=0.0, ptq=True, cali=True, use_aq=True, wq=4, aq=8,
                          q_mode=[2, 1], softmax_a_bit=8, interval_length=5, running_stat=True, asym=True, cali_save_path=ck,
                          cali_batch=NB, cali_iters=ITERS, max_images=256, fid=True)
phases = {}
marks = []
orig_gen = None
import tfmq_dm_amd.quant.data_generate as DG
_g = DG.generate_cali_data_ddim
def timed_gen(*a, **k):
    t0 = time.time(); r = _g(*a, **k); torch.cuda.synchronize(); phases["calibration_set_generation_s"] = round(time.time() - t0, 2); return r
import tfmq_dm_amd.ddim.runner as RUN
_walk, _act = CAL._recon_walk, CAL._calibrate_activations
def timed_walk(*a, **k):
    torch.cuda.synchronize(); t0 = time.time(); r = _walk(*a, **k); torch.cuda.synchronize(); phases["reconstruction_s"] = round(time.time() - t0, 2); return r
def timed_act(*a, **k):
    torch.cuda.synchronize(); t0 = time.time(); r = _act(*a, **k); torch.cuda.synchronize(); phases["finite_set_activation_calibration_s"] = round(time.time() - t0, 2); return r
CAL._recon_walk, CAL._calibrate_activations = timed_walk, timed_act
DG.generate_cali_data_ddim = timed_gen
trace = {"counts": (1, int(0.2 * ITERS), ITERS // 2, ITERS), "rows": [], "unit": 0}
REC.LOSS_TRACE = trace
torch.manual_seed(1234); np.random.seed(1234)
t_all = time.time()
import tfmq_dm_amd.quant.data_generate  # noqa
# Diffusion.sample imports generate_cali_data_ddim from the module at call time -> patched version is used
qnn, _ = Diffusion(args, cfg, device=dev).sample(model)
torch.cuda.synchronize()
total = time.time() - t_all
rows = trace["rows"]
units = {}
for (u, c, r, q) in rows:
    units.setdefault(int(u), {})[int(c)] = {"rec": r, "round": q}
ckpt = torch.load(ck, map_location="cpu")
n_alpha = sum(v.numel() for k, v in ckpt["weight"].items() if k.endswith("alpha"))
rep = {"recipe": f"CIFAR-10 DDPM UNet 35.7M, w4a8, DDIM-{T} quad, {NB} samples/timestep, interval_length 5, {ITERS} iterations/unit, batch 32, running_stat",
       "wall_clock_s": round(total, 1), "phases_s": phases, "reconstruction_units": trace["unit"], "iterations_per_unit": ITERS,
       "ms_per_iteration_all_units": round(phases.get("reconstruction_s", 0) * 1e3 / max(ITERS, 1), 3),
       "adaround_parameters": n_alpha, "act_groups": len([k for k in ckpt if k.startswith("act_")]),
       "checkpoint_MB": round(os.path.getsize(ck) / 1e6, 1), "losses_per_unit": units,
       "weights": "random init (no checkpoints offline)", "device": torch.cuda.get_device_name(0)}
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(rep, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in rep.items() if k != "losses_per_unit"}))

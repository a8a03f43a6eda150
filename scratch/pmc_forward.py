"""Eager UNet forwards of a bench workload, for rocprofv3 --pmc passes (counter collection segfaults on the
hipGraph replay of bench.py, so the same forwards run eagerly here).  usage: pmc_forward.py [sd|cifar]"""
import sys, os, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import tfmq_dm_amd.ops as ops
wl = sys.argv[1] if len(sys.argv) > 1 else "sd"
dev = torch.device("cuda", 0)
args = argparse.Namespace(batch=0, ddim_steps=4, first_sampling=False)
run, fwd, cpu, info = (bench.setup_sd if wl == "sd" else bench.setup_cifar)(args, dev, 0, lambda *a: None)
# marker dispatch: make_traffic_json.py keeps only what follows the last k_upsample2x kernel (= the 3 forwards below)
torch.cuda.synchronize()
ops.upsample2x(torch.zeros(1, 2, 2, 4, device=dev))
torch.cuda.synchronize()
with torch.cuda.stream(info["stream"]):
    info["step"].zero_()
    for _ in range(3):
        fwd()
    info["stream"].synchronize()
torch.cuda.synchronize()

"""Which kernel each conv / linear shape of a bench workload runs on (the engine's autotune cache) and its measured time:
usage: tile_report.py [sd|cifar|cin256|celeba].  Prints one line per distinct launch shape, slowest total first."""
import sys, os, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import tfmq_dm_amd.ops as ops
wl = sys.argv[1] if len(sys.argv) > 1 else "sd"
dev = torch.device("cuda", 0)
args = argparse.Namespace(batch=0, ddim_steps=4, first_sampling=False)
if wl == "cifar":
    run, fwd, cpu, info = bench.setup_cifar(args, dev, 0, lambda *a: None)
else:
    run, fwd, cpu, info = bench.setup_sd(args, dev, 0, lambda *a: None, preset=wl)
eng = info["oracle_state"]["eng"] if "oracle_state" in info else info["eng"]
names = ops._TILE_NAMES
keys = ("kind", "B", "H", "W", "Cin", "Cout", "KH", "stride", "up2x", "out_mode", "res", "stats", "seg", "x_f16", "yt")
for k, v in sorted(eng.tiles.items(), key=lambda kv: str(kv[0])):
    print(ops.tile_name(v), "|", " ".join(f"{a}={b}" for a, b in zip(keys, k)))

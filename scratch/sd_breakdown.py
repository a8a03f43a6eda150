"""Per-shape timing of every conv / linear launch of one SD UNet forward (HIP events per launch)."""
import sys, os, collections, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import tfmq_dm_amd.ops as ops
dev = torch.device("cuda", 0)
args = argparse.Namespace(batch=int(os.environ.get("SD_BATCH", "8")), ddim_steps=int(os.environ.get("SD_STEPS", "4")), first_sampling=False)
run, fwd, cpu, info = bench.setup_sd(args, dev, 0, lambda *a: print(*a, file=sys.stderr))
rec, shapes = [], []
orig = ops._profiled_conv
def prof(name, kind, d, dsc, nops, nbytes=0.0):
    shapes.append((kind, dsc.B, dsc.H, dsc.W, dsc.Cin, dsc.Cout, dsc.KH, dsc.stride, dsc.up2x, bool(dsc.residual), bool(dsc.stats)))
    return orig(name, kind, d, dsc, nops, nbytes)
ops._profiled_conv = prof
with torch.cuda.stream(info["stream"]):
    for it in range(3):
        shapes.clear(); rec.clear()
        ops.set_conv_profile(rec)
        fwd()
        info["stream"].synchronize()
        ops.set_conv_profile(None)
agg = collections.OrderedDict()
for s, (e0, e1, nops, kind, nbytes) in zip(shapes, rec):
    ms = ops.event_elapsed_ms(e0, e1)
    a = agg.setdefault(s, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += nops; a[3] += nbytes
tot = sum(a[1] for a in agg.values())
print("kind B H W Cin Cout K stride up res stats")
for s, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{s}: n={a[0]} total={a[1]*1e3:8.1f} us avg={a[1]/a[0]*1e3:7.1f} us  {a[2]/a[1]/1e9:8.1f} TOP/s  {a[3]/a[1]/1e9:7.2f} TB/s {100*a[1]/tot:4.1f}%")
print("total conv ms per forward", tot)

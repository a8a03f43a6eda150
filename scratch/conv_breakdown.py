import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import tfmq_dm_amd.ops as ops
dev = torch.device("cuda", 0)
eng, cfg, sd, wq, names, seq, betas = bench.build_quantized_engine(dev, 256, 100)
eng.build_tib_table([float(i) for i in reversed(seq)])
x = torch.randn(256, 32, 32, 3, device=dev)
# monkeypatch to record shapes
rec = []
orig = ops._profiled_conv
shapes = []
def prof(name, kind, d, dsc, nops):
    shapes.append((kind, dsc.B, dsc.H, dsc.W, dsc.Cin, dsc.Cout, dsc.KH, dsc.stride, dsc.up2x))
    return orig(name, kind, d, dsc, nops)
ops._profiled_conv = prof
for it in range(3):
    shapes.clear(); rec.clear()
    ops.set_conv_profile(rec)
    eng.forward(x, None)
    torch.cuda.synchronize()
    ops.set_conv_profile(None)
agg = collections.OrderedDict()
for s, (e0, e1, nops, kind) in zip(shapes, rec):
    ms = ops.event_elapsed_ms(e0, e1)
    a = agg.setdefault(s, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += nops
tot = sum(a[1] for a in agg.values())
for s, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{s}: n={a[0]} total={a[1]*1e3:8.1f} us avg={a[1]/a[0]*1e3:7.1f} us  {a[2]/a[1]/1e9:8.1f} TOP/s  {100*a[1]/tot:4.1f}%")
print("total conv ms per forward", tot)

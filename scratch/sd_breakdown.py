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
# the token-per-lane launches of round 4 append their own records (ops.row_chain / ops.ff_fused): name them so that the zip below stays aligned
_rc, _ff = ops.row_chain, ops.ff_fused
def rc(x, T, gemms, gn=None, ln=None):
    shapes.append(("row_chain:" + ("gn+" if gn is not None else "") + "+".join(f"{g['pw'].cin}->{g['pw'].cout}" + ("+res" if g.get("residual") is not None else "") + ("+ln" if g.get("ln") else "") for g in gemms),
                   x.shape[0] // T, T, 1, x.shape[1], sum(g["pw"].cout for g in gemms), 1, 1, 0, any(g.get("residual") is not None for g in gemms), False))
    return _rc(x, T, gemms, gn=gn, ln=ln)
def ff(x, gamma, beta, eps, aq0, pw1, aq2, pw2, out_q8=None, pre=None, post=None):
    src = x if pre is None else pre["xq"]
    M = src.numel() // src.shape[-1]
    shapes.append(("ff_fused" + ("+to_out" if pre is not None else "") + ("+proj_out" if post is not None else ""), M, 1, 1, src.shape[-1], pw1.cout // 2, 1, 1, 0, True, post is not None))
    return _ff(x, gamma, beta, eps, aq0, pw1, aq2, pw2, out_q8=out_q8, pre=pre, post=post)
ops.row_chain, ops.ff_fused = rc, ff
with torch.cuda.stream(info["stream"]):
    for it in range(3):
        shapes.clear(); rec.clear()
        ops.set_conv_profile(rec)
        fwd()
        info["stream"].synchronize()
        ops.set_conv_profile(None)
agg = collections.OrderedDict()
for s, (e0, e1, nops, kind, nbytes, _fam) in zip(shapes, rec):
    ms = ops.event_elapsed_ms(e0, e1)
    a = agg.setdefault(s, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += nops; a[3] += nbytes
tot = sum(a[1] for a in agg.values())
print("kind B H W Cin Cout K stride up res stats")
for s, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{s}: n={a[0]} total={a[1]*1e3:8.1f} us avg={a[1]/a[0]*1e3:7.1f} us  {a[2]/a[1]/1e9:8.1f} TOP/s  {a[3]/a[1]/1e9:7.2f} TB/s {100*a[1]/tot:4.1f}%")
print("total conv ms per forward", tot)

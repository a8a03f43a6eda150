"""Per-quantizer bin-flip rates of the W8A8 forward (fixture F20 tables) in the exact / fast modes vs the oracle's trace."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tfmq_oracle as O
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.asarray(a))
which = sys.argv[1] if len(sys.argv) > 1 else "ddim"
from tfmq_dm_amd.engine import DdimUNetEngine, LayerQ, LdmUNetEngine
import tfmq_dm_amd.ops as ops
g = np.load(os.path.join(ROOT, "tests/golden/f20_w8a8.npz"))
base = np.load(os.path.join(ROOT, "tests/golden", ("f7_ddim_tiny" if which == "ddim" else "f11_ldm_tiny") + ".npz"))
sd = {k[3:]: T(base[k]) for k in base.files if k.startswith("sd/")}
cfg = dict(ch=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[8], resolution=16) if which == "ddim" else dict(model_channels=32, num_heads=2, in_channels=4)
eng = (DdimUNetEngine if which == "ddim" else LdmUNetEngine)(sd, cfg, DEV)
print("exact_fp", eng.exact_fp)
pre = which + "/"
act_names = sorted(k[len(pre) + 3:-6] for k in g.files if k.startswith(pre + "aq/") and k.endswith("/delta"))
qid = {n: i for i, n in enumerate(act_names)}
wqd = {}
for k in g.files:
    if k.startswith(pre + "wq/") and k.endswith("/delta"):
        n = k[len(pre) + 3:-6]
        wqd[n] = LayerQ(T(g[k]), T(g[f"{pre}wq/{n}/zp"]), None, qid.get(n), level=256)
qtable = torch.tensor([[[float(g[f"{pre}aq/{n}/delta"]), float(g[f"{pre}aq/{n}/zp"])] for n in act_names]])
x, t = T(base["x"]), T(base["t"]).float()
args = (x.permute(0, 2, 3, 1).contiguous().to(DEV), t.to(DEV)) + ((T(base["ctx"]).to(DEV),) if which == "ldm" else ())
if os.environ.get("WIDE_VIA_EXACT"):
    from tfmq_dm_amd.engine import ddim_unet as D
    one = torch.ones(1, device=DEV)
    def rw(self, xq, out=None, y_coff=0, want_stats=None, **kw):
        cin, cout = self.p.cin, self.p.cout
        if getattr(self, "_dbg_w32", None) is None:
            self._dbg_w32 = (self.p.w16[:, :, :cin].float() * self.p.wscale.reshape(cout, 1, 1)).reshape(cout, -1).contiguous()
        x = ops.bins_to_grid(xq, self.aq, half=False) * ops.scale_by_qdelta(one, self.aq)
        self.w32 = self._dbg_w32
        y = self._run_exact(x, **kw)
        self.w32 = None
        if out is not None:
            out[..., y_coff:y_coff + cout] = y
            return out
        return y
    D._Layer._run_wide = rw
eng.prepare(wqd, qtable.to(DEV))
eng.set_calibration("record", 0)
eng.forward(*args)
eng.set_calibration(None)
owq = {n: {"delta": q.delta.reshape((-1,) + (1,) * (sd[n + ".weight"].dim() - 1)),
           "zp": q.zp.reshape((-1,) + (1,) * (sd[n + ".weight"].dim() - 1)), "alpha": None} for n, q in wqd.items()}
qs = O.QuantSpec(wq=owq, aq={n: (qtable[0, i, 0], qtable[0, i, 1]) for i, n in enumerate(act_names)}, w_level=256)
qs.trace = {}
with torch.no_grad():
    if which == "ddim":
        eo = O.ddim_unet_forward(sd, dict(cfg), x, T(base["t"]), qs)
    else:
        eo = O.ldm_unet_forward(sd, dict(cfg), x, T(base["t"]).long(), T(base["ctx"]), qs)
for n in qs.trace:        # the oracle's call order
    i = qid[n]
    if i not in eng.observed:
        print(f"{n:60s} not observed"); continue
    be = (ops.quantize_act(eng.observed[i].float().contiguous(), ops.qsel(qtable[:, i:i + 1].contiguous().to(DEV))).to(torch.int32) + 128).cpu()
    bo = qs.trace[n].to(torch.int32)
    if bo.dim() == 4:
        bo = bo.permute(0, 2, 3, 1)
    if bo.numel() == 4 * be.numel():
        bo = bo[:, ::2, ::2, :]
    diff = (be - bo.reshape(be.shape)).abs()
    print(f"{n:60s} moved {float((diff > 0).float().mean()):.4f}  >1: {float((diff > 1).float().mean()):.4f}  max {int(diff.max())}")
e = eng.forward(*args).permute(0, 3, 1, 2).cpu()
print("eps rel-L2 vs oracle", float((e - eo).norm() / eo.norm()))

"""pair_prefix vs the materialised guidance pair on the F11 tiny SD-style UNet: max |diff| of eps per engine state, under the env
switches given on the command line (run once per setting: the engine reads them at construction)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tfmq_dm_amd.engine import LayerQ, LdmUNetEngine
DEV = "cuda:0"
g = np.load(os.path.join(ROOT, "tests", "golden", "f11_ldm_tiny.npz"))
T = lambda a: torch.from_numpy(np.asarray(a))
nhwc = lambda x: x.permute(0, 2, 3, 1).contiguous().to(DEV)
sd = {k[3:]: T(g[k]) for k in g.files if k.startswith("sd/")}
act_names = sorted(k[3:-6] for k in g.files if k.startswith("aq/") and k.endswith("/delta"))
qid = {n: i for i, n in enumerate(act_names)}
wq = {k[3:-6]: LayerQ(T(g[k]), T(g[f"wq/{k[3:-6]}/zp"]), None, qid.get(k[3:-6])) for k in g.files if k.startswith("wq/") and k.endswith("/delta")}
qtable = torch.tensor([[[float(g[f"aq/{n}/delta"]), float(g[f"aq/{n}/zp"])] for n in act_names]])
x, t, ctx, uc = T(g["x"]), T(g["t"]).float(), T(g["ctx"]), T(g["traj_uc"])
B = x.shape[0]
t = t[:1].repeat(B)
x2, t2, c2 = torch.cat([x, x]), torch.cat([t, t]).to(DEV), torch.cat([uc, ctx]).to(DEV)
step = torch.zeros(1, dtype=torch.int32, device=DEV)
eng = LdmUNetEngine(sd, dict(model_channels=32, num_heads=2, in_channels=4), DEV)
for state in ("fp", "w4a8"):
    if state == "fp":
        eng.prepare()
    else:
        eng.prepare(wq, qtable.repeat(4, 1, 1).contiguous().to(DEV), step)
    full = eng.forward(nhwc(x2), t2, c2)
    pair = eng.forward(nhwc(x), t2, c2, pair_prefix=True)
    half = eng.forward(nhwc(x), t2[:B], c2[B:])          # batch independence itself: the cond half alone
    print(f"[{' '.join(sys.argv[1:]) or 'default'}] {state}: pair vs full max|d| = {float((pair - full).abs().max()):.3e}   "
          f"cond half alone vs full[B:] = {float((half - full[B:]).abs().max()):.3e}   (|eps| max {float(full.abs().max()):.3f})")
    import tfmq_dm_amd.ops as ops
    print("   tiles:", {k[1:8]: v for k, v in eng.tiles.items() if k[0] == "f16"} if state == "fp" else "")

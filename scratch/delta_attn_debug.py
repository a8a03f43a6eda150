"""Debug: first-iteration dL/ddelta of mid.attn_1 (F22 state) -- unit vs torch autograd on the unit's own captured tensors."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["TFMQ_RECON_GEMM"] = "f32"
os.environ["TFMQ_EXACT_FP"] = "1"
import numpy as np, torch
import torch.nn.functional as F
import test_delta_learning_gpu as TD
GOLD = os.path.join(ROOT, "tests", "golden")
golden = lambda n: np.load(os.path.join(GOLD, n + ".npz"), allow_pickle=False)
qnn, g8, g = TD._state(golden, False)
import quant.reconstruction as REC
from quant.data_utill import save_inout
from tfmq_dm_amd.engine import recon as R
block = dict(qnn.model.named_modules())["mid.attn_1"]
qnn.set_quant_state(False, False); block.set_quant_state(True, True)
ds = REC._DeltaSet()
fl = [ds.fixed(getattr(block, n)) for n in ("q", "k", "v", "proj_out")]
data = (TD.T(g8["cali_x"]), TD.T(g8["cali_t"]))
ci, co = save_inout(qnn, block, data, True, True, 48, True)
unit = R.DeltaAttnUnit(fl[0], fl[1], fl[2], fl[3], (block.norm.weight.data.float(), block.norm.bias.data.float()), ci[0], co, **ds.kw(30, 5e-4, False))
rec, grads = unit._forward_backward(torch.arange(48, device="cuda:0"))
print("unit loss", float(rec), "grads", [float(x) for x in grads])
x = ci[0].cpu().permute(0, 3, 1, 2).contiguous(); y = co.cpu().permute(0, 3, 1, 2).contiguous()
Cc = x.shape[1]; B, H, W = x.shape[0], x.shape[2], x.shape[3]
d = [torch.tensor(float(v), requires_grad=True) for v in unit.delta.cpu()]
zp = [float(v) for v in unit.zp.cpu()]
Wt = [f.wg.cpu().reshape(Cc, Cc, 1, 1) for f in fl]; bt = [f.bias.cpu() for f in fl]
hn = F.group_norm(x, 32, block.norm.weight.data.cpu().float(), block.norm.bias.data.cpu().float(), 1e-6)
q = F.conv2d(TD._fq(hn, d[0], zp[0]), Wt[0], bt[0]).reshape(B, Cc, H * W).permute(0, 2, 1)
k = F.conv2d(TD._fq(hn, d[1], zp[1]), Wt[1], bt[1]).reshape(B, Cc, H * W)
v = F.conv2d(TD._fq(hn, d[2], zp[2]), Wt[2], bt[2]).reshape(B, Cc, H * W)
w_ = torch.softmax(torch.bmm(q, k) * (int(Cc) ** (-0.5)), dim=2)
h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(B, Cc, H, W)
out = x + F.conv2d(TD._fq(h_, d[3], zp[3]), Wt[3], bt[3])
loss = ((out - y) ** 2).sum(1).mean()
loss.backward()
print("torch loss", float(loss), "grads", [float(t.grad) for t in d])
print("reference first-step signs (trajectory[0] - before):", (g["mid.attn_1/trajectory"][0] - g["mid.attn_1/before"]).tolist(), "reference loss[0]", g["mid.attn_1/loss"][0])

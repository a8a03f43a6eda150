import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd"))
os.environ["TFMQ_RECON_GEMM"] = "f32"; os.environ["TFMQ_EXACT_FP"] = "1"
import numpy as np, torch
import test_delta_learning_gpu as TD
GOLD = os.path.join(ROOT, "tests", "golden")
cache = {}
def golden(n):
    if n not in cache: cache[n] = np.load(os.path.join(GOLD, n + ".npz"), allow_pickle=False)
    return cache[n]
import quant.reconstruction as REC
from tfmq_dm_amd.engine import recon as R
orig = R._DeltaUnit.iterate
def it(self, idx):
    rec, grads = self._forward_backward(idx)
    print("attn_q", getattr(self, "attn_q", None), "grads", [None if g is None else float(g) for g in grads], "rec", float(rec), "delta", self.delta.tolist())
    return orig(self, idx)
R._DeltaUnit.iterate = it
class MP:
    def setenv(self, k, v): os.environ[k] = v
    def setattr(self, o, n, v): setattr(o, n, v)
g = golden("f25_delta_learning_attention")
try:
    TD.test_delta_learning_through_live_attention_quantizers(golden, MP(), "down.1.attn.0", False)
except AssertionError as e:
    print("assert", str(e)[:200])
n = "down.1.attn.0"
print("ref traj first 3:", g[n + "/trajectory"][:3].tolist())
print("ref loss:", g[n + "/loss"][:4])

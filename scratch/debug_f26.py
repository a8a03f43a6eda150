"""F26 debugging: the mirror's cached inputs / target of the QK seam of input_blocks.1.1 against the reference's (dumped by hand into
tests/golden/_debug_f26_inout.npz)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tfmq-dm_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["TFMQ_RECON_GEMM"] = "f32"; os.environ["TFMQ_EXACT_FP"] = "1"
import numpy as np, torch
from test_ldm_attnblock import qnn_of
from quant.calibration import load_cali_model
from quant.data_utill import save_inout
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.asarray(a))
G = lambda n: np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"), allow_pickle=False)
g13, g16, g, dbg = G("f13_ldm_attnblock_tiny"), G("f16_attnblock_cali_tiny"), G("f26_delta_learning_qk_smv"), G("_debug_f26_inout")
name, pre = "input_blocks.1.1.attention.qkv_matmul", "attnblock/"
qnn = qnn_of(g13, DEV, cali=False).to(DEV)
ck = {"weight": {str(k): T(g16["ck/weight/" + str(k)]) for k in g16["weight_keys"] if "ck/weight/" + str(k) in g16.files}}
for n, mod in qnn.model.named_modules():
    if hasattr(mod, "original_w") and ("model." + n + ".w") in set(map(str, g16["weight_keys"])):
        ck["weight"]["model." + n + ".w"] = mod.original_w.detach().cpu().clone()
        if getattr(mod, "original_b", None) is not None:
            ck["weight"]["model." + n + ".b"] = mod.original_b.detach().cpu().clone()
akeys = [str(k) for k in g16["act_keys"]]
dk, zk = [k for k in akeys if k.endswith("delta")], [k for k in akeys if k.endswith("zero_point")]
for gi in range(3):
    d, z = T(g16[f"ck/act_{gi}/delta"]), T(g16[f"ck/act_{gi}/zp"])
    ck[f"act_{gi}"] = {**{k: d[i].clone() for i, k in enumerate(dk)}, **{k: z[i].clone() for i, k in enumerate(zk)}}
path = os.path.join(tempfile.mkdtemp(), "c.pth"); torch.save(ck, path)
load_cali_model(qnn, (T(g[pre + "init_x"]), T(g[pre + "init_t"]).float()), use_aq=True, path=path)
qnn.load_state_dict(ck["act_1"], strict=False)
unit = dict(qnn.model.named_modules())[name]
unit.use_aq = True
for an in ("aqtizer_q", "aqtizer_k"):
    q = getattr(unit, an)
    q.delta = torch.nn.Parameter(T(g[f"{pre}{name}/attn_q/{an}/delta"]).reshape(()).clone().to(DEV))
    q.zero_point = torch.tensor(float(g[f"{pre}{name}/attn_q/{an}/zp"]), device=DEV); q.init = True
qnn.invalidate()
qnn.set_quant_state(False, False); unit.set_quant_state(True, True)
data = (T(g16["cali_x"]), T(g16["cali_t"]))
for asym in (True, False):
    (q, k), S = save_inout(qnn, unit, data, asym, True, 48, True)
    H = S.shape[1]
    rows = lambda x: x.reshape(x.shape[0], x.shape[1], H, -1).permute(0, 2, 3, 1).reshape(x.shape[0] * H, -1, x.shape[1]).cpu()    # -> [(b h), ch, T]
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print(f"asym={asym}: q rel-L2 {rel(rows(q), T(dbg['q'])):.3e}  k {rel(rows(k), T(dbg['k'])):.3e}  target S {rel(S.reshape(-1, S.shape[2], S.shape[3]).cpu(), T(dbg['S'])):.3e}")

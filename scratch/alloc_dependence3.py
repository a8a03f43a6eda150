"""Narrowing, part 3: a fresh sampler's step body run EAGERLY S times (no graph) -- with its arena (recorded pass + replays), and without."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import tfmq_dm_amd.ops as ops
dev = torch.device("cuda", 0)
args = argparse.Namespace(batch=int(os.environ.get("SD_BATCH", "8")), ddim_steps=int(os.environ.get("SD_STEPS", "20")))
run, fwd, cpu, info = bench.setup_sd(args, dev, 0, lambda *a: None)
smp = info["sampler"]
eng = smp.eng
x_T, cond, uncond = info["inputs"]
run(); smp.stream.synchronize()
ref = smp.x.clone()
tib0 = eng.tib_table.clone()
S = smp.coef.shape[0]


def poison(pattern):
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    t = torch.empty(int(free * 0.9) // 4, dtype=torch.int32, device=dev)
    t.fill_(pattern)
    torch.cuda.synchronize()
    del t


def eager(s2, use_ar, tune):
    with torch.cuda.stream(s2.stream):
        s2.x.copy_(x_T); s2.ctx2[:s2.batch].copy_(uncond); s2.ctx2[s2.batch:].copy_(cond)
        s2.step.zero_()
        ops.set_conv_autotune(eng.tiles if tune else None)
        try:
            for i in range(S):
                with ops.use_arena(s2.arena if use_ar else None):
                    s2._step_body()
        finally:
            ops.set_conv_autotune(None)
        s2.stream.synchronize()
    d = (s2.x - ref).abs()
    return "same" if float(d.max()) == 0 else f"{float(d.max()):.1e}"


for pat in (0x7f7f7f7f, -1, 0x7f7f7f7f, 0x3c003c00, 0x7f7f7f7f, 0x01010101, 0x3c003c00):
    poison(pat)
    s2 = info["new_sampler"]()
    if os.environ.get("SYNC") == "1":
        torch.cuda.synchronize()
    tib_now = eng.tib_table
    r = [f"{k}: {eager(s2, *v)}" for k, v in (("no arena, rule tiles", (False, False)), ("no arena, tuned tiles", (False, True)), ("arena, tuned tiles", (True, True)))]
    s2.capture()
    o = s2.sample_nhwc(x_T, cond, uncond); s2.stream.synchronize()
    d = (o - ref).abs()
    r.append("graph: " + ("same" if float(d.max()) == 0 else f"{float(d.max()):.1e}"))
    torch.cuda.synchronize()
    r.append("TIB table " + ("same" if torch.equal(tib_now, tib0) else f"DIFFERENT ({float((tib_now - tib0).abs().max()):.2e})"))
    print(f"{pat & 0xffffffff:08x}: " + "; ".join(r), flush=True)
    del s2, o

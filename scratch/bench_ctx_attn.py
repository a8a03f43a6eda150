"""Cross attention over the 77-token context with int8 output at the SD levels: k_attention_ctx (TFMQ_ATTN_CTX=1, default) vs k_attention_h (=0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tfmq_dm_amd.ops as ops
dev = torch.device("cuda", 0)
B = int(os.environ.get("BATCH", "128"))
qt = torch.tensor([[[0.02, 128.0]]], device=dev)
sel = ops.qsel(qt, 0)
for (T, heads, d) in ((4096, 8, 40), (1024, 8, 80)):
    C = heads * d
    q = torch.randn(B, T, C, device=dev).half()
    k = torch.randn(B, 80, C, device=dev).half()
    vt = torch.randn(B, C, 80, device=dev).half()
    f = lambda: ops.attention_f16(q, k, vt, heads, d ** -0.5, sel, want_f32=False, n_keys=77)
    if os.environ.get("HEAD_MAJOR") == "1":       # q stored [B][heads][T][d]: timing of the layout only (the values are a permutation)
        import ctypes as C_
        from tfmq_dm_amd.ops import _p, _stream, handle, QSel, _alloc
        yq = _alloc(B, T, C, dtype=torch.int8, device=dev)
        f = lambda: handle(0).call("attention_f16", _p(q), _p(k), _p(vt), d, k.stride(1), None, C, _p(yq), sel, B, heads, T, 77, 80, d, float(d ** -0.5), _stream(0))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record(); torch.cuda.synchronize()
    print(f"ATTN_CTX={os.environ.get('TFMQ_ATTN_CTX', '1')} cross attention B {B} T {T} d {d}: {e0.elapsed_time(e1) / 5 * 1e3:8.1f} us", flush=True)

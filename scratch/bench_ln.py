"""LayerNorm + 8-bit quantise at the SD token shapes: TB/s of (4 B read + 1 B written) per element."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tfmq_dm_amd.ops as ops
DEV = "cuda:0"
qt = torch.tensor([[0.02, 128.0]], device=DEV); sel = ops.qsel(qt)
for rows, C in [(128 * 4096, 320), (128 * 1024, 640), (128 * 256, 1280), (32 * 4096, 320)]:
    x = torch.randn(rows, C, device=DEV); g = torch.ones(C, device=DEV); b = torch.zeros(C, device=DEV)
    if os.environ.get("F16", "1") == "1":          # the fp16 activation stream (2 B read + 1 B written per element)
        x = x.half()
    fn = lambda: ops.layernorm(x, g, b, 1e-5, sel)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"rows {rows} C {C}: {ms*1e3:7.1f} us  {rows*C*(x.element_size()+1)/ms/1e9:5.2f} TB/s")

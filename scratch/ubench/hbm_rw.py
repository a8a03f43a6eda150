"""HBM write / copy / read rates at 1 GB through torch's own kernels (what a write-dominated epilogue can hope for)."""
import torch
dev = "cuda:0"
n = 512 * 1024 * 1024
x = torch.empty(n, dtype=torch.float16, device=dev).normal_()
y = torch.empty_like(x)
def t(fn, nbytes, name, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"{name}: {ms*1e3:8.1f} us  {nbytes/ms/1e9:5.2f} TB/s")
t(lambda: y.zero_(), 2 * n, "memset 1 GB")
t(lambda: y.fill_(1.5), 2 * n, "fill 1 GB")
t(lambda: y.copy_(x), 4 * n, "copy 1 GB -> 1 GB (read + write bytes)")
t(lambda: x.sum(), 2 * n, "sum 1 GB (read)")
xs = x[: n // 4]
t(lambda: torch.add(xs, 1.0, out=y[: n // 4]), n, "add 256 MB -> 256 MB (read + write bytes)")

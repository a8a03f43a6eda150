import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda", 0)
eng, cfg, sd, wq, names, seq, betas = bench.build_quantized_engine(dev, 256, 4)
eng.build_tib_table([float(i) for i in reversed(seq)])
x = torch.randn(256, 32, 32, 3, device=dev)
for _ in range(4):
    eng.forward(x, None)
torch.cuda.synchronize()

#!/bin/bash
# round 6, call 5: host memory of the box, HostRows tests, pinned-allocation and host -> device row-copy rates
mkdir -p gpurun_out/r06
O=gpurun_out/r06/run5_host_probe.txt
(free -g; nproc; ulimit -l) > $O 2>&1
python -m pytest tests/test_cache_host_rows_gpu.py tests/test_calibration_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 >> $O
python - >> $O 2>&1 <<'PY'
import time, torch
for gb in (8, 32):
    t = time.perf_counter()
    try:
        b = torch.empty((gb * 64, 4096, 1024), dtype=torch.float32, pin_memory=True)     # gb GiB: rows of 16 MiB
    except Exception as e:
        print(f"pinned {gb} GiB: FAILED {type(e).__name__}: {str(e)[:120]}"); break
    dt = time.perf_counter() - t
    out = torch.empty((8, 4096, 1024), dtype=torch.float32, device="cuda:0")
    idx = torch.randint(0, gb * 64, (200, 8))
    torch.cuda.synchronize(); t = time.perf_counter()
    for r in idx.tolist():
        for j, i in enumerate(r):
            out[j].copy_(b[i], non_blocking=True)
    torch.cuda.synchronize(); dc = time.perf_counter() - t
    print(f"pinned {gb} GiB: allocated in {dt:.1f} s; 8 random 16-MiB rows host -> device: {dc / 200 * 1e3:.2f} ms per mini-batch = {200 * 8 * 16 / 1024 / dc:.1f} GiB/s")
    del b
PY
cat $O

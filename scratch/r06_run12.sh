#!/bin/bash
# round 6, call 12: d = 40 attention with 53 KB of LDS (three 4-wave blocks per CU) vs 56 KB (two), tests + same-box A/B; then the default bench line
mkdir -p gpurun_out/r06
O=gpurun_out/r06/run12_attn_lds3.txt; : > $O
timeout 900 python -m pytest tests/test_attention_f16_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 >> $O
for s in 0 1 0 1; do
  echo "== TFMQ_ATTN_LDS3=$s" >> $O
  TFMQ_ATTN_LDS3=$s BATCH=128 ONLY40=1 timeout 300 python scratch/bench_attn.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
timeout 1500 python bench.py --steps 2 --warmup 1 > gpurun_out/r06/run12_bench_line.json 2> gpurun_out/r06/run12_bench.err
tail -c 1500 gpurun_out/r06/run12_bench_line.json | cut -c1-600; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/run12_bench_line.json').read().strip().splitlines()[-1])
print("VALUE", d["value"], d.get("value_gelu_exact"), d["roofline"]["frac"], d["ms_per_step"])
PY

#!/bin/bash
# round 6, call 9: ping-pong d = 40 self-attention (TFMQ_ATTN_PP), tests first, then same-box A/B at UNet batch 128
mkdir -p gpurun_out/r06
O=gpurun_out/r06/run9_attn_pp.txt; : > $O
timeout 900 python -m pytest tests/test_attention_f16_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 >> $O
for s in 0 1 0 1; do
  echo "== TFMQ_ATTN_PP=$s" >> $O
  TFMQ_ATTN_PP=$s BATCH=128 ONLY40=1 timeout 300 python scratch/bench_attn.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O

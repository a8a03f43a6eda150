#!/bin/bash
# round 3: the pipelined d = 40 attention kernel -- correctness, then same-box A/B against the tile-at-a-time kernel
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r03; cd $R
for nw in 8 4; do
TFMQ_ATTN_PIPE_NW=$nw timeout 600 python -m pytest tests/test_attention_f16_gpu.py -x -q 2>&1 | tail -5 | tee -a gpurun_out/r03/attn_tests.txt
  echo "== pipelined nw=$nw" | tee -a gpurun_out/r03/attn_ab.txt
  TFMQ_ATTN_PIPE_NW=$nw BATCH=128 ONLY40=1 timeout 300 python scratch/bench_attn.py 2>&1 | tail -1 | tee -a gpurun_out/r03/attn_ab.txt
done
echo "== tile-at-a-time (round 2)" | tee -a gpurun_out/r03/attn_ab.txt
TFMQ_ATTN_PIPE=0 BATCH=128 ONLY40=1 timeout 300 python scratch/bench_attn.py 2>&1 | tail -1 | tee -a gpurun_out/r03/attn_ab.txt

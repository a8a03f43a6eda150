#!/bin/bash
# round 3: timing-only ablations of k_attention_d40 (scratch/ab/libtfmq_abl.so = the library with -DTFMQ_ATTN_ABLATE)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r03; cd $R
export TFMQ_LIB_PATH=$R/scratch/ab/libtfmq_abl.so BATCH=128 ONLY40=1
for dbg in ${DBGS:-0 128 1 2 4 6 8 16 32 64 49 14 78}; do
  echo -n "dbg=$dbg  " | tee -a gpurun_out/r03/attn_abl${TAG}.txt
  TFMQ_ATTN_DBG=$dbg timeout 120 python scratch/bench_attn.py 2>&1 | grep "d40" | tee -a gpurun_out/r03/attn_abl${TAG}.txt
done

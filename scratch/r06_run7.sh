#!/bin/bash
# round 6, call 7: LDS-DMA issue of a K-step split between the two wave halves (slab: TFMQ_SLAB_ISSUE_SPLIT, pointwise: TFMQ_LIN_ISSUE_SPLIT), same-box A/B
mkdir -p gpurun_out/r06
O=gpurun_out/r06/run7_issue_split.txt; : > $O
for s in 0 1 0 1; do
  echo "== TFMQ_SLAB_ISSUE_SPLIT=$s" >> $O
  TFMQ_SLAB_ISSUE_SPLIT=$s TILES=5 timeout 300 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids >> $O
done
for s in 0 1 0 1; do
  echo "== TFMQ_LIN_ISSUE_SPLIT=$s" >> $O
  TFMQ_LIN_ISSUE_SPLIT=$s TILES=6 timeout 300 python scratch/bench_lin.py 2>&1 | grep -v amdgpu.ids >> $O
done
python -m pytest tests/test_conv_epilogue_modes_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 >> $O
TFMQ_SLAB_ISSUE_SPLIT=1 TFMQ_LIN_ISSUE_SPLIT=1 python -m pytest tests/test_conv_epilogue_modes_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 >> $O
tail -4 $O

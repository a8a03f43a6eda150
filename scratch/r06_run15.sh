#!/bin/bash
# round 6, call 15: GEGLU projections on the register-direct kernel with TWO operand stages (34 KB of LDS, four blocks per CU) vs three stages (three blocks): tests + same-box A/B
mkdir -p gpurun_out/r06
O=gpurun_out/r06/run15_lin_geglu_nst2.txt; : > $O
TFMQ_LIN_GEGLU_NST2=1 timeout 600 python -m pytest tests/test_geglu_fast_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 >> $O
for s in 0 1 0 1 0 1; do
  echo "== TFMQ_LIN_GEGLU_NST2=$s" >> $O
  TFMQ_LIN_GEGLU_NST2=$s TILES=6 ONLY=0,4,7 timeout 300 python scratch/bench_lin.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O

#!/bin/bash
# round 6, call 17: the exact-GELU GEGLU epilogue (TFMQ_OUT_GEGLU_Q8) on two stages / four blocks per CU as well: tests, same-box A/B (TFMQ_GELU_EXACT=1 selects that mode in bench_lin's geglu shapes)
mkdir -p gpurun_out/r06
O=gpurun_out/r06/run17_geglu_exact_nst2.txt; : > $O
timeout 900 python -m pytest tests/test_conv_epilogue_modes_gpu.py tests/test_geglu_fast_gpu.py tests/test_ff_fused_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 >> $O
for s in 0 1 0 1; do
  echo "== TFMQ_LIN_GEGLU_NST2=$s (TFMQ_GELU_EXACT=1)" >> $O
  TFMQ_GELU_EXACT=1 TFMQ_LIN_GEGLU_NST2=$s TILES=6 ONLY=0,4,7 timeout 300 python scratch/bench_lin.py 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O

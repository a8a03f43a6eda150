#!/bin/bash
# round 6, call 14: the register-direct pointwise kernel on 256 x 128 tiles (TFMQ_TILE_DIRECT256 = 9) vs 128 x 128 (6): tests, same-box A/B per SD shape
mkdir -p gpurun_out/r06
O=gpurun_out/r06/run14_lin_m256.txt; : > $O
timeout 900 python -m pytest tests/test_conv_epilogue_modes_gpu.py tests/test_geglu_fast_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 >> $O
for rep in 1 2; do
  TILES=6,9 timeout 300 python scratch/bench_lin.py 2>&1 | grep -v amdgpu.ids >> $O
  TILES=6,9 SHAPES=qkv timeout 300 python scratch/bench_lin.py 2>&1 | grep -v amdgpu.ids >> $O
done
TILES=6,9 SHAPES=modes timeout 300 python scratch/bench_lin.py 2>&1 | grep -v amdgpu.ids >> $O
echo '== fp16-operand slab kernel, TFMQ_SLAB_PP=0 / 1 / 0 / 1' >> $O
for s in 0 1 0 1; do TFMQ_SLAB_PP=$s timeout 300 python scratch/bench_slab_f16.py 2>&1 | grep -v amdgpu.ids >> $O; done
cat $O

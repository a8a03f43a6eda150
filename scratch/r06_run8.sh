#!/bin/bash
# round 6, call 8: ping-pong K loop of the 8-wave slab kernel (TFMQ_SLAB_PP), same-box A/B + bit identity against tile 1 + epilogue-mode tests
mkdir -p gpurun_out/r06
O=gpurun_out/r06/run8_slab_pp.txt; : > $O
echo "== TFMQ_SLAB_PP=1 vs tile 1 (bit identity of outputs and statistics)" >> $O
TFMQ_SLAB_PP=1 TILES=1,5 SHAPES=0,1,3,6,8,9 timeout 600 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids >> $O
for s in 0 1 0 1; do
  echo "== TFMQ_SLAB_PP=$s" >> $O
  TFMQ_SLAB_PP=$s TILES=5 timeout 300 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids >> $O
done
echo "== UP=1 (fused nearest-2x), PP=0 then PP=1" >> $O
for s in 0 1; do UP=1 TFMQ_SLAB_PP=$s TILES=1,5 timeout 300 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids >> $O; done
python -m pytest tests/test_conv_epilogue_modes_gpu.py tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 >> $O
cat $O

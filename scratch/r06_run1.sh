#!/bin/bash
# round 6, call 1: LDS-DMA streaming ubench, slab store-staging A/B, life of a pointwise block
mkdir -p gpurun_out/r06
./scratch/ubench/ldsdma_stream 256 > gpurun_out/r06/ldsdma_stream.txt 2>&1
python -m pytest tests/test_conv_epilogue_modes_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 > gpurun_out/r06/run1_tests.txt
for s in 0 1 0 1; do
  echo "== TFMQ_SLAB_STG=$s" >> gpurun_out/r06/run1_slab_ab.txt
  TFMQ_SLAB_STG=$s TILES=5,7 timeout 300 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06/run1_slab_ab.txt
done
echo "== TFMQ_SLAB_STG=1 vs tile 1 (bit identity)" >> gpurun_out/r06/run1_slab_ab.txt
TFMQ_SLAB_STG=1 TILES=1,5,7 SHAPES=0,3,6,8 timeout 300 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06/run1_slab_ab.txt
timeout 300 python scratch/lat_lin.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/run1_lat_lin.txt
cat gpurun_out/r06/run1_tests.txt

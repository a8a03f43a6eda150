#!/bin/bash
# round 6, call 3: scalar-cache qparam loads + late DMA issue / late residual loads in k_lin_direct (A/B against HEAD's build), slab start stagger
mkdir -p gpurun_out/r06
O=gpurun_out/r06/run3_lin_ab.txt; : > $O
for rep in 1 2; do
for lib in head lin_00 lin_10 lin_01 product; do
  if [ $lib = product ]; then unset TFMQ_LIB_PATH; else export TFMQ_LIB_PATH=$PWD/scratch/ab/libtfmq_$lib.so; fi
  echo "== $lib" >> $O
  TILES=6 timeout 300 python scratch/bench_lin.py 2>&1 | grep -v amdgpu.ids >> $O
  TILES=6 SHAPES=qkv timeout 300 python scratch/bench_lin.py 2>&1 | grep -v amdgpu.ids >> $O
done
done
unset TFMQ_LIB_PATH
O=gpurun_out/r06/run3_slab_stagger.txt; : > $O
for us in 0 20 40 80 0 40; do
  echo "== TFMQ_SLAB_STAGGER_US=$us" >> $O
  TFMQ_SLAB_STAGGER_US=$us TILES=5 timeout 300 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids >> $O
done
python -m pytest tests/test_conv_epilogue_modes_gpu.py tests/test_ff_fused_gpu.py tests/test_row_chain_gpu.py tests/test_hip_kernels.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 > gpurun_out/r06/run3_tests.txt
cat gpurun_out/r06/run3_tests.txt

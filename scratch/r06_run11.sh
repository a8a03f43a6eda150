#!/bin/bash
# round 6, call 11: ping-pong slab loop with the LDS-DMA issue between the last MFMAs of the compute phase (TFMQ_SLAB_PP=2) vs in the load phase (=1) vs lockstep (=0)
mkdir -p gpurun_out/r06
O=gpurun_out/r06/run11_slab_pp2.txt; : > $O
echo "== TFMQ_SLAB_PP=2 vs tile 1 (bit identity of outputs and statistics)" >> $O
TFMQ_SLAB_PP=2 TILES=1,5 SHAPES=0,1,3,6,8,9 timeout 600 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids >> $O
for s in 0 1 2 0 1 2; do
  echo "== TFMQ_SLAB_PP=$s" >> $O
  TFMQ_SLAB_PP=$s TILES=5 timeout 300 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids >> $O
done
echo "== UP=1, PP=2" >> $O
UP=1 TFMQ_SLAB_PP=2 TILES=1,5 timeout 300 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids >> $O
TFMQ_SLAB_PP=2 python -m pytest tests/test_conv_epilogue_modes_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 >> $O
echo "== phases, PP=2 (diagnostics build)" >> $O
TFMQ_LIB_PATH=$PWD/scratch/ab/libtfmq_phase.so TFMQ_SLAB_PP=2 timeout 300 python scratch/phase_slab.py 2>&1 | grep "PING\|blocks" >> $O
cat $O

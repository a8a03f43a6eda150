#!/bin/bash
# round 6, call 2: MFMA || VALU ping-pong ubench; per-block phase times of the pointwise and slab kernels (diagnostics build)
mkdir -p gpurun_out/r06
./scratch/ubench/pingpong_mfma_valu > gpurun_out/r06/pingpong_mfma_valu.txt 2>&1
export TFMQ_LIB_PATH=$PWD/scratch/ab/libtfmq_phase.so
TFMQ_PHASE_PRINT=1 TILES=6 timeout 300 python scratch/bench_lin.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/run2_phase_lin.txt
TFMQ_PHASE_PRINT=1 TILES=6 SHAPES=qkv timeout 300 python scratch/bench_lin.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06/run2_phase_lin.txt
timeout 300 python scratch/phase_slab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/run2_phase_slab.txt
tail -3 gpurun_out/r06/run2_phase_slab.txt

#!/bin/bash
# round 6, call 10: where a step of the ping-pong loops goes (diagnostics build, s_memtime stamps, wave 0 = early half, wave 4 = late half)
mkdir -p gpurun_out/r06
export TFMQ_LIB_PATH=$PWD/scratch/ab/libtfmq_phase.so
O=gpurun_out/r06/run10_pp_phases.txt; : > $O
TFMQ_SLAB_PP=1 timeout 300 python scratch/phase_slab.py 2>&1 | grep -v amdgpu.ids >> $O
TFMQ_SLAB_PP=0 timeout 300 python scratch/phase_slab.py 2>&1 | grep -v amdgpu.ids | grep "K-step\|blocks" >> $O
TFMQ_PHASE_PRINT=1 TFMQ_ATTN_PP=1 BATCH=128 ONLY40=1 timeout 300 python scratch/bench_attn.py 2>&1 | grep -v amdgpu.ids | tail -4 >> $O
cat $O

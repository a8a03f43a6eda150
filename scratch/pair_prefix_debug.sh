#!/bin/bash
# same-box: which switch makes the tiny-UNet pair_prefix forward differ from the materialised pair
cd $GRAFT_REPO_ROOT
python scratch/pair_prefix_debug.py 2>&1 | grep -v amdgpu.ids
TFMQ_CONV_AUTOTUNE=0 python scratch/pair_prefix_debug.py TFMQ_CONV_AUTOTUNE=0 2>&1 | grep -v amdgpu.ids
TFMQ_STREAM_F32=1 python scratch/pair_prefix_debug.py TFMQ_STREAM_F32=1 2>&1 | grep -v amdgpu.ids
TFMQ_NARROW_CONV_GEMM=0 python scratch/pair_prefix_debug.py TFMQ_NARROW_CONV_GEMM=0 2>&1 | grep -v amdgpu.ids
TFMQ_EXACT_FP=1 python scratch/pair_prefix_debug.py TFMQ_EXACT_FP=1 2>&1 | grep -v amdgpu.ids

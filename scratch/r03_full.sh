#!/bin/bash
# round 3: the full GPU suite, then the SD-width exact-vs-fast learned-rounding comparison
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r03; cd $R
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "amdgpu.ids" | tail -15 > gpurun_out/r03/full_gpu_suite.txt
cat gpurun_out/r03/full_gpu_suite.txt | tail -5
timeout 1500 python scratch/sd_masks_exact_vs_fast.py 2000 32 > gpurun_out/r03/sd_masks_log.txt 2>&1
tail -5 gpurun_out/r03/sd_masks_log.txt

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sd_trajectory_gpu.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/run8_f27.txt
PART=a bash scratch/r05_cin256_part.sh

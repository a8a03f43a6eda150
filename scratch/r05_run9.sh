#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_sd_trajectory_gpu.py -q -s 2>&1 | grep -v amdgpu.ids > $O/run9_f27_full.txt; grep "F27\]" $O/run9_f27_full.txt | grep -v print; tail -3 $O/run9_f27_full.txt
PART=b bash scratch/r05_cin256_part.sh

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_sd_trajectory_gpu.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/run7_f27.txt
timeout 3000 python -m pytest tests/test_bench_multi_rank_gpu.py -x -q --durations=6 2>&1 | tail -25 | tee $O/run7_multirank.txt
timeout 600 python -m pytest tests/test_fisher_gpu.py tests/test_recon_units_gpu.py tests/test_configs_r02_gpu.py -x -q 2>&1 | tail -4 | tee $O/run7_recon.txt

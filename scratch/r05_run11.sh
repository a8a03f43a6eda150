#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_calibration_gpu.py tests/test_recon_units_gpu.py tests/test_delta_learning_gpu.py tests/test_fisher_gpu.py tests/test_configs_r02_gpu.py tests/test_ldm_runner_gpu.py -x -q 2>&1 | tail -5 | tee $O/run11_tests.txt
PART=b bash scratch/r05_cin256_part.sh

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_recon_graph_gpu.py tests/test_recon_units_gpu.py tests/test_calibration_gpu.py tests/test_fisher_gpu.py -q 2>&1 | tail -6 | tee $O/run13_tests.txt
bash scratch/r05_run10.sh

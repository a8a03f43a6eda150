#!/bin/bash
# round 5, GPU run 6: the new / changed tests, static s_setprio A/B, the whole -m gpu suite, the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_sd_trajectory_gpu.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/run6_f27.txt
timeout 900 python -m pytest tests/test_fisher_gpu.py tests/test_gemm_bx3_gpu.py -x -q -s 2>&1 | grep -E "\[bf16x3\]|passed|failed|Error|assert" | tail -60 | tee $O/run6_fisher.txt
timeout 2400 python -m pytest tests/test_bench_multi_rank_gpu.py -x -q 2>&1 | tail -15 | tee $O/run6_multirank.txt
echo "== setprio A/B" | tee $O/run6_setprio.txt
for r in 1 2; do for pr in 0 1; do
  echo "-- TFMQ_SETPRIO=$pr" | tee -a $O/run6_setprio.txt
  TFMQ_SETPRIO=$pr timeout 300 python scratch/bench_ff.py 2>&1 | grep -v amdgpu.ids | tee -a $O/run6_setprio.txt
  TFMQ_SETPRIO=$pr timeout 300 python scratch/bench_chain.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/run6_setprio.txt
  TFMQ_SETPRIO=$pr timeout 300 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids | tail -14 | tee -a $O/run6_setprio.txt
done; done
echo "== whole suite" | tee $O/run6_suite.txt
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_bench_multi_rank_gpu.py --deselect tests/test_sd_trajectory_gpu.py 2>&1 | tail -15 | tee -a $O/run6_suite.txt

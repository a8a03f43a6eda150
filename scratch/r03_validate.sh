#!/bin/bash
# round 3: the new test files with their printed numbers, then the reconstruction-GEMM precision modes (GEMM rates, AdaRound iteration times)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r03; cd $R
timeout 1500 python -m pytest tests/test_exact_fp_mode_gpu.py tests/test_ldm_runner_gpu.py tests/test_w8a8_gpu.py tests/test_attention_quant_gpu.py \
   tests/test_recon_precision_gpu.py tests/test_calibration_multi_gpu.py tests/test_configs_r02_gpu.py -q -s -rA 2>&1 | grep -v "^$" > gpurun_out/r03/new_tests.txt
grep -n "passed\|failed\|error" gpurun_out/r03/new_tests.txt | tail -5
grep -n "^\[\|FAILED\|Error\|rel-L2\|yardstick\|bins moved\|max-normalised\|deviation\|median rel" gpurun_out/r03/new_tests.txt | head -80
for m in f32 bf16x3 f16; do
  GEMM_PREC=$m timeout 300 python scratch/bench_gemm_f32.py 2>&1 | grep -v "^$" | tee gpurun_out/r03/bench_gemm_$m.txt | sed 's/(torch.*//' | head -12
  echo "== TFMQ_RECON_GEMM=$m" | tee gpurun_out/r03/bench_recon_$m.txt
  TFMQ_RECON_GEMM=$m timeout 300 python scratch/bench_recon.py 2>&1 | tee -a gpurun_out/r03/bench_recon_$m.txt | tail -4
done

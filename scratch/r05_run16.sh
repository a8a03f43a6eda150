#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_bx3_gpu.py tests/test_exact_fp_mode_gpu.py tests/test_recon_graph_gpu.py -q 2>&1 | tail -6 | tee $O/run16_tests.txt
echo "== bench_gemm short-K form" | tee $O/run16_gemm.txt
for r in 1 2; do for m in 1 3; do
  echo "-- TFMQ_GEMM_BX3=$m" | tee -a $O/run16_gemm.txt
  TFMQ_GEMM_BX3=$m GEMM_PREC=bf16x3 NOLIB=1 ONLY=1,3,4 timeout 300 python scratch/bench_gemm_f32.py 2>&1 | grep -v "amdgpu.ids\|operand" | tee -a $O/run16_gemm.txt
done; done
echo "== cali A/B" | tee $O/run16_cali.txt
for m in 3 1; do
  echo "-- TFMQ_GEMM_BX3=$m" | tee -a $O/run16_cali.txt
  TFMQ_GEMM_BX3=$m timeout 900 python bench.py --workload cali --cali-iters 300 --cali-samples 32 --cali-groups 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['calibration']['adaround_iterations_per_s'], j['roofline']['achieved'])" | tee -a $O/run16_cali.txt
done
bash scratch/r05_pmc_gemm_bx3.sh > $O/run16_pmc.log 2>&1; tail -30 $O/run16_pmc.log

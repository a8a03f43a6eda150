#!/bin/bash
# round 5, GPU run 2: k_gemm_bx3 -- tests, same-box A/B against the in-register split (TFMQ_GEMM_BX3=0), the calibration job with both,
# and the rocprofv3 summary of a calibration slice (a 64x64-level ResBlock + SpatialTransformer, 1000 iterations per unit).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
echo "== tests" | tee $O/run2_tests.txt
timeout 900 python -m pytest tests/test_gemm_bx3_gpu.py tests/test_gemm_f32_gpu.py tests/test_recon_precision_gpu.py -x -q -s 2>&1 | tail -40 | tee -a $O/run2_tests.txt
echo "== bench_gemm" | tee $O/run2_gemm.txt
for r in 1 2; do
for bx in 1 0; do
  echo "-- TFMQ_GEMM_BX3=$bx" | tee -a $O/run2_gemm.txt
  TFMQ_GEMM_BX3=$bx GEMM_PREC=bf16x3 timeout 300 python scratch/bench_gemm_f32.py 2>&1 | grep -v amdgpu.ids | tee -a $O/run2_gemm.txt
done
done
echo "== cali workload A/B (2 groups x 32, 300 iterations per unit)" | tee $O/run2_cali.txt
for bx in 1 0; do
  echo "-- TFMQ_GEMM_BX3=$bx" | tee -a $O/run2_cali.txt
  TFMQ_GEMM_BX3=$bx timeout 900 python bench.py --workload cali --cali-iters 300 --cali-samples 32 --cali-groups 2 2>$O/run2_cali_bx$bx.err | tee -a $O/run2_cali.txt
done
echo "== rocprofv3 of a calibration slice" | tee -a $O/run2_cali.txt
cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $O/prof_cali -o cali -- python $R/bench.py --workload cali --cali-only model.input_blocks.1 --cali-iters 1000 --cali-samples 32 --cali-groups 2 2>$O/run2_prof.err | tee -a $O/run2_cali.txt
cd $R
find $O/prof_cali -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/r05_cali_sd_kernel_stats.csv
find $O/prof_cali -name "*agent_info.csv" | head -1 | xargs -I{} cp {} $O/r05_agent_info.csv
rm -rf $O/prof_cali
head -12 $O/r05_cali_sd_kernel_stats.csv

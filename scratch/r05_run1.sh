#!/bin/bash
# round 5, GPU run 1: A/B of (a) the d = 40 attention with the P V accumulators in named AccVGPRs, (b) GEMM-epilogue files without packed-fp32
# VALU, (c) 128 x 128 tiles for the bf16x3 reconstruction GEMM; + this box's sampling baseline.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
echo "== attention tests, ACC=1" | tee $O/run1_attn.txt
TFMQ_ATTN_ACC=1 timeout 600 python -m pytest tests/test_attention_f16_gpu.py -x -q 2>&1 | tail -5 | tee -a $O/run1_attn.txt
for r in 1 2 3; do
  for acc in 0 1; do
    echo "-- round $r ACC=$acc" | tee -a $O/run1_attn.txt
    TFMQ_ATTN_ACC=$acc BATCH=128 ONLY40=1 timeout 300 python scratch/bench_attn.py 2>&1 | tail -1 | tee -a $O/run1_attn.txt
  done
done
echo "== ff_fused: packed fp32 vs scalar fp32 epilogues" | tee $O/run1_nopk.txt
for r in 1 2; do
  for lib in tfmq-dm_amd/libtfmq_hip.so scratch/ab/libtfmq_nopk.so; do
    echo "-- round $r $lib" | tee -a $O/run1_nopk.txt
    TFMQ_LIB_PATH=$R/$lib timeout 300 python scratch/bench_ff.py 2>&1 | grep -v amdgpu.ids | tee -a $O/run1_nopk.txt
    TFMQ_LIB_PATH=$R/$lib TILES=6 ONLY=0,4,7,1,5,6 timeout 300 python scratch/bench_lin.py 2>&1 | grep -v amdgpu.ids | tee -a $O/run1_nopk.txt
    TFMQ_LIB_PATH=$R/$lib timeout 300 python scratch/bench_chain.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $O/run1_nopk.txt
  done
done
echo "== bf16x3 GEMM: 128x64 (1x2 MFMA tiles per wave) vs 128x128 (2x2)" | tee $O/run1_gemm.txt
echo "-- 128x64" | tee -a $O/run1_gemm.txt
GEMM_PREC=bf16x3 timeout 300 python scratch/bench_gemm_f32.py 2>&1 | grep -v amdgpu.ids | tee -a $O/run1_gemm.txt
echo "-- 128x128 (TFMQ_GEMM_BN128=1)" | tee -a $O/run1_gemm.txt
TFMQ_GEMM_BN128=1 GEMM_PREC=bf16x3 timeout 300 python scratch/bench_gemm_f32.py 2>&1 | grep -v amdgpu.ids | tee -a $O/run1_gemm.txt
echo "== sampling baseline of this box" | tee $O/run1_bench.txt
timeout 900 python bench.py --no-cpu-baseline --no-cali-leg --steps 2 --warmup 1 2>$O/run1_bench.err | tee -a $O/run1_bench.txt
tail -3 $O/run1_bench.err

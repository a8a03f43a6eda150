#!/bin/bash
# round 5: timing ablations of k_gemm_bx3 (results of the ablated builds are garbage, times are not).
#   CPU side:  bash scratch/r05_gemm_abl.sh build      GPU side:  bash scratch/r05_gemm_abl.sh run
cd $(dirname $0)/..
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math"
if [ "$1" = build ]; then
  mkdir -p scratch/ab
  rest=$(ls tfmq-dm_amd/build/*.o | grep -v gemm_f32_mfma.o)
  build() { /opt/rocm/bin/hipcc $FL $2 -c tfmq-dm_amd/csrc/gemm_f32_mfma.hip -o /tmp/gemm_$1.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/ab/libtfmq_gemm_$1.so $rest /tmp/gemm_$1.o -ldl && echo built $1; }
  build noload "-DTFMQ_DBG_GEMM_NO_LOAD"
  build nostore "-DTFMQ_DBG_GEMM_NO_STORE"
  build nopk "-Xclang -target-feature -Xclang -packed-fp32-ops"
  build nomfma "-DTFMQ_DBG_GEMM_NO_MFMA"
else
  O=gpurun_out/r05; mkdir -p $O
  for f in tfmq-dm_amd/libtfmq_hip.so scratch/ab/libtfmq_gemm_*.so; do
    echo "== $f" | tee -a $O/gemm_abl.txt
    TFMQ_LIB_PATH=$PWD/$f GEMM_PREC=bf16x3 NOLIB=1 ONLY=${ONLY:-0,1,2,4,5} timeout 200 python scratch/bench_gemm_f32.py 2>&1 | grep -v "amdgpu.ids\|operand" | tee -a $O/gemm_abl.txt
  done
fi

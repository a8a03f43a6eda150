#!/bin/bash
# round 6, call 6: K-step and epilogue breakdown of the slab kernel (diagnostics build); slab statistics through LDS vs DPP (same-box A/B);
# full GPU suite on the product library; refreshed profiles
mkdir -p gpurun_out/r06
TFMQ_LIB_PATH=$PWD/scratch/ab/libtfmq_phase.so timeout 300 python scratch/phase_slab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/run6_phase_slab.txt
O=gpurun_out/r06/run6_slab_stats_ab.txt; : > $O
for s in 0 1 0 1; do
  echo "== TFMQ_SLAB_STATS_LDS=$s" >> $O
  TFMQ_SLAB_STATS_LDS=$s TILES=5,7 timeout 300 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids >> $O
done
echo "== TFMQ_SLAB_STATS_LDS=1 vs tile 1 (bit identity of outputs and statistics)" >> $O
TFMQ_SLAB_STATS_LDS=1 TILES=1,5,7 SHAPES=0,3,6,8 timeout 300 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids >> $O
python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -a -E "passed|failed|error|FAILED|ERROR|^E  " | tail -20 > gpurun_out/r06/run6_suite.txt
cat gpurun_out/r06/run6_suite.txt
RR=r06 WLS=sd bash scratch/refresh_profiles.sh > gpurun_out/r06/run6_refresh.log 2>&1
tail -3 gpurun_out/r06/run6_refresh.log

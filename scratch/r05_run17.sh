#!/bin/bash
# round 5: hot kernels under other LLVM scheduler strategies, same-box A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O; cd $R
export TMPDIR=/tmp
rm -f $O/run17_sched.txt
for lib in tfmq-dm_amd/libtfmq_hip.so scratch/ab/libtfmq_sched_max-ilp.so scratch/ab/libtfmq_sched_max-memory-clause.so tfmq-dm_amd/libtfmq_hip.so; do
  echo "== $lib" | tee -a $O/run17_sched.txt
  export TFMQ_LIB_PATH=$R/$lib
  BATCH=128 ONLY40=1 timeout 200 python scratch/bench_attn.py 2>&1 | tail -1 | tee -a $O/run17_sched.txt
  timeout 200 python scratch/bench_ff.py 2>&1 | grep -v amdgpu.ids | tee -a $O/run17_sched.txt
  TILES=6 ONLY=0,4,7,1,5,6 timeout 200 python scratch/bench_lin.py 2>&1 | grep -v amdgpu.ids | tee -a $O/run17_sched.txt
  TILES=5 timeout 300 python scratch/bench_slab.py 2>&1 | grep -v amdgpu.ids | tee -a $O/run17_sched.txt
  timeout 200 python scratch/bench_chain.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/run17_sched.txt
done

#!/bin/bash
# round 6, call 6: K-step and epilogue breakdown of the slab kernel (diagnostics build); full GPU suite on the product library; refreshed profiles
mkdir -p gpurun_out/r06
TFMQ_LIB_PATH=$PWD/scratch/ab/libtfmq_phase.so timeout 300 python scratch/phase_slab.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/run6_phase_slab.txt
python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -a -E "passed|failed|error|FAILED|ERROR|^E  " | tail -20 > gpurun_out/r06/run6_suite.txt
cat gpurun_out/r06/run6_suite.txt
RR=r06 WLS=sd bash scratch/refresh_profiles.sh > gpurun_out/r06/run6_refresh.log 2>&1
tail -3 gpurun_out/r06/run6_refresh.log

#!/bin/bash
# round 6, call 4: where a wave's K-step of k_lin_direct goes (diagnostics build with s_memtime stamps)
mkdir -p gpurun_out/r06
export TFMQ_LIB_PATH=$PWD/scratch/ab/libtfmq_phase.so
O=gpurun_out/r06/run4_kstep_lin.txt
TFMQ_PHASE_PRINT=1 TILES=6 timeout 300 python scratch/bench_lin.py 2>&1 | grep "K-step\|start->" | awk '{k=$2 $3 $4 $5; c[k]++; if (c[k]==3 || c[k]==4) print}' > $O
TFMQ_PHASE_PRINT=1 TILES=6 SHAPES=qkv timeout 300 python scratch/bench_lin.py 2>&1 | grep "K-step\|start->" | awk '{k=$2 $3 $4 $5; c[k]++; if (c[k]==3 || c[k]==4) print}' >> $O
cat $O | head -50

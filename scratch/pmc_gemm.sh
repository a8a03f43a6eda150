#!/bin/bash
# usage (on the GPU box): bash scratch/pmc_gemm.sh  -> gpurun_out/r01/pmc_gemm_<pass>.csv
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r01; cd /tmp; export TMPDIR=/tmp
i=0
for c in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d /tmp/pg$i -- python $R/scratch/pmc_gemm.py > /tmp/pg$i.log 2>&1
  f=$(ls /tmp/pg$i/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep "k_gemm_f32_mfma" $f | tail -40 > $R/gpurun_out/r01/pmc_gemm_$i.csv || tail -5 /tmp/pg$i.log
done
head -1 $f > $R/gpurun_out/r01/pmc_gemm_header.csv

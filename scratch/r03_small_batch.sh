#!/bin/bash
# where a UNet(2) forward (1 image / GPU under guidance) spends its time: kernel trace of the timed samplings at --batch 1
RR=${RR:-r03}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$RR; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for b in ${BATCHES:-1 4}; do
  rm -rf /tmp/tr_b$b
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_b$b -- python $R/bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-cali-leg > $O/small_b$b.log 2> $O/small_b$b.err
  grep '^{"metric"' $O/small_b$b.log | cut -c1-160
  t=$(ls /tmp/tr_b$b/*/*kernel_trace.csv | head -1)
  python $R/scratch/trace_gaps.py $t 24000 > $O/${RR}_trace_family_shares_sd_b$b.txt 2>&1
  head -16 $O/${RR}_trace_family_shares_sd_b$b.txt
done

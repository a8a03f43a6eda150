#!/bin/bash
# On the GPU box (gpurun): PMC traffic passes, then the bench lines under rocprofv3 --kernel-trace --stats.
# Outputs land in gpurun_out/$RR/final/ ; copy them to profiles/ afterwards (see scratch/README.md).  RR = round tag (default r02).
RR=${RR:-r02}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$RR/final; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for wl in ${WLS:-sd cifar}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${wl}_$c -- python $R/scratch/pmc_forward.py $wl > $O/pmc_${wl}_$c.log 2>&1
  done
  f=$(ls /tmp/pmc_${wl}_FETCH_SIZE/*/*counter_collection.csv | head -1); w=$(ls /tmp/pmc_${wl}_WRITE_SIZE/*/*counter_collection.csv | head -1)
  python $R/scratch/make_traffic_json.py $f $w $O/${RR}_traffic_$wl.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over 'python scratch/pmc_forward.py $wl': 3 eager UNet forwards of the $wl bench workload at its default batch (only dispatches after the script's marker kernel are kept, i.e. engine set-up and calibration are excluded; counter collection segfaults on the hipGraph replay of bench.py itself). Counter unit = KiB; gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x -> doubled. hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024." > $O/traffic_$wl.txt 2>&1
  cp $O/${RR}_traffic_$wl.json $R/profiles/${RR}_traffic_$wl.json
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$wl -- python $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-cali-leg > $O/bench_$wl.log 2> $O/bench_$wl.err
  s=$(ls /tmp/st_$wl/*/*kernel_stats.csv | head -1); cp $s $O/${RR}_bench_${wl}_kernel_stats.csv
  # (the profiled command leaves out the legs outside the timed region -- CPU baseline, batch sweep, calibration legs --
  # so that the per-kernel averages are those of the sampling forwards)
  # the committed bench line is a plain run (rocprofv3 costs 2-3 %), with the fresh traffic file in place
  python $R/bench.py --workload $wl --steps 2 --warmup 1 2> $O/plain_$wl.err | grep '^{"metric"' | tail -1 > $O/${RR}_bench_line_$wl.json
  a=$(ls /tmp/st_$wl/*/*agent_info.csv | head -1); cp $a $O/${RR}_agent_info.csv
done
ls -la $O | head -30

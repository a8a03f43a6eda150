#!/bin/bash
# final state: whole GPU suite, then the default bench line
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -a -E "passed|failed|error|FAILED|ERROR|^E  " | tail -40 > gpurun_out/r05/run21_suite.txt
cat gpurun_out/r05/run21_suite.txt
timeout 420 python bench.py > gpurun_out/r05/run21_bench.json 2> gpurun_out/r05/run21_bench.err
echo "bench rc $?"
head -c 600 gpurun_out/r05/run21_bench.json

#!/bin/bash
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -a -E "passed|failed|error|FAILED|ERROR|^E  " | tail -40 > gpurun_out/r05/run22_suite.txt
cat gpurun_out/r05/run22_suite.txt

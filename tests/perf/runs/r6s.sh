#!/bin/bash
mkdir -p gpurun_out/r6s
timeout 1500 python -m pytest tests -q -m gpu --timeout=300 --durations=8 > gpurun_out/r6s/gpu_suite.log 2>&1
tail -3 gpurun_out/r6s/gpu_suite.log
timeout 900 python bench.py > gpurun_out/r6s/bench.log 2>&1
tail -c 1500 gpurun_out/r6s/bench.log

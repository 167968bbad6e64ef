#!/bin/bash
mkdir -p gpurun_out/r6u
timeout 1500 python -m pytest tests -q -m gpu --timeout=300 --durations=8 > gpurun_out/r6u/gpu_suite.log 2>&1
tail -3 gpurun_out/r6u/gpu_suite.log

#!/bin/bash
# final default bench line (HLLL leg behind --hlll)
mkdir -p gpurun_out/r4r
( time timeout 270 python bench.py > gpurun_out/r4r/bench.log 2> gpurun_out/r4r/bench.err ) 2>&1 | tail -4
echo "bench rc=$?"
tail -c 600 gpurun_out/r4r/bench.log

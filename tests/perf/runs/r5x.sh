#!/bin/bash
# round 5, call x: PMC passes over the enumeration walk (the kernel without scratch memory)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 500 bash tests/perf/enum_pmc.sh > gpurun_out/r5x_summary.txt 2>&1; echo "rc=$?"; cat gpurun_out/r5x_summary.txt | cut -c1-600
mkdir -p gpurun_out/r5x
for p in pmc1 pmc2; do f=$(find gpurun_out/exp/$p -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Correlation_Id|enum_phase_kernel" "$f" > gpurun_out/r5x/$p.csv; done
ls -la gpurun_out/r5x

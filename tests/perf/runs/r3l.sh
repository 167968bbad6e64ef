set -x
bash tests/perf/enum_pmc.sh > gpurun_out/exp_pmc.txt 2>&1
mkdir -p gpurun_out/r3l; cp gpurun_out/exp_pmc.txt gpurun_out/r3l/pmc_summary.txt
for p in pmc1 pmc2; do f=$(find gpurun_out/exp/$p -name "*counter_collection.csv" | head -1); grep -E "Dispatch_Id|enum_phase_kernel" $f > gpurun_out/r3l/${p}_phase.csv; done
rm -rf gpurun_out/exp
cat gpurun_out/r3l/pmc_summary.txt | tail -8

#!/bin/bash
# round 4, call l (after the final suite run; no product code changes): SQ counters of the round-4 sweep kernel,
# config 3's tour with hand-off + in-loop pruning
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4l; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
(cd $R && timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES -f csv -d $O/pmc_sq -- python tests/perf/gso_roof.py --once 8192 > $O/pmc_sq.log 2>&1)
cd $R
python - <<'PY'
import csv,glob
best={}
for f in glob.glob("gpurun_out/r4l/pmc_sq/**/*counter_collection.csv", recursive=True):
    rows=[r for r in csv.DictReader(open(f)) if "gso_sweep2_kernel" in r["Kernel_Name"]]
    # the sweep launch = the dispatch with the largest SQ_WAVE_CYCLES
    byd={}
    for r in rows: byd.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"])
    d=max(byd.values(), key=lambda x:x.get("SQ_WAVE_CYCLES",0))
    print("sweep launch SQ counters:", d)
    open("gpurun_out/r4l/sweep_sq_summary.txt","w").write("\n".join("%s %.6g"%kv for kv in sorted(d.items()))+"\n")
PY
find $O -name "*.db" -delete 2>/dev/null; find $O -type f -size +4M -delete 2>/dev/null
( time timeout 250 python tests/perf/c3_inloop.py ) > $O/c3_inloop.log 2>&1; echo "c3 inloop rc=$?"; tail -4 $O/c3_inloop.log | cut -c1-900

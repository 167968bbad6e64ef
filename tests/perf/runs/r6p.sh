#!/bin/bash
# round 6, call p: PMC passes of the Householder kernels (rows kernel: where do its cycles go; blocked MFMA mode: MFMA counters)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6p; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp
( cd $R && timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -f csv -d $O/p1 -- python tests/perf/hh_bench.py > $O/p1.log 2>&1 )
( cd $R && timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU -f csv -d $O/p2 -- python tests/perf/hh_bench.py > $O/p2.log 2>&1 )
( cd $R && timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $O/p3 -- python tests/perf/hh_bench.py > $O/p3.log 2>&1 )
cd $R
python - <<'PY'
import csv, glob, collections
O="gpurun_out/r6p"
for p in ("p1","p2","p3"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for f in glob.glob("%s/%s/**/*counter_collection.csv"%(O,p), recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][-40:]
            if "hh_" in k:
                acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in acc.items():
        print(p,k,{a:"%.3e"%b for a,b in v.items()})
tail=open(O+"/p1.log").read().strip().split("\n")[-4:]
print("\n".join(tail))
PY

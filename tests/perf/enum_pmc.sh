#!/bin/bash
# two PMC passes over the enumeration-only bench (1 step); summary per kernel name
rm -rf gpurun_out/exp/pmc1 gpurun_out/exp/pmc2; mkdir -p gpurun_out/exp/pmc1 gpurun_out/exp/pmc2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --no-cpu --no-gso --no-tour --no-pmc --no-batch --steps 1 --warmup 0"
(cd $R && rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES -f csv -d gpurun_out/exp/pmc1 -- $B > gpurun_out/exp/pmc1.log 2>&1)
(cd $R && rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH -f csv -d gpurun_out/exp/pmc2 -- $B > gpurun_out/exp/pmc2.log 2>&1)
cd $R
python - <<'PY'
import csv,glob,collections,json
for p in ("pmc1","pmc2"):
    acc=collections.defaultdict(float)
    for f in glob.glob("gpurun_out/exp/%s/**/*counter_collection.csv"%p, recursive=True):
        for r in csv.DictReader(open(f)):
            if "enum_phase_kernel" in r["Kernel_Name"] or "enum_walk_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(p, dict(acc))
    l=[x for x in open("gpurun_out/exp/%s.log"%p) if x.startswith("{")]
    if l:
        d=json.loads(l[-1]); print("  nodes/step", d["value"]*d["ms_per_step"]/1e3)
    else:
        print(open("gpurun_out/exp/%s.log"%p).read()[-800:])
PY

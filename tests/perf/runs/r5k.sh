#!/bin/bash
# round 5, call k: the default bench line (new legs), kernel stats of a short bench under rocprofv3, PMC passes of
# the batched LLL kernel (instructions per iteration, HBM bytes), MFMA counters of the blocked Householder mode
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5k; mkdir -p $O
R=$GRAFT_REPO_ROOT
( time timeout 900 python bench.py ) > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
tail -3 $O/bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r5k/bench.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1])
    r=j.get("roofline") or {}
    print("value %.4g ms/step %.1f roofline frac %s kernel_ms %s traffic/alg %s" % (j["value"], j["ms_per_step"], r.get("frac"), r.get("kernel_ms"), r.get("traffic_over_algorithmic")))
    for k in ("lll_batch","bkz40_strategies_batch","bkz20_batch","hlll_batch","householder","c2_dropin_resident","pruner_regime"):
        print(k, json.dumps(j.get(k))[:520])
    print("tour", json.dumps(j.get("bkz60_tour"))[:400])
PY
cd /tmp
( cd $R && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_bench -- python bench.py --steps 3 --warmup 1 --no-cpu --no-tour --no-pmc --no-batch > $O/prof_bench.log 2>&1 )
B="python $R/tests/perf/lll_bench.py 120 2048 0"
( cd $R && timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -f csv -d $O/lllpmc1 -- $B > $O/lllpmc1.log 2>&1 )
( cd $R && timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU -f csv -d $O/lllpmc2 -- $B > $O/lllpmc2.log 2>&1 )
( cd $R && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/lllpmc3 -- $B > $O/lllpmc3.log 2>&1 )
( cd $R && timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/lllpmc4 -- $B > $O/lllpmc4.log 2>&1 )
( cd $R && timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -f csv -d $O/hhpmc -- python $R/tests/perf/hh_bench.py > $O/hhpmc.log 2>&1 )
cd $R
python - <<'PY'
import csv, glob, collections
O="gpurun_out/r5k"
acc=collections.defaultdict(float); dur=[]
for p in ("lllpmc1","lllpmc2","lllpmc3","lllpmc4"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv"%(O,p), recursive=True):
        for r in csv.DictReader(open(f)):
            if "lll_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"])
with open(O+"/lll_kernel_pmc_summary.txt","w") as out:
    for k in sorted(acc): out.write("%s %g\n"%(k,acc[k]))
    it=2048*184716.0
    if "SQ_ACTIVE_INST_ANY" in acc or "SQ_INSTS_VALU" in acc:
        tot = acc.get("SQ_INSTS_VALU",0)+acc.get("SQ_INSTS_SALU",0)+acc.get("SQ_INSTS_VMEM",0)+acc.get("SQ_INSTS_SMEM",0)+acc.get("SQ_INSTS_LDS",0)+acc.get("SQ_INSTS_BRANCH",0)
        out.write("# per LLL iteration (2048 lattices x 184716 iterations): VALU %.0f SALU %.0f VMEM %.0f LDS %.0f BRANCH %.0f SMEM %.0f = %.0f instructions\n"%(acc.get("SQ_INSTS_VALU",0)/it,acc.get("SQ_INSTS_SALU",0)/it,acc.get("SQ_INSTS_VMEM",0)/it,acc.get("SQ_INSTS_LDS",0)/it,acc.get("SQ_INSTS_BRANCH",0)/it,acc.get("SQ_INSTS_SMEM",0)/it,tot/it))
    if "FETCH_SIZE" in acc:
        out.write("# HBM traffic: 2 x FETCH_SIZE KiB + WRITE_SIZE KiB = %.3e bytes per launch (gfx950 correction of the guide)\n"%((2*acc["FETCH_SIZE"]+acc.get("WRITE_SIZE",0))*1024))
print(open(O+"/lll_kernel_pmc_summary.txt").read())
acc=collections.defaultdict(float)
for f in glob.glob(O+"/hhpmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hh_blocked" in r["Kernel_Name"]:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"])
print("hh_blocked", dict(acc))
PY
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
ls $O/prof_bench/*/ 2>/dev/null | head

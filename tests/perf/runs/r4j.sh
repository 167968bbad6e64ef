#!/bin/bash
# round 4, call j: what bounds the batched LLL kernel (lll_kernel<2>, 2048 x 120-dim)?  Four PMC passes over the
# lll_batch leg; then the default bench line once more (PMC passes now behind every timed leg)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4j; mkdir -p $O
R=$GRAFT_REPO_ROOT
B="python $R/tests/perf/bench_leg.py lll_batch 2048"
cd /tmp
(cd $R && timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES -f csv -d $O/pmc1 -- $B > $O/pmc1.log 2>&1)
(cd $R && timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH -f csv -d $O/pmc2 -- $B > $O/pmc2.log 2>&1)
(cd $R && timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc3 -- $B > $O/pmc3.log 2>&1)
(cd $R && timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc4 -- $B > $O/pmc4.log 2>&1)
cd $R
python - <<'PY'
import csv,glob,collections
O="gpurun_out/r4j"
tot=collections.defaultdict(float)
for p in ("pmc1","pmc2","pmc3","pmc4"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv"%(O,p), recursive=True):
        for r in csv.DictReader(open(f)):
            if "lll_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]]+=float(r["Counter_Value"])
print("lll_kernel<2> PMC totals:", dict(tot))
with open(O+"/lll_pmc_summary.txt","w") as f:
    for k,v in sorted(tot.items()): f.write("%s %.6g\n"%(k,v))
PY
find $O -name "*.db" -delete 2>/dev/null; find $O -type f -size +4M -delete 2>/dev/null
( time FPHIP_BENCH_KEEP_PMC=$R/$O/pmc timeout 900 python bench.py ) > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r4j/bench.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); r=j.get("roofline") or {}
    print("value %.4g roofline frac %s kernel_ms %s traffic/alg %s mirror %s" % (j["value"], r.get("frac"), r.get("kernel_ms"), r.get("traffic_over_algorithmic"), {k:round(v["frac"],3) for k,v in (r.get("by_mirror_width") or {}).items()}))
    print({k:(j.get(k) or {}).get("reductions_per_s", (j.get(k) or {}).get("lattices_per_s")) for k in ("lll_batch","bkz40_strategies_batch","bkz20_batch","hlll_batch")}, (j.get("bkz60_tour") or {}).get("speedup"))
PY

#!/bin/bash
# round 6, call m: the default bench line, and the kernel stats of a short bench under rocprofv3 (the timed
# gso_sweep2 dispatches and the walk kernel's launches extracted from its kernel trace)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6m; mkdir -p $O
R=$GRAFT_REPO_ROOT
( time timeout 700 python bench.py --steps 20 --warmup 2 ) > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
tail -4 $O/bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r6m/bench.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1])
    r=j.get("roofline") or {}
    print("value %.4g ms/step %.1f parity %s roofline frac %s kernel_ms %s traffic/alg %s traffic/model %s" % (j["value"], j["ms_per_step"], j["parity"]["final_norm_equal_to_reference"], r.get("frac"), r.get("kernel_ms"), r.get("traffic_over_algorithmic"), r.get("traffic_over_kernel_model")))
    for k in ("second_headline","lll_batch","bkz40_strategies_batch","bkz20_batch","hlll_batch","householder","c2_dropin_resident"):
        print(k, json.dumps(j.get(k))[:600])
    print("tour", json.dumps(j.get("bkz60_tour"))[:900])
    print("cpu", json.dumps(j.get("cpu_baseline"))[:500])
PY
cd /tmp
( cd $R && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_bench -- python bench.py --steps 3 --warmup 1 --no-cpu --no-tour --no-pmc --no-batch > $O/prof_bench.log 2>&1 )
cd $R
python - <<'PY'
import csv, glob
O="gpurun_out/r6m"
for f in glob.glob(O+"/prof_bench/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    open(O+"/bench_kernel_stats.csv","w").write(open(f).read())
    for r in rows[:8]: print({k:r[k] for k in list(r)[:6]})
for f in glob.glob(O+"/prof_bench/**/*kernel_trace.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    with open(O+"/bench_gso_sweep2_dispatches.csv","w") as out, open(O+"/bench_enum_walk_dispatches.csv","w") as out2:
        out.write("kernel,duration_ms\n"); out2.write("kernel,duration_ms\n")
        for r in rows:
            dur=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
            if "gso_sweep2_kernel" in r["Kernel_Name"] and dur>20: out.write("%s,%.3f\n"%(r["Kernel_Name"][:40],dur))
            if "enum_walk_kernel" in r["Kernel_Name"] and dur>20: out2.write("%s,%.3f\n"%(r["Kernel_Name"][:60],dur))
    print(open(O+"/bench_gso_sweep2_dispatches.csv").read()[:400]); print(open(O+"/bench_enum_walk_dispatches.csv").read()[:500])
PY
tail -2 $O/prof_bench.log | cut -c1-300

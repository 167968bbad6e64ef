#!/bin/bash
# round 4, call h: what the driver runs — the whole GPU suite in one process, smoke, the default bench line — plus
# the kernel trace of the bench and the PMC passes of the sweep kernel for profiles/
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O
R=$GRAFT_REPO_ROOT
( time timeout 1300 python -m pytest tests -x -q -m gpu --durations=25 ) > $O/suite.log 2>&1
echo "suite rc=$?" >> $O/suite.log
tail -4 $O/suite.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
( time FPHIP_BENCH_KEEP_PMC=$R/$O/pmc timeout 900 python bench.py ) > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r4h/bench.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1])
    r=j.get("roofline") or {}
    print("value %.4g roofline frac %s kernel_ms %s traffic/alg %s mirror %s" % (j["value"], r.get("frac"), r.get("kernel_ms"), r.get("traffic_over_algorithmic"), {k:round(v["frac"],3) for k,v in (r.get("by_mirror_width") or {}).items()}))
    for k in ("lll_batch","bkz40_strategies_batch","bkz20_batch","hlll_batch","pruner","bkz60_tour"):
        print(k, json.dumps(j.get(k))[:300])
PY
cd /tmp
( cd $R && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_bench -- python bench.py --steps 3 --warmup 1 --no-cpu --no-tour --no-pmc --no-batch > $O/prof_bench.log 2>&1 )
cd $R
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null

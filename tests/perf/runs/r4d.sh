#!/bin/bash
# round 4, call d: block-parallel slide reduction, pruner / in-loop tests with the final thresholds, the bench
# line with the new legs, kernel stats of the bench under rocprofv3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O
R=$GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_zz_slide_gpu.py tests/test_pruner_gpu.py tests/test_bkzs_gpu.py -q -s -m gpu -k "slide or pruner or prune or inloop or volumes" --durations=8 ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
grep -E "passed|failed|rc=|slide potential|block-parallel|prune\(\) of" $O/tests.log | cut -c1-300 | tail -8
( time timeout 1200 python bench.py ) > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err
tail -4 $O/bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r4d/bench.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1])
    r=j.get("roofline") or {}
    print("value %.4g roofline frac %s kernel_ms %s traffic/alg %s mirror %s" % (j["value"], r.get("frac"), r.get("kernel_ms"), r.get("traffic_over_algorithmic"), {k:v["frac"] for k,v in (r.get("by_mirror_width") or {}).items()}))
    for k in ("lll_batch","bkz40_strategies_batch","bkz20_batch","hlll_batch","pruner"):
        print(k, json.dumps(j.get(k))[:420])
PY
cd /tmp
( cd $R && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_bench -- python bench.py --steps 3 --warmup 1 --no-cpu --no-tour --no-pmc --no-batch > $O/prof_bench.log 2>&1 )
cd $R
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
ls $O/prof_bench/* 2>/dev/null | head

#!/bin/bash
# round 4, call m: the default bench line on a fresh box with nothing before it (call r4j's line was measured right
# after four PMC passes of another kernel), 20 timed steps like round 3's committed line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4m; mkdir -p $O
( time FPHIP_BENCH_KEEP_PMC=$GRAFT_REPO_ROOT/$O/pmc timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2> $O/bench.err
echo "bench rc=$?" >> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r4m/bench.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); r=j.get("roofline") or {}
    print("value %.4g parity %s roofline frac %s kernel_ms %s traffic/alg %s mirror %s" % (j["value"], j["parity"]["final_norm_equal_to_reference"], r.get("frac"), r.get("kernel_ms"), r.get("traffic_over_algorithmic"), {k:round(v["frac"],3) for k,v in (r.get("by_mirror_width") or {}).items()}))
    print({k:(j.get(k) or {}).get("reductions_per_s", (j.get(k) or {}).get("lattices_per_s")) for k in ("lll_batch","bkz40_strategies_batch","bkz20_batch","hlll_batch")}, (j.get("bkz60_tour") or {}).get("speedup"), j["pruner_regime"]["ms_per_call"])
PY

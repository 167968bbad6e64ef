#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5u; mkdir -p $O
python - > /tmp/b.txt <<'PY'
import sys, os
sys.path.insert(0, "tests")
import conftest as C
f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "bkz_q60_b16.json"))
print("[" + "\n".join("[" + " ".join(str(int(x)) for x in row) + "]" for row in f["b_in"]) + "]")
PY
DROPIN_U=1 FPLLL_HIP_CHECK_U=1 timeout 200 oracle/_ref/dropin_driver bkz /tmp/b.txt 16 hip 2> $O/err.log > $O/out.json; echo "rc=$?"; head -12 $O/err.log; grep -c "u b_0" $O/err.log

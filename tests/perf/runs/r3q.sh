set -x
O=gpurun_out/r3q; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_pruner -- python bench.py --regime pruner --no-cpu --no-gso --no-tour --no-pmc --steps 12 --warmup 2 > $O/prof_pruner.log 2>&1 )
cd $R
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
cut -c1-300 $O/prof_pruner.log | tail -2

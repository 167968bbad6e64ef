set -x
O=gpurun_out/r3p; mkdir -p $O
R=$GRAFT_REPO_ROOT
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.log 2> $O/bench.err
cut -c1-300 $O/bench.log | tail -1
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_bench -- python bench.py --steps 5 --warmup 2 --no-cpu --no-tour --no-pmc > $O/prof_bench.log 2>&1 )
B="python $R/bench.py --no-cpu --no-gso --no-tour --no-pmc --steps 1 --warmup 0"
( cd $R && timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES -f csv -d $O/pmc1 -- $B > $O/pmc1.log 2>&1 )
( cd $R && timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH -f csv -d $O/pmc2 -- $B > $O/pmc2.log 2>&1 )
cd $R
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
( time timeout 1200 python -m pytest tests -x -q -m gpu ) > $O/suite.log 2>&1
tail -5 $O/suite.log

set -x
O=gpurun_out/r3i; mkdir -p $O
( time timeout 600 python -m pytest tests/test_dd_gpu.py -x -q -m gpu -s -k "config5_lattice_lll_ladder" ) > $O/tests_c5ladder.log 2>&1
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/suite.log 2>&1
# profiles: kernel stats of the bench (enumeration + GSO legs), and PMC passes of one enumeration step
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_bench -- python bench.py --steps 5 --warmup 2 --no-cpu --no-tour --no-pmc > $O/prof_bench.log 2>&1 )
( cd $R && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_pruner -- python bench.py --regime pruner --no-cpu --no-gso --no-tour --no-pmc --steps 12 --warmup 2 > $O/prof_pruner.log 2>&1 )
B="python $R/bench.py --no-cpu --no-gso --no-tour --no-pmc --steps 1 --warmup 0"
( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES -f csv -d $O/pmc1 -- $B > $O/pmc1.log 2>&1 )
( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH -f csv -d $O/pmc2 -- $B > $O/pmc2.log 2>&1 )
cd $R
# keep only the small csv summaries (the merged directory is capped at 64 MiB)
find $O -name "*.db" -delete 2>/dev/null
find $O -type f -size +8M -delete 2>/dev/null
ls -laR $O | head -60

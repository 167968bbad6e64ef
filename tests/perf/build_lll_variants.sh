#!/bin/bash
# A/B libraries for one-box runs of the slot-mode reduction kernels (the default library must be built first):
#   exp/libOLD.so   -DFPHIP_LLL_STREAM=0: the first generation's ring of single rows instead of the block streams
#   exp/libPROF.so  -DFPHIP_LLL_PROF=1: the block streams with per-phase timers (tests/perf/lll_prof.py)
set -e
cd "$(dirname "$0")/../.."
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-inline-asm"
U="-mllvm -structurizecfg-skip-uniform-regions=1"
OBJ=fplll_amd/lib/obj
build() { # name, files..., then -D flags after --
  name=$1; shift
  files=(); while [ "$1" != "--" ]; do files+=("$1"); shift; done; shift
  mkdir -p exp/obj_$name
  for f in "${files[@]}"; do
    extra="$U"; [ "$f" = bkzs_kernel.hip ] && extra="$U -Xclang -disable-lifetime-markers"; [ "$f" = gso_host.hip ] && extra=""
    hipcc $FL "$@" $extra -c -o exp/obj_$name/$f.o fplll_amd/csrc/$f &
  done
  wait
  objs=""
  for o in $OBJ/*.hip.o; do
    b=$(basename $o .o); skip=0
    for f in "${files[@]}"; do [ "$b" = "$f" ] && skip=1; done
    [ $skip = 0 ] && objs="$objs $o"
  done
  hipcc --offload-arch=gfx950 -fPIC -shared -pthread -o exp/lib$name.so $objs exp/obj_$name/*.hip.o
  ls -la exp/lib$name.so
}
for v in "$@"; do
  case $v in
    OLD) build OLD lll_kernel.hip bkz_kernel.hip bkzs_kernel.hip gso_host.hip -- -DFPHIP_LLL_STREAM=0 ;;
    PROF) build PROF lll_kernel.hip -- -DFPHIP_LLL_PROF=1 ;;
  esac
done

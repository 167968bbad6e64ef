#!/bin/bash
# sweep-kernel geometry variants for one-box A/B runs (tests/perf/runs/r4g.sh): exp/libV*.so
set -e
cd "$(dirname "$0")/../.."
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-inline-asm"
OBJ=fplll_amd/lib/obj
build() { # name, defines
  name=$1; shift
  mkdir -p exp/obj_$name
  hipcc $FL "$@" -c -o exp/obj_$name/gso_sweep2.hip.o fplll_amd/csrc/gso_sweep2.hip &
  hipcc $FL "$@" -c -o exp/obj_$name/gso_host.hip.o fplll_amd/csrc/gso_host.hip &
  wait
  objs=$(ls $OBJ/*.hip.o | grep -v "gso_sweep2.hip.o\|gso_host.hip.o")
  hipcc --offload-arch=gfx950 -fPIC -shared -pthread -o exp/lib$name.so $objs exp/obj_$name/gso_sweep2.hip.o exp/obj_$name/gso_host.hip.o
}
build V0
build V1 -DFPHIP_S2_NQ3_WPS=3
build V2 -DFPHIP_S2_NQ3_WPS=3 -DFPHIP_S2_NQ3_LDS=13312
ls -la exp/*.so

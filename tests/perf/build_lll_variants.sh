#!/bin/bash
# A/B library for one-box runs of the slot-mode reduction kernels: exp/libOLD.so is the same tree built with
# -DFPHIP_LLL_STREAM=0 (the first generation's ring of single rows instead of the block streams of lll_stream.h)
set -e
cd "$(dirname "$0")/../.."
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-inline-asm"
OBJ=fplll_amd/lib/obj
name=OLD
mkdir -p exp/obj_$name
hipcc $FL -DFPHIP_LLL_STREAM=0 -mllvm -structurizecfg-skip-uniform-regions=1 -c -o exp/obj_$name/lll_kernel.hip.o fplll_amd/csrc/lll_kernel.hip &
hipcc $FL -DFPHIP_LLL_STREAM=0 -mllvm -structurizecfg-skip-uniform-regions=1 -c -o exp/obj_$name/bkz_kernel.hip.o fplll_amd/csrc/bkz_kernel.hip &
hipcc $FL -DFPHIP_LLL_STREAM=0 -mllvm -structurizecfg-skip-uniform-regions=1 -Xclang -disable-lifetime-markers -c -o exp/obj_$name/bkzs_kernel.hip.o fplll_amd/csrc/bkzs_kernel.hip &
hipcc $FL -DFPHIP_LLL_STREAM=0 -c -o exp/obj_$name/gso_host.hip.o fplll_amd/csrc/gso_host.hip &
wait
objs=$(ls $OBJ/*.hip.o | grep -v "lll_kernel.hip.o\|bkz_kernel.hip.o\|bkzs_kernel.hip.o\|gso_host.hip.o")
hipcc --offload-arch=gfx950 -fPIC -shared -pthread -o exp/lib$name.so $objs exp/obj_$name/*.hip.o
ls -la exp/lib$name.so

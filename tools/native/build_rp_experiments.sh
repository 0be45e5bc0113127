#!/bin/bash
# tools/native/libina_rpexp.so: the tree's library with gemm_rowpanel.hip / gemm.hip rebuilt under -DINA_RP_EXPERIMENTS (ablation and schedule
# variants of the row-panel GEMM as tile configs 41-53, see csrc/gemm_rowpanel.hip) - a probe library, never loaded by the product
set -e
cd "$(dirname "$0")/../.."
tmp=$(mktemp -d)
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -ffp-contract=fast -Iinternnav_amd/csrc -DINA_RP_EXPERIMENTS"
hipcc $FL -x hip -c internnav_amd/csrc/gemm_rowpanel.hip -o $tmp/rp.o &
hipcc $FL -x hip -c internnav_amd/csrc/gemm.hip -o $tmp/g.o &
wait
objs=$(ls internnav_amd/csrc/build/*.o | grep -v "/gemm_rowpanel.hip.o" | grep -v "/gemm.hip.o")
hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $objs $tmp/rp.o $tmp/g.o -o tools/native/libina_rpexp.so
echo tools/native/libina_rpexp.so

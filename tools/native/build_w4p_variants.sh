#!/bin/bash
# tools/native/build_w4p_variants.sh "rd,dma rd,dma ...": the tree's library with gemm_w4.hip rebuilt for other positions of the A read / DMA piece
# inside a group of 8 MFMAs of tile config 40 -> tools/native/libina_w4p_<rd>_<dma>.so (probe with tools/native/gemm_sweep <lib> <spec>)
set -e
cd "$(dirname "$0")/../.."
objs=$(ls internnav_amd/csrc/build/*.o | grep -v "/gemm_w4.hip.o")
for v in $1; do
  rd=${v%,*}; dma=${v#*,}
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -ffp-contract=fast -Iinternnav_amd/csrc -Iinclude -DW4P_RD_POS=$rd -DW4P_DMA_POS=$dma \
        -x hip -c internnav_amd/csrc/gemm_w4.hip -o /tmp/gemm_w4_${rd}_${dma}.o
  hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $objs /tmp/gemm_w4_${rd}_${dma}.o -o tools/native/libina_w4p_${rd}_${dma}.so
  echo tools/native/libina_w4p_${rd}_${dma}.so
done

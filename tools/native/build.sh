#!/bin/bash
# builds the torch-free native probes (hipcc, gfx950) next to their sources; the binaries travel to the GPU box with the snapshot
set -e
cd "$(dirname "$0")"
for f in gemm_sweep chain_sweep skinny_sweep head3_probe embed3_probe dit_attn_probe attn_probe rowchain_probe issue_cost_probe; do
  hipcc -O2 -std=c++17 --offload-arch=gfx950 $f.cpp -o $f -ldl
done

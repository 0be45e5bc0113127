#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | grep -v "^decode\|windows" > gpurun_out/r03l_bench_attn.log; cat gpurun_out/r03l_bench_attn.log
echo done

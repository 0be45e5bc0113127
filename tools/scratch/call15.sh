#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_attention_wide_gpu.py -x -q 2>&1 | tail -4
timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | grep -v "^decode\|windows" > gpurun_out/r03m_bench_attn.log; head -24 gpurun_out/r03m_bench_attn.log
echo done

#!/bin/bash
mkdir -p gpurun_out
timeout 60 tools/scratch/tr_probe > gpurun_out/r03j_tr_probe.log 2>&1; echo "probe rc $?"; grep "mismatches" gpurun_out/r03j_tr_probe.log
timeout 900 python -m pytest tests/test_attention_wide_gpu.py -x -q -s 2>&1 | tail -60 > gpurun_out/r03j_pytest_attn_wide.log; tail -5 gpurun_out/r03j_pytest_attn_wide.log
timeout 300 python tools/bench_attn.py > gpurun_out/r03j_bench_attn.log 2>&1; cat gpurun_out/r03j_bench_attn.log | grep -v amdgpu.ids
echo done

#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03n_pytest_gpu_tail.log; cat gpurun_out/r03n_pytest_gpu_tail.log
timeout 400 python bench.py > gpurun_out/r03n_bench_n1_dual_b64.json 2> gpurun_out/r03n_bench.err; cut -c1-400 gpurun_out/r03n_bench_n1_dual_b64.json
INA_ATTN_WIDE=0 timeout 400 python bench.py > gpurun_out/r03n_bench_n1_dual_b64_attn16.json 2>> gpurun_out/r03n_bench.err; cut -c1-400 gpurun_out/r03n_bench_n1_dual_b64_attn16.json
echo done

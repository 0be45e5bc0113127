#!/bin/bash
mkdir -p gpurun_out
for cfg in "INA_ATTN_WIDE=0" "INA_ATTN_WIDE=1 INA_ATTN_DEFER=0" "INA_ATTN_WIDE=1 INA_ATTN_DEFER=1"; do
  echo "== $cfg"; env $cfg timeout 300 python -m pytest tests/test_b64_spotcheck_gpu.py -q -s 2>&1 | grep "B=64\|passed\|failed"
done > gpurun_out/r03o_spotcheck_attn_variants.log 2>&1; cat gpurun_out/r03o_spotcheck_attn_variants.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r03o_pytest_gpu_tail.log; cat gpurun_out/r03o_pytest_gpu_tail.log
echo done

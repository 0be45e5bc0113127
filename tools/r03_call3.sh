# round 3, GPU call 3: new System-1 branches (non-async, CFG), pp32 3-stage tile on the K = 384 shapes, fused row-norm A/B in the bench on ONE box
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_nextdit_gpu.py tests/test_navdp_gpu.py tests/test_agent_gpu.py tests/test_ops_gpu.py -q -m gpu -s > $R/gpurun_out/r03c_pytest.log 2>&1
tail -3 $R/gpurun_out/r03c_pytest.log; grep -n "FAILED\|Error\|nextdit cfg\|nextdit plain\|non-async" $R/gpurun_out/r03c_pytest.log | head -20
GEMM_ITERS=40 timeout 300 python tools/bench_gemm.py 22 26 27 25 29 s1 > $R/gpurun_out/r03c_gemm_s1_pp32.log 2>&1
cat $R/gpurun_out/r03c_gemm_s1_pp32.log
timeout 100 python tools/bench_dit_attn.py 2>&1 | tail -1
for v in "" "--fuse-rownorm" "" "--fuse-rownorm"; do
timeout 600 python bench.py --no-cpu-baseline $v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d['config']['calibration'])"
done > $R/gpurun_out/r03c_bench_ab.log 2>&1
cat $R/gpurun_out/r03c_bench_ab.log

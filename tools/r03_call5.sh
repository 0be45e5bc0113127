# round 3, GPU call 5: new tests (prefix cache through the policy, fused-norm GEMM op), schedule / ring-depth experiments in the bench on one box
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_prefix_kv_gpu.py tests/test_vit_cache_gpu.py tests/test_agent_gpu.py tests/test_nextdit_gpu.py tests/test_ops_gpu.py -q -m gpu -s -k "prefix or vit or agent or nextdit or skinny" > $R/gpurun_out/r03e_pytest.log 2>&1
tail -3 $R/gpurun_out/r03e_pytest.log; grep -n "FAILED\|Error" $R/gpurun_out/r03e_pytest.log | head
run() { timeout 600 env $1 python bench.py --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$1 $2]', d['value'], d['ms_per_step'], d['config']['calibration']['gemm_8192_tflops'])"; }
{ run "X=0" ""; run "X=0" "--s1-early-images"; run "INA_SKINNY_DEEP=1" ""; run "INA_SKINNY_DEEP=1" "--s1-early-images"; run "X=0" ""; } > $R/gpurun_out/r03e_bench_experiments.log 2>&1
cat $R/gpurun_out/r03e_bench_experiments.log

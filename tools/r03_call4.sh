# round 3, GPU call 4: full GPU suite on the current tree, decode-chain timing with the fused input norms, bench variants on one box
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
rm -f $R/gpurun_out/qwen_full_drift.txt
timeout 1500 python -m pytest tests -q -m gpu > $R/gpurun_out/r03d_pytest_gpu.log 2>&1
tail -4 $R/gpurun_out/r03d_pytest_gpu.log; grep -n "FAILED\|Error" $R/gpurun_out/r03d_pytest_gpu.log | head -20
timeout 300 python tools/step_breakdown.py > $R/gpurun_out/r03d_step_breakdown.log 2>&1
tail -14 $R/gpurun_out/r03d_step_breakdown.log
for v in "" "--priority decode" "--prefix-kv" "--vit-cache"; do
timeout 600 python bench.py --no-cpu-baseline $v 2>/dev/null > $R/gpurun_out/r03d_bench_tmp.json
python -c "
import json,sys
d=json.loads(open('$R/gpurun_out/r03d_bench_tmp.json').read().strip().splitlines()[-1]); r=d['roofline']; print('[$v]', d['value'], d['ms_per_step'], d['config']['calibration']['gemm_8192_tflops'], r['achieved'], r['whole_step'])"
n=$(echo "$v" | tr -d ' -'); cp $R/gpurun_out/r03d_bench_tmp.json $R/gpurun_out/r03d_bench_n1_dual_b64_${n:-default}.json
done > $R/gpurun_out/r03d_bench_variants.log 2>&1
cat $R/gpurun_out/r03d_bench_variants.log

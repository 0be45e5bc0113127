# round 3, GPU call 2: changed GPU tests (all, no -x), prefix-KV test + bench variant, persistent-launch experiments, fused row-norm in the bench
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 1200 python -m pytest tests/test_prefix_kv_gpu.py tests/test_unet1d_gpu.py tests/test_sft_full_gpu.py tests/test_sft_navdp_gpu.py tests/test_train_ops_gpu.py tests/test_sft_gpu.py tests/test_trainer_gpu.py tests/test_agent_gpu.py tests/test_sft_llm_gpu.py tests/test_nextdit_gpu.py -q -m gpu -s > $R/gpurun_out/r03b_pytest_changed.log 2>&1
tail -4 $R/gpurun_out/r03b_pytest_changed.log
grep -n "FAILED\|Error\|prefix\|bit-equal" $R/gpurun_out/r03b_pytest_changed.log | head -30
for v in 0 1; do
echo "== INA_GEMM_PERSIST=$v"
INA_GEMM_PERSIST=$v GEMM_ITERS=40 timeout 200 python tools/bench_gemm.py 22 26 27 s1 2>&1 | tail -6
done > $R/gpurun_out/r03b_gemm_persist.log 2>&1
cat $R/gpurun_out/r03b_gemm_persist.log
for v in "0 3" "2 3" "3 3" "0 4" "3 4" "4 4"; do
set -- $v
echo "== INA_DIT_PERSIST=$1 INA_DIT_OCC=$2"
INA_DIT_PERSIST=$1 INA_DIT_OCC=$2 timeout 100 python tools/bench_dit_attn.py 2>&1 | tail -1
done > $R/gpurun_out/r03b_dit_attn_variants.log 2>&1
cat $R/gpurun_out/r03b_dit_attn_variants.log
timeout 600 python bench.py --no-cpu-baseline --fuse-rownorm > $R/gpurun_out/r03b_bench_fuse_rownorm.json 2> $R/gpurun_out/r03b_bench.err
head -c 330 $R/gpurun_out/r03b_bench_fuse_rownorm.json; echo
timeout 600 python bench.py --no-cpu-baseline --prefix-kv > $R/gpurun_out/r03b_bench_prefix_kv.json 2>> $R/gpurun_out/r03b_bench.err
head -c 330 $R/gpurun_out/r03b_bench_prefix_kv.json; echo
INA_GEMM_PERSIST=1 INA_DIT_PERSIST=3 timeout 600 python bench.py --no-cpu-baseline --fuse-rownorm > $R/gpurun_out/r03b_bench_persist.json 2>> $R/gpurun_out/r03b_bench.err
head -c 330 $R/gpurun_out/r03b_bench_persist.json; echo
tail -5 $R/gpurun_out/r03b_bench.err

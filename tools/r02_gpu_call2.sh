# round-2 GPU call 2: fused FFN kernel (op test, engine parity, isolated timing), agent-from-config tests, B=64 spot checks, bench
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "dit_ffn" > $R/gpurun_out/r02b_ffn_op.log 2>&1; tail -15 $R/gpurun_out/r02b_ffn_op.log
timeout 600 python -m pytest tests/test_nextdit_gpu.py tests/test_agent_gpu.py tests/test_b64_spotcheck_gpu.py tests/test_navdp_gpu.py -q -s > $R/gpurun_out/r02b_tests.log 2>&1; tail -30 $R/gpurun_out/r02b_tests.log
timeout 300 python tools/bench_ffn.py > $R/gpurun_out/r02b_bench_ffn.log 2>&1; cat $R/gpurun_out/r02b_bench_ffn.log
timeout 600 python bench.py --no-cpu-baseline > $R/gpurun_out/r02b_bench_n1_dual_b64.log 2>&1; tail -3 $R/gpurun_out/r02b_bench_n1_dual_b64.log

# round-2 GPU call 7: dit_ffn with rotated chunk order
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "dit_ffn" > $R/gpurun_out/r02g_ffn_op.log 2>&1; tail -3 $R/gpurun_out/r02g_ffn_op.log
timeout 300 python tools/bench_ffn.py > $R/gpurun_out/r02g_bench_ffn.log 2>&1; cat $R/gpurun_out/r02g_bench_ffn.log

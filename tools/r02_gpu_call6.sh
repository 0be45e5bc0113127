# round-2 GPU call 6: dit_ffn v2 (weights global->VGPR), UNet1D head with fp32 conv outputs
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "dit_ffn" > $R/gpurun_out/r02f_ffn_op.log 2>&1; tail -5 $R/gpurun_out/r02f_ffn_op.log
timeout 300 python tools/bench_ffn.py > $R/gpurun_out/r02f_bench_ffn.log 2>&1; cat $R/gpurun_out/r02f_bench_ffn.log
timeout 900 python -m pytest tests/test_nextdit_gpu.py tests/test_unet1d_gpu.py -q -s > $R/gpurun_out/r02f_tests.log 2>&1; grep -E "passed|failed|unet1d|nextdit latents" $R/gpurun_out/r02f_tests.log | head -20

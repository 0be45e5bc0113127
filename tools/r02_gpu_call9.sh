export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "glds_tile_configs" > $R/gpurun_out/r02j_gemm_cfg_tests.log 2>&1; tail -4 $R/gpurun_out/r02j_gemm_cfg_tests.log
timeout 600 python tools/bench_gemm.py 18 21 23 24 25 > $R/gpurun_out/r02j_gemm_deep_ring_sweep.log 2>&1; cat $R/gpurun_out/r02j_gemm_deep_ring_sweep.log

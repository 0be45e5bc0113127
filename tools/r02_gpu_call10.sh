export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_nextdit_gpu.py tests/test_navdp_gpu.py tests/test_b64_spotcheck_gpu.py tests/test_unet1d_gpu.py -q -k "glds_tile_configs or nextdit or navdp or dinov2 or b64 or unet1d" > $R/gpurun_out/r02m_tests.log 2>&1; tail -4 $R/gpurun_out/r02m_tests.log
timeout 300 python tools/step_breakdown.py 2>&1 | tail -13
timeout 600 python bench.py --no-cpu-baseline > $R/gpurun_out/r02m_bench_n1_dual_b64.log 2>&1; tail -1 $R/gpurun_out/r02m_bench_n1_dual_b64.log | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline --workload navdp_s1 > $R/gpurun_out/r02m_bench_navdp_s1_b64.log 2>&1; tail -1 $R/gpurun_out/r02m_bench_navdp_s1_b64.log | cut -c1-200

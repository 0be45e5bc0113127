# round-2 GPU call 3: ViT cache exactness, scripted agent end-to-end, spot checks, bench from raw frames (both workloads) + kernel trace
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_vit_cache_gpu.py tests/test_agent_gpu.py tests/test_b64_spotcheck_gpu.py tests/test_qwen_gpu.py tests/test_rccl_gpu.py -q -s > $R/gpurun_out/r02c_tests.log 2>&1; tail -25 $R/gpurun_out/r02c_tests.log
timeout 900 python bench.py > $R/gpurun_out/r02c_bench_n1_dual_b64.log 2>&1; tail -2 $R/gpurun_out/r02c_bench_n1_dual_b64.log
timeout 600 python bench.py --no-raw-frames --no-cpu-baseline > $R/gpurun_out/r02c_bench_n1_dual_b64_noraw.log 2>&1; tail -1 $R/gpurun_out/r02c_bench_n1_dual_b64_noraw.log | cut -c1-200
timeout 600 python bench.py --workload navdp_s1 > $R/gpurun_out/r02c_bench_navdp_s1_b64.log 2>&1; tail -1 $R/gpurun_out/r02c_bench_navdp_s1_b64.log | cut -c1-300

export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > $R/gpurun_out/r02q_pytest_gpu.log 2>&1
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/r02q_smoke.log 2>&1
timeout 400 python bench.py --no-cpu-baseline > $R/gpurun_out/r02q_bench_n1_dual_b64.log 2>&1
tail -2 $R/gpurun_out/r02q_pytest_gpu.log; tail -1 $R/gpurun_out/r02q_smoke.log; tail -1 $R/gpurun_out/r02q_bench_n1_dual_b64.log | cut -c1-200

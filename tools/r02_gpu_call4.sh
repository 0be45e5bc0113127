# round-2 GPU call 4: whole -m gpu suite on the current tree + bench (per-kernel roofline) + kernel-trace summaries
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 1500 python -m pytest tests -q -m gpu > $R/gpurun_out/r02d_pytest_gpu.log 2>&1; tail -12 $R/gpurun_out/r02d_pytest_gpu.log
timeout 900 python bench.py > $R/gpurun_out/r02d_bench_n1_dual_b64.log 2>&1; tail -1 $R/gpurun_out/r02d_bench_n1_dual_b64.log | cut -c1-400
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/kt.log 2>&1
python $R/tools/rocprof_summary.py $R/gpurun_out/kt/kt_results.db 45 > $R/gpurun_out/r02d_n1_dual_b64_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/kt
head -12 $R/gpurun_out/r02d_n1_dual_b64_kernel_stats.txt

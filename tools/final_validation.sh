# round-end validation on the GPU box: tests, smoke, both bench workloads, rocprof kernel stats (all outputs -> gpurun_out/)
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r01}
mkdir -p $R/gpurun_out
timeout 600 python -m pytest tests -q -m gpu > $R/gpurun_out/${TAG}_pytest_gpu.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/${TAG}_smoke.log 2>&1
timeout 600 python bench.py > $R/gpurun_out/${TAG}_bench_n1_dual_b64.log 2>&1
timeout 600 python bench.py --workload navdp_s1 > $R/gpurun_out/${TAG}_bench_navdp_s1_b64.log 2>&1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/kt.log 2>&1
python $R/tools/rocprof_summary.py $R/gpurun_out/kt/kt_results.db 45 > $R/gpurun_out/${TAG}_n1_dual_b64_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/kt
for w in s1 s2; do
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$w -o kt -- python $R/tools/profile_phases.py $w 3 > $R/gpurun_out/kt_$w.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/kt_$w/*.db | head -1) 40 > $R/gpurun_out/${TAG}_${w}_3calls_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/kt_$w
done
tail -2 $R/gpurun_out/${TAG}_pytest_gpu.log

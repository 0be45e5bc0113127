set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_dual -o kt -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/kt_dual_bench.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/kt_dual/*.db $R/gpurun_out/kt_dual/*/*.db 2>/dev/null | head -1) 40 > $R/gpurun_out/kt_dual_stats.txt 2>&1
rm -rf $R/gpurun_out/kt_dual

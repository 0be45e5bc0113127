export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_sft -o kt -- python $R/bench_sft.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/kt_sft.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/kt_sft/*.db | head -1) 40 > $R/gpurun_out/r02o_sft_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/kt_sft
tail -1 $R/gpurun_out/kt_sft.log | cut -c1-400
head -50 $R/gpurun_out/r02o_sft_kernel_stats.txt

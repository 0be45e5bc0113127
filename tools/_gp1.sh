set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_dual -o kt -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/kt_dual_bench.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/kt_dual/*.db $R/gpurun_out/kt_dual/*/*.db 2>/dev/null | head -1) 40 > $R/gpurun_out/kt_dual_stats.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_f -o f -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_w -o w -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $R/gpurun_out/pmc_w.log 2>&1
F=$(ls $R/gpurun_out/pmc_f/*.db $R/gpurun_out/pmc_f/*/*.db 2>/dev/null | head -1); W=$(ls $R/gpurun_out/pmc_w/*.db $R/gpurun_out/pmc_w/*/*.db 2>/dev/null | head -1)
python $R/tools/pmc_summary.py $F $W 16 > $R/gpurun_out/pmc_dual.txt 2>&1
rm -rf $R/gpurun_out/kt_dual $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
tail -3 $R/gpurun_out/kt_dual_bench.log

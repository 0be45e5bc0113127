export TMPDIR=/tmp
R=$PWD
cd /tmp
for w in s1 s2; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$w -o kt -- python $R/tools/profile_phases.py $w 3 > $R/gpurun_out/kt_$w.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/kt_$w/*.db | head -1) 40 > $R/gpurun_out/kt_${w}_stats.txt 2>&1
rm -rf $R/gpurun_out/kt_$w
done

# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of every kernel of ONE eager System-2 call over a 7-env micro-batch
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pf -o f -- python $R/tools/profile_phases.py s2 1 > $R/gpurun_out/pf.log 2>&1
timeout 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pw -o w -- python $R/tools/profile_phases.py s2 1 > $R/gpurun_out/pw.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pf/f_results.db $R/gpurun_out/pw/w_results.db 14 > $R/gpurun_out/pmc_s2_traffic.txt 2>&1
rm -rf $R/gpurun_out/pf $R/gpurun_out/pw

export TMPDIR=/tmp
R=$PWD
cd /tmp
rm -f $R/gpurun_out/pmc_gateup.txt
run() { # counter cfg group_m
  timeout 200 rocprofv3 --pmc $1 --kernel-trace -d $R/gpurun_out/pg -o g -- python $R/tools/gemm_one.py 6440 37888 3584 glu $2 5 $3 > /dev/null 2>&1
  echo "== $1 cfg$2 group_m=$3" >> $R/gpurun_out/pmc_gateup.txt
  python $R/tools/pmc_dump.py $R/gpurun_out/pg/g_results.db gemm >> $R/gpurun_out/pmc_gateup.txt 2>&1
  rm -rf $R/gpurun_out/pg
}
run FETCH_SIZE 17 1
run FETCH_SIZE 18 0
run WRITE_SIZE 18 0

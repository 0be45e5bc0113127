export TMPDIR=/tmp
R=$PWD
cd /tmp
rm -f $R/gpurun_out/pmc_k384.txt
for shape in "65536 1536 384 bf16 11" "65536 384 1024 bf16 11"; do
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pg -o g -- python $R/tools/gemm_one.py $shape 5 > /dev/null 2>&1
echo "== $shape" >> $R/gpurun_out/pmc_k384.txt
python $R/tools/pmc_dump.py $R/gpurun_out/pg/g_results.db gemm >> $R/gpurun_out/pmc_k384.txt 2>&1
rm -rf $R/gpurun_out/pg
done

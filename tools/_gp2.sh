export TMPDIR=/tmp
R=$PWD
cd /tmp
for shape in "6440 37888 3584 glu 17" "8192 8192 8192 bf16 17"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/pmc_g -o g -- python $R/tools/gemm_one.py $shape > /dev/null 2>&1
  echo "== $shape" >> $R/gpurun_out/pmc_gemm.txt
  python $R/tools/pmc_dump.py $R/gpurun_out/pmc_g/g_results.db gemm >> $R/gpurun_out/pmc_gemm.txt 2>&1
  rm -rf $R/gpurun_out/pmc_g
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/pmc_g -o g -- python $R/tools/gemm_one.py $shape > /dev/null 2>&1
  python $R/tools/pmc_dump.py $R/gpurun_out/pmc_g/g_results.db gemm >> $R/gpurun_out/pmc_gemm.txt 2>&1
  rm -rf $R/gpurun_out/pmc_g
done

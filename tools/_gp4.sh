export TMPDIR=/tmp
R=$PWD
cd /tmp
rm -f $R/gpurun_out/pmc_da.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/gpurun_out/pmc_g -o g -- python $R/tools/bench_dit_attn.py > /dev/null 2>&1
python $R/tools/pmc_dump.py $R/gpurun_out/pmc_g/g_results.db dit_attn >> $R/gpurun_out/pmc_da.txt 2>&1
rm -rf $R/gpurun_out/pmc_g
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmc_g -o g -- python $R/tools/bench_dit_attn.py > /dev/null 2>&1
python $R/tools/pmc_dump.py $R/gpurun_out/pmc_g/g_results.db dit_attn >> $R/gpurun_out/pmc_da.txt 2>&1
rm -rf $R/gpurun_out/pmc_g
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC --kernel-trace -d $R/gpurun_out/pmc_g -o g -- python $R/tools/bench_dit_attn.py > /dev/null 2>&1
python $R/tools/pmc_dump.py $R/gpurun_out/pmc_g/g_results.db dit_attn >> $R/gpurun_out/pmc_da.txt 2>&1
rm -rf $R/gpurun_out/pmc_g

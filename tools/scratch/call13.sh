#!/bin/bash
R=$(pwd); mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | head -12 > $R/gpurun_out/r03k_bench_attn.log; cat $R/gpurun_out/r03k_bench_attn.log
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pa1 -o a -- python $R/tools/scratch/attn_one.py > $R/gpurun_out/pa1.log 2>&1
python $R/tools/pmc_table.py $(ls $R/gpurun_out/pa1/*.db | head -1) 6 attn > $R/gpurun_out/r03k_pmc_sq_attn_wide.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pa2 -o a -- python $R/tools/scratch/attn_one.py > $R/gpurun_out/pa2.log 2>&1
python $R/tools/pmc_table.py $(ls $R/gpurun_out/pa2/*.db | head -1) 6 attn > $R/gpurun_out/r03k_pmc_lds_attn_wide.txt 2>&1
cat $R/gpurun_out/r03k_pmc_sq_attn_wide.txt $R/gpurun_out/r03k_pmc_lds_attn_wide.txt | cut -c1-400
tail -3 $R/gpurun_out/pa2.log
rm -rf $R/gpurun_out/pa1 $R/gpurun_out/pa2
echo done

# round-2 GPU call 1: parity (full-config drift report) + whole -m gpu suite + SQ PMC on the shipped kernels inside one S2 / one S1 call
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
rm -f $R/gpurun_out/qwen_full_drift.txt
timeout 900 python -m pytest tests/test_qwen_full_gpu.py -x -q -s > $R/gpurun_out/r02a_qwen_full.log 2>&1
tail -5 $R/gpurun_out/r02a_qwen_full.log
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_qwen_full_gpu.py > $R/gpurun_out/r02a_pytest_gpu.log 2>&1
tail -5 $R/gpurun_out/r02a_pytest_gpu.log
cd /tmp
for w in s2 s1; do
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace -d $R/gpurun_out/sq_$w -o sq -- python $R/tools/profile_phases.py $w 1 > $R/gpurun_out/sq_$w.log 2>&1
python $R/tools/pmc_sq_summary.py $(ls $R/gpurun_out/sq_$w/*.db | head -1) 14 > $R/gpurun_out/r02a_pmc_sq_${w}_call.txt 2>&1
rm -rf $R/gpurun_out/sq_$w
done
cat $R/gpurun_out/r02a_pmc_sq_s2_call.txt

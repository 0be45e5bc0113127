# round 3, GPU call 1: changed GPU tests, baseline bench on this box, S1 experiments (fused row-norm GEMM in situ, PMC of the S1 kernels)
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_unet1d_gpu.py tests/test_sft_full_gpu.py tests/test_sft_navdp_gpu.py tests/test_train_ops_gpu.py tests/test_sft_gpu.py tests/test_trainer_gpu.py tests/test_agent_gpu.py tests/test_sft_llm_gpu.py -x -q -m gpu -s > $R/gpurun_out/r03a_pytest_changed.log 2>&1
tail -5 $R/gpurun_out/r03a_pytest_changed.log
timeout 600 python bench.py --no-cpu-baseline > $R/gpurun_out/r03a_bench_n1_dual_b64.json 2> $R/gpurun_out/r03a_bench.err
tail -c 1500 $R/gpurun_out/r03a_bench_n1_dual_b64.json | head -c 600; echo
timeout 300 python tools/step_breakdown.py > $R/gpurun_out/r03a_step_breakdown.log 2>&1
tail -12 $R/gpurun_out/r03a_step_breakdown.log
timeout 200 python tools/bench_rownorm.py > $R/gpurun_out/r03a_bench_rownorm.log 2>&1
cat $R/gpurun_out/r03a_bench_rownorm.log
timeout 200 python tools/bench_s1_variants.py > $R/gpurun_out/r03a_s1_variants.log 2>&1
cat $R/gpurun_out/r03a_s1_variants.log
cd /tmp
rocprofv3 -L > $R/gpurun_out/r03a_counters_list.txt 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pq -o q -- python $R/tools/profile_phases.py s1 1 > $R/gpurun_out/pq.log 2>&1
python $R/tools/pmc_table.py $(ls $R/gpurun_out/pq/*.db | head -1) 12 > $R/gpurun_out/r03a_pmc_sq_s1_call.txt 2>&1
rm -rf $R/gpurun_out/pq
cat $R/gpurun_out/r03a_pmc_sq_s1_call.txt
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum --kernel-trace -d $R/gpurun_out/pt -o t -- python $R/tools/profile_phases.py s1 1 > $R/gpurun_out/pt.log 2>&1
python $R/tools/pmc_table.py $(ls $R/gpurun_out/pt/*.db | head -1) 12 > $R/gpurun_out/r03a_pmc_tcc_s1_call.txt 2>&1
rm -rf $R/gpurun_out/pt
cat $R/gpurun_out/r03a_pmc_tcc_s1_call.txt

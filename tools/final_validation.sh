# round-end validation on the GPU box: tests, smoke, the bench workloads and variants, rocprofv3 kernel stats, PMC traffic (all outputs -> gpurun_out/)
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r03}
mkdir -p $R/gpurun_out
rm -f $R/gpurun_out/qwen_full_drift.txt $R/gpurun_out/sft_full_drift.txt $R/gpurun_out/sft_navdp_drift.txt
timeout 1500 python -m pytest tests -q -m gpu > $R/gpurun_out/${TAG}_pytest_gpu.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/${TAG}_smoke.log 2>&1
timeout 900 python bench.py > $R/gpurun_out/${TAG}_bench_n1_dual_b64.json 2> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --no-cpu-baseline --prefix-kv > $R/gpurun_out/${TAG}_bench_n1_dual_b64_prefixkv.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --workload navdp_s1 > $R/gpurun_out/${TAG}_bench_navdp_s1_b64.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --workload unet1d_s1 > $R/gpurun_out/${TAG}_bench_unet1d_s1_b64.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --workload sft --steps 10 --warmup 2 > $R/gpurun_out/${TAG}_bench_sft.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 300 python tools/step_breakdown.py > $R/gpurun_out/${TAG}_step_breakdown.log 2>&1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/kt.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/kt/*.db | head -1) 45 > $R/gpurun_out/${TAG}_n1_dual_b64_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/kt
for w in s1 s2; do
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$w -o kt -- python $R/tools/profile_phases.py $w 3 > $R/gpurun_out/kt_$w.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/kt_$w/*.db | head -1) 40 > $R/gpurun_out/${TAG}_${w}_3calls_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/kt_$w
done
# HBM-side traffic of one System-2 call (separate FETCH_SIZE / WRITE_SIZE passes, guides/MI355X_MICROARCH.md HBM section)
timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pf -o f -- python $R/tools/profile_phases.py s2 1 > $R/gpurun_out/pf.log 2>&1
timeout 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pw -o w -- python $R/tools/profile_phases.py s2 1 > $R/gpurun_out/pw.log 2>&1
python $R/tools/pmc_summary.py $(ls $R/gpurun_out/pf/*.db | head -1) $(ls $R/gpurun_out/pw/*.db | head -1) 14 > $R/gpurun_out/${TAG}_pmc_hbm_traffic_s2_call.txt 2>&1
rm -rf $R/gpurun_out/pf $R/gpurun_out/pw
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pq -o q -- python $R/tools/profile_phases.py s2 1 > $R/gpurun_out/pq.log 2>&1
python $R/tools/pmc_table.py $(ls $R/gpurun_out/pq/*.db | head -1) 10 > $R/gpurun_out/${TAG}_pmc_sq_s2_call.txt 2>&1
rm -rf $R/gpurun_out/pq
tail -3 $R/gpurun_out/${TAG}_pytest_gpu.log
tail -2 $R/gpurun_out/${TAG}_smoke.log
head -c 400 $R/gpurun_out/${TAG}_bench_n1_dual_b64.json; echo
head -8 $R/gpurun_out/${TAG}_pmc_hbm_traffic_s2_call.txt
tail -6 $R/gpurun_out/${TAG}_bench.err

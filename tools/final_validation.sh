# round-end validation on the GPU box: tests, smoke, the bench workloads and variants, rocprofv3 kernel stats (all outputs -> gpurun_out/)
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r03}
mkdir -p $R/gpurun_out
rm -f $R/gpurun_out/qwen_full_drift.txt $R/gpurun_out/sft_full_drift.txt $R/gpurun_out/sft_navdp_drift.txt
timeout 1500 python -m pytest tests -q -m gpu > $R/gpurun_out/${TAG}_pytest_gpu.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/${TAG}_smoke.log 2>&1
timeout 900 python bench.py > $R/gpurun_out/${TAG}_bench_n1_dual_b64.json 2> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --no-cpu-baseline --no-split-prefill > $R/gpurun_out/${TAG}_bench_n1_dual_b64_joint_prefill.json 2>> $R/gpurun_out/${TAG}_bench.err
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
# torch-free native probes over the C-ABI (seconds each; tools/native/build.sh builds them next to their sources before the snapshot is sent)
cd $R
if [ -x tools/native/chain_sweep ]; then
  timeout 60 tools/native/gemm_sweep internnav_amd/libinternnav_amd.so > $R/gpurun_out/${TAG}_native_gemm_sweep.log 2>&1
  timeout 60 tools/native/chain_sweep internnav_amd/libinternnav_amd.so all part > $R/gpurun_out/${TAG}_native_chain_partitions.log 2>&1
  timeout 20 tools/native/skinny_sweep internnav_amd/libinternnav_amd.so 6 > $R/gpurun_out/${TAG}_native_skinny.log 2>&1
  timeout 20 tools/native/head3_probe internnav_amd/libinternnav_amd.so > $R/gpurun_out/${TAG}_native_head3.log 2>&1
  timeout 20 tools/native/dit_attn_probe internnav_amd/libinternnav_amd.so > $R/gpurun_out/${TAG}_native_dit_attn.log 2>&1
  timeout 20 tools/native/attn_probe internnav_amd/libinternnav_amd.so > $R/gpurun_out/${TAG}_native_attn.log 2>&1
fi
# (PMC passes: tools/pmc_s2_traffic.sh / profiles/r03h_pmc_* - collected before the tile-selection changes of the end of round 3, see profiles/INDEX.md)
tail -3 $R/gpurun_out/${TAG}_pytest_gpu.log
tail -2 $R/gpurun_out/${TAG}_smoke.log
head -c 400 $R/gpurun_out/${TAG}_bench_n1_dual_b64.json; echo
head -12 $R/gpurun_out/${TAG}_step_breakdown.log
tail -6 $R/gpurun_out/${TAG}_bench.err

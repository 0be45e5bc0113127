# round-end validation on the GPU box: tests, smoke, the bench workloads and variants, rocprofv3 kernel stats, PMC passes (all outputs -> gpurun_out/)
#   bash tools/final_validation.sh <tag> [quick]      quick: skip the variant benches and the PMC passes
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r06}
QUICK=${2:-}
mkdir -p $R/gpurun_out
rm -f $R/gpurun_out/qwen_full_drift.txt $R/gpurun_out/qwen_outliers_drift.txt $R/gpurun_out/sft_full_drift.txt $R/gpurun_out/sft_navdp_drift.txt $R/gpurun_out/s1_b64_distribution.txt
timeout 2400 python -m pytest tests -q -m gpu > $R/gpurun_out/${TAG}_pytest_gpu.log 2>&1
for f in qwen_full_drift qwen_outliers_drift sft_full_drift sft_navdp_drift s1_b64_distribution; do cp $R/gpurun_out/$f.txt $R/gpurun_out/${TAG}_$f.txt 2>/dev/null; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/${TAG}_smoke.log 2>&1
timeout 900 python bench.py > $R/gpurun_out/${TAG}_bench_n1_dual_b64.json 2> $R/gpurun_out/${TAG}_bench.err
if [ -z "$QUICK" ]; then
timeout 600 python bench.py --no-cpu-baseline --no-variants --cadence reference > $R/gpurun_out/${TAG}_bench_n1_dual_b64_reference.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --workload s2_only > $R/gpurun_out/${TAG}_bench_s2_only_b7.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --no-cpu-baseline --no-variants --num-history 8 --lookdown --steps 10 > $R/gpurun_out/${TAG}_bench_n1_dual_b64_h8_lookdown.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --no-cpu-baseline --prefix-kv --steps 20 > $R/gpurun_out/${TAG}_bench_n1_dual_b64_prefixkv.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --workload navdp_s1 > $R/gpurun_out/${TAG}_bench_navdp_s1_b64.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --workload unet1d_s1 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_unet1d_s1_b64.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --workload sft --steps 30 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_sft.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --workload sft --steps 30 --warmup 3 --no-cpu-baseline --no-prefetch > $R/gpurun_out/${TAG}_bench_sft_noprefetch.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --workload sft --steps 30 --warmup 3 --no-cpu-baseline --no-priority > $R/gpurun_out/${TAG}_bench_sft_nopriority.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --workload sft --steps 30 --warmup 3 --no-cpu-baseline --no-graph-prefix > $R/gpurun_out/${TAG}_bench_sft_eagerprefix.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --no-cpu-baseline --no-variants --dit-ffn 1024 > $R/gpurun_out/${TAG}_bench_n1_dual_b64_ffn1024.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --no-cpu-baseline --no-variants --no-frag-weights > $R/gpurun_out/${TAG}_bench_n1_dual_b64_no_frag_weights.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --no-cpu-baseline --no-variants --no-row-chain > $R/gpurun_out/${TAG}_bench_n1_dual_b64_no_row_chain.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --no-cpu-baseline --no-variants --no-fuse-decode-rope > $R/gpurun_out/${TAG}_bench_n1_dual_b64_no_fuse_decode_rope.json 2>> $R/gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --workload host_stub --gpus 8 --steps 20 > $R/gpurun_out/${TAG}_bench_host_stub_8ranks.json 2>> $R/gpurun_out/${TAG}_bench.err
fi
timeout 300 python tools/step_breakdown.py > $R/gpurun_out/${TAG}_step_breakdown.log 2>&1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt -o kt -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-variants > $R/gpurun_out/kt.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/kt/*.db | head -1) 45 > $R/gpurun_out/${TAG}_n1_dual_b64_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/kt
for w in s1 s2; do
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$w -o kt -- python $R/tools/profile_phases.py $w 3 > $R/gpurun_out/kt_$w.log 2>&1
python $R/tools/rocprof_summary.py $(ls $R/gpurun_out/kt_$w/*.db | head -1) 40 > $R/gpurun_out/${TAG}_${w}_3calls_kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/kt_$w
done
if [ -z "$QUICK" ]; then
# PMC passes (counters alone, --kernel-trace only): HBM-side traffic and SQ counters of one eager System-2 call (the timed step's launches)
timeout 420 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pf -o f -- python $R/tools/profile_phases.py s2 1 > $R/gpurun_out/pf.log 2>&1
timeout 420 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pw -o w -- python $R/tools/profile_phases.py s2 1 > $R/gpurun_out/pw.log 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/pf/f_results.db $R/gpurun_out/pw/w_results.db 14 > $R/gpurun_out/${TAG}_pmc_hbm_traffic_s2_call.txt 2>&1
rm -rf $R/gpurun_out/pf $R/gpurun_out/pw
for w in s1 s2; do
timeout 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace -d $R/gpurun_out/ps -o s -- python $R/tools/profile_phases.py $w 1 > $R/gpurun_out/ps.log 2>&1
{ python $R/tools/pmc_sq_summary.py $R/gpurun_out/ps/s_results.db 14; python $R/tools/pmc_table.py $R/gpurun_out/ps/s_results.db 14; } > $R/gpurun_out/${TAG}_pmc_sq_${w}_call.txt 2>&1
rm -rf $R/gpurun_out/ps
done
fi
# torch-free native probes over the C-ABI (seconds each). Rebuilt here against the header of THIS tree: a probe compiled before a struct of
# include/internnav_amd.h changed hands the library a mis-laid argument block (a GPU memory fault in round 6)
cd $R
bash tools/native/build.sh > $R/gpurun_out/${TAG}_native_build.log 2>&1
if [ -x tools/native/chain_sweep ]; then
  timeout 90 tools/native/gemm_sweep internnav_amd/libinternnav_amd.so tools/native/specs_r04_rowpanel.txt > $R/gpurun_out/${TAG}_native_rowpanel.log 2>&1
  timeout 90 tools/native/gemm_sweep internnav_amd/libinternnav_amd.so tools/native/specs_r04_w4.txt > $R/gpurun_out/${TAG}_native_w4.log 2>&1
  timeout 90 tools/native/chain_sweep internnav_amd/libinternnav_amd.so tools/native/specs_r04_w4_chain.txt spec > $R/gpurun_out/${TAG}_native_w4_chain.log 2>&1
  timeout 60 tools/native/chain_sweep internnav_amd/libinternnav_amd.so all part > $R/gpurun_out/${TAG}_native_chain_partitions.log 2>&1
  timeout 60 tools/native/chain_sweep internnav_amd/libinternnav_amd.so tools/native/specs_r04_phase_mix.txt spec > $R/gpurun_out/${TAG}_native_phase_mix.log 2>&1
  timeout 30 tools/native/skinny_sweep internnav_amd/libinternnav_amd.so 6 > $R/gpurun_out/${TAG}_native_skinny.log 2>&1
  timeout 20 tools/native/attn_probe internnav_amd/libinternnav_amd.so > $R/gpurun_out/${TAG}_native_attn.log 2>&1
  timeout 120 tools/native/gemm_sweep internnav_amd/libinternnav_amd.so tools/native/specs_r05_w4p.txt > $R/gpurun_out/${TAG}_native_w4p.log 2>&1
  timeout 60 tools/native/dit_attn_probe internnav_amd/libinternnav_amd.so > $R/gpurun_out/${TAG}_native_dit_attn.log 2>&1
  timeout 60 tools/native/rowchain_probe internnav_amd/libinternnav_amd.so 1536 > $R/gpurun_out/${TAG}_native_rowchain.log 2>&1
  timeout 60 tools/native/rowchain_probe internnav_amd/libinternnav_amd.so 1024 > $R/gpurun_out/${TAG}_native_rowchain_ffn1024.log 2>&1
  timeout 60 tools/native/issue_cost_probe > $R/gpurun_out/${TAG}_native_issue_cost.log 2>&1
fi
tail -3 $R/gpurun_out/${TAG}_pytest_gpu.log
tail -2 $R/gpurun_out/${TAG}_smoke.log
head -c 400 $R/gpurun_out/${TAG}_bench_n1_dual_b64.json; echo
head -12 $R/gpurun_out/${TAG}_step_breakdown.log
tail -6 $R/gpurun_out/${TAG}_bench.err

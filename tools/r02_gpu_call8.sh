# round-2 GPU call 8: bench variants (ViT cache, UNet1D head workload, NavDP) + smoke
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 600 python bench.py --vit-cache --no-cpu-baseline > $R/gpurun_out/r02i_bench_n1_dual_b64_vitcache.log 2>&1; tail -1 $R/gpurun_out/r02i_bench_n1_dual_b64_vitcache.log | cut -c1-330
timeout 600 python bench.py --workload unet1d_s1 > $R/gpurun_out/r02i_bench_unet1d_s1_b64.log 2>&1; tail -1 $R/gpurun_out/r02i_bench_unet1d_s1_b64.log | cut -c1-330
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $R/gpurun_out/r02i_smoke.log 2>&1; tail -3 $R/gpurun_out/r02i_smoke.log

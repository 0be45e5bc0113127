# round 3, GPU call 7: decode-chain footprint experiments inside the two-stream step on one box
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
run() { timeout 600 env $1 python bench.py --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$1 $2]', d['value'], d['ms_per_step'], d['config']['calibration']['gemm_8192_tflops'])"; }
{ run "X=0" ""; run "X=0" "--no-fuse-decode-norm"; run "INA_SKINNY_SMALL=1" ""; run "INA_SKINNY_SMALL=1" "--no-fuse-decode-norm"; run "X=0" ""; } > $R/gpurun_out/r03g_bench_decode_footprint.log 2>&1
cat $R/gpurun_out/r03g_bench_decode_footprint.log

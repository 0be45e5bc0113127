# round 3, GPU call 6: stream-priority / overlap-start experiments of the two-stream schedule on one box
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
run() { timeout 600 python bench.py --no-cpu-baseline $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$1]', d['value'], d['ms_per_step'], d['config']['calibration']['gemm_8192_tflops'], d['config'].get('side_stream_priority'))"; }
{ run ""; run "--priority main"; run "--priority main --overlap-at start"; run "--priority side-low"; run "--priority side-low --overlap-at start"; run "--overlap-at start"; } > $R/gpurun_out/r03f_bench_priority.log 2>&1
cat $R/gpurun_out/r03f_bench_priority.log

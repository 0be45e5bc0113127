# round-2 GPU call 5: ConditionalUnet1D head (op test, fixture parity, B=64 spot check + timing)
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests/test_unet1d_gpu.py -q -s > $R/gpurun_out/r02e_unet1d.log 2>&1; tail -25 $R/gpurun_out/r02e_unet1d.log

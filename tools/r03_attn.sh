export TMPDIR=/tmp
echo "== default"; python tools/bench_attn.py 2>/dev/null | head -3
echo "== INA_ATTN_NW8=1"; INA_ATTN_NW8=1 python tools/bench_attn.py 2>/dev/null | head -3
INA_ATTN_NW8=1 python -m pytest tests/test_ops_gpu.py tests/test_qwen_gpu.py -q -m gpu -k "attention or qwen" 2>&1 | tail -3

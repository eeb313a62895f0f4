export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_guided_fused_gpu.py -x -q -m gpu 2>&1 | tail -3
MNV_FUSED_DIAG=1 python tools/guided_bench.py 32 4 2>/dev/null | tail -1
python tools/guided_bench.py 32 10 2>/dev/null | tail -1

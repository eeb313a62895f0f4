export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_guided_fused_gpu.py -x -q -m gpu 2>&1 | tail -25

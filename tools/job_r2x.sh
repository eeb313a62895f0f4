export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_scale_gpu.py -x -q -m gpu -k together 2>&1 | tail -5

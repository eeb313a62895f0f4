export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_scale_gpu.py -x -q -m gpu --durations=5 2>&1 | tail -25

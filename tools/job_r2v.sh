export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_scale_gpu.py -x -q -m gpu -k depth11 --durations=3 2>&1 | tail -8
free -g | head -2

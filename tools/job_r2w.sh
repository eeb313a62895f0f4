export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_guided_fused_gpu.py tests/test_renderer_refine_gpu.py tests/test_scale_gpu.py -x -q -m gpu 2>&1 | tail -6
python tools/guided_bench.py 32 4 2>/dev/null | tail -1

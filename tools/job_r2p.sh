export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_renderer_refine_gpu.py tests/test_scale_gpu.py tests/test_guided_fused_gpu.py tests/test_cli_gpu.py -x -q -m gpu -k "visit or prune or refine or guided or scale or cli or tracker" 2>&1 | tail -12

export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
timeout 900 python -m pytest tests/test_guided_fused_gpu.py tests/test_scale_gpu.py tests/test_renderer_refine_gpu.py tests/test_guided_gpu.py tests/test_mlp_gpu.py -x -q -m gpu 2>&1 | tail -8
for b in 16 32 40 48 56 64; do MNV_FUSED_BATCH_MIN=$b python tools/guided_bench.py 32 4 2>/dev/null | tee -a $O/guided_bench.jsonl; done
python tools/guided_bench.py 128 4 2>/dev/null | tee -a $O/guided_bench.jsonl
python tools/guided_bench.py 32 10 2>/dev/null | tee -a $O/guided_bench.jsonl

export TMPDIR=/tmp
O=gpurun_out/r2k; mkdir -p $O
timeout 900 python -m pytest tests/test_guided_fused_gpu.py -x -q -m gpu 2>&1 | tail -6
for b in 64 48 32; do MNV_FUSED_DIAG=1 MNV_FUSED_BATCH_MIN=$b timeout 300 python tools/guided_bench.py 32 4 2>/dev/null | tee -a $O/guided_bench.jsonl; done
timeout 300 python tools/guided_bench.py 32 10 2>/dev/null | tee -a $O/guided_bench.jsonl
timeout 300 python tools/guided_bench.py 128 4 2>/dev/null | tee -a $O/guided_bench.jsonl

export TMPDIR=/tmp
O=gpurun_out/r2j; mkdir -p $O
timeout 900 python -m pytest tests/test_guided_fused_gpu.py -x -q -m gpu 2>&1 | tail -4
for b in 8 16 24 32; do MNV_FUSED_DIAG=1 MNV_FUSED_BATCH_MIN=$b python tools/guided_bench.py 32 4 2>/dev/null | tee -a $O/guided_bench.jsonl; done
MNV_FUSED_BATCH_MIN=16 python tools/guided_bench.py 32 10 2>/dev/null | tee -a $O/guided_bench.jsonl

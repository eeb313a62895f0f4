export TMPDIR=/tmp
MNV_FUSED_DIAG=1 python tools/guided_bench.py 32 4 2>/dev/null | tail -1

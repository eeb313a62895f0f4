export TMPDIR=/tmp
O=gpurun_out/r2i; mkdir -p $O
for b in 1 4 8 16; do MNV_FUSED_DIAG=1 MNV_FUSED_BATCH_MIN=$b python tools/guided_bench.py 32 4 2>/dev/null | tee -a $O/guided_bench.jsonl; done
for b in 8 16 32; do MNV_LIB_PATH=$PWD/variants/libmnv_fw2.so MNV_FUSED_DIAG=1 MNV_FUSED_BATCH_MIN=$b python tools/guided_bench.py 32 4 2>/dev/null | tee -a $O/guided_bench.jsonl; done

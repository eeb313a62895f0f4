export TMPDIR=/tmp
for b in 2 3 4 6; do MNV_BLOCKS_PER_CU=$b python tools/sample_march_time.py 2>/dev/null | tail -1; done

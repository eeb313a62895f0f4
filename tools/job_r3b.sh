export TMPDIR=/tmp
for v in w7 w6; do
  export MNV_LIB_PATH=$PWD/variants/libmnv_$v.so
  B=7; [ $v = w6 ] && B=6
  MNV_BLOCKS_PER_CU=$B python bench.py --steps 6 --warmup 2 --cpu-poses 2 --frame-streams 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['avg_launch_ms'], 'bad', d['parity']['pixels_not_bit_identical'])"
done

export TMPDIR=/tmp
for v in base rp1 rp2 rp3; do
  if [ $v = base ]; then unset MNV_LIB_PATH; else export MNV_LIB_PATH=$PWD/variants/libmnv_$v.so; fi
  python bench.py --steps 6 --warmup 2 --cpu-poses 2 --frame-streams 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['avg_launch_ms'], 'bad', d['parity']['pixels_not_bit_identical'])"
done

export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "ref_layout or trackers_and_visited or stress or depth_extremes or cfg2_full or goldens or binding" 2>&1 | tail -4
for v in rc0 rc6 base rc8 rc9; do
  if [ $v = base ]; then unset MNV_LIB_PATH; else export MNV_LIB_PATH=$PWD/variants/libmnv_$v.so; fi
  python bench.py --kernel ref_layout --steps 3 --warmup 1 --cpu-poses 2 --laps 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['avg_launch_ms'], 'bad', d['parity']['pixels_not_bit_identical'])"
done

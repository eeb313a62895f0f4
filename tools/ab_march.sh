for i in 1 2 3; do
 for lib in head new; do
  if [ $lib = head ]; then export MNV_LIB_PATH=$PWD/variants/libmnv_head.so; else unset MNV_LIB_PATH; fi
  for wl in cfg2 cfg3; do
   python bench.py --no-extras --no-cpu-baseline --workload $wl --laps $([ $wl = cfg2 ] && echo 4 || echo 1) --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib $wl batch', d['value'], d['ms_per_step'])"
  done
  python bench.py --no-extras --no-cpu-baseline --per-frame --frame-streams 1 --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib cfg2 per-frame 1 stream', d['value'], d['ms_per_step'])"
  python bench.py --no-extras --no-cpu-baseline --per-frame --frame-streams 3 --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib cfg2 per-frame 3 streams', d['value'], d['ms_per_step'])"
 done
done

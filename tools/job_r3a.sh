export TMPDIR=/tmp
python bench.py --steps 6 --warmup 2 --cpu-poses 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['per_frame']['value'], d['parity'])"

export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_guided_gpu.py tests/test_renderer_refine_gpu.py -x -q -m gpu 2>&1 | tail -5
python bench.py --steps 6 --warmup 2 --cpu-poses 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_ms'], d['per_frame']['value'], d['parity'])"
python bench.py --workload cfg3 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3', d['value'], d['per_frame']['value'])"

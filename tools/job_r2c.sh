export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
timeout 1200 python -m pytest tests/test_renderer_refine_gpu.py tests/test_cli_gpu.py tests/test_bench_multirank_gpu.py -x -q -m gpu 2>&1 | tail -15
python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err; cat $O/bench.json; tail -3 $O/bench.err

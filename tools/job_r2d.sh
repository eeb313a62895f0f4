export TMPDIR=/tmp
O=gpurun_out/r2d; mkdir -p $O
timeout 1200 python -m pytest tests/test_comm_gpu.py tests/test_cli_gpu.py -x -q -m gpu 2>&1 | tail -25
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8

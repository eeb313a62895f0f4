export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
for k in 1 2 3 4 6; do python bench.py --per-frame --frame-streams $k --steps 4 --warmup 1 --no-cpu-baseline 2>>$O/err.txt | tee -a $O/perframe_streams.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $k', d['value'], d['ms_per_step'])"; done
python bench.py --per-frame --frame-streams 3 --steps 2 --warmup 1 --cpu-poses 16 > $O/perframe_parity.json 2>>$O/err.txt; python -c "import json; d=json.load(open('$O/perframe_parity.json')); print(d['value'], d['parity'])"
python bench.py --workload cfg3 --per-frame --frame-streams 3 --steps 4 --warmup 1 --no-cpu-baseline 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 pf', d['value'], d['ms_per_step'])"
python bench.py --workload cfg3 --per-frame --frame-streams 1 --steps 4 --warmup 1 --no-cpu-baseline 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3 pf1', d['value'], d['ms_per_step'])"
tail -3 $O/err.txt

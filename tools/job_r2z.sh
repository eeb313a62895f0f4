export TMPDIR=/tmp
for w in cfg2 cfg3; do echo "== $w"; MNV_STATS=1 python bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --frame-streams 0 --laps 1 2>&1 | grep "mnv stats"; done

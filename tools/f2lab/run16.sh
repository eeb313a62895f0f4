#!/bin/bash
cd "$(dirname "$0")/../.."
R=r04
for wl in cfg3 cfg4; do
  bash tools/prof_traffic.sh ${R}_${wl} --workload $wl --laps 1 > /dev/null 2>&1
  python3 tools/make_traffic_json.py gpurun_out/${R}_${wl}/summary.txt 16 "--workload $wl --laps 1" > gpurun_out/${R}_${wl}/traffic.json
  cp gpurun_out/${R}_${wl}/traffic.json profiles/${R}_traffic_${wl}.json
  grep "mean of all" gpurun_out/${R}_${wl}/summary.txt
done
cp gpurun_out/${R}_batch/traffic.json profiles/${R}_traffic.json 2>/dev/null
python3 bench.py --steps 10 --warmup 2 > gpurun_out/${R}_bench_n1.json 2> /dev/null
python3 -c "
import json;d=json.load(open('gpurun_out/r04_bench_n1.json'))
print(d['value'], d['roofline'])
for k in ('cfg3','cfg4_n1'): print(k, d[k]['value'], d[k]['roofline'])
print(d['cfg5']); print(d['ref_layout']); print(d['cpu_baseline'])"

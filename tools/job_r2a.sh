export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
python bench.py --per-frame --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_perframe.json 2> $O/bench_perframe.err
python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_batch.json 2> $O/bench_batch.err
for p in 0 3 7; do MNV_TIMELINE=$O/tl_cfg2_p$p.bin python tools/timeline.py --pose $p >> $O/timeline.jsonl 2>> $O/timeline.err; done
MNV_TIMELINE=$O/tl_cfg3_p3.bin python tools/timeline.py --workload cfg3 --pose 3 >> $O/timeline.jsonl 2>> $O/timeline.err
MNV_TIMELINE=$O/tl_cfg2_b4.bin python tools/timeline.py --pose 0 --frames 4 >> $O/timeline.jsonl 2>> $O/timeline.err
python tools/frame_latency.py > $O/frame_latency.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pf -- python3 bench.py --per-frame --steps 2 --warmup 1 --no-cpu-baseline > $O/trace_pf.log 2>&1
rm -f $O/*.bin.keep; ls -la $O
cat $O/bench_perframe.json $O/timeline.jsonl $O/frame_latency.txt

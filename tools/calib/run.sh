#!/bin/bash
export TMPDIR=/tmp
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o /tmp/fetch_calib || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/calib; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- /tmp/fetch_calib > $OUT/log.txt 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $OUT/tcc -- /tmp/fetch_calib >> $OUT/log.txt 2>&1
grep -h "known" $OUT/log.txt | head -1
python3 - <<PY
import csv,glob
for d in ("fetch","tcc"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            print(d, r["Kernel_Name"][:24], r["Counter_Name"], r["Counter_Value"])
PY

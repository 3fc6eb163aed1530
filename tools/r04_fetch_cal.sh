#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$PWD}"
root=$PWD
export TMPDIR=/tmp
out=$root/gpurun_out/r04_fetch_calibration_n1.txt
cd /tmp
$root/tools/_bin/fetch_calibration > $out 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $root/gpurun_out/r04_fcal_$c -o run -- $root/tools/_bin/fetch_calibration > /dev/null 2>&1
    f=$(find $root/gpurun_out/r04_fcal_$c -name "*counter_collection.csv" | head -1)
    python3 - $f $c >> $out <<'PY'
import csv, sys
by = {}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]: continue
    k = r["Kernel_Name"].split("(")[0]
    by.setdefault(k, {}).setdefault(r["Dispatch_Id"], 0.0)
    by[k][r["Dispatch_Id"]] += float(r["Counter_Value"])
for k, d in by.items():
    v = list(d.values())
    print("%-10s %-18s %10.1f MiB per launch (counter value x 1 KiB)" % (sys.argv[2], k, sum(v) / len(v) / 1024))
PY
    rm -rf $root/gpurun_out/r04_fcal_$c
done
cat $out

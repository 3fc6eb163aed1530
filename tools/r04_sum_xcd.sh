#!/bin/bash
# round 4: ordered_sum_kernel with runs of workgroups kept on one XCD (RVB_SUM_XCD_CHUNK = workgroups per run, 0 = off): exact-mode stage time + kernel stats
cd "${GRAFT_REPO_ROOT:-$PWD}"
root=$PWD
export TMPDIR=/tmp
out=gpurun_out/r04_ordered_sum_xcd_n1.txt
: > $out
for c in 0 64 256 512 1024 2048; do
    echo "RVB_SUM_XCD_CHUNK=$c: $(RVB_SUM_XCD_CHUNK=$c python tools/mode_bench.py 2>/dev/null | grep exact)" >> $out
done
cd /tmp
for c in 0 512; do
    export RVB_SUM_XCD_CHUNK=$c
    rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r04_sumx_$c -o run -- python3 $root/tools/mode_bench.py > /dev/null 2>&1
    f=$(find $root/gpurun_out/r04_sumx_$c -name "*kernel_stats.csv" | head -1)
    echo "chunk $c: $(grep ordered_sum $f | cut -d, -f1-4 | cut -c1-40,200-)" >> $root/$out
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/r04_sumxf_$c -o run -- python3 $root/tools/mode_bench.py > /dev/null 2>&1
    f=$(find $root/gpurun_out/r04_sumxf_$c -name "*counter_collection.csv" | head -1)
    python3 - $f $c >> $root/$out <<'PY'
import csv, sys
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "ordered_sum" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
# FETCH_SIZE rows are per XCD? sum per dispatch
by = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "ordered_sum" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        by[r["Dispatch_Id"]] = by.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
print("chunk %s: ordered_sum FETCH_SIZE per launch %.1f MB (x2 by the guide's correction: %.1f MB)" % (sys.argv[2], sum(by.values()) / len(by) / 1024, 2 * sum(by.values()) / len(by) / 1024))
PY
    rm -rf $root/gpurun_out/r04_sumx_$c $root/gpurun_out/r04_sumxf_$c
done
cat $root/$out

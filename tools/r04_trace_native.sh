#!/bin/bash
# round 4: kernel trace of the native pipeline (bench.py --native) — pipeline_timeline summary + the raw schedule of two cycles
cd "${GRAFT_REPO_ROOT:-$PWD}"
root=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/r04_trace_native -o run -- python3 $root/bench.py --steps 24 --warmup 4 --native --no-cpu-baseline --no-extras > /dev/null 2> $root/gpurun_out/r04_trace_native.err
cd $root
f=$(find gpurun_out/r04_trace_native -name "*kernel_trace.csv" | head -1)
python3 tools/pipeline_timeline.py $f 24 > gpurun_out/r04_pipeline_timeline_native_n1.txt
python3 - $f >> gpurun_out/r04_pipeline_timeline_native_n1.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"]) for r in rows)
paths = [e for e in ev if "path_pair_group" in e[2]]
# the window: from the 7th-last group launch to the 3rd-last
t0, t1 = paths[-7][0], paths[-3][0]
print("\nraw schedule of two cycles (start ms, duration ms, stream, kernel):")
for s, e, n, st in ev:
    if s < t0 or s > t1: continue
    name = n.split("(")[0].replace("(anonymous namespace)::", "")[-60:]
    print("  %8.3f %7.3f  s%-3s %s" % ((s - t0) / 1e6, (e - s) / 1e6, st, name))
PY
head -30 gpurun_out/r04_pipeline_timeline_native_n1.txt
rm -rf gpurun_out/r04_trace_native

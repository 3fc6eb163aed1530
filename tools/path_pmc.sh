#!/bin/bash
# Developer tool (GPU box, through gpurun): PMC passes of the path kernel ALONE on the GPU, per lanes-per-ray form and launch size —
# what its waves wait for (SQ), how busy the texture addresser / L1 are (TA, TCP) and what the L1 asks of L2.
#   tools/path_pmc.sh <tag> <rays> [lanes ...]        -> gpurun_out/<tag>_path_pmc.txt
set -u
tag=$1; rays=$2; shift 2
lanes=${*:-"1 2 4"}
cd "${GRAFT_REPO_ROOT:-$PWD}"
root=$PWD
export TMPDIR=/tmp
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp
for L in $lanes; do
    export RVB_PATH_LANES=$L
    i=0
    for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD" \
               "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL" \
               "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
               "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
               "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
               "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
               "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
        i=$((i + 1))
        rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/${tag}_pp_l${L}_$i" -o run -- python3 $root/tools/rays_sweep.py $rays > /dev/null 2> "$out/${tag}_pp_l${L}_$i.err" || echo "pass $i lanes $L failed"
    done
done
cd "$root"
python3 - "$tag" <<'PY' > "$out/${tag}_path_pmc.txt"
import csv, glob, os, sys
tag = sys.argv[1]
res = {}
for d in sorted(glob.glob("gpurun_out/%s_pp_l*_*" % tag)):
    if not os.path.isdir(d): continue
    lanes = d.split("_pp_l")[1].split("_")[0]
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(path)):
            if "path_" not in r["Kernel_Name"]: continue
            per.setdefault((r["Counter_Name"], r["Dispatch_Id"]), 0.0)
            per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        names = sorted(set(k[0] for k in per))
        for n in names:
            v = [x for (c, _), x in per.items() if c == n]
            res.setdefault(lanes, {})[n] = sum(v) / len(v)
for lanes, c in sorted(res.items()):
    print("lanes per ray %s" % lanes)
    for n, v in sorted(c.items()):
        print("   %-40s %16.0f" % (n, v))
PY
cat "$out/${tag}_path_pmc.txt"

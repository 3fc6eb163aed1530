#!/bin/bash
# Profiles of the default bench command on the GPU box (run through gpurun): one kernel-trace + stats pass and separate
# PMC passes (a pass never mixes --pmc with a trace domain other than --kernel-trace; FETCH_SIZE and WRITE_SIZE do not fit
# in one pass — MI355X_MICROARCH.md "rocprofv3 PMC slots").  Raw output under gpurun_out/<tag>_*, summaries by
# tools/profile_summary.py.        tools/profile.sh <tag> [ENV=value ...]
set -u
tag=$1
shift
for kv in "$@"; do export "$kv"; done      # extra environment for every pass, e.g. RVB_PATH_LANES=2 (the two-lane path kernel alone on the GPU)
cd "${GRAFT_REPO_ROOT:-$PWD}"
root=$PWD
export TMPDIR=/tmp
out=$root/gpurun_out
mkdir -p "$out"
# Kernel durations and counters are taken with ONE IR at a time (--contexts 1): in the default mode the kernels of two IRs share
# the GPU, and a kernel's span then includes its neighbour's work.  One more stats pass records the default command as it is.
BENCH="python3 $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --contexts 1"
# issue cost per VALU instruction class (input of the issue model in profile_summary.py)
[ -x "$root/tools/_bin/inst_probe" ] && "$root/tools/_bin/inst_probe" > "$out/${tag}_inst_probe.log" 2>&1 && python3 "$root/tools/inst_costs.py" "$out/${tag}_inst_probe.log" > "$out/${tag}_inst_costs.json"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats" -o run -- $BENCH > "$out/${tag}_bench_under_rocprof.json" 2> "$out/${tag}_stats.err" || echo "stats pass failed"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_stats_default" -o run -- python3 $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > "$out/${tag}_bench_default_under_rocprof.json" 2> "$out/${tag}_stats_default.err" || echo "default stats pass failed"
# (every SQ set fits the 8 SQ slots; the per-class VALU counters give the dynamic instruction mix, THREAD_CYCLES the lane utilisation)
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVES"; do
    name=$(echo "$set" | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/${tag}_pmc_$name" -o run -- $BENCH > /dev/null 2> "$out/${tag}_pmc_$name.err" || echo "pmc pass $name failed"
    echo "pass $name done"
done
cd "$root"
python3 tools/profile_summary.py "$tag"

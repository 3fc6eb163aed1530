#!/bin/bash
# Developer tool: A/B runs of bench.py on the GPU box over builds of the same C-ABI (RVB_LIB) and env switches.
#   tools/ab_bench.sh <tag> "name|lib|ENV=1 ENV2=2" ...
# Writes gpurun_out/ab_<tag>_<name>.json and prints one summary line per variant.  A variant first has to pass the
# bit-exact trace parity test, otherwise its timing is not reported.
set -u
tag=$1; shift
mkdir -p gpurun_out
for spec in "$@"; do
    IFS='|' read -r name lib envs <<< "$spec"
    out=gpurun_out/ab_${tag}_${name}
    if ! env RVB_LIB="$PWD/$lib" $envs timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "seeded or golden" > ${out}.parity.log 2>&1; then
        echo "$name PARITY-FAIL (see ${out}.parity.log)"; tail -5 ${out}.parity.log
        continue
    fi
    env RVB_LIB="$PWD/$lib" $envs timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --contexts 1 > ${out}.json 2> ${out}.err || { echo "$name BENCH-FAIL"; tail -3 ${out}.err; continue; }
    python - "$name" ${out}.json <<'EOF'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["kernel_ms"]
print("%-14s step %.3f ms | " % (sys.argv[1], d["ms_per_step"]) + " ".join("%s %.3f" % (n.replace("_kernel", "").replace("histogram", "hist"), v) for n, v in k.items()))
EOF
done

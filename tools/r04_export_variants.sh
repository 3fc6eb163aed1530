#!/bin/bash
# round 4: how the finished histogram leaves for the host — bin-range slices of the exact mode's fold (rvb_ir_accumulate_export)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r04_export_slices_n1.txt
: > $out
for cfg in "1 0" "2 0" "4 0" "8 0" "4 1" "8 1"; do
    set -- $cfg
    echo "slices $1 rows $2" >> $out
    RVB_EXPORT_SLICES=$1 RVB_EXPORT_ROWS=$2 timeout -k 10 300 python bench.py --steps 100 --warmup 8 --no-extras --no-cpu-baseline > /tmp/b.json 2> /tmp/b.err
    grep "timed region" /tmp/b.err >> $out
    python -c "import json; d=json.load(open('/tmp/b.json')); print('   to host %.3f ms, in HBM %.3f ms, exact_mode %.3f' % (d['ir_gen_to_host_ms'], d['ir_gen_wall_ms_histogram_in_hbm'], d['kernel_ms']['exact_mode']))" >> $out
done
cat $out

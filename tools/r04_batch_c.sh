#!/bin/bash
# round 4, batch c: launch sizing (98 304 rays per IR = one resident round per two-trace launch), Python vs native pipeline, the whole GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r04_launch_size_n1.txt
: > $out
for rays in 100000 98304 100000 98304; do
    timeout -k 10 300 python bench.py --steps 200 --warmup 8 --rays $rays --no-extras --no-cpu-baseline > /tmp/b.json 2> /tmp/b.err
    echo "rays $rays: $(grep 'timed region' /tmp/b.err) -> $(python -c "import json; d=json.load(open('/tmp/b.json')); print('%.4f us per 1000 rays, value %.4g' % (d['ms_per_step'] * 1e6 / $rays, d['value']))")" >> $out
done
for drv in "" "--native" "" "--native"; do
    timeout -k 10 300 python bench.py --steps 200 --warmup 8 $drv --no-extras --no-cpu-baseline > /tmp/b.json 2> /tmp/b.err
    echo "driver '$drv': $(grep 'timed region' /tmp/b.err)" >> $out
done
cat $out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests_c.log 2>&1
echo "gpu tests rc $?"; tail -5 gpurun_out/r04_gputests_c.log

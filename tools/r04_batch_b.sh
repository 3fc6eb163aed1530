#!/bin/bash
# round 4, batch b: the new tests, the bench line with the native leg, the native pipeline as the driver, a C++ caller timing itself
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_multi.py tests/test_pipeline_cpp.py tests/test_formats.py tests/test_cpp_api.py -x -q -m gpu > gpurun_out/r04_tests_b.log 2>&1
echo "tests rc $?"; tail -15 gpurun_out/r04_tests_b.log
timeout -k 10 600 python bench.py --steps 100 > gpurun_out/r04_bench_b.json 2> gpurun_out/r04_bench_b.err
echo "bench rc $?"; tail -3 gpurun_out/r04_bench_b.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r04_bench_b.json"))
    for k in ("value", "ms_per_step", "ir_gen_to_host_ms", "ir_gen_wall_ms_histogram_in_hbm", "native_pipeline", "timed_region_check", "fast_mode", "kernel_ms"):
        print(k, d.get(k))
except Exception as e:
    print("no bench line", e)
PY
timeout -k 10 300 python bench.py --steps 100 --native --no-extras --no-cpu-baseline > gpurun_out/r04_bench_native_b.json 2> gpurun_out/r04_bench_native_b.err
echo "native bench rc $?"; grep "timed region" gpurun_out/r04_bench_native_b.err
timeout -k 10 300 tests/cpp/_build/test_pipeline time 60 > gpurun_out/r04_cpp_pipeline_time_b.txt 2>&1; cat gpurun_out/r04_cpp_pipeline_time_b.txt

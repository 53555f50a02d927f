#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3b; mkdir -p $O
export S2C_GOLDEN_REPORT=$PWD/$O/reports
timeout 900 python -m pytest tests/test_bench_launch_gpu.py -m gpu -x -q > $O/launch.log 2>&1; tail -3 $O/launch.log
timeout 1200 python -m pytest tests/test_modules_cfg3_gpu.py -m gpu -q > $O/modules.log 2>&1; tail -15 $O/modules.log
timeout 900 python -m pytest tests/test_directional_gpu.py -m gpu -q > $O/dir.log 2>&1; tail -15 $O/dir.log
timeout 600 python bench.py --workload cfg2 --no-cpu-baseline > $O/cfg2.json 2> $O/cfg2.err; tail -c 300 $O/cfg2.json
timeout 600 python bench.py --no-cpu-baseline --no-fed > $O/cfg3.json 2> $O/cfg3.err

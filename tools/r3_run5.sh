#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3e; mkdir -p $O
export S2C_GOLDEN_REPORT=$PWD/$O/rep
timeout 900 python -m pytest tests/test_modules_cfg3_gpu.py -m gpu -q > $O/modules.log 2>&1; tail -12 $O/modules.log | cut -c1-700
timeout 900 python -m pytest tests/test_directional_gpu.py -m gpu -q > $O/dir.log 2>&1; tail -6 $O/dir.log | cut -c1-1500
timeout 900 python -m pytest tests/test_configs_gpu.py -m gpu -q -k "cfg3_train" > $O/cfg3.log 2>&1; tail -6 $O/cfg3.log | cut -c1-2500
timeout 900 python -m pytest tests/test_bench_launch_gpu.py -m gpu -q > $O/launch.log 2>&1; tail -3 $O/launch.log | cut -c1-600

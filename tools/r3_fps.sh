#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3g; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "fps" > $O/fps.log 2>&1; tail -3 $O/fps.log
timeout 600 python tools/bench_fps.py > $O/bench_fps.log 2>&1; grep -v amdgpu $O/bench_fps.log
timeout 600 python tools/prof_fps.py > $O/prof_fps.log 2>&1; grep -v amdgpu $O/prof_fps.log | head -4

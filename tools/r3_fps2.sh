#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3k; mkdir -p $O
timeout 600 python tools/bench_fps.py > $O/bench_fps.log 2>&1; grep -v amdgpu $O/bench_fps.log

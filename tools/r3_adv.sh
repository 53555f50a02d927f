#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3h; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_fused_gpu.py -m gpu -x -q > $O/ops.log 2>&1; tail -5 $O/ops.log | cut -c1-600

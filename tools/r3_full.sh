#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3f; mkdir -p $O
export S2C_GOLDEN_REPORT=$PWD/$O/rep
timeout 900 python -m pytest tests/test_directional_gpu.py -m gpu -q > $O/dir.log 2>&1; tail -4 $O/dir.log | cut -c1-1800
timeout 3000 python -m pytest tests -m gpu -q -x --deselect tests/test_directional_gpu.py > $O/all.log 2>&1; tail -8 $O/all.log | cut -c1-1500

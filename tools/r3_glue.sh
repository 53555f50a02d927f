#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3l; mkdir -p $O
timeout 900 python tools/glue_census.py > $O/glue.log 2>&1; grep -v amdgpu $O/glue.log | tail -130

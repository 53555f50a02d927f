#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3m; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_bench_launch_gpu.py > $O/all.log 2>&1; tail -4 $O/all.log | cut -c1-600
timeout 600 python bench.py --no-cpu-baseline --no-fed > $O/cfg3.json 2> $O/cfg3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3m/cfg3.json').read().strip().splitlines()[-1])
print(round(d['value'],1), round(d['ms_per_step'],3), d['windows']['ms_per_step'])
PY

#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3j; mkdir -p $O
timeout 1500 python -m pytest tests/test_fused_gpu.py tests/test_capnet_golden.py tests/test_configs_gpu.py -m gpu -x -q -k "decoder or golden or cfg3_train" > $O/tests.log 2>&1; tail -4 $O/tests.log | cut -c1-800
timeout 600 python bench.py --no-cpu-baseline --no-fed > $O/cfg3.json 2> $O/cfg3.err
python - <<'PY'
import json
for f in ("cfg3",):
    d=json.loads(open('gpurun_out/r3j/%s.json'%f).read().strip().splitlines()[-1])
    print(f, round(d['value'],1), round(d['ms_per_step'],3), d['windows']['median_ms_per_step'])
PY

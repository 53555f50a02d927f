#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3i; mkdir -p $O
export S2C_GOLDEN_REPORT=$PWD/$O/rep
timeout 1500 python -m pytest tests/test_capnet_golden.py tests/test_configs_gpu.py tests/test_directional_gpu.py tests/test_train_loop_gpu.py tests/test_plain_caption.py tests/test_variants.py -m gpu -x -q > $O/tests.log 2>&1; tail -4 $O/tests.log | cut -c1-800
timeout 600 python bench.py --no-cpu-baseline --no-fed > $O/cfg3.json 2> $O/cfg3.err
S2C_LOCAL_TRAIN_ATTN=0 timeout 600 python bench.py --no-cpu-baseline --no-fed > $O/cfg3_dense.json 2> $O/cfg3_dense.err
python - <<'PY'
import json
for f in ("cfg3","cfg3_dense"):
    d=json.loads(open('gpurun_out/r3i/%s.json'%f).read().strip().splitlines()[-1])
    print(f, round(d['value'],1), round(d['ms_per_step'],3), d['windows']['median_ms_per_step'])
    print([(k['kernel'], round(k['ms_per_step'],3), k['calls_per_step']) for k in d['kernels'][:10]])
PY

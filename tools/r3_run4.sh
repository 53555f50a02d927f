#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3d; mkdir -p $O
timeout 600 python tools/diag_fp2.py fp2 > $O/diag_fp2.log 2>&1; grep -v amdgpu $O/diag_fp2.log | tail -40
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "fps" > $O/fps.log 2>&1; tail -2 $O/fps.log
S2C_GOLDEN_REPORT=$PWD/$O/rep timeout 900 python -m pytest tests/test_directional_gpu.py -m gpu -q > $O/dir.log 2>&1; tail -8 $O/dir.log | cut -c1-600
timeout 600 python bench.py --workload cfg2 --no-cpu-baseline > $O/cfg2.json 2> $O/cfg2.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3d/cfg2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step']); print([(k['kernel'], round(k['ms_per_step'],3), k['calls_per_step']) for k in d['kernels'][:8]])
PY

timeout 300 python -m pytest tests/test_fused_gpu.py -x -q -k "persistent_decoder" 2>&1 | tail -1
timeout 300 python tools/bench_decoder_persist.py | head -2
R=$(pwd); cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o s -- python $R/bench.py --no-cpu-baseline --no-fed --steps 5 --warmup 2 --no-graph > /dev/null 2>&1; grep -i "persist" /tmp/pp/s_kernel_stats.csv | cut -c1-120

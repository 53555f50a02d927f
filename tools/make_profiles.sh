#!/bin/bash
# Regenerates the judged profile set of a round on the GPU box (run through gpurun from the
# repo root): tools/make_profiles.sh r02   -> gpurun_out/<round>/...  (copy into profiles/)
R=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py > $OUT/${R}_bench_default.json 2> $OUT/bench_default.err
rm -rf /tmp/p1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o s -- python $ROOT/bench.py --no-cpu-baseline --no-fed > $OUT/p1.log 2>&1
cp /tmp/p1/s_kernel_stats.csv $OUT/${R}_bench_default_kernel_stats.csv
rm -rf /tmp/p2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -o s -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-fed > $OUT/p2.log 2>&1
cp /tmp/p2/s_kernel_stats.csv $OUT/${R}_bench_cfg3_eager_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/p_$c -o s -- python $ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-fed > $OUT/pmc_$c.log 2>&1
done
python $ROOT/tools/pmc_summary.py $OUT/${R}_pmc_bench.json FETCH_SIZE=$(ls /tmp/p_FETCH_SIZE/*counter_collection.csv | head -1) WRITE_SIZE=$(ls /tmp/p_WRITE_SIZE/*counter_collection.csv | head -1) > $OUT/pmc_summary.log 2>&1
for w in cfg2 cfg5 cfg3e; do
  python $ROOT/bench.py --workload $w --no-cpu-baseline --no-fed > $OUT/${R}_bench_$w.json 2> $OUT/bench_$w.err
done
ls -la $OUT

#!/bin/bash
# Regenerates the judged profile set of a round on the GPU box (run through gpurun from the
# repo root): tools/make_profiles.sh r02   -> gpurun_out/<round>/...  (copy into profiles/)
R=${1:-r06}
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
# MFMA utilisation and the wave-cycle split of every kernel (round 3: streaming GEMM, pooled-layer
# kernels, s2c_bn_bwd_gemm, dw_x3): SQ counters in their own pass, --kernel-trace only
rm -rf /tmp/p_sq && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/p_sq -o s -- python $ROOT/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-fed > $OUT/pmc_sq.log 2>&1
python $ROOT/tools/pmc_sq_summary.py $(ls /tmp/p_sq/*counter_collection.csv | head -1) $OUT/${R}_pmc_sq_bench.json > $OUT/pmc_sq_summary.log 2>&1
# the N > 1 path on this one GPU: two ranks over gloo (RCCL refuses two ranks per device)
S2C_DIST_BACKEND=gloo python $ROOT/bench.py --gpus 2 --no-cpu-baseline > $OUT/${R}_bench_2ranks_gloo_1gpu.json 2> $OUT/bench_2ranks.err
# counters of the evaluation workloads (round 5): FETCH / WRITE and the SQ MFMA-busy pass for cfg3e and
# cfg5 (planes_gemm_kernel, attn_local_kernel), each in its own run; bench.py's roofline_decode reads
# the SQ summary of its workload
for w in cfg3e cfg5; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/q_${w}_$c && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/q_${w}_$c -o s -- python $ROOT/bench.py --workload $w --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-fed > $OUT/pmc_${w}_$c.log 2>&1
  done
  python $ROOT/tools/pmc_summary.py $OUT/${R}_pmc_$w.json FETCH_SIZE=$(ls /tmp/q_${w}_FETCH_SIZE/*counter_collection.csv | head -1) WRITE_SIZE=$(ls /tmp/q_${w}_WRITE_SIZE/*counter_collection.csv | head -1) > $OUT/pmc_summary_$w.log 2>&1
  rm -rf /tmp/q_${w}_sq && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/q_${w}_sq -o s -- python $ROOT/bench.py --workload $w --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-fed > $OUT/pmc_sq_$w.log 2>&1
  python $ROOT/tools/pmc_sq_summary.py $(ls /tmp/q_${w}_sq/*counter_collection.csv | head -1) $OUT/${R}_pmc_sq_$w.json > $OUT/pmc_sq_summary_$w.log 2>&1
  # bench.py looks the summaries up under profiles/: make this run's visible to the lines below
  cp $OUT/${R}_pmc_sq_$w.json $ROOT/profiles/ 2>/dev/null
done
cp $OUT/${R}_pmc_bench.json $OUT/${R}_pmc_sq_bench.json $ROOT/profiles/ 2>/dev/null
for w in cfg2 cfg5 cfg3e; do
  python $ROOT/bench.py --workload $w --no-cpu-baseline --no-fed > $OUT/${R}_bench_$w.json 2> $OUT/bench_$w.err
  # per-kernel table of the same workload, launched eagerly (kernel names inside a hipGraph replay
  # are not attributed): what runs in the forward + greedy decode, and that no library GEMM does
  rm -rf /tmp/pe_$w && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe_$w -o s -- python $ROOT/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-fed > $OUT/pe_$w.log 2>&1
  cp /tmp/pe_$w/s_kernel_stats.csv $OUT/${R}_${w}_eager_kernel_stats.csv
done
# the reference's own training batch sizes (README.md:145: 12; slurm/train.job:24: 16): the persistent
# decoder kernels take up to 8 rows, beyond that the launch chain runs
for b in 12 16; do
  python $ROOT/bench.py --batch $b --no-cpu-baseline --no-fed > $OUT/${R}_bench_cfg3_batch$b.json 2> $OUT/bench_b$b.err
done
# the reference's own default --num_locals -1 (scripts/train.py:322, benchmark/predict.py:249: attention over
# all K proposals): beyond the persistent / planes kernels' 32 objects -> the launch chain and the `_step`
# loop; what that costs, with the kernel tables
python $ROOT/bench.py --num-locals -1 --no-cpu-baseline --no-fed > $OUT/${R}_bench_cfg3_locals_all.json 2> $OUT/bench_locals_all.err
python $ROOT/bench.py --workload cfg3e --num-locals -1 --no-cpu-baseline --no-fed > $OUT/${R}_bench_cfg3e_locals_all.json 2> $OUT/bench_e_locals_all.err
for w in cfg3 cfg3e; do
  rm -rf /tmp/pl_$w && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl_$w -o s -- python $ROOT/bench.py --workload $w --num-locals -1 --steps 5 --warmup 2 --no-cpu-baseline --no-graph --no-fed > $OUT/pl_$w.log 2>&1
  cp /tmp/pl_$w/s_kernel_stats.csv $OUT/${R}_${w}_locals_all_eager_kernel_stats.csv
done
# the RCCL code path on ONE rank (nccl backend, two-stage backward graphs, both buckets all-reduced)
S2C_FORCE_DDP=1 python $ROOT/bench.py --no-cpu-baseline --no-fed > $OUT/${R}_bench_force_ddp_1rank.json 2> $OUT/bench_force_ddp.err
# no library GEMM in the train step: the census of torch.mm / addmm / bmm / linear calls
python $ROOT/tools/lib_gemm_census.py cfg3 > $OUT/${R}_lib_gemm_census_cfg3.txt 2>&1
python $ROOT/tools/count_launches.py > $OUT/${R}_launches_cfg3.txt 2>&1
python $ROOT/tools/glue_census.py > $OUT/${R}_glue_census_cfg3.txt 2>&1
# every library-GEMM call site of the DEFAULT command (capture, replay, instrumented pass, stream probe):
# round 6: none (round 5: 120 `Cijk_*` launches of pipeline.py's stream probe, outside the step)
python $ROOT/tools/trace_lib_gemm_bench.py > /dev/null 2> $OUT/${R}_lib_gemm_sites_default_command.txt
# cross-workgroup exchange costs inside one kernel, all workgroups / XCD-local groups (the number the
# multi-workgroup FPS and the persistent decoder are priced with: DESIGN 4.1, 4.4)
python $ROOT/tools/probe_sync.py > $OUT/${R}_probe_sync.txt 2>&1
ls -la $OUT

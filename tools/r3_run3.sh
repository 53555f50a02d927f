#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3c; mkdir -p $O
S2C_DEBUG=1 S2C_FORCE_DDP=1 timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fed > $O/force.json 2> $O/force.err; echo "force rc=$?"; grep "bench\]" $O/force.err | tail -4
S2C_DEBUG=1 S2C_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 3 --warmup 2 > $O/g2.json 2> $O/g2.err; echo "gloo2 rc=$?"; grep "bench\]" $O/g2.err | tail -4
run() {  # name, env...
  n=$1; shift
  env "$@" S2C_GOLDEN_REPORT=$PWD/$O/rep_$n timeout 600 python -m pytest tests/test_modules_cfg3_gpu.py -m gpu -q -k "sa2 or fp2" > $O/mod_$n.log 2>&1
  tail -1 $O/mod_$n.log
}
run base X=1
run nosplit S2C_GEMM_SPLIT=0
run nostream S2C_GEMM_STREAM=0
run nofusebwd S2C_FUSE_BWD_GEMM=0
run nohanddw S2C_HAND_DW=0
run nohandda S2C_HAND_DA=0
run nobnrelugemm S2C_FUSE_BNRELU_GEMM=0
run nodyscatter S2C_FUSE_DY_SCATTER=0
timeout 600 python tools/lib_gemm_census.py > $O/census.log 2>&1; tail -45 $O/census.log

import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tests import test_train_loop_gpu as T
bench, wl, model, opt, dd, cfg, dev = T._setup()
step = bench.make_step(model, wl, cfg, opt, None, dev)
print(os.environ.get("S2C_GEMM_STREAM"), os.environ.get("S2C_FUSE_BNRELU_GEMM"), [float(step(dd).detach()) for _ in range(3)])

"""The whole inference SA stage in one kernel (csrc/s2c_sa_fused.hip) vs the three per-layer
kernels it replaces, at the SA1 shape of BASELINE configs[1] (cfg2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd.pointnet2 import fused
from scan2cap_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
from scan2cap_amd.synthetic import scene_xyz
from tools.bench_ops import timeit

B, N, C, m, r, ns, mlp = 8, 40000, 4, 2048, 0.2, 64, [64, 64, 128]
torch.manual_seed(0)
sa = PointnetSAModuleVotes(npoint=m, radius=r, nsample=ns, mlp=[C] + mlp, use_xyz=True,
                           normalize_xyz=True).cuda().eval()
xyz = torch.from_numpy(scene_xyz(B, N, seed=4)).cuda()
pc = torch.cat([xyz, torch.randn(B, N, C, device="cuda")], -1)
feats = pc[..., 3:].transpose(1, 2)
with torch.no_grad():
    geom = sa.geometry(xyz)
    f = lambda: sa(xyz, feats, geom=geom)
    f(); torch.cuda.synchronize()
    t1 = timeit(f, iters=int(os.environ.get("ITERS", 20)))
    fused.FUSE_EVAL_STAGE = False
    f(); torch.cuda.synchronize()
    t2 = timeit(f, iters=int(os.environ.get("ITERS", 20)))
M = B * m * ns
gf = 2.0 * M * (7 * 64 + 64 * 64 + 64 * 128) / 1e9
print("SA1 cfg2 stage: fused %.1f us (%.0f TF fp32-equiv, %.2f of the bf16x3 roof) | per-layer %.1f us" % (
    t1, gf / t1 * 1e3, gf / t1 * 1e3 / (2500.0 / 6), t2))

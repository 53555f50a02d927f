"""SA1-size last layer (1M rows, 64 -> 128 channels, 64 rows per centre): the materialised path
(GEMM + bn_relu_max; bn_relu_max_bwd + dX GEMM + dW) vs the pooled-layer algebra, piece by piece."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from scan2cap_amd.pointnet2 import fused
from scan2cap_amd import _C
J, ns, K, C3 = 16384, 64, 64, 128
M = J * ns
torch.manual_seed(0)
A = torch.relu(torch.randn(M, K, device="cuda")); W = torch.randn(C3, K, device="cuda") * 0.2
Y = A @ W.t()
gamma = torch.rand(C3, device="cuda") + 0.5; beta = torch.randn(C3, device="cuda") * 0.3
mean, var = Y.mean(0), Y.var(0, unbiased=False); invstd = 1 / torch.sqrt(var + 1e-5)
scale = gamma * invstd; shift = beta - mean * scale
out = torch.empty(J, C3, device="cuda"); arg = torch.empty(J, C3, dtype=torch.int32, device="cuda"); ymax = torch.empty(J, C3, device="cuda")
fused._call("s2c_bn_relu_max", Y, J, ns, C3, Y.data_ptr(), scale.data_ptr(), shift.data_ptr(), out.data_ptr(), arg.data_ptr(), ymax.data_ptr())
dOut = torch.randn(J, C3, device="cuda")
def algebra():
    return fused.pooled_layer_backward(dOut, arg, ymax, scale, shift, mean, invstd, gamma, False, A, W, ns)
for _ in range(3): algebra()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    algebra(); torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
tot = sum(e.device_time for e in evs)
print("pooled_layer_backward: %d kernels, %.1f us of kernel time" % (len(evs), tot))
agg = {}
for e in evs:
    a = agg.setdefault(e.name[:80], [0, 0.0]); a[0] += 1; a[1] += e.device_time
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %6.1f us x%d  %s" % (v[1], v[0], k))

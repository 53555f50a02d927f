import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd.pointnet2 import _ext
from scan2cap_amd.synthetic import scene_xyz
xyz = torch.from_numpy(scene_xyz(8, 40000)).cuda()
def chain():
    return _ext.furthest_point_sampling(xyz, 2048)
for _ in range(3): chain()
torch.cuda.synchronize()
for ns in (1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R = 12
    for i in range(R):
        with torch.cuda.stream(streams[i % ns]):
            chain()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("streams=%d: %.2f ms per FPS" % (ns, dt / R * 1e3))
print("env GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))

import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd import _C
from scan2cap_amd.synthetic import scene_xyz
lib = _C.load()
P, I = ctypes.c_void_p, ctypes.c_int
lib.s2c_fps_cells_profile.argtypes = [I, I, I, P, P, P, I, P, P]
for waves in (17, 16, 8, 4):
    for mode in ("volume", "surface"):
        B, N, m = 8, 40000, 2048
        xyz = torch.from_numpy(scene_xyz(B, N, mode=mode)).cuda()
        ws = torch.empty(lib.s2c_fps_cells_workspace_bytes(B, N), dtype=torch.uint8, device="cuda")
        out = torch.empty((B, m), dtype=torch.int32, device="cuda")
        prof = torch.zeros((16, 8), dtype=torch.int64, device="cuda")
        lib.s2c_fps_cells_profile(B, N, m, xyz.data_ptr(), ws.data_ptr(), out.data_ptr(), waves, prof.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        p = prof.cpu().numpy()[:min(waves, 16)].astype(float) / (m - 1)
        print("waves=%d %-7s per round (s_memtime ticks, mean over waves / max): cells %.0f/%.0f  own-argmax %.0f  barrier-wait %.0f/%.0f  decode %.0f | active cells per wave-round %.2f (max wave %.2f)" % (
            waves, mode, p[:, 0].mean(), p[:, 0].max(), p[:, 1].mean(), p[:, 2].mean(), p[:, 2].max(), p[:, 3].mean(), p[:, 4].mean(), p[:, 4].max()))

for mode in ("volume", "surface"):
    B, N, m = 8, 40000, 2048
    xyz = torch.from_numpy(scene_xyz(B, N, mode=mode)).cuda()
    ws = torch.empty(lib.s2c_fps_cells_workspace_bytes(B, N), dtype=torch.uint8, device="cuda")
    out = torch.empty((B, m), dtype=torch.int32, device="cuda")
    prof = torch.zeros((16, 8), dtype=torch.int64, device="cuda")
    lib.s2c_fps_cells_profile(B, N, m, xyz.data_ptr(), ws.data_ptr(), out.data_ptr(), -16, prof.data_ptr(),
                              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    p = prof.cpu().numpy().astype(float)
    r = p[0, 0]
    print("multi-pick %-7s rounds %d (%.2f picks/round) | active cells per wave-round %.2f | cycles per round: update %.0f (max wave %.0f) candidates+barrier %.0f" % (
        mode, r, (m - 1) / r, p[:, 1].mean() / r, p[:, 2].mean() / r, p[:, 2].max() / r, p[:, 3].mean() / r))

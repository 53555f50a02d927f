"""parse_predictions at the cfg3 / cfg5 shapes: HIP path vs the numpy
restatement (oracle/post.py, which mirrors the reference's CPU algorithm minus scipy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import post_common as pc
from scan2cap_amd import ap_helper
from oracle import post

dev = torch.device("cuda")
for (B, K, N) in [(8, 256, 40000), (16, 512, 80000)]:
    pc.B, pc.K, pc.N = B, K, N
    inputs = pc.make_inputs(seed=3)
    cfg = dict(pc.POST_DICTS["predict"], dataset_config=pc.dataset_config())
    ep = {k: torch.from_numpy(v).to(dev) for k, v in inputs.items()}
    for _ in range(2):
        ap_helper.parse_predictions(dict(ep), cfg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ap_helper.parse_predictions(dict(ep), cfg)
    torch.cuda.synchronize()
    t_gpu = (time.perf_counter() - t0) / 5
    # device-only part (decode + empty-box + NMS), no host list building
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        boxes = ap_helper.decode_boxes(ep, cfg["dataset_config"])
        ne = ap_helper.nonempty_box_mask(ep["point_clouds"], boxes)
        obj = torch.softmax(ep["objectness_scores"], -1)[:, :, 1]
        ap_helper.nms_mask(boxes, obj, ne, cfg)
    e1.record(); torch.cuda.synchronize()
    t_dev = e0.elapsed_time(e1) / 10 / 1e3
    # CPU: one scene of the same size
    one = {k: v[:1] for k, v in inputs.items()}
    t0 = time.perf_counter()
    post.parse_predictions(one, cfg)
    t_cpu = (time.perf_counter() - t0) * B
    print("B=%d K=%d N=%d: HIP %.2f ms end-to-end (%.3f ms device work), numpy port %.0f ms "
          "(extrapolated from 1 scene) -> %.0fx" % (B, K, N, t_gpu * 1e3, t_dev * 1e3,
                                                   t_cpu * 1e3, t_cpu / t_gpu))

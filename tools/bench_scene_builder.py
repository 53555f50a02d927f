"""Item assembly at the cfg3 size: B=8 items of N=40000 points x 135
channels (xyz + normal + 128 multiview + height), augmentation on, from synthetic
150k-vertex scenes resident in HBM.  Prints ONE JSON line: items/s of
`SceneBatchBuilder.build` (device-side vertex sampling), the per-kernel durations (HIP
events), the gather kernel against the 8 TB/s HBM roof, and beside it the numpy oracle
(= the reference's per-item algorithm, oracle/scene_builder.py) on this host.

    python tools/bench_scene_builder.py [--scenes 16] [--steps 50]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from scan2cap_amd import _C, scene_builder as sb  # noqa: E402
from tests import scene_common as sc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=16)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--points", type=int, default=40000)
    ap.add_argument("--cpu-items", type=int, default=6)
    args = ap.parse_args()
    B, N, mvw = args.batch, args.points, 128
    scenes = [sc.make_scene(500 + i, 140000 + 1777 * i, mvw, num_instances=48)
              for i in range(args.scenes)]
    store = sb.SceneStore("cuda:0", multiview_width=mvw)
    for i, s in enumerate(scenes):
        store.add_scene("s%d" % i, s["mesh_vertices"], s["instance_labels"],
                        s["semantic_labels"], s["instance_bboxes"], s["multiview"])
    t0 = time.time()
    store.finalize()
    torch.cuda.synchronize()
    t_final = time.time() - t0
    opts = dict(use_color=False, use_height=True, use_normal=True, use_multiview=True,
                augment=True)
    msa = np.full((18, 3), 0.6)
    builder = sb.SceneBatchBuilder(store, msa, num_points=N, **opts)
    rs = np.random.RandomState(0)
    pick = lambda k: ["s%d" % ((k * B + b) % args.scenes) for b in range(B)]
    oid_of = lambda ids: [int(scenes[int(s[1:])]["instance_bboxes"][0, 7]) for s in ids]

    def step(k, device_choices=True):
        ids = pick(k)
        return builder.build(ids, oid_of(ids), builder.draw(ids, rng=rs,
                                                            device_choices=device_choices))

    for k in range(5):
        step(k)
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(args.steps):
        out = step(k)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / args.steps
    # per-kernel durations
    _C.TIMER.start()
    for k in range(args.steps):
        step(k)
    per = _C.TIMER.stop()
    kern = {n: {"us": 1e3 * v["total_ms"] / v["calls"]} for n, v in per.items()}
    g = per["s2c_scene_gather"]
    gbs = g["alg_bytes"] / (g["total_ms"] * 1e-3) / 1e9
    # host-drawn (numpy stream replay) variant
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(5):
        step(k, device_choices=False)
    torch.cuda.synchronize()
    dt_host = (time.time() - t0) / 5
    # CPU: the oracle, one item at a time on one core
    t0 = time.time()
    for k in range(args.cpu_items):
        i = k % args.scenes
        d = osb.draw(len(scenes[i]["mesh_vertices"]), N, True, rng=rs)
        osb.build_item(scenes[i], d, 0, N, msa, **opts)
    cpu_item = (time.time() - t0) / args.cpu_items
    print(json.dumps({
        "metric": "items_per_s", "value": B / dt, "unit": "items/s", "batch": B,
        "points": N, "channels": builder.Cout, "ms_per_batch": dt * 1e3,
        "ms_per_batch_host_choices": dt_host * 1e3,
        "kernels": kern,
        "roofline": {"kernel": "s2c_scene_gather", "bound": "hbm", "achieved": gbs,
                     "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0},
        "store": {"scenes": len(store), "resident_GB": store.resident_bytes() / 1e9,
                  "finalize_s": t_final},
        "cpu_baseline": {"value": 1.0 / cpu_item, "unit": "items/s", "cores": 1,
                         "kind": "port", "sample": "%d items of the numpy oracle"
                         % args.cpu_items},
        "bytes_per_batch_h2d": int(B * 256 + B * 8 + 8 * ((B * 4 + 7) // 8)),
    }))


if __name__ == "__main__":
    from oracle import scene_builder as osb  # noqa: E402  (CPU baseline leg only)
    main()

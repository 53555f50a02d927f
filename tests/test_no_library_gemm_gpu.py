"""Round 6: "no library GEMM on the BASELINE workloads" as an ASSERTION (it was a tool run by hand,
tools/lib_gemm_census.py): one eager step of cfg2 / cfg3 / cfg3e and of cfg3 / cfg3e at the
reference's default `--num_locals -1` under torch.profiler must launch no rocBLAS / hipBLASLt /
Tensile (`Cijk_*`) / MIOpen kernel -- the shape-guarded library fall-backs of pointnet2/fused.py,
models/decoder_fused.py and models/caption_module.py are never taken at these shapes -- and must load
the product library's kernels (a silent framework fall-back would also be "no Cijk").
Reference path covered: models/capnet.py:60-117 forward (+ lib/loss_helper.py:381-491 and the
backward for the train workloads)."""
import copy
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

LIBRARY_MARKERS = ("Cijk_", "rocblas", "hipblaslt", "Tensile", "miopen", "MIOpen", "gemm_kernel_lib")


def _kernels_of_one_step(name, num_locals=None):
    import bench
    from torch.profiler import ProfilerActivity, profile
    from scan2cap_amd.loss_helper import get_scene_cap_loss
    import numpy as np
    wl = copy.deepcopy(bench.WORKLOADS[name])
    if num_locals is not None:
        wl["num_locals"] = num_locals
    dev = torch.device("cuda")
    vocabulary, embeddings, table = bench.make_vocab(wl["V"])
    msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
    torch.manual_seed(0)
    model = bench.build_model(wl, vocabulary, embeddings, msa).to(dev)
    model.train(wl["train"])
    dd = bench.to_device(bench.make_batch(wl, wl["B"], 42, table, msa), dev)
    cfg = bench.LossConfig(msa)
    if wl["train"]:
        from scan2cap_amd.synthetic import aim_reference_boxes_at_proposals
        dd = aim_reference_boxes_at_proposals(model, dd)     # a live caption branch (IoU 1 boxes)

    def step():
        if not wl["train"]:
            with torch.no_grad():
                return model(dict(dd), use_tf=False, is_eval=True)
        model.zero_grad(set_to_none=True)
        d = model(dict(dd), use_tf=True, is_eval=False)
        d = get_scene_cap_loss(d, dev, cfg, None, detection=True, caption=True, orientation=False,
                               distance=False)
        d["loss"].backward()
        return d
    for _ in range(2):
        out = step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = step()
        torch.cuda.synchronize()
    names = {}
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            names[e.name] = names.get(e.name, 0) + 1
    if wl["train"]:
        assert bool(out["good_bbox_masks"].any()) and float(out["cap_loss"]) > 0.0
    return names


@pytest.mark.parametrize("name,num_locals", [("cfg3", None), ("cfg3", -1), ("cfg2", None),
                                             ("cfg3e", None), ("cfg3e", -1)])
def test_one_step_launches_no_library_gemm(name, num_locals):
    names = _kernels_of_one_step(name, num_locals)
    assert len(names) > 20, names
    lib = {n: c for n, c in names.items() if any(m in n for m in LIBRARY_MARKERS)}
    assert not lib, "library kernels in one eager %s step (num_locals=%s): %s" % (name, num_locals, lib)
    hand = [n for n in names if "rows_gemm" in n or "point_gemm" in n or "planes_gemm" in n
            or "rows_stream_gemm" in n or "sgemm_kernel" in n or "sa_fused_eval" in n
            or "sa_gather_add" in n]
    assert hand, "none of the product library's GEMM kernels ran: %s" % sorted(names)[:40]

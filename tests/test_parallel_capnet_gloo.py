"""world_size-2 CPU (gloo) test of the N>1 training path on the REAL CapNet: scenes sharded
over the ranks, backward in two stages (captioner + relation graph first, detector second:
parallel.backward_in_two_stages), the captioner bucket all-reduced asynchronously while the
detector's backward still runs, BatchNorm statistics per replica, parameters that receive no
gradient (use_orientation heads with the orientation loss off) reduced as zeros.  Checked
against plain `loss.backward()` + a per-tensor all-reduce.  The oracle ops stand in for the
HIP kernels (CPU test double, as in tests/test_capnet_golden.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import torch_ext
    from scan2cap_amd.pointnet2 import _ext
    for n in torch_ext.NAMES:
        setattr(_ext, n, getattr(torch_ext, n))
    import bench
    from scan2cap_amd import parallel as par
    from scan2cap_amd.loss_helper import get_scene_cap_loss
    from scan2cap_amd.models import CapNet
    wl = dict(B=2, N=1024, C=4, K=16, V=60, train=True, desc="gloo test")
    vocabulary, embeddings, table = bench.make_vocab(wl["V"])
    msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
    torch.manual_seed(50 + rank)               # different init per rank: broadcast fixes it
    model = CapNet(18, vocabulary, embeddings, 1, 18, msa, input_feature_dim=wl["C"],
                   num_proposal=wl["K"], num_locals=5, use_topdown=True,
                   graph_mode="edge_conv", num_graph_steps=2, use_relation=True,
                   use_orientation=True, use_distance=True).train()
    early, late = par.split_detector_captioner(model)
    ddp = par.BucketedGradAllReduce(model, [early, late])
    w0 = torch.cat([p.detach().flatten() for p in model.parameters()])
    dd = bench.to_device(bench.make_batch(wl, wl["B"], 300 + rank, table, msa), "cpu")
    cfg = bench.LossConfig(msa)
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def forward():
        model.load_state_dict(state)           # same BN running statistics for both passes
        d = model(dict(dd), use_tf=True, is_eval=False)
        # orientation / distance heads exist but their losses are off: unused parameters
        return get_scene_cap_loss(d, torch.device("cpu"), cfg, None, detection=True,
                                  caption=True, orientation=False, distance=False)

    # reference: plain backward, then average every gradient tensor across the ranks
    model.zero_grad(set_to_none=True)
    d = forward()
    d["loss"].backward()
    want, unused = {}, []
    for n, p in model.named_parameters():
        g = p.grad.clone() if p.grad is not None else torch.zeros_like(p)
        if p.grad is None:
            unused.append(n)
        dist.all_reduce(g)
        want[n] = g / world
    # path under test
    ddp.drop_grads()
    d2 = forward()
    order = []

    def between():
        order.append("early bucket on the wire")
        ddp.pack_grads(0)
        ddp.reduce(0, async_op=True)
    par.backward_in_two_stages(d2, early, late, between)
    order.append("detector backward done")
    ddp.pack_grads(1)
    ddp.reduce(1, async_op=True)
    ddp.wait()
    worst = 0.0
    for n, p in model.named_parameters():
        scale = max(1.0, float(want[n].abs().max()))
        worst = max(worst, float((p.grad - want[n]).abs().max()) / scale)
    views_ok = all(any(f.data_ptr() <= p.grad.data_ptr() < f.data_ptr() + f.numel() * 4
                       for f in ddp.flats) for p in model.parameters())
    out[rank] = dict(w0=w0, worst=worst, unused=unused, order=order, views=views_ok,
                     loss=(float(d["loss"]), float(d2["loss"])),
                     n_early=sum(p.numel() for p in early), n_late=sum(p.numel() for p in late),
                     bn=model.backbone_net.sa1.mlp_module.layer0.bn.bn.running_mean.clone())
    dist.destroy_process_group()


def test_capnet_two_stage_backward_bucketed_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert torch.equal(a["w0"], b["w0"])                      # rank-0 broadcast
    assert a["loss"][0] == a["loss"][1] and a["loss"][0] != b["loss"][0]   # own shard each
    assert a["worst"] <= 1e-5 and b["worst"] <= 1e-5, (a["worst"], b["worst"])
    assert a["order"] == ["early bucket on the wire", "detector backward done"]
    assert a["views"] and b["views"]
    # the orientation / distance heads got no gradient and were reduced as zeros
    assert any("edge_predict" in n for n in a["unused"]), a["unused"]
    assert a["n_early"] > 2 * a["n_late"]                     # the early bucket is the big one
    assert not torch.equal(a["bn"], b["bn"])                  # BN statistics stay per replica

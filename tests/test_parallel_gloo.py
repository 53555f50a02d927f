"""world_size-2 CPU (gloo) test of the data-parallel pieces: scene sharding and
the flat-bucket gradient all-reduce (the N>1 path of bench.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from scan2cap_amd.parallel import FlatGradAllReduce, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)  # different init per rank: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5),
                              torch.nn.Linear(5, 3))
    ddp = FlatGradAllReduce(net)
    w0 = torch.cat([p.detach().flatten() for p in net.parameters()])
    # each rank owns its shard of 6 "scenes"
    torch.manual_seed(7)
    data = torch.randn(6, 6)
    lo, hi = shard_range(6, rank, world)
    ddp.zero_grad()
    net(data[lo:hi]).pow(2).sum().backward()
    local = ddp.flat.clone()
    # the drop/pack flavour (fresh grads + one multi-tensor copy) must agree
    ddp.drop_grads()
    net(data[lo:hi]).pow(2).sum().backward()
    ddp.pack_grads()
    assert torch.allclose(ddp.flat, local, atol=1e-6)
    ddp.reduce()
    out[rank] = dict(w0=w0, local=local, reduced=ddp.flat.clone(), shard=(lo, hi),
                     views=all(p.grad.data_ptr() >= ddp.flat.data_ptr()
                               for p in net.parameters()))
    dist.destroy_process_group()


def test_flat_grad_allreduce_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert a["shard"] == (0, 3) and b["shard"] == (3, 6)
    assert torch.equal(a["w0"], b["w0"])                      # rank-0 broadcast
    assert a["views"] and b["views"]
    want = (a["local"] + b["local"]) / 2
    assert torch.allclose(a["reduced"], want, atol=1e-6)
    assert torch.equal(a["reduced"], b["reduced"])


def test_shard_range_covers_everything():
    for total in (1, 7, 8, 64):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = shard_range(total, r, world)
                got += list(range(lo, hi))
            assert got == list(range(total))

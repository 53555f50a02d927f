"""End-to-end training steps on the GPU: the solver's step (lib/solver.py:293-302:
forward -> get_scene_cap_loss -> zero_grad / backward / Adam.step) repeated on one fixed
batch.  The first updates must track the op-by-op CPU path (torch autograd over the
oracle ops) and the captured hipGraph + geometry-slot pipeline of bench.py must replay
the eager steps.  (At random init with 2 scenes the loss itself is not monotone -- the
label assignment and the sampled proposals move with the weights -- on the CPU path
just as here, so "the loss falls" is not asserted.)"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _setup(B=2, N=4096, C=4, K=64, V=200):
    import bench
    wl = dict(B=B, N=N, C=C, K=K, V=V, train=True, desc="test")
    dev = torch.device("cuda")
    vocabulary, embeddings, table = bench.make_vocab(V)
    msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
    torch.manual_seed(0)
    model = bench.build_model(wl, vocabulary, embeddings, msa).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True,
                           fused=True)
    dd = bench.to_device(bench.make_batch(wl, B, 7, table, msa), dev)
    return bench, wl, model, opt, dd, bench.LossConfig(msa), dev


def test_first_updates_track_the_cpu_path():
    """Loss of steps 1..3 (i.e. after 0, 1, 2 Adam updates through every hand-written
    backward kernel) vs the same loop on the CPU: torch autograd + oracle ops."""
    from oracle import torch_ext
    from scan2cap_amd.pointnet2 import _ext
    bench, wl, model, opt, dd, cfg, dev = _setup()
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    step = bench.make_step(model, wl, cfg, opt, None, dev)
    got = [float(step(dd).detach()) for _ in range(3)]
    # CPU replica
    cpu = torch.device("cpu")
    torch.manual_seed(0)
    vocabulary, embeddings, table = bench.make_vocab(wl["V"])
    msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
    ref_model = bench.build_model(wl, vocabulary, embeddings, msa).train()
    ref_model.load_state_dict(state)
    ref_opt = torch.optim.Adam(ref_model.parameters(), lr=1e-3, weight_decay=1e-5)
    ref_dd = bench.to_device(bench.make_batch(wl, wl["B"], 7, table, msa), cpu)
    saved = {n: getattr(_ext, n) for n in torch_ext.NAMES}
    try:
        for n in torch_ext.NAMES:
            setattr(_ext, n, getattr(torch_ext, n))
        ref_step = bench.make_step(ref_model, wl, bench.LossConfig(msa), ref_opt, None, cpu)
        want = [float(ref_step(ref_dd).detach()) for _ in range(3)]
    finally:
        for n, f in saved.items():
            setattr(_ext, n, f)
    assert all(np.isfinite(got))
    np.testing.assert_allclose(got[0], want[0], rtol=1e-4)     # same weights: forward parity
    # one update through all gradients.  The gradients are ill-conditioned at random init
    # (max-pool arg-max / ReLU / label flips: a 1e-7 relative change of the first-layer
    # pre-activations moves them by 1e-3..1e-2): the same step on the tiled GEMM,
    # the streaming GEMM (K walked in another order) and the exact fp32 MFMA chain gives
    # 32.39 / 31.39 / 32.22 here against 32.20 on the CPU (tools/diag_first_updates.py)
    # (sanity bound only: with the pooled-layer algebra the same step gives 30.16)
    np.testing.assert_allclose(got[1], want[1], rtol=1.5e-1)
    # from the third step on the two trajectories separate (discontinuous label
    # assignment amplifies last-bit differences of the float atomics): finite is all
    # that can be asserted


def test_graphed_training_replays_eager_steps():
    """One hipGraph replay per step (scan2cap_amd/graphs.py) with the geometry one batch
    ahead in static slots (pipeline.GeometrySlots): same losses as the eager steps."""
    from scan2cap_amd.graphs import GraphedCallable
    from scan2cap_amd.pipeline import GeometrySlots
    bench, wl, model, opt, dd, cfg, dev = _setup()
    # (slots first: they size the persistent GEMM grid, which fixes the order of the BatchNorm
    # partial sums for the eager steps and the replays alike)
    slots = GeometrySlots(model.backbone_net, dd["point_clouds"], 1)
    eager = bench.make_step(model, wl, cfg, opt, None, dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    ref = [float(eager(dd).detach()) for _ in range(3)]
    model.load_state_dict(state)
    opt2 = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True,
                            fused=True)
    step = bench.make_step(model, wl, cfg, opt2, None, dev)

    def body():
        d = dict(dd)
        d["_geometry"] = slots.geometry(0)
        return step(d)
    # capture warms up with real steps: restore the weights / a fresh optimizer after it
    g = GraphedCallable(body).capture()
    model.load_state_dict(state)
    for st in opt2.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()
    slots.refill(0, dd["point_clouds"])
    losses = []
    for _ in range(12):
        slots.acquire(0)
        losses.append(float(g().detach()))
        slots.release(0)
        slots.refill(0, dd["point_clouds"])
    assert all(np.isfinite(losses))
    # float atomics make two runs differ in the last bits; by the third step the
    # discontinuous label assignment may amplify that (see above)
    # step 1: same weights, same kernels.  Step 2 runs on weights that differ by the float
    # atomics' last bits of ONE update, and at random init that is amplified without bound
    # (1.4 % seen): a sanity bound only -- that a replay reproduces the eager
    # GRADIENTS is asserted by test_graph_replays_reproduce_the_eager_gradients below
    np.testing.assert_allclose(losses[0], ref[0], rtol=1e-5)
    np.testing.assert_allclose(losses[1], ref[1], rtol=1e-1)


def test_two_stage_graph_pair_matches_single_backward():
    """The N>1 step of bench.py -- forward + captioner/graph backward in one hipGraph, the
    detector's backward in a second one (the early gradient bucket goes on the wire in
    between), gradients packed into two flat buckets -- gives the gradients of the plain
    eager `loss.backward()`."""
    from scan2cap_amd.graphs import GraphedPair
    from scan2cap_amd.loss_helper import get_scene_cap_loss
    from scan2cap_amd.parallel import (BucketedGradAllReduce, TwoStageBackward,
                                       split_detector_captioner)
    bench, wl, model, opt, dd, cfg, dev = _setup()
    # (the eager reference runs AFTER the capture: a backward on the default stream first would
    # leave AccumulateGrad nodes bound to that stream, and syncing with the legacy stream inside
    # a capture crashes hipStreamEndCapture)

    early, late = split_detector_captioner(model)
    ddp = BucketedGradAllReduce(model, [early, late])
    two = TwoStageBackward(early, late)

    def first():                                  # (no optimizer step: the weights stay put)
        ddp.drop_grads()
        x = model(dict(dd), use_tf=True, is_eval=False)
        x = get_scene_cap_loss(x, dev, cfg, None)
        two.stage1(x)
        ddp.pack_grads(0)
        return x["loss"]

    def second():
        two.stage2()
        ddp.pack_grads(1)
    pair = GraphedPair(first, second).capture()
    runs = []
    for _ in range(3):
        loss = pair.replay_first()
        ddp.reduce(0, async_op=True)              # world == 1: no-op, same call sequence
        pair.replay_second()
        ddp.reduce(1, async_op=True)
        ddp.wait()
        torch.cuda.synchronize()
        runs.append((float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()}))
    # reference: plain eager backward from the same weights
    for p in model.parameters():
        p.grad = None
    d = model(dict(dd), use_tf=True, is_eval=False)
    d = get_scene_cap_loss(d, dev, cfg, None)
    d["loss"].backward()
    want = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    want_loss = float(d["loss"])
    for loss, grads in runs:
        np.testing.assert_allclose(loss, want_loss, rtol=1e-5)
        for n, g in grads.items():
            if n not in want:
                assert float(g.abs().max()) == 0.0, n            # unused: reduced as zeros
                continue
            scale = max(1.0, float(want[n].abs().max()))
            # float atomics: last-bit noise between two evaluations of the same backward
            assert float((g - want[n]).abs().max()) <= 2e-3 * scale, n


def test_graph_replays_reproduce_the_eager_gradients():
    """Regression test of the round-2 finding: `hipMemsetAsync` nodes inside a captured hipGraph
    are not ordered against the kernels that accumulate into the zeroed buffers (ROCm 7.2):
    with them 88 % of the replayed steps returned non-finite gradients while eager steps were
    fine.  Every zero-fill of the library is a kernel now (csrc/s2c_common.h: zero_async): 40
    replays from identical weights must reproduce the eager gradients up to the last-bit noise
    of the float atomics."""
    from scan2cap_amd.graphs import GraphedCallable
    from scan2cap_amd.loss_helper import get_scene_cap_loss
    bench, wl, model, opt, dd, cfg, dev = _setup()

    def step():
        model.zero_grad(set_to_none=True)
        d = model(dict(dd), use_tf=True, is_eval=False)
        d = get_scene_cap_loss(d, dev, cfg, None)
        d["loss"].backward()
        return d["loss"]
    g = GraphedCallable(step).capture()
    graph_grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    worst = 0.0
    snaps = []
    for _ in range(40):
        loss = float(g().detach())
        assert np.isfinite(loss)
        snaps.append({n: t.clone() for n, t in graph_grads.items()})
    float(step().detach())                       # eager reference (after the capture)
    ref = {n: p.grad.detach() for n, p in model.named_parameters() if p.grad is not None}
    assert set(ref) == set(graph_grads)
    for snap in snaps:
        for n, t in snap.items():
            assert torch.isfinite(t).all(), n
            worst = max(worst, float((t - ref[n]).abs().max()) / max(1.0, float(ref[n].abs().max())))
    assert worst < 1e-3, worst

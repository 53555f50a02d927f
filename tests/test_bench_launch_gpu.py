"""The N>1 path on ONE leased GPU: bench.py's own launcher (`python bench.py --gpus 2`, no
torch.distributed.run around it) and the two-graph data-parallel step with two REAL ranks.
RCCL refuses two ranks on one device, so the ranks talk gloo here (S2C_DIST_BACKEND=gloo);
everything else -- rank spawning, env:// rendezvous, the broadcast of the weights, two flat
gradient buckets, forward + captioner backward / detector backward as two hipGraphs with the
first bucket's asynchronous all-reduce enqueued in between, optimizer graph, the rank-0 JSON
line -- is the code the driver runs over RCCL on an 8-GPU node (cfg4).  The reference has no
multi-GPU path (scripts/train.py:132), so the check is against this build's own single-rank
gradients: averaged over the ranks they must equal the mean of the per-shard gradients."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_self_launches_two_ranks_on_one_gpu():
    env = dict(os.environ, S2C_DIST_BACKEND="gloo", S2C_BENCH_WINDOWS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--steps", "3", "--warmup", "2"], env=env, capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    # round 6: stdout is the JSON line and NOTHING else (gloo's "[Gloo] Rank ..." line and RCCL's version
    # banner go to stderr: bench._main_with_one_line_stdout)
    assert len(res.stdout.strip().splitlines()) == 1, res.stdout[:500]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16
    assert line["config"]["parallelism"] == "dp2" and line["scaling"] == "weak"
    d = line["ddp"]
    assert d["ranks_seen"] == 2 and d["world_size"] == 2 and d["backend"] == "gloo"
    assert sum(d["bucket_bytes"]) == line["config"]["grad_allreduce_bytes"] > 20e6
    assert d["bucket_bytes"][0] > 3 * d["bucket_bytes"][1]       # captioner bucket first
    assert len(d["allreduce_alone_ms"]) == 2 and d["stage2_graph_ms"] > 0
    assert np.isfinite(line["value"]) and line["value"] > 0
    assert "cpu_baseline" not in line                            # rank 0, N == 1 only


def test_bench_self_launches_eight_ranks_on_one_gpu():
    """The driver's `python bench.py --gpus 8` end to end on the one leased GPU (gloo, the small
    cfg1 workload): eight ranks rendezvous on their own port, every rank is seen by the collective,
    only rank 0 prints, the OpenMP pools are divided between the ranks, and bucket 0's timing
    against the detector's backward graph is reported."""
    env = dict(os.environ, S2C_DIST_BACKEND="gloo", S2C_BENCH_WINDOWS="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "OMP_NUM_THREADS"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "cfg1",
           "--steps", "2", "--warmup", "1"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    if res.returncode != 0 and "HSA_STATUS_ERROR" in res.stderr:
        # Eight processes on ONE device: one rank's queue aborts ("HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION")
        # inside the replayed step in ~1 of 10 cold starts (tools/repro_cold_start.sh: 1/12, 4/40, also
        # 4/40 with every large-LDS kernel off and 1/40 without the overlapped all-reduce) -- and in 0 of
        # 25 + 25 runs of tools/repro_oversubscribe.py, eight processes replaying a hipGraph of plain torch
        # kernels or of this library's train step WITHOUT torch.distributed: it needs gloo's device-tensor
        # collectives beside eight processes' graphs on one device, never seen with one rank per device
        # (DESIGN section 7, item 7).  Not a wrong result: repeat ONCE; any other failure, or a second
        # abort, fails the test.
        print("first attempt aborted by the runtime:\n" + res.stderr[-1500:])
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0): %d" % len(lines)
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["config"]["parallelism"] == "dp8"
    assert line["config"]["global_batch"] == 8 * line["config"]["scenes_per_gpu"]
    d = line["ddp"]
    assert d["ranks_seen"] == 8 and d["world_size"] == 8 and d["backend"] == "gloo"
    assert isinstance(d["bucket0_hidden"], bool)
    assert d["stage2_graph_ms"] > 0 and len(d["allreduce_alone_ms"]) == 2
    assert np.isfinite(line["value"]) and line["value"] > 0


def test_a_dying_rank_takes_the_launch_down():
    """bench.py's launcher: a rank that exits non-zero after the rendezvous terminates the other
    ranks (which would otherwise wait in a collective for ever) and its code is returned."""
    env = dict(os.environ, S2C_DIST_BACKEND="gloo", S2C_BENCH_FAIL_RANK="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                          "--workload", "cfg1", "--steps", "1", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 17, (res.returncode, res.stderr[-2000:])
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def _rank_main(rank, world, port, out_path):
    """One rank of the two-graph step (what bench.py's `replay` does), small shapes."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), S2C_DIST_BACKEND="gloo")
    import torch.distributed as dist
    import bench
    from scan2cap_amd.graphs import GraphedPair
    from scan2cap_amd.loss_helper import get_scene_cap_loss
    from scan2cap_amd.parallel import (BucketedGradAllReduce, TwoStageBackward, init_from_env,
                                       split_detector_captioner)
    r, w, _ = init_from_env()
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    wl = dict(B=2, N=4096, C=4, K=64, V=200, train=True, desc="test")
    vocabulary, embeddings, table = bench.make_vocab(wl["V"])
    msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
    torch.manual_seed(10 + rank)              # different init per rank: the broadcast fixes it
    model = bench.build_model(wl, vocabulary, embeddings, msa).to(dev).train()
    early, late = split_detector_captioner(model)
    ddp = BucketedGradAllReduce(model, [early, late])
    two = TwoStageBackward(early, late)
    dd = bench.to_device(bench.make_batch(wl, wl["B"], 70 + rank, table, msa), dev)
    cfg = bench.LossConfig(msa)
    weights = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}

    def first():
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
    model.load_state_dict({k: v.to(dev) for k, v in weights.items()})   # BN running stats
    for _ in range(2):                                # a replay after a replay, as in training
        loss = pair.replay_first()
        ddp.reduce(0, async_op=True)
        pair.replay_second()
        ddp.reduce(1, async_op=True)
        ddp.wait()
    torch.cuda.synchronize()
    torch.save({"weights": weights, "loss": float(loss),
                "grads": {n: p.grad.detach().cpu() for n, p in model.named_parameters()}},
               out_path % rank)
    dist.destroy_process_group()


def test_two_ranks_average_the_single_rank_gradients(tmp_path):
    import torch.multiprocessing as mp
    import bench
    from scan2cap_amd.loss_helper import get_scene_cap_loss
    out_path = str(tmp_path / "rank%d.pt")
    mp.spawn(_rank_main, args=(2, _free_port(), out_path), nprocs=2, join=True)
    got = [torch.load(out_path % r) for r in range(2)]
    for k, v in got[0]["weights"].items():
        assert torch.equal(v, got[1]["weights"][k]), k            # rank 0's weights everywhere
    assert got[0]["loss"] != got[1]["loss"]                       # each rank its own shard
    for n, g in got[0]["grads"].items():
        assert torch.equal(g, got[1]["grads"][n]), n              # same averaged gradient
    # single-rank gradients of both shards from the same weights, plain eager backward
    dev = torch.device("cuda")
    wl = dict(B=2, N=4096, C=4, K=64, V=200, train=True, desc="test")
    vocabulary, embeddings, table = bench.make_vocab(wl["V"])
    msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
    model = bench.build_model(wl, vocabulary, embeddings, msa).to(dev).train()
    cfg = bench.LossConfig(msa)
    mean = None
    for rank in range(2):
        model.load_state_dict(got[0]["weights"])
        model.zero_grad(set_to_none=True)
        dd = bench.to_device(bench.make_batch(wl, wl["B"], 70 + rank, table, msa), dev)
        d = get_scene_cap_loss(model(dict(dd), use_tf=True, is_eval=False), dev, cfg, None)
        d["loss"].backward()
        np.testing.assert_allclose(float(d["loss"]), got[rank]["loss"], rtol=1e-5)
        g = {n: (p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(p).cpu())
             for n, p in model.named_parameters()}
        mean = g if mean is None else {n: (mean[n] + g[n]) / 2 for n in g}
    for n, want in mean.items():
        scale = max(1.0, float(want.abs().max()))
        # float atomics: last-bit noise between two evaluations of the same backward
        assert float((got[0]["grads"][n] - want).abs().max()) <= 2e-3 * scale, n

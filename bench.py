#!/usr/bin/env python
"""bench.py -- scenes/s of the Scan2Cap hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic scenes:
CapNet forward + get_scene_cap_loss + backward + Adam (the reference's train
step, lib/solver.py:293-302) on the workload BASELINE.json's metric is quoted
on: B=8 scenes per GPU, N=40000 points, XYZ + multiview(128) + normal + height,
256 proposals, --use_topdown --use_relation --num_graph_steps 2 --num_locals 10
(BASELINE.json configs[2]; configs[1] is the forward-only detection case,
`--workload cfg2`).  Inputs are resident in HBM before the timed region.

Scenes shard data-parallel: one process per GPU, B scenes each (weak scaling),
one flat-bucket gradient all-reduce per step over RCCL (scan2cap_amd/parallel.py).

Prints ONE JSON line (rank 0).  `roofline` describes the kernel with the largest
total time per step -- whichever stream it runs on -- (HIP events on the launch stream,
live in this run); `roofline_gemm` the hand-written MFMA GEMM entry points taken together;
`roofline_main_stream` the largest one of the main stream when the top
one is overlapped on a side stream; `roofline_named` north_star's ball_query + grouping
pair; `fed` the same step on a new device-assembled batch every step;
`cpu_baseline` is the same step run through the CPU oracle ops + torch CPU on a
bounded sample (rank 0, N=1 only) -- a reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from scan2cap_amd import _C  # noqa: E402
from scan2cap_amd.loss_helper import get_scene_cap_loss  # noqa: E402
from scan2cap_amd.models import CapNet  # noqa: E402
from scan2cap_amd.parallel import (BucketedGradAllReduce, FlatGradAllReduce,  # noqa: E402
                                   TwoStageBackward, init_from_env, split_detector_captioner)
from scan2cap_amd.synthetic import (aim_reference_boxes_at_proposals, scene_labels,  # noqa: E402
                                    scene_xyz)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # dense f32-input MFMA peak
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
# the rows GEMMs form an fp32-accurate product from 6 bf16 MFMAs (csrc/s2c_gemm.hip)
GEMM_SPLIT = os.environ.get("S2C_GEMM_SPLIT", "1") != "0"
MFMA_GEMM_PEAK_TF = MFMA_BF16_PEAK_TF / 6.0 if GEMM_SPLIT else MFMA_F32_PEAK_TF

WORKLOADS = {
    # name: (B, N, feature channels C, proposals K, vocab V, train?)
    "cfg3": dict(B=8, N=40000, C=132, K=256, V=3500, train=True,
                 desc="B=8 N=40000 XYZ+multiview(128)+normal+height, 256 proposals, "
                      "topdown+relation graph(2 steps, 10 locals), train step"),
    "cfg2": dict(B=8, N=40000, C=4, K=256, V=3500, train=False,
                 desc="B=8 N=40000 XYZ+normal+height, VoteNet 256 proposals, forward only"),
    "cfg1": dict(B=1, N=4096, C=1, K=32, V=3500, train=True,
                 desc="1 scene XYZ+height N=4096, 32 proposals"),
    "cfg3e": dict(B=8, N=40000, C=132, K=256, V=3500, train=False, caption=True,
                  desc="cfg3 shapes, forward only: detection + graph + greedy decode "
                       "of every proposal"),
    "cfg5": dict(B=16, N=80000, C=132, K=512, V=3500, train=False, caption=True,
                 desc="B=16 N=80000 XYZ+multiview+normal+height, 512 proposals, "
                      "relation graph + greedy top-down decode (len 30) of every "
                      "proposal, forward only"),
}


class LossConfig(object):
    def __init__(self, msa):
        self.num_heading_bin, self.num_size_cluster, self.num_class = 1, 18, 18
        self.mean_size_arr = msa


# entry point -> device kernels it launches (for the PMC traffic lookup)
_GEMM_PLAIN = tuple("rows_gemm%s_kernel<%s, %d>" % (v, g, p)
                    for v in ("", "_x3") for g in ("2, 2", "4, 1") for p in (0, 1)) + \
    ("rows_gemm_c64_kernel<0>", "rows_gemm_c64_kernel<1>")
_GEMM_GATHER = tuple("rows_gemm%s_kernel<%s, 2>" % (v, g)
                     for v in ("", "_x3") for g in ("2, 2", "4, 1")) + ("rows_gemm_c64_kernel<2>",)
KERNELS_OF = {
    "s2c_bn_relu_bwd": ("bn_bwd_stats_kernel", "bn_bwd_apply_kernel"),
    "s2c_bn_relu_max_bwd": ("pool_bwd_stats_kernel", "pool_bwd_apply_kernel"),
    "s2c_bn_relu": ("bn_relu_kernel",),
    "s2c_bn_relu_max": ("bn_relu_max_kernel",),
    "s2c_bn_train_stats": ("col_stats_kernel",),
    "s2c_rows_gemm": _GEMM_PLAIN,
    "s2c_sa_gather_gemm": _GEMM_GATHER,
    "s2c_sa_point_gemm": ("point_gemm_kernel",),
    "s2c_sa_gather_add": ("sa_gather_add_kernel",),
    "s2c_sa_fused_eval": ("sa_fused_eval_kernel",),
    "s2c_sa_gather_rows": ("sa_gather_rows_kernel",),
    "s2c_sa_scatter_rows": ("sa_scatter_rows_kernel",),
    "s2c_sa_scatter_sum": ("sa_scatter_sum_kernel",),
    "s2c_fp_interp_rows": ("fp_interp_rows_kernel",),
    "s2c_fp_interp_rows_grad": ("fp_interp_rows_grad_kernel",),
    "s2c_small_linear": ("small_linear_kernel",),
    "s2c_small_linear_pair": ("small_linear_kernel",),
    "s2c_gru_fwd": ("gru_fwd_kernel",),
    "s2c_attn_bwd": ("attn_bwd_kernel",),
    "s2c_attn_bwd_x2": ("attn_bwd_x2_kernel",),
    "s2c_attn_x2_fwd": ("attn_x2_kernel",),
    "s2c_decoder_fwd_persist": ("decoder_fwd_persist_kernel",),
    "s2c_decoder_bwd_persist": ("decoder_bwd_persist_kernel",),
    "s2c_planes_gemm": ("planes_gemm_kernel",),
    "s2c_mgemm": ("mgemm_kernel",),
    "s2c_weight_grad": ("dw_x3_kernel",),
    "s2c_weight_grad_multi": ("dw_x3_multi_kernel",),
    "s2c_weight_grad_stream": ("dw_private_kernel",),
    "s2c_bn_bwd_dx_dw64": ("bn_bwd_dx_dw64_kernel",),
    "s2c_small_gemm": ("sgemm_kernel",),
    "s2c_attn_local_fwd_planes": ("attn_local_kernel",),
    "s2c_ball_query": ("ball_query_kernel",),
    "s2c_ball_query_grid": ("bq_grid_build_kernel", "ball_query_grid_kernel"),
    "s2c_furthest_point_sampling_bucketed": ("fps_bucket_kernel",),
    "s2c_furthest_point_sampling_cells": ("fps_cells_prep_kernel", "fps_cells_rounds_kernel"),
    "s2c_furthest_point_sampling_small": ("fps_small_kernel",),
}


def pmc_traffic(entry):
    """Mean HBM bytes per launch of `entry` from the committed rocprofv3 PMC
    summary (profiles/rNN_pmc_bench.json, newest round: FETCH_SIZE and WRITE_SIZE collected in
    separate passes over `bench.py --no-graph --steps 2 --warmup 1`).  gfx950
    correction (MI355X_MICROARCH.md, re-calibrated in
    profiles/r01_pmc_sa_kernels_calibration.json on bn_relu, whose byte count is
    exact): FETCH_SIZE counts half of a 16-byte-per-lane streaming read, so
    bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  None when not available."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_bench.json")))
    path = found[-1] if found else ""          # the newest round's summary
    keys = KERNELS_OF.get(entry)
    if not keys or not os.path.exists(path):
        return None
    try:
        table = json.load(open(path))
    except Exception:
        return None
    total, hit = 0.0, False
    for kname, c in table.items():
        if any(k in kname for k in keys) and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            n = c["FETCH_SIZE"]["dispatches"]
            total += (2 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024 * n
            hit = True
    if not hit:
        return None
    calls = max(c["FETCH_SIZE"]["dispatches"] for kname, c in table.items()
                if any(k in kname for k in keys) and "FETCH_SIZE" in c)
    return total / max(calls, 1)


def pmc_mfma_busy(name_parts, workload="bench"):
    """MFMA-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs)) of the kernels
    whose names contain one of `name_parts`, time-weighted, from the committed SQ-counter summary
    (profiles/rNN_pmc_sq_<workload>.json, newest round -- "bench" = the cfg3 train step, "cfg3e" / "cfg5"
    = the evaluation workloads; each its own rocprofv3 pass: tools/make_profiles.sh).
    None when not available."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_sq_%s.json" % workload)))
    if not found:
        return None
    try:
        table = json.load(open(found[-1]))
    except Exception:
        return None
    num = den = 0.0
    per = {}
    for kname, e in table.items():
        if not any(p in kname for p in name_parts) or "mfma_util" not in e:
            continue
        cyc = e["mean"].get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * e["dispatches"]
        num += e["mfma_util"] * cyc
        den += cyc
        import re
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", kname)
        per[m.group(1) if m else kname[:60]] = round(e["mfma_util"], 4)
    if den <= 0:
        return None
    return {"busy": num / den, "source": os.path.basename(found[-1]), "per_kernel": per}


def make_vocab(V, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    words = ["pad_", "unk", "sos", "eos"] + ["w%d" % i for i in range(V - 4)]
    vocabulary = {"word2idx": {w: i for i, w in enumerate(words)},
                  "idx2word": {str(i): w for i, w in enumerate(words)}}
    table = (rng.standard_normal((V, 300)) * 0.3).astype(np.float32)
    embeddings = {w: table[i] for i, w in enumerate(words)}
    return vocabulary, embeddings, table


def make_batch(wl, B, seed, table, msa):
    """Synthetic data_dict (numpy) with the reference's keys/dtypes."""
    N, C, V = wl["N"], wl["C"], wl["V"]
    rng = np.random.Generator(np.random.PCG64(seed))
    xyz = scene_xyz(B, N, seed=seed, mode="volume", adversarial=True)
    feats = []
    if C >= 4:
        nrm = rng.standard_normal((B, N, 3)).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True) + 1e-9
        feats.append(nrm)
    if C >= 132:
        mv = np.maximum(rng.standard_normal((B, N, 128)).astype(np.float32) * 0.5, 0)
        feats.append(mv)
    height = xyz[..., 2:3] - np.percentile(xyz[..., 2], 0.99)
    feats.append(height.astype(np.float32))
    pc = np.concatenate([xyz] + feats, -1).astype(np.float32)
    assert pc.shape[-1] == 3 + C, pc.shape
    T = 32
    lang_len = rng.integers(8, T + 1, B).astype(np.int64)
    lang_ids = np.zeros((B, T), np.int64)
    for b in range(B):
        toks = [2] + list(rng.integers(4, V, lang_len[b] - 2)) + [3]
        lang_ids[b, :len(toks)] = toks
    lang_feat = table[lang_ids] * (lang_ids != 0)[..., None]
    out = dict(point_clouds=pc, lang_feat=lang_feat.astype(np.float32),
               lang_len=lang_len, lang_ids=lang_ids)
    labels = scene_labels(xyz, num_boxes=32, seed=seed, mean_size_arr=msa)
    out.update(labels)
    out["ref_box_corner_label"] = labels["gt_box_corner_label"][:, 0].copy()
    return out


def build_model(wl, vocabulary, embeddings, msa):
    cap = wl["train"] or wl.get("caption", False)
    return CapNet(num_class=18, vocabulary=vocabulary, embeddings=embeddings,
                  num_heading_bin=1, num_size_cluster=18, mean_size_arr=msa,
                  input_feature_dim=wl["C"], num_proposal=wl["K"],
                  num_locals=wl.get("num_locals", 10) if cap else -1,
                  no_caption=not cap, use_topdown=True,
                  query_mode="corner", graph_mode="edge_conv",
                  # --num-locals -1 = the reference's DEFAULT command line (scripts/train.py:318-323:
                  # no --use_relation, --num_graph_steps 0): its relational graph takes a positive
                  # num_locals only (graph_module.py:216: torch.topk(pc_dist, self.num_locals))
                  num_graph_steps=2 if cap and wl.get("num_locals", 10) > 0 else 0,
                  use_relation=cap and wl.get("num_locals", 10) > 0)


def make_feeder(wl, dd, depth, msa, device, num_scenes, rank, stream=None):
    """--feed builder: synthetic ScanNet-like scenes resident in HBM (150k vertices, the
    workload's channels), `depth` static copies of the batch tensors the builder writes, and
    the item picker (round-robin over scenes; the language entries of `dd` are reused so
    that the decoder length of every step equals the resident run's)."""
    from scan2cap_amd import scene_builder as sb
    from scan2cap_amd.synthetic import make_scene
    B, N, C = wl["B"], wl["N"], wl["C"]
    mvw = 128 if C >= 132 else 0
    scenes = [make_scene(900 + 31 * rank + i, 140000 + 1777 * i, mvw, num_instances=48)
              for i in range(num_scenes)]
    store = sb.SceneStore(device, multiview_width=mvw)
    for i, sc in enumerate(scenes):
        store.add_scene(i, sc["mesh_vertices"], sc["instance_labels"], sc["semantic_labels"],
                        sc["instance_bboxes"], sc.get("multiview"))
    store.finalize()
    builder = sb.SceneBatchBuilder(store, msa, num_points=N, use_color=False, use_height=True,
                                   use_normal=C >= 4, use_multiview=C >= 132, augment=True)
    assert builder.Cout == 3 + C
    probe = builder.build([0] * B, [0] * B, builder.draw([0] * B,
                                                         rng=np.random.RandomState(1)))
    produced = [k for k in probe if k in dd and torch.is_tensor(dd[k])
                and dd[k].shape == probe[k].shape and dd[k].dtype == probe[k].dtype]
    assert "point_clouds" in produced and "vote_label" in produced, produced
    sets = []
    for p in range(depth):
        d = dict(dd)
        d["_follow"] = False          # the builder writes the described box of every new batch
        for k in produced:
            d[k] = dd[k].clone()
        sets.append(d)
    feeder = sb.BatchFeeder(builder, None, sets, device_choices=True,
                            rng=np.random.RandomState(7 + rank), stream=stream)
    box_ids = [sc["instance_bboxes"][:, 7].astype(int) for sc in scenes]

    def pick(k):
        ids = [(k * B + b) % num_scenes for b in range(B)]
        return ids, [int(box_ids[s][(k + b) % len(box_ids[s])]) for b, s in enumerate(ids)]

    return feeder, sets, pick


def to_device(batch, device):
    dd = {k: torch.from_numpy(v).to(device) for k, v in batch.items()}
    dd["_num_words"] = int(batch["lang_len"].max())   # host-known: no device read
    return dd


# what the last executed train step saw of the caption branch: tensors (static outputs of a
# captured graph, rewritten by every replay), read once after the timed region
CAPTION_PROBE = {}


def probe_caption(d, slot=0):
    if "cap_loss" in d and "good_bbox_masks" in d:
        CAPTION_PROBE[slot] = (d["cap_loss"].detach(), d["good_bbox_masks"])


# Resident batch only: the described box of every scene FOLLOWS the model -- after each forward the
# box predicted for proposal 0 becomes the next step's `ref_box_corner_label` (one device-side copy
# into the static input buffer).  A reference run captions with a pretrained detector whose proposals
# overlap the annotated boxes; a randomly initialised one never reaches IoU 0.25 with a fixed box for
# more than a few updates, `good_bbox_masks` goes all False and the caption loss and every
# captioner / relation-graph gradient are exactly zero (lib/loss_helper.py:189-230).
FOLLOW_MODEL = os.environ.get("S2C_BENCH_FOLLOW", "1") != "0"


def follow_model(d, static):
    if FOLLOW_MODEL and static.get("_follow", False) and "bbox_corner" in d:
        with torch.no_grad():
            static["ref_box_corner_label"].copy_(d["bbox_corner"][:, 0])


def make_step(model, wl, cfg_loss, optimizer, ddp, device, two_stage=None):
    """Eager step (also the un-captured body of the graphed step)."""
    def train_step(dd):
        slot = dd.get("_slot", 0)
        dd = dict(dd)
        # fresh gradient tensors every step: no zero-fill and no accumulate
        # kernels (2 launches per parameter with preallocated .grad)
        if ddp is not None:
            ddp.drop_grads()
        else:
            optimizer.zero_grad(set_to_none=True)
        dd = model(dd, use_tf=True, is_eval=False)
        dd = get_scene_cap_loss(dd, device, cfg_loss, None, detection=True,
                                caption=True, orientation=False, distance=False)
        probe_caption(dd, slot)
        follow_model(dd, dd)
        dd["loss"].backward()
        if two_stage is not None:
            ddp.pack_grads(0)
            ddp.pack_grads(1)
            ddp.reduce(0)
            ddp.reduce(1)
            ddp.wait()
        elif ddp is not None:
            ddp.pack_grads()
            ddp.reduce()
        optimizer.step()
        return dd["loss"]

    def fwd_step(dd):
        with torch.no_grad():
            return model(dict(dd), use_tf=False, is_eval=True)["objectness_scores"]

    return train_step if wl["train"] else fwd_step


def cpu_baseline(wl, vocabulary, embeddings, table, msa, sample_B=1):
    """The same step through the CPU oracle ops (oracle/s2c_oracle.c, OpenMP) +
    torch CPU, on a bounded sample of the workload.  kind = "port"."""
    from oracle import torch_ext
    from scan2cap_amd.pointnet2 import _ext
    saved = {n: getattr(_ext, n) for n in torch_ext.NAMES}
    # more threads than this only add OpenMP fork/join and NUMA noise on the
    # 256-core hosts of the GPU boxes (measured: 143 s/step at 256 threads)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    try:
        for n in torch_ext.NAMES:
            setattr(_ext, n, getattr(torch_ext, n))
        torch.manual_seed(0)
        model = build_model(wl, vocabulary, embeddings, msa)
        cfg_loss = LossConfig(msa)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        model.train(wl["train"])
        step = make_step(model, wl, cfg_loss, opt, None, torch.device("cpu"))
        dd = to_device(make_batch(wl, sample_B, 4242, table, msa), "cpu")
        step(dd)                                  # warm-up (allocator, OpenMP pool, page-in)
        reps = 3
        t0 = time.time()
        for _ in range(reps):
            step(dd)
        dt = (time.time() - t0) / reps
    finally:
        for n, f in saved.items():
            setattr(_ext, n, f)
    return {"value": sample_B / dt, "unit": "scenes/s", "cores": cores,
            "kind": "port",
            "sample": "3 steps (after 1 warm-up) of the same workload at B=%d (N=%d, C=%d, K=%d), "
                      "%.1f s/step wall; oracle C ops (OpenMP) + torch CPU fp32"
                      % (sample_B, wl["N"], wl["C"], wl["K"], dt)}


# FPS is a chain of dependent rounds: per round one pick -> min-distance update of the points
# it can affect -> arg-max.  Floor used for the latency regime: the register-resident
# single-barrier kernel of csrc/s2c_fps_small.hip sustains 0.5 us per round at N <= 8192
# (DESIGN 4.1), i.e. the dependent-instruction chain of one round with no memory traffic.
FPS_ROUND_FLOOR_US = 0.5
_FPS_ROUNDS = {"s2c_furthest_point_sampling_bucketed": 2047,       # SA1: npoint - 1
               "s2c_furthest_point_sampling_cells": 2047}


def roofline_of(top, ms_per_step, wl):
    """`roofline` object of one kernel-table entry (the contract's keys; `traffic` from the
    committed PMC summary).  GEMM kernels report the roof they sit closer to; FPS reports
    its HBM fraction (as the contract asks: algorithmic bytes / duration) AND the regime it
    is really in: a latency-bound chain of rounds."""
    hbm = {"bound": "hbm", "achieved": top["alg_GBps"], "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": top["alg_GBps"] / HBM_PEAK_GBS}
    roof = dict(hbm)
    if top["alg_TFLOPs"] > 0:
        mfma = {"bound": "mfma", "achieved": top["alg_TFLOPs"],
                "peak": MFMA_GEMM_PEAK_TF, "unit": "TFLOP/s",
                "frac": top["alg_TFLOPs"] / MFMA_GEMM_PEAK_TF,
                "note": ("fp32-equivalent FLOPs; peak = bf16 MFMA peak / 6 (bf16x3 "
                         "split products)") if GEMM_SPLIT else "fp32 MFMA"}
        roof = dict(mfma, other_roof=hbm) if mfma["frac"] > hbm["frac"] \
            else dict(hbm, other_roof=mfma)
    roof = dict({"kernel": top["kernel"]}, **roof)
    roof.update({"traffic": pmc_traffic(top["kernel"]),
                 "alg_bytes_per_launch": top["alg_bytes_per_launch"],
                 "avg_launch_us": top["avg_us"],
                 "share_of_step": top["ms_per_step"] / ms_per_step})
    if top["kernel"] in _DECODER_CHAIN:
        roof["regime"] = {"kind": "latency",
                          "note": "teacher-forced decoder: %d rows x ~30 strictly sequential "
                                  "steps of 5 + 5 dependent mat-vec stages -- two persistent "
                                  "kernels exchanging tagged vectors between 128 workgroups "
                                  "(~2.5 us per stage), or 10 launches per step (DESIGN 4.4)"
                                  % wl["B"]}
    rounds = _FPS_ROUNDS.get(top["kernel"])
    if rounds:
        us = top["avg_us"] / rounds
        roof["regime"] = {"kind": "latency", "rounds_per_launch": rounds,
                          "us_per_round": us, "rounds_per_s": 1e6 / us,
                          "floor_us_per_round": FPS_ROUND_FLOOR_US,
                          "frac_of_floor": FPS_ROUND_FLOOR_US / us,
                          "note": "serial rounds (one workgroup per scene, %d scenes in "
                                  "flight); overlapped with the previous step on a side "
                                  "stream" % wl["B"]}
    return roof


# the FUSED contract of the same five stages (query -> gather -> MLP -> max, nothing
# materialised: compulsory inputs + pooled outputs + idx), MB per step at the workload's own batch size
SURVEY_FUSED_MB = {"cfg3": (232.5, 8), "cfg3e": (232.5, 8), "cfg2": (68.7, 8), "cfg5": (813.0, 16), "cfg1": (7.5, 1)}


def named_roofline(table_k, workload=None, batch=None):
    """north_star's named target: ball_query + grouping as a fraction of the HBM roof.
    Grouping is fused into the first GEMM of every set-abstraction stage
    (s2c_sa_gather_gemm: the gathered (rows, 3+C) operand is never materialised), so the
    pair is  s2c_ball_query (+ its grid variant for SA1) + s2c_sa_gather_gemm  with the op-contract algorithmic bytes
    of both (xyz + centres + idx;  unique source rows + idx + Y)."""
    parts = [k for k in table_k if k["kernel"] in ("s2c_ball_query", "s2c_ball_query_grid",
                                                   "s2c_sa_gather_gemm",
                                                   # training path (round 4): the first layer in
                                                   # point space = per-point product + gather-add
                                                   "s2c_sa_point_gemm", "s2c_sa_gather_add",
                                                   "s2c_sa_gather_gemm_bn_eval",
                                                   "s2c_sa_fused_eval")]
    if not parts:
        return None
    ms = sum(k["ms_per_step"] for k in parts)
    nbytes = sum(k["alg_bytes_per_launch"] * k["calls_per_step"] for k in parts)
    gbs = nbytes / max(ms, 1e-9) / 1e6
    tflops = sum(k["alg_TFLOPs"] * k["ms_per_step"] for k in parts) / max(ms, 1e-9)
    survey = None
    if workload in SURVEY_FUSED_MB:
        mb, b0 = SURVEY_FUSED_MB[workload]
        sb = mb * 1e6 * (batch or b0) / b0
        sg = sb / max(ms, 1e-9) / 1e6
        survey = {"alg_bytes_per_step": sb, "achieved": sg, "frac": sg / HBM_PEAK_GBS,
                  "note": "SURVEY.md 8(d) fused contract (nothing between the query and the pooled output "
                          "counts); `frac` above uses the op contract of the launches that exist: xyz + "
                          "centres + idx, unique source rows + idx + the first layer's output"}
    return {"kernels": [k["kernel"] for k in parts], "bound": "hbm", "achieved": gbs,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "ms_per_step": ms, "alg_bytes_per_step": nbytes, "survey_fused_contract": survey,
            # the fully fused inference stage (s2c_sa_fused_eval) moves only its compulsory
            # bytes and sits on the matrix pipe instead: the other roof of the same pair
            "other_roof": {"bound": "mfma", "achieved": tflops, "peak": MFMA_GEMM_PEAK_TF,
                           "unit": "TFLOP/s", "frac": tflops / MFMA_GEMM_PEAK_TF},
            "parts": {k["kernel"]: {"ms_per_step": k["ms_per_step"],
                                    "alg_GBps": k["alg_GBps"], "alg_TFLOPs": k["alg_TFLOPs"]}
                      for k in parts}}


GEMM_FAMILY = ("s2c_rows_gemm", "s2c_rows_gemm_bn_relu_side", "s2c_sa_gather_gemm", "s2c_sa_point_gemm",
               "s2c_bn_bwd_gemm", "s2c_bn_bwd_gemm_next_stats", "s2c_rows_gemm_next_stats",
               "s2c_rows_gemm_bn_eval", "s2c_sa_gather_gemm_bn_eval", "s2c_sa_fused_eval",
               # round 5: the backward products that were library GEMMs until then
               "s2c_weight_grad", "s2c_weight_grad_multi", "s2c_weight_grad_stream", "s2c_small_gemm",
               "s2c_bn_bwd_dx_dw64")
_DECODER_CHAIN = ("s2c_small_linear", "s2c_small_linear_pair", "s2c_gru_fwd", "s2c_attn_fwd",
                  "s2c_attn_bwd", "s2c_gru_gates_bwd", "s2c_attn_x2_fwd", "s2c_attn_bwd_x2",
                  "s2c_decoder_fwd_persist", "s2c_decoder_bwd_persist")


def family_roofline(table_k, ms_per_step):
    """The hand-written bf16x3 MFMA GEMM entry points taken together (tiled kernel of
    csrc/s2c_gemm.hip + streaming kernel of csrc/s2c_gemm2.hip): algorithmic bytes and
    fp32-equivalent FLOPs per step over their summed kernel time."""
    parts = [k for k in table_k if k["kernel"] in GEMM_FAMILY]
    if not parts:
        return None
    ms = sum(k["ms_per_step"] for k in parts)
    nbytes = sum(k["alg_bytes_per_launch"] * k["calls_per_step"] for k in parts)
    tflops = sum(k["alg_TFLOPs"] * k["ms_per_step"] for k in parts) / max(ms, 1e-9)
    gbs = nbytes / max(ms, 1e-9) / 1e6
    return {"kernels": [k["kernel"] for k in parts], "bound": "hbm", "achieved": gbs,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "ms_per_step": ms, "share_of_step": ms / ms_per_step,
            "launches_per_step": sum(k["calls_per_step"] for k in parts),
            "other_roof": {"bound": "mfma", "achieved": tflops, "peak": MFMA_GEMM_PEAK_TF,
                           "unit": "TFLOP/s", "frac": tflops / MFMA_GEMM_PEAK_TF},
            # hardware view of the same kernels: matrix-pipe busy cycles (PMC, bf16 pipe:
            # 6 plane products per fp32 product)
            "mfma_busy": pmc_mfma_busy(("rows_stream_gemm_kernel", "rows_gemm_x3_kernel",
                                        "rows_gemm_c64_kernel", "rows_gemm_kernel", "dw_x3",
                                        "dw_private_kernel", "sgemm_kernel", "point_gemm_kernel")),
            "parts": {k["kernel"]: {"ms_per_step": k["ms_per_step"], "alg_GBps": k["alg_GBps"],
                                    "avg_launch_us": k["avg_us"]} for k in parts}}


def decode_roofline(table_k, ms_per_step, workload="cfg5"):
    """Greedy decoding (cfg3e / cfg5): the planes GEMMs of csrc/s2c_planes.hip taken together -- every
    product of the token loop -- against the bf16 matrix roof (6 plane products per fp32 product),
    with the PMC matrix-pipe busy fraction where the committed SQ summary has the kernel."""
    parts = [k for k in table_k if k["kernel"] == "s2c_planes_gemm"]
    if not parts:
        return None
    k = parts[0]
    return {"kernel": "s2c_planes_gemm", "bound": "mfma", "achieved": k["alg_TFLOPs"],
            "peak": MFMA_GEMM_PEAK_TF, "unit": "TFLOP/s", "frac": k["alg_TFLOPs"] / MFMA_GEMM_PEAK_TF,
            "note": "fp32-equivalent FLOPs of every product (K as padded to 32); peak = bf16 MFMA "
                    "peak / 6 (bf16x3 plane products)",
            "ms_per_step": k["ms_per_step"], "share_of_step": k["ms_per_step"] / ms_per_step,
            "launches_per_step": k["calls_per_step"], "avg_launch_us": k["avg_us"],
            "mfma_busy": pmc_mfma_busy(("planes_gemm_kernel",), workload)}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks here (one process
    per GPU, env:// rendezvous on 127.0.0.1 -- the same environment
    `python -m torch.distributed.run --nproc-per-node N` would give them) and pass rank 0's
    JSON line through.  A rank that dies takes the others down with it."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                      env=env, stdout=(_REAL_STDOUT[0] if r == 0
                                                       else subprocess.DEVNULL)))
    code = 0
    try:
        live = list(procs)
        while live:
            for pr in list(live):
                rc = pr.poll()
                if rc is None:
                    continue
                live.remove(pr)
                if rc != 0 and code == 0:
                    code = rc
                    for other in live:
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return code


def ddp_evidence(ddp, head, device, world, trace=lambda m: None):
    """What the N>1 step did, measured after the timed region: how many ranks the collective
    really spans, the bytes it moves, each bucket's all-reduce timed on its own and the
    graphs it is meant to hide under (bucket 0 goes on the wire between the two backward
    graphs; the detector's backward = `stage2_graph_ms` is its cover)."""
    def timed(fn, reps=5):
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / reps * 1e3], device=device,
                         dtype=torch.float64)
        if dist.is_initialized():
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    seen = torch.ones(1, device=device)
    if dist.is_initialized():
        dist.all_reduce(seen)
    trace("evidence: ranks counted")
    flats = ddp.flats if hasattr(ddp, "flats") else [ddp.flat]
    info = {"backend": dist.get_backend() if dist.is_initialized() else None,
            "world_size": dist.get_world_size() if dist.is_initialized() else 1,
            "ranks_seen": int(seen.item()),
            "bucket_bytes": [f.numel() * 4 for f in flats]}
    if hasattr(ddp, "flats"):
        def reduce_alone(i):
            ddp.reduce(i, async_op=True)
            ddp.wait(i)
        info["allreduce_alone_ms"] = [timed(lambda i=i: reduce_alone(i))
                                      for i in range(len(flats))]
    else:
        info["allreduce_alone_ms"] = [timed(lambda: ddp.reduce())]
    trace("evidence: buckets reduced alone")
    if head["pairs"]:
        # the two backward graphs and the optimizer graph, timed in their normal order (HIP
        # events on the launch stream, no collective in between): what bucket 0 hides under
        pair, g2 = head["pairs"][0], head["g2"]
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(5)]
        torch.cuda.synchronize()
        for e in ev:
            e[0].record()
            pair.replay_first()
            e[1].record()
            pair.replay_second()
            e[2].record()
            g2()
            e[3].record()
        torch.cuda.synchronize()
        trace("evidence: graphs timed")
        for k, name in enumerate(("stage1_graph_ms", "stage2_graph_ms", "optimizer_graph_ms")):
            info[name] = float(np.median([e[k].elapsed_time(e[k + 1]) for e in ev]))
        info["bucket0_hidden"] = info["allreduce_alone_ms"][0] <= info["stage2_graph_ms"]
    return info


_LAST_LINE = [None]
_REAL_STDOUT = [None]       # the caller's stdout while fd 1 points at stderr (_main_with_one_line_stdout)


def main(_emit=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="compute the xyz-only geometry stage (FPS chain, ball "
                         "queries, 3-NN) inside the step instead of one batch ahead "
                         "on a side stream")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel eagerly instead of replaying the "
                         "captured hipGraph of the step")
    ap.add_argument("--cpu-sample-scenes", type=int, default=8)
    ap.add_argument("--feed", choices=["resident", "builder"], default="resident",
                    help="resident: one batch resident in HBM is replayed (the contract's "
                         "headline); builder: every step trains on a NEW batch assembled on "
                         "the device from HBM-resident synthetic scenes "
                         "(scan2cap_amd/scene_builder.py, SURVEY 8 f3), one batch ahead")
    ap.add_argument("--feed-scenes", type=int, default=12)
    ap.add_argument("--batch", type=int, default=0,
                    help="scenes per GPU instead of the workload's (README.md:145 trains at 12, "
                         "slurm/train.job:24 at 16); the headline stays the workload's own B")
    ap.add_argument("--num-locals", type=int, default=0,
                    help="attended objects per proposal instead of the workload's 10 (scripts/train.py:322, "
                         "benchmark/predict.py:249 default to -1 = all K proposals); the headline stays "
                         "BASELINE's --num_locals 10")
    ap.add_argument("--no-fed", action="store_true",
                    help="skip the second measurement (builder-fed step) of the default run")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    rank, world, local_rank = init_from_env()
    assert world == args.gpus, "--gpus must match the launched world size"
    if os.environ.get("S2C_BENCH_FAIL_RANK") == str(rank) and world > 1:
        # fault injection for tests/test_bench_launch_gpu.py: this rank dies after the rendezvous;
        # the launcher must take the other ranks down and return its exit code
        os._exit(17)
    # one GPU per rank; several ranks share a device only under S2C_DIST_BACKEND=gloo (the
    # one-GPU rehearsal of the N>1 path)
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    _C.load()
    from scan2cap_amd.models import decoder_fused   # (ranks sharing a device: init_from_env has
    #                                                  switched the persistent decoder kernels off)

    wl = dict(WORKLOADS[args.workload])
    if args.num_locals != 0 and args.num_locals != 10:
        wl["desc"] += " [--num-locals %d%s: not the BASELINE configuration]" % (
            args.num_locals, ", no relational graph (the reference's default command)"
            if args.num_locals < 0 else "")
        wl["num_locals"] = args.num_locals
    if args.batch > 0 and args.batch != wl["B"]:
        wl["desc"] = wl["desc"].replace("B=%d" % wl["B"], "B=%d" % args.batch) + \
            " [--batch %d: not the BASELINE batch]" % args.batch
        wl["B"] = args.batch
    B = wl["B"]
    vocabulary, embeddings, table = make_vocab(wl["V"])
    msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))

    torch.manual_seed(0)
    model = build_model(wl, vocabulary, embeddings, msa).to(device)
    model.train(wl["train"])
    cfg_loss = LossConfig(msa)
    use_graph = not args.no_graph
    # scripts/train.py:138's optim.Adam(lr, weight_decay) as ONE launch over every parameter tensor
    # (scan2cap_amd/optim.py: FusedAdam, csrc/s2c_optim.hip; same update, same state_dict layout);
    # S2C_ADAM_TORCH=1: torch's fused multi-tensor Adam (3 launches + the step counters')
    if os.environ.get("S2C_ADAM_TORCH") == "1":
        optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5,
                                     capturable=use_graph, fused=True)
    else:
        from scan2cap_amd.optim import FusedAdam
        optimizer = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    # S2C_FORCE_DDP=1 exercises the multi-GPU code path (flat gradient bucket,
    # fwd/bwd graph + eager all-reduce + optimizer graph) on a single rank
    force_ddp = os.environ.get("S2C_FORCE_DDP") == "1"
    if force_ddp and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=0, world_size=1)
    # N > 1: two gradient buckets -- captioner + relation graph (80 % of the bytes, complete
    # early in backward) and detector -- the first one all-reduced on RCCL's stream while the
    # detector's backward still runs (S2C_DDP_OVERLAP=0: one flat bucket after backward)
    ddp_overlap = os.environ.get("S2C_DDP_OVERLAP", "1") != "0" and use_graph
    ddp, two_stage = None, None
    if (world > 1 or force_ddp) and wl["train"]:
        if ddp_overlap:
            early, late = split_detector_captioner(model)
            ddp = BucketedGradAllReduce(model, [early, late])
            two_stage = TwoStageBackward(early, late)
        else:
            ddp = FlatGradAllReduce(model)
    if ddp is not None and force_ddp:
        ddp.world = 2          # make reduce() issue the collective
    eager_step = make_step(model, wl, cfg_loss, optimizer, ddp, device, two_stage)
    dd = to_device(make_batch(wl, B, 42 + rank, table, msa), device)
    if wl["train"]:
        # a step that learns to caption: at random init no proposal reaches IoU 0.25 with the
        # synthetic described box (good_bbox_masks all False, caption loss and every captioner /
        # graph gradient exactly 0); describe the box the model predicts for proposal 0 instead
        dd = aim_reference_boxes_at_proposals(model, dd)
        dd["_follow"] = True          # ... and keep following it (follow_model above)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dbg = os.environ.get("S2C_DEBUG") == "1"

    def trace(msg):
        if dbg:
            torch.cuda.synchronize()
            print("[bench] ok:", msg, file=sys.stderr, flush=True)

    def read_caption_probe():
        """cap_loss / number of scenes whose target box passes the IoU threshold, of the most
        recently executed step of every slot (host read: only outside the timed region)."""
        if not CAPTION_PROBE:
            return None
        torch.cuda.synchronize()
        vals = [(float(c.item()), int(g.sum().item())) for c, g in CAPTION_PROBE.values()]
        return {"cap_loss": max(v[0] for v in vals), "good_bbox_masks_sum": max(v[1] for v in vals)}

    def measure(feed, steps, warmup):
        """Set up the pipeline for `feed` ("resident": one batch in HBM replayed; "builder":
        a new device-assembled batch per step), run `warmup` + `steps` steps, return the
        timing and a description of what ran."""
        overlap = use_graph and not args.no_overlap
        slots, depth = None, 0
        if overlap:
            # software pipeline across batches: geometry of step i+depth on a side
            # stream while step i runs; each slot has its own captured step graph that
            # reads the slot's static geometry tensors (scan2cap_amd/pipeline.py)
            from scan2cap_amd.pipeline import GeometrySlots
            # a forward-only step is shorter than one FPS chain: keep 3 batches of
            # geometry in flight; a train step (~12.6 ms) hides one chain (~5.8 ms)
            depth = 1 if wl["train"] else 3
            if feed == "builder":
                depth = 3          # three static batch sets / graphs: host-paced hand-over
            depth = int(os.environ.get("S2C_GEO_DEPTH", depth))
            # forward-only steps are shorter than one FPS chain even with 3 chains in
            # flight: compute the geometry of `group` batches per pass (stacked clouds),
            # two groups alternating
            # (detection only -- cfg2 -- is short enough that eight batches per pass pay: 5900 vs 5380
            # scenes/s at three, measured three times each; with the greedy decoder behind it the
            # step is 9-20 ms and three per pass is as good as any)
            group = int(os.environ.get("S2C_GEO_GROUP",
                                       1 if wl["train"] else (3 if wl.get("caption") else 8)))
            if group > 1:
                depth = 2 * group
            # N > 1: RCCL's kernels (one workgroup per channel) run under the detector's
            # backward; a persistent GEMM workgroup without a free CU doubles a launch
            rccl_cus = int(os.environ.get("S2C_RCCL_CUS", "16")) if ddp is not None else 0
            # beside the FPS workgroups (one per scene of a geometry pass): measured optimum of the
            # persistent GEMM grid = 248 of 256 for the train step (8 scenes; 240: +0.05 ms, 256:
            # +0.4 ms) and 216-224 for cfg2's 24 scenes per pass (232: +0.05 ms)
            slots = GeometrySlots(model.backbone_net, dd["point_clouds"], depth, group,
                                  reserve_cus=(0 if wl["train"] else 8) + rccl_cus)

        feeder = None
        dd_sets = [dd] * max(depth, 1)
        if feed == "builder":
            if not (overlap and wl["train"] and slots.group == 1 and wl["C"] in (1, 4, 132)):
                raise SystemExit("--feed builder needs the default graphed train step")
            from scan2cap_amd.pipeline import independent_streams
            feeder, dd_sets, feed_pick = make_feeder(
                wl, dd, depth, msa, device, args.feed_scenes, rank,
                independent_streams(1, avoid=slots.streams)[0])
        if use_graph:
            # whole step = one hipGraph replay (fwd + loss + bwd [+ Adam]); with N>1
            # the RCCL all-reduce stays an eager call between two graphs
            from scan2cap_amd.graphs import GraphedCallable

            def with_geometry(p):
                d = dict(dd_sets[p])
                d["_slot"] = p
                if slots is not None:
                    d["_geometry"] = slots.geometry(p)
                return d

            replays = []
            g2 = None
            pairs = []
            for p in range(max(depth, 1)):
                if wl["train"] and two_stage is not None:
                    from scan2cap_amd.graphs import GraphedPair

                    def first(p=p):
                        d = with_geometry(p)
                        ddp.drop_grads()
                        d = model(d, use_tf=True, is_eval=False)
                        d = get_scene_cap_loss(d, device, cfg_loss, None)
                        probe_caption(d, p)
                        follow_model(d, d)
                        two_stage.stage1(d)       # captioner + graph gradients, d loss / d X
                        ddp.pack_grads(0)
                        return d["loss"]

                    def second():
                        two_stage.stage2()        # ... through the detector
                        ddp.pack_grads(1)
                    pair = GraphedPair(first, second).capture()
                    pairs.append(pair)
                    if g2 is None:
                        g2 = GraphedCallable(lambda: optimizer.step()).capture()

                    def replay(pair=pair):
                        loss = pair.replay_first()
                        ddp.reduce(0, async_op=True)   # 80 % of the bytes: on the wire while
                        pair.replay_second()           # the detector's backward runs
                        ddp.reduce(1, async_op=True)
                        ddp.wait()
                        g2()
                        return loss
                elif wl["train"] and ddp is not None:
                    def fwd_bwd(p=p):
                        d = with_geometry(p)
                        ddp.drop_grads()
                        d = model(d, use_tf=True, is_eval=False)
                        d = get_scene_cap_loss(d, device, cfg_loss, None)
                        probe_caption(d, p)
                        follow_model(d, d)
                        d["loss"].backward()
                        ddp.pack_grads()      # one multi-tensor copy into the flat bucket
                        return d["loss"]
                    g1 = GraphedCallable(fwd_bwd).capture()
                    if g2 is None:
                        g2 = GraphedCallable(lambda: optimizer.step()).capture()

                    def replay(g1=g1):
                        loss = g1()
                        ddp.reduce()
                        g2()
                        return loss
                else:
                    replay = GraphedCallable(lambda p=p: eager_step(with_geometry(p))).capture()
                replays.append(replay)
            if overlap:
                G = slots.group
                if G > 1:
                    for g in range(depth // G):
                        slots.refill_group(g, [dd["point_clouds"]] * G)
                else:
                    for p in range(depth):
                        slots.refill(p, dd["point_clouds"])
                torch.cuda.synchronize()     # from here on the resident batch is complete
                counter = {"i": 0}
                if feeder is not None:
                    for p in range(depth - 1 if depth >= 3 else depth):
                        feeder.produce(p, *feed_pick(p))
                        slots.refill(p, dd_sets[p]["point_clouds"], ready=feeder.ready[p])

                host_ms = {"replay": 0.0, "produce": 0.0, "refill": 0.0, "n": 0}
                host_paced = feeder is not None and depth >= 3

                def fed_step(_dd):
                    """Step i trains on buffer set i % depth; batch i + depth - 1 (host-paced,
                    depth 3: the host waits for step i-1 itself, so that no stream parks on an
                    event) or i + depth (stream-paced) is assembled and its geometry computed
                    while the following steps run."""
                    i = counter["i"]
                    p = i % depth
                    counter["i"] += 1
                    t_a = time.perf_counter()
                    feeder.acquire(p)                      # batch i is assembled
                    slots.acquire(p)                       # ... and its geometry published
                    out = replays[p]()
                    slots.release(p)
                    feeder.release(p)
                    t_b = time.perf_counter()
                    q, nxt = ((i + depth - 1) % depth, i + depth - 1) if host_paced else (p, i + depth)
                    feeder.produce(q, *feed_pick(nxt), host_wait=host_paced)
                    t_c = time.perf_counter()
                    slots.refill(q, dd_sets[q]["point_clouds"], ready=feeder.ready[q])
                    t_d = time.perf_counter()
                    host_ms["replay"] += 1e3 * (t_b - t_a)
                    host_ms["produce"] += 1e3 * (t_c - t_b)
                    host_ms["refill"] += 1e3 * (t_d - t_c)
                    host_ms["n"] += 1
                    return out

                def step(_dd):
                    if feeder is not None:
                        return fed_step(_dd)
                    i = counter["i"]
                    p = i % depth
                    counter["i"] += 1
                    slots.acquire(p)                       # geometry of this step is published
                    out = replays[p]()
                    slots.release(p)
                    # ready=None: the resident batch is complete (synchronised above); the
                    # refill must NOT wait for the step graph just launched -- it overlaps it
                    if G == 1:
                        slots.refill(p, dd["point_clouds"], ready=None)   # step i+depth
                    elif (p + 1) % G == 0:                 # group fully consumed: next G batches
                        slots.refill_group(p // G, [dd["point_clouds"]] * G, ready=None)
                    return out
            else:
                def step(_dd):
                    return replays[0]()
        else:
            step = eager_step

        trace("setup / capture done")
        for _ in range(warmup):
            step(dd)
            trace("warmup step")
        barrier()
        caption_first = read_caption_probe()
        if not use_graph:
            _C.TIMER.start()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(dd)
            trace("timed step")
        barrier()
        elapsed = time.perf_counter() - t0
        trace("timed region done")
        # the contract's K steps are the headline; the same K-step window repeated a few more
        # times says how far one 0.2-second sample can be trusted (median + spread reported)
        windows = []
        if use_graph and int(os.environ.get("S2C_BENCH_WINDOWS", "5")) > 0:
            for _ in range(int(os.environ.get("S2C_BENCH_WINDOWS", "5"))):
                barrier()
                tw = time.perf_counter()
                for _ in range(steps):
                    step(dd)
                barrier()
                windows.append((time.perf_counter() - tw) / steps * 1e3)
        if feeder is not None and os.environ.get("S2C_DEBUG_HOST") == "1":
            print("[bench] host ms per step:", {k: round(v / host_ms["n"], 3) for k, v in
                                                host_ms.items() if k != "n"}, file=sys.stderr)
        return {"elapsed": elapsed, "overlap": overlap, "depth": depth,
                "caption_first": caption_first, "caption_last": read_caption_probe(),
                "group": slots.group if slots is not None else 1,
                "fed": feeder is not None, "windows_ms": windows,
                "pairs": pairs if use_graph else [], "g2": g2 if use_graph else None,
                # the captured graphs read these static tensors by address: whoever replays
                # a graph after this function returns (ddp_evidence) must keep them alive
                "keepalive": (slots, dd_sets, feeder)}

    head = measure(args.feed, args.steps, args.warmup)
    elapsed, overlap, depth = head["elapsed"], head["overlap"], head["depth"]
    if use_graph:
        # per-kernel durations: HIP events cannot be read back from inside a graph
        # replay, so the same steps are run once more eagerly, un-timed for the
        # headline, with the event timer on (same kernels, same shapes)
        _C.TIMER.start()
        dd.pop("_geometry", None)   # instrumented pass computes geometry in-line
        for _ in range(min(args.steps, 3)):
            eager_step(dd)
            trace("instrumented eager step")
        kern_steps = min(args.steps, 3)
    else:
        kern_steps = args.steps
    kern = _C.TIMER.stop()
    windows_ms = head["windows_ms"]
    if world > 1:
        t = torch.tensor([elapsed] + windows_ms, device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, windows_ms = float(t[0].item()), [float(x) for x in t[1:].tolist()]
    ddp_info = ddp_evidence(ddp, head, device, world, trace) if ddp is not None else None
    # the same step on a NEW device-assembled batch every step, in the same
    # line: the headline replays one resident 187 MB batch, which fits the 256 MiB
    # Infinity Cache
    fed = None
    if (world == 1 and args.feed == "resident" and not args.no_fed and head["overlap"]
            and wl["train"] and wl["C"] in (1, 4, 132)):
        try:
            # in a fresh process, after this one has gone idle: a second pipeline set up next
            # to the first one's graphs / slots / side streams measured 8 % low (639 vs 693
            # scenes/s standalone), whatever was freed in between
            import subprocess
            torch.cuda.synchronize()
            decoder_fused.release_device_locks()     # the child runs the persistent decoder kernels
            cmd = [sys.executable, os.path.abspath(__file__), "--feed", "builder", "--no-fed",
                   "--no-cpu-baseline", "--workload", args.workload, "--steps", str(args.steps),
                   "--warmup", str(args.warmup), "--feed-scenes", str(args.feed_scenes)]
            env = {k: v for k, v in os.environ.items()
                   if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "S2C_FORCE_DDP")}
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            f = json.loads(res.stdout.strip().splitlines()[-1])
            fed = {"value": f["value"], "unit": "scenes/s", "ms_per_step": f["ms_per_step"],
                   "feed": f["config"]["feed"] + "; geometry " + f["config"]["geometry"],
                   "how": "python bench.py --feed builder (own process, same GPU, run after "
                          "the headline measurement)"}
        except Exception as e:          # reported, never fatal for the headline
            fed = {"error": "%s: %s" % (type(e).__name__, e)}

    if decoder_fused.persist_failed():
        raise SystemExit("bench.py: a persistent decoder kernel gave up on a poll -- results invalid")
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        table_k = []
        for name, r in sorted(kern.items(), key=lambda kv: -kv[1]["total_ms"]):
            avg_us = r["total_ms"] / max(r["calls"], 1) * 1e3
            gbs = r["alg_bytes"] / max(r["total_ms"], 1e-9) / 1e6
            table_k.append({"kernel": name, "calls_per_step": r["calls"] / kern_steps,
                            "ms_per_step": r["total_ms"] / kern_steps,
                            "avg_us": avg_us, "alg_GBps": gbs,
                            "alg_TFLOPs": r.get("alg_flops", 0) / max(r["total_ms"], 1e-9) / 1e9,
                            "alg_bytes_per_launch": r["alg_bytes"] / max(r["calls"], 1)})
        roof = roofline_of(table_k[0], ms_per_step, wl) if table_k else None
        # the dominant kernel of the MAIN stream as well when the top one runs ahead on a
        # side stream (the geometry stage, overlapped with the previous step)
        side = ("s2c_furthest_point_sampling", "s2c_ball_query", "s2c_three_nn")
        main_k = [k for k in table_k if not k["kernel"].startswith(side)]
        roof_main = roofline_of(main_k[0], ms_per_step, wl) if (overlap and main_k) else None
        roof_gemm = family_roofline(table_k, ms_per_step)
        out = {
            "metric": ("scenes/sec forward+backward, B=%d N=%d pts" if wl["train"]
                       else "scenes/sec forward, B=%d N=%d pts") % (wl["B"], wl["N"]),
            "value": value, "unit": "scenes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "windows": ({"ms_per_step": windows_ms,
                         "median_ms_per_step": float(np.median(windows_ms)),
                         "median_value": B * world / float(np.median(windows_ms)) * 1e3,
                         "note": "%d more windows of %d steps after the contract's timed "
                                 "region" % (len(windows_ms), args.steps)}
                        if windows_ms else None),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, wl["desc"]),
                       "scenes_per_gpu": B, "global_batch": B * world,
                       "parallelism": "dp%d" % world,
                       "arithmetic": ("fp32 in / fp32 accumulate everywhere; the rows GEMMs form "
                                      "each fp32 product from 6 bf16 MFMA products of a 3-way "
                                      "split (error = an fp32 FMA chain's, tests/test_fused_gpu.py)"
                                      if GEMM_SPLIT else "fp32 MFMA chain"),
                       "launch": "hipGraph replay" if use_graph else "eager",
                       "geometry": ("%d batch(es) ahead on side stream(s)%s" % (
                           depth, ", %d batches per geometry pass" % head["group"]
                           if head["group"] > 1 else ""))
                                   if overlap else "in-line",
                       "feed": ("a new batch per step assembled on the device from %d "
                                "HBM-resident scenes, one batch ahead" % args.feed_scenes)
                               if head["fed"] else "one batch resident in HBM",
                       "grad_allreduce_bytes": ddp.nbytes if ddp else 0,
                       "grad_allreduce": (("2 buckets (captioner+graph %d B async under the "
                                           "detector's backward, detector %d B)"
                                           % (ddp.flats[0].numel() * 4, ddp.flats[1].numel() * 4))
                                          if two_stage is not None else "1 flat bucket after "
                                          "backward") if ddp else "none"},
            "caption_branch": {"after_warmup": head["caption_first"],
                               "after_last_step": head["caption_last"],
                               "note": "ref_box_corner_label follows the model: the box predicted for "
                                       "proposal 0 in step i is the described box of step i + 1 "
                                       "(bench.follow_model, one device copy per step); cap_loss > 0 "
                                       "and good_bbox_masks.sum() > 0 mean the captioner / relation-"
                                       "graph gradients of the timed steps are live"}
                              if wl["train"] else None,
            "roofline": roof,
            "roofline_main_stream": roof_main,
            "roofline_named": named_roofline(table_k, args.workload, B),
            "roofline_gemm": roof_gemm,
            "roofline_decode": decode_roofline(table_k, ms_per_step, args.workload),
            "fed": fed,
            "ddp": ddp_info,
            "kernels": table_k[:10],
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, vocabulary, embeddings, table, msa,
                                               args.cpu_sample_scenes)
            out["vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        line = json.dumps(out)
    else:
        line = None
    if dist.is_initialized():
        dist.destroy_process_group()
    if line is not None:
        # the JSON must be the LAST stdout line: push out whatever native libraries
        # (e.g. the RCCL version banner) still hold in C stdio buffers first
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        _LAST_LINE[0] = line
        if _emit is None:
            print(line, flush=True)
        else:
            _emit(line)


def _main_with_one_line_stdout():
    """RCCL prints a five-line version banner and gloo a "[Gloo] Rank ..." line on STDOUT; the driver
    reads ONE JSON line there.  File descriptor 1 points at stderr while main() runs and is restored
    for the JSON line alone (everything else a native library writes lands on stderr)."""
    import ctypes
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    _REAL_STDOUT[0] = saved
    try:
        main(_emit=lambda line: None)
    finally:
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
        _REAL_STDOUT[0] = None
    if _LAST_LINE[0] is not None:
        print(_LAST_LINE[0], flush=True)


if __name__ == "__main__":
    _main_with_one_line_stdout()

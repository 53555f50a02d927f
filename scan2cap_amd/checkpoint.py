"""Checkpoint / pretrained-weight compatibility.

The files are the reference's, byte for byte in layout:

* `pretrained/PRETRAIN_VOTENET_*/model.pth`   -- a VoteNet state_dict, loaded with
  `load_state_dict(..., strict=False)` into a `no_caption` CapNet whose `backbone_net`,
  `vgen`, `proposal` are then mounted on the captioning model (scripts/train.py:84-118);
* `outputs/<stamp>/checkpoint.tar`            -- `{"epoch", "model_state_dict",
  "optimizer_state_dict", "best"}` (lib/solver.py:501-510), resumed by
  scripts/train.py:138-145;
* `model.pth` / `model_last.pth`              -- plain `state_dict()` (lib/solver.py:185-188).

What is specific to this implementation is the captured step: a hipGraph holds raw pointers
to the parameter, gradient and optimizer-state tensors.  `model.load_state_dict` copies
into the existing parameters (the graph stays valid), but `optimizer.load_state_dict`
REPLACES the state tensors, so a captured optimizer step would keep updating the old
ones.  `load_optimizer_state_inplace` copies into the live tensors instead;
`load_checkpoint(..., inplace=True)` is the resume call for a model whose step is already
captured.  (Re-capturing after a plain load is equally valid; both are tested.)
"""
import os

import torch

PRETRAINED_PREFIXES = ("backbone_net.", "vgen.", "proposal.")


def input_channels(use_multiview=False, use_normal=False, use_color=False, no_height=False):
    """scripts/train.py:58."""
    return int(use_multiview) * 128 + int(use_normal) * 3 + int(use_color) * 3 \
        + int(not no_height)


def pretrained_name(use_color=False, use_multiview=False, use_normal=False):
    """scripts/train.py:94-97."""
    name = "PRETRAIN_VOTENET_XYZ"
    if use_color:
        name += "_COLOR"
    if use_multiview:
        name += "_MULTIVIEW"
    if use_normal:
        name += "_NORMAL"
    return name


def mount_pretrained_votenet(model, pretrained_path, no_detection=False, map_location="cpu"):
    """scripts/train.py:84-118: build the `no_caption` twin of `model`, load the VoteNet
    weights into it (strict=False, the reference's call), mount its detector stages on
    `model`, optionally freeze them.  Returns torch's (missing_keys, unexpected_keys)."""
    from .models import CapNet
    twin = CapNet(num_class=model.num_class, vocabulary=None, embeddings=None,
                  num_heading_bin=model.num_heading_bin,
                  num_size_cluster=model.num_size_cluster,
                  mean_size_arr=model.mean_size_arr,
                  num_proposal=model.num_proposal,
                  input_feature_dim=model.input_feature_dim, no_caption=True)
    result = twin.load_state_dict(torch.load(pretrained_path, map_location=map_location),
                                  strict=False)
    model.backbone_net = twin.backbone_net
    model.vgen = twin.vgen
    model.proposal = twin.proposal
    if no_detection:
        for stage in (model.backbone_net, model.vgen, model.proposal):
            for p in stage.parameters():
                p.requires_grad = False
    return result


def save_checkpoint(root, epoch, model, optimizer, best):
    """lib/solver.py:501-515: checkpoint.tar + model_last.pth under `root`."""
    os.makedirs(root, exist_ok=True)
    torch.save({"epoch": epoch, "model_state_dict": model.state_dict(),
                "optimizer_state_dict": optimizer.state_dict(), "best": best},
               os.path.join(root, "checkpoint.tar"))
    torch.save(model.state_dict(), os.path.join(root, "model_last.pth"))


def load_optimizer_state_inplace(optimizer, state_dict):
    """`optimizer.load_state_dict` semantics, but tensors of an already-populated state
    are overwritten IN PLACE (a captured hipGraph keeps pointing at them)."""
    groups, saved_groups = optimizer.param_groups, state_dict["param_groups"]
    if len(groups) != len(saved_groups) or any(
            len(g["params"]) != len(s["params"]) for g, s in zip(groups, saved_groups)):
        raise ValueError("loaded state dict has different parameter groups")
    id_map = {}
    for g, s in zip(groups, saved_groups):
        for p, sid in zip(g["params"], s["params"]):
            id_map[sid] = p
        for k, v in s.items():
            if k != "params":
                if torch.is_tensor(g.get(k)) and torch.is_tensor(v):
                    g[k].copy_(v)
                else:
                    g[k] = v
    for sid, st in state_dict["state"].items():
        p = id_map[sid]
        live = optimizer.state[p]
        for k, v in st.items():
            if torch.is_tensor(v) and torch.is_tensor(live.get(k)) \
                    and live[k].shape == v.shape:
                live[k].copy_(v)
            elif torch.is_tensor(v):
                live[k] = v.to(device=p.device) if v.dim() else v.to(
                    device=live[k].device if torch.is_tensor(live.get(k)) else p.device)
            else:
                live[k] = v


def load_checkpoint(root, model, optimizer, map_location=None, inplace=False):
    """scripts/train.py:138-145.  Returns (epoch, best)."""
    ckpt = torch.load(os.path.join(root, "checkpoint.tar"), map_location=map_location)
    model.load_state_dict(ckpt["model_state_dict"])
    if inplace:
        load_optimizer_state_inplace(optimizer, ckpt["optimizer_state_dict"])
    else:
        optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    return ckpt.get("epoch"), ckpt.get("best")

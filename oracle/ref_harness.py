"""Imports the REFERENCE's Python layers in THIS container (CPU only) so that
golden vectors can be generated from the reference's own code.

TEST INFRASTRUCTURE ONLY, and only usable where /root/reference exists (never
on the GPU box -- nothing under tests -m gpu / smoke() / bench.py imports this).

The reference hard-codes CUDA and a few absent third-party packages; this
harness supplies the minimum around it WITHOUT touching its sources:

  1. `easydict`            -- 6-line attribute-dict stub (lib/config.py:3).
  2. `pointnet2._ext`      -- the nine ops backed by oracle/s2c_oracle.c (the
                              reference ships them CUDA-only, bindings.cpp:6-19).
  3. `.cuda()`             -- no-ops on Tensor / Module; torch.cuda.FloatTensor
                              aliases (proposal_module.py:99,135, graph_module.py
                              :211-259, caption_module.py:35-36 ...).
  4. `torch_geometric`     -- a shim implementing exactly the members
                              models/graph_module.py touches (:5-12, :74-100),
                              following PyG's documented source_to_target
                              semantics.  PyG is un-vendored and unpinned by the
                              reference, so goldens that pass through it are
                              "parity unpinned" at that boundary.
  5. cwd / CONF.PATH.SCANNET -> /root/reference (lib/config.py:9,
                              model_util_scannet.py:90).
"""
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = os.environ.get("S2C_REFERENCE_ROOT", "/root/reference")
_installed = False


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


# --------------------------------------------------------------------------
def _install_easydict():
    mod = types.ModuleType("easydict")

    class EasyDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

    mod.EasyDict = EasyDict
    sys.modules["easydict"] = mod


def _install_ext():
    from . import torch_ext
    ext = torch_ext.as_module("pointnet2._ext")
    pkg = types.ModuleType("pointnet2")
    pkg._ext = ext
    pkg.__path__ = []
    sys.modules["pointnet2"] = pkg
    sys.modules["pointnet2._ext"] = ext


def _install_cuda_noops():
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.cuda.LongTensor = torch.LongTensor


def _install_pyg_shim():
    import inspect

    pyg = types.ModuleType("torch_geometric")
    pyg.__path__ = []
    utils = types.ModuleType("torch_geometric.utils")
    data = types.ModuleType("torch_geometric.data")
    nn_mod = types.ModuleType("torch_geometric.nn")
    typing_mod = types.ModuleType("torch_geometric.typing")

    def from_scipy_sparse_matrix(A):
        A = A.tocoo()
        row = torch.from_numpy(A.row).to(torch.long)
        col = torch.from_numpy(A.col).to(torch.long)
        return torch.stack([row, col], dim=0), torch.from_numpy(A.data)

    def add_self_loops(edge_index, num_nodes=None):
        n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
        loop = torch.arange(n, dtype=torch.long).unsqueeze(0).repeat(2, 1)
        return torch.cat([edge_index, loop], 1), None

    def degree(index, num_nodes=None, dtype=None):
        n = int(index.max()) + 1 if num_nodes is None else num_nodes
        out = torch.zeros(n, dtype=dtype or torch.float)
        return out.scatter_add_(0, index, torch.ones_like(index, dtype=out.dtype))

    utils.from_scipy_sparse_matrix = from_scipy_sparse_matrix
    utils.add_self_loops = add_self_loops
    utils.degree = degree

    class Data(object):
        def __init__(self, x=None, edge_index=None, **kw):
            self.x, self.edge_index = x, edge_index
            self.__dict__.update(kw)

    data.Data = Data
    data.DataLoader = object

    class _Inspector(object):
        def __init__(self, owner):
            self.owner = owner

        def distribute(self, name, coll):
            fn = getattr(self.owner, name)
            params = [p for p in inspect.signature(fn).parameters]
            if name in ("aggregate", "update"):
                # PyG inspects these with pop_first=True: the first parameter is
                # the positional tensor handed over by propagate()
                params = params[1:]
            return {p: coll[p] for p in params if p in coll}

    class MessagePassing(torch.nn.Module):
        """source_to_target flow: x_j = x[edge_index[0]], x_i = x[edge_index[1]],
        aggregation at edge_index[1] over dim 0 (node_dim)."""

        def __init__(self, aggr="add", flow="source_to_target", node_dim=0):
            super().__init__()
            self.aggr, self.flow, self.node_dim = aggr, flow, node_dim
            self.inspector = _Inspector(self)
            self.__user_args__ = ["x_i", "x_j"]
            self.__explain__ = False

        def __check_input__(self, edge_index, size):
            assert edge_index.dtype == torch.long and edge_index.dim() == 2
            return [None, None] if size is None else list(size)

        def __collect__(self, args, edge_index, size, kwargs):
            x = kwargs["x"]
            n = x.size(self.node_dim)
            return {"x_j": x.index_select(self.node_dim, edge_index[0]),
                    "x_i": x.index_select(self.node_dim, edge_index[1]),
                    "index": edge_index[1], "dim_size": n, "ptr": None}

        def aggregate(self, inputs, index, ptr=None, dim_size=None):
            # torch_scatter's reductions: "add"; "mean" = sum / max(count, 1); "max" with
            # 0 for nodes that receive nothing
            out = torch.zeros(dim_size, inputs.size(1), dtype=inputs.dtype)
            if self.aggr == "add":
                return out.index_add(0, index, inputs)
            cnt = torch.zeros(dim_size, dtype=inputs.dtype).index_add(
                0, index, torch.ones_like(index, dtype=inputs.dtype))
            if self.aggr == "mean":
                return out.index_add(0, index, inputs) / cnt.clamp(min=1).unsqueeze(-1)
            assert self.aggr == "max"
            ix = index.view(-1, 1).expand_as(inputs)
            red = out.scatter_reduce(0, ix, inputs, reduce="amax", include_self=False)
            return torch.where((cnt > 0).unsqueeze(-1), red, out)

        def update(self, inputs):
            return inputs

    class GCNConv(torch.nn.Module):
        def __init__(self, in_channels, out_channels):
            super().__init__()
            self.weight = torch.nn.Parameter(torch.empty(in_channels, out_channels))
            self.bias = torch.nn.Parameter(torch.zeros(out_channels))
            torch.nn.init.xavier_uniform_(self.weight)

        def forward(self, x, edge_index):
            n = x.size(0)
            ei, _ = add_self_loops(edge_index, n)
            deg = degree(ei[1], n, x.dtype)
            dinv = deg.pow(-0.5)
            norm = dinv[ei[0]] * dinv[ei[1]]
            xw = x @ self.weight
            out = torch.zeros_like(xw).index_add(0, ei[1], xw[ei[0]] * norm.unsqueeze(-1))
            return out + self.bias

    nn_mod.MessagePassing = MessagePassing
    nn_mod.GCNConv = GCNConv
    typing_mod.Adj = object
    typing_mod.Size = object
    pyg.utils, pyg.data, pyg.nn, pyg.typing = utils, data, nn_mod, typing_mod
    for name, m in (("torch_geometric", pyg), ("torch_geometric.utils", utils),
                    ("torch_geometric.data", data), ("torch_geometric.nn", nn_mod),
                    ("torch_geometric.typing", typing_mod)):
        sys.modules[name] = m


def _install_absent_stubs():
    """Empty stand-ins for third-party packages the reference imports at module
    scope but never touches on this path (utils/metric_util.py:17 `trimesh`,
    pc_utils `plyfile`, ...).  Only installed when the real package is absent."""
    import importlib.util
    for name in ("trimesh", "plyfile", "h5py", "tensorboardX"):
        if name in sys.modules:
            continue
        try:
            found = importlib.util.find_spec(name) is not None
        except (ImportError, ValueError):
            found = False
        if not found:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    if "plyfile" in sys.modules and not hasattr(sys.modules["plyfile"], "PlyData"):
        sys.modules["plyfile"].PlyData = object
        sys.modules["plyfile"].PlyElement = object


def install():
    """Make `import models.capnet` etc. (the reference's modules) work here."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("reference tree not found at %s" % REF_ROOT)
    _install_easydict()
    _install_ext()
    _install_cuda_noops()
    _install_pyg_shim()
    _install_absent_stubs()
    os.chdir(REF_ROOT)
    for p in (REF_ROOT, os.path.join(REF_ROOT, "lib"),
              os.path.join(REF_ROOT, "lib", "pointnet2")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import importlib
    conf = importlib.import_module("lib.config")
    conf.CONF.PATH.SCANNET = os.path.join(REF_ROOT, "data", "scannet")
    _installed = True


def reference_modules():
    """Returns a namespace of the reference's classes / functions."""
    install()
    import importlib
    ns = types.SimpleNamespace()
    ns.pointnet2_utils = importlib.import_module("lib.pointnet2.pointnet2_utils")
    ns.pointnet2_modules = importlib.import_module("lib.pointnet2.pointnet2_modules")
    ns.backbone = importlib.import_module("models.backbone_module")
    ns.voting = importlib.import_module("models.voting_module")
    ns.proposal = importlib.import_module("models.proposal_module")
    ns.graph = importlib.import_module("models.graph_module")
    ns.caption = importlib.import_module("models.caption_module")
    ns.capnet = importlib.import_module("models.capnet")
    ns.box_util = importlib.import_module("utils.box_util")
    ns.loss_helper = importlib.import_module("lib.loss_helper")
    ns.DC = ns.proposal.DC
    return ns

"""A/B switch between the MI355X-first fused paths and the reference's op-by-op
formulation of the same modules (QueryAndGroup -> Conv2d/BatchNorm/ReLU -> max_pool,
three_interpolate -> SharedMLP, per-step decoder loop, op-by-op losses ...), all still on
the GPU through the nine `_ext` ops.  Used by the parity tests to hold the fused kernels
to 1e-4 against the op-by-op path at the BASELINE workload sizes
(tests/test_configs_gpu.py); never used by bench.py.

    with op_by_op():
        ref = model(data_dict)
"""
import contextlib

_FLAGS = (
    ("scan2cap_amd.pointnet2.fused", "ENABLED"),
    ("scan2cap_amd.pointnet2.pointnet2_modules", "FUSE_FP"),
    ("scan2cap_amd.models.voting_module", "FUSE_VOTE_HEAD"),
    ("scan2cap_amd.models.proposal_module", "FUSE_BOX_DECODE"),
    ("scan2cap_amd.models.graph_module", "USE_QUERY_KERNEL"),
    ("scan2cap_amd.models.graph_module", "USE_EDGE_KERNELS"),
    ("scan2cap_amd.models.caption_module", "FUSE_EVAL_STEP"),
    ("scan2cap_amd.models.caption_module", "USE_SELECT_TARGET_KERNEL"),
    ("scan2cap_amd.models.decoder_fused", "ENABLED"),
    ("scan2cap_amd.loss_helper", "FUSED_DETECTION_LOSS"),
    ("scan2cap_amd.loss_helper", "FUSED_CAPTION_LOSS"),
)


@contextlib.contextmanager
def op_by_op():
    import importlib
    saved = []
    for mod, name in _FLAGS:
        m = importlib.import_module(mod)
        saved.append((m, name, getattr(m, name)))
        setattr(m, name, False)
    try:
        yield
    finally:
        for m, name, v in saved:
            setattr(m, name, v)

"""Geometry-ahead pipeline.

FPS is inherently serial (npoint-1 dependent rounds) and can only use one
workgroup per scene: at B=8 it keeps 8 of 256 CUs busy for milliseconds while
everything else waits -- unless it does not have to wait.  The whole geometry
stage of the backbone (FPS chain, ball queries, 3-NN) depends on xyz alone, so it
is computed for batch i+1 on a second HIP stream while batch i's GEMM-heavy
forward/backward owns the other CUs.  This is a software pipeline across batches:
all work of every batch is still executed, none is cached or skipped.

    pipe = GeometryPipeline(model.backbone_net)
    h = pipe.submit(next_batch["point_clouds"])         # side stream, async
    ...
    pipe.attach(data_dict, h)                           # main stream waits on the event
    model(data_dict)
"""
import torch


class GeometryPipeline(object):
    def __init__(self, backbone, stream=None):
        self.backbone = backbone
        self.stream = stream if stream is not None else torch.cuda.Stream()

    def submit(self, point_clouds):
        """Launch the geometry stage of `point_clouds` on the side stream."""
        main = torch.cuda.current_stream()
        self.stream.wait_stream(main)          # inputs were produced on `main`
        with torch.cuda.stream(self.stream):
            geo = self.backbone.compute_geometry(point_clouds)
            done = torch.cuda.Event()
            done.record(self.stream)
        return geo, done

    def attach(self, data_dict, handle):
        """Make the current stream wait for the geometry, then hand it to the model."""
        geo, done = handle
        cur = torch.cuda.current_stream()
        cur.wait_event(done)
        for v in geo.values():
            for t in v:
                t.record_stream(cur)
        data_dict["_geometry"] = geo
        return data_dict


def flatten_geometry(geo):
    """Stable flat list of the geometry tensors (for copying into the static
    input buffers of a captured hipGraph)."""
    out = []
    for k in ("sa1", "sa2", "sa3", "sa4", "fp1", "fp2"):
        out += list(geo[k])
    return out


def unflatten_geometry(flat):
    it = iter(flat)
    geo = {}
    for k in ("sa1", "sa2", "sa3", "sa4"):
        geo[k] = (next(it), next(it), next(it))
    for k in ("fp1", "fp2"):
        geo[k] = (next(it), next(it))
    return geo

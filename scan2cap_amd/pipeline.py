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
import ctypes as _ctypes
import os as _os

import torch


def _shares_queue(a, b, probe):
    """True if work on stream `b` is held up by earlier work on stream `a` (HIP
    maps streams onto a small number of in-order hardware queues: two streams on
    the same queue cannot overlap).  Timed with a ~1 ms probe kernel."""
    torch.cuda.synchronize()
    with torch.cuda.stream(b):
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(a):
        busy0 = torch.cuda.Event(enable_timing=True)
        busy1 = torch.cuda.Event(enable_timing=True)
        busy0.record(a)
        probe()
        busy1.record(a)
    with torch.cuda.stream(b):
        t0.record(b)
        torch.empty(1, device="cuda").fill_(0)
        t1.record(b)
    torch.cuda.synchronize()
    # b's trivial kernel finished only after a's probe => same queue
    return busy0.elapsed_time(t1) > 0.8 * busy0.elapsed_time(busy1)


def independent_streams(n, candidates=12, avoid=()):
    """`n` side streams that share a hardware queue neither with the current
    stream, nor with each other, nor with the streams in `avoid` (best effort: falls
    back to plain new streams)."""
    if not torch.cuda.is_available():
        return [torch.cuda.Stream() for _ in range(n)]
    main = torch.cuda.current_stream()
    # ~1 ms of element-wise work (NOT a library GEMM: until round 6 this probe was the only
    # rocBLAS / Tensile launch of the default bench.py command -- 120 `Cijk_*` rows in its rocprofv3
    # table, tools/trace_lib_gemm_bench.py)
    x = torch.randn(1 << 24, device="cuda")

    def probe():
        y = x
        for _ in range(24):
            y = torch.sin(y)
        return y

    for s in (main,):
        with torch.cuda.stream(s):
            probe()
    picked, pool = [], [torch.cuda.Stream() for _ in range(candidates)]
    for c in pool:
        with torch.cuda.stream(c):
            probe()                       # warm the library on this stream
    for c in pool:
        if len(picked) == n:
            break
        if _shares_queue(c, main, probe) or _shares_queue(main, c, probe):
            continue
        if any(_shares_queue(c, p, probe) for p in list(picked) + list(avoid)):
            continue
        picked.append(c)
    while len(picked) < n:
        picked.append(torch.cuda.Stream())
    return picked


class GeometryPipeline(object):
    """`depth` side streams used round-robin: with depth d the geometry of d
    batches is in flight at once (each FPS chain occupies only B CUs), which is
    what a forward-only pipeline needs when a step is shorter than one FPS chain."""

    def __init__(self, backbone, stream=None, depth=1):
        self.backbone = backbone
        self.streams = [stream] if stream is not None else \
            independent_streams(max(1, depth))
        self._next = 0

    @property
    def stream(self):
        return self.streams[0]

    def submit(self, point_clouds):
        """Launch the geometry stage of `point_clouds` on the next side stream."""
        side = self.streams[self._next % len(self.streams)]
        self._next += 1
        main = torch.cuda.current_stream()
        side.wait_stream(main)                 # inputs were produced on `main`
        with torch.cuda.stream(side):
            geo = self.backbone.compute_geometry(point_clouds)
            done = torch.cuda.Event()
            done.record(side)
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


class GeometrySlots(object):
    """Geometry-ahead pipeline for hipGraph-captured steps.

    `depth` slots of STATIC geometry tensors.  A step graph is captured once per
    slot with `data_dict["_geometry"] = slots.geometry(p)`, so the main stream never
    runs an eager kernel between waiting for the geometry and launching the graph
    (eager device-to-device copies directly in front of a hipGraph launch faulted
    on ROCm 7.x at the cfg5 sizes).  The side stream computes the geometry of batch
    i+depth into temporaries while batch i runs, and publishes it into the slot
    with one multi-tensor copy AFTER the step that last read the slot has finished.

        slots = GeometrySlots(backbone, point_clouds, depth)
        graphs = [capture(step_fn reading slots.geometry(p)) for p in range(depth)]
        for p in range(depth): slots.refill(p, point_clouds)          # prime (waits for
                                         # the current stream; ready=None / an Event otherwise)
        for i in ...:
            p = i % depth
            slots.acquire(p)          # main waits until slot p is published
            graphs[p].replay()
            slots.release(p)          # marks the slot consumed ...
            slots.refill(p, next_point_clouds)   # ... and starts batch i+depth
    """

    def __init__(self, backbone, point_clouds, depth=1, group=1, reserve_cus=8):
        """reserve_cus: compute units the persistent GEMM grid of the main stream leaves free
        on top of the FPS workgroups of this stage (one per scene of a pass) -- 8 for the
        ball-query / 3-NN launches; pass 8 + the RCCL channel count when a collective runs
        beside the step (bench.py does for N > 1).  The grid is a process-wide setting of
        libs2c_hip.so: `close()` restores the value found here.
        group > 1: the geometry of `group` consecutive batches is computed by ONE
        set of launches on the stacked clouds (the FPS kernels run one workgroup per
        scene for ~6 ms whatever the batch: stacking G batches gives G times the
        geometry throughput from a single side stream); `depth` must be a multiple of
        `group`, and `refill_group` replaces `refill`."""
        self.backbone = backbone
        self.depth = max(1, depth)
        self.group = max(1, group)
        if self.depth % self.group:
            raise ValueError("GeometrySlots: depth must be a multiple of group")
        self.streams = independent_streams(self.depth // self.group if self.group > 1
                                           else self.depth)
        # the persistent GEMM grid on the main stream must fit beside this stage's FPS
        # workgroups (one per scene of a pass, resident for milliseconds): measured optimum
        # 256 - scenes - 8 (cfg3: 240, cfg2 with 3 batches per pass: 224)
        self._old_grid = None
        if point_clouds.is_cuda and not _os.environ.get("S2C_GEMM_STREAM_GRID"):
            from . import _C
            lib = _C.load()
            lib.s2c_gemm_set_stream_grid.argtypes = [_ctypes.c_int]
            cus = torch.cuda.get_device_properties(point_clouds.device).multi_processor_count
            grid = max(cus // 4, cus - point_clouds.shape[0] * self.group - int(reserve_cus))
            self._old_grid = lib.s2c_gemm_set_stream_grid(grid)
            # ... and so must the streaming weight-gradient kernel's (csrc/s2c_dwstream.hip)
            lib.s2c_weight_grad_stream_set_grid.argtypes = [_ctypes.c_int]
            self._old_dws_grid = lib.s2c_weight_grad_stream_set_grid(grid)
        geo0 = backbone.compute_geometry(point_clouds)
        self._slots = []
        for _ in range(self.depth):
            flat = [t.clone() for t in flatten_geometry(geo0)]
            self._slots.append(flat)
        self._published = [None] * self.depth
        self._consumed = [None] * self.depth
        torch.cuda.synchronize()

    def close(self):
        """Give the persistent GEMM grid back the size it had before these slots existed."""
        if self._old_grid:
            from . import _C
            _C.load().s2c_gemm_set_stream_grid(self._old_grid)
            _C.load().s2c_weight_grad_stream_set_grid(self._old_dws_grid)
            self._old_grid = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def geometry(self, p):
        return unflatten_geometry(self._slots[p])

    def _order_after_producer(self, side, tensors, ready):
        """The geometry kernels read `tensors` on `side`: order them after the producer.
        ready = "current" (default): whatever is enqueued on the caller's current stream
        (an H2D copy, a builder kernel, a previous step) finishes first; an Event: wait
        for it (a producer on a third stream); None: the caller guarantees the data is
        already complete (e.g. a resident batch after a synchronize) -- no wait, so a
        refill issued right behind a step graph still overlaps that step."""
        if ready == "current":
            side.wait_stream(torch.cuda.current_stream())
        elif ready is not None:
            side.wait_event(ready)
        for t in tensors:
            t.record_stream(side)

    def refill(self, p, point_clouds, ready="current"):
        if self.group > 1:
            raise RuntimeError("GeometrySlots(group=%d): use refill_group(g, clouds); "
                               "refill(p, ...) is the group == 1 interface" % self.group)
        side = self.streams[p]
        self._order_after_producer(side, [point_clouds], ready)
        with torch.cuda.stream(side):
            geo = self.backbone.compute_geometry(point_clouds)
            if self._consumed[p] is not None:
                side.wait_event(self._consumed[p])   # the reader of slot p is done
            torch._foreach_copy_(self._slots[p], flatten_geometry(geo))
            ev = torch.cuda.Event()
            ev.record(side)
        self._published[p] = ev

    def refill_group(self, g, clouds, ready="current"):
        """Geometry of slots g*group .. (g+1)*group-1 from `group` point clouds (a list
        of (B,N,3+C) tensors) in one pass on group g's side stream.  `ready`: see
        `_order_after_producer`."""
        G = self.group
        assert len(clouds) == G
        side = self.streams[g]
        first = g * G
        self._order_after_producer(side, clouds, ready)
        with torch.cuda.stream(side):
            xyz = torch.cat([c[..., :3] for c in clouds], 0) if G > 1 else clouds[0]
            flat = flatten_geometry(self.backbone.compute_geometry(xyz))
            ev = None
            for k in range(G):
                p = first + k
                if self._consumed[p] is not None:
                    side.wait_event(self._consumed[p])
                torch._foreach_copy_(self._slots[p], [t.chunk(G, 0)[k] for t in flat])
            ev = torch.cuda.Event()
            ev.record(side)
        for k in range(G):
            self._published[first + k] = ev

    def acquire(self, p):
        torch.cuda.current_stream().wait_event(self._published[p])

    def release(self, p):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._consumed[p] = ev

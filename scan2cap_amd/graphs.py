"""hipGraph capture of whole pipeline stages.

The hot path is launch-bound outside the big kernels (the teacher-forced decoder
alone is ~30 steps x ~25 small kernels; the reference additionally forces
CUDA_LAUNCH_BLOCKING=1, scripts/train.py:354).  Because this implementation has
no host synchronisation and only fixed-shape tensors inside forward / loss /
backward, a complete train step can be captured once into a hipGraph and
replayed with a single launch: HIP graphs instead of a tracing compiler.  The
hand-written kernels are captured like any other launch (they are enqueued on
torch's current stream through the C ABI and never allocate).

Shapes that depend on the batch (the decode length T = max(lang_len) - 1) key a
small cache of graphs.
"""
import torch


class GraphedCallable(object):
    """Capture `fn()` (no arguments: it reads static input buffers and writes
    static outputs) after `warmup` eager runs on a side stream."""

    def __init__(self, fn, warmup=3):
        self.fn = fn
        self.warmup = warmup
        self.graph = None
        self.out = None

    def capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self.fn()
        return self

    def __call__(self):
        if self.graph is None:
            self.capture()
        self.graph.replay()
        return self.out


class GraphCache(object):
    """key -> GraphedCallable, e.g. keyed by the decode length of the batch."""

    def __init__(self, make_fn, warmup=3):
        self.make_fn = make_fn
        self.warmup = warmup
        self.graphs = {}

    def __call__(self, key):
        g = self.graphs.get(key)
        if g is None:
            g = GraphedCallable(self.make_fn(key), self.warmup).capture()
            self.graphs[key] = g
        return g()


class GraphedPair(object):
    """Two functions that must run back to back (`second` consumes state `first` leaves
    behind, e.g. the autograd graph of a two-stage backward), captured as TWO hipGraphs so
    that the host can enqueue something between them on every replay -- the asynchronous
    all-reduce of the gradients the first half has finished (scan2cap_amd/parallel.py).
    Both graphs allocate from one memory pool."""

    def __init__(self, first, second, warmup=3):
        self.first, self.second, self.warmup = first, second, warmup
        self.g1 = self.g2 = None
        self.out = None

    def capture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.first()
                self.second()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g1):
            self.out = self.first()
        self.g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g2, pool=self.g1.pool()):
            self.second()
        return self

    def replay_first(self):
        self.g1.replay()
        return self.out

    def replay_second(self):
        self.g2.replay()

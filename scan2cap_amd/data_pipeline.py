"""Host -> device staging for the training / prediction loops.

The reference moves every batch synchronously on the compute stream
(`data_dict[key] = data_dict[key].cuda()`, lib/solver.py:280-287, benchmark/predict.py:
178-181): 187 MB per cfg3 batch = 3.5-3.9 ms of PCIe time in front of a 12.6 ms step.
`DevicePrefetcher` wraps any iterable of host `data_dict`s (e.g. the reference's
DataLoader) and keeps `depth` batches in flight: reusable pinned staging buffers, copies
on a dedicated stream, an event per batch that the consumer's stream waits on -- the
copy of batch i+1 runs under the compute of batch i and no host synchronisation is
added.  Non-tensor entries pass through untouched.
"""
import collections

import torch


class DevicePrefetcher(object):
    def __init__(self, batches, device, depth=2, pin=True):
        self.it = iter(batches)
        self.device = torch.device(device)
        self.depth = max(1, int(depth))
        self.pin = pin and self.device.type == "cuda"
        self.stream = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self.queue = collections.deque()
        self._staging = [dict() for _ in range(self.depth + 1)]   # pinned buffers, reused
        self._slot_event = [None] * (self.depth + 1)              # last H2D out of a slot
        self._slot = 0
        for _ in range(self.depth):
            self._enqueue()

    def _stage(self, slot, key, t):
        """Copy a host tensor into this slot's pinned buffer (allocated once per shape)."""
        buf = self._staging[slot].get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._staging[slot][key] = buf
        buf.copy_(t)
        return buf

    def _enqueue(self):
        try:
            host = next(self.it)
        except StopIteration:
            return
        slot = self._slot
        self._slot = (self._slot + 1) % len(self._staging)
        out, event = {}, None
        if self._slot_event[slot] is not None:
            # the copies that read this slot's pinned buffers were issued depth+1 batches
            # ago; normally long finished -- make it certain before overwriting them
            self._slot_event[slot].synchronize()
        if self.stream is None:
            out = dict(host)
        else:
            with torch.cuda.stream(self.stream):
                for k, v in host.items():
                    if torch.is_tensor(v) and not v.is_cuda:
                        src = self._stage(slot, k, v) if self.pin else v
                        out[k] = src.to(self.device, non_blocking=True)
                    else:
                        out[k] = v
                event = torch.cuda.Event()
                event.record(self.stream)
            self._slot_event[slot] = event
        self.queue.append((out, event))

    def __iter__(self):
        return self

    def __next__(self):
        if not self.queue:
            raise StopIteration
        out, event = self.queue.popleft()
        if event is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(event)
            for v in out.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(cur)        # allocated on the copy stream
        self._enqueue()
        return out

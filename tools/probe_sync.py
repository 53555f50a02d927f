"""Cost of a grid-wide exchange inside one kernel on MI355X (tools/probes/sync_probe.hip):
us per iteration for a counter barrier, a flag-array barrier and tagged-data polling, over all
workgroups and XCD-local.  The number to hold against the ~6 us of a dependent kernel launch in
the decoder chain."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "probes", "sync_probe.hip")
SO = os.path.join(HERE, "probes", "_sync_probe.so")


def build():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", SRC,
                               "-o", SO])
    return ctypes.CDLL(SO)


def main():
    lib = build()
    lib.sync_probe.restype = ctypes.c_float
    lib.sync_probe.argtypes = [ctypes.c_int] * 5 + [ctypes.c_void_p]
    scratch = torch.zeros(8 << 20, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    iters = 2000
    for G in (256, 128, 64, 32):
        for mode, name in ((0, "counter"), (1, "flags"), (2, "tagged data"), (3, "tagged x8"), (4, "x8 sys store")):
            for group in ((1,) if mode == 0 else (1, 8)):
                for V in ((0,) if mode < 2 else (512, 4096)):
                    if mode >= 2 and V // (G // group) > 256:
                        continue
                    if G // group < 1 or (mode >= 2 and V % (G // group)):
                        continue
                    best = min(lib.sync_probe(mode, G, iters, V, group, scratch.data_ptr())
                               for _ in range(3))
                    print("G=%3d  %-12s group=%d V=%4d : %.2f us per exchange"
                          % (G, name, group, V, best), flush=True)


if __name__ == "__main__":
    main()

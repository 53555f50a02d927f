"""The optimizer of the train step: torch.optim.Adam's update in ONE launch (csrc/s2c_optim.hip).

Reference: scripts/train.py:138 `optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.wd)`,
stepped once per batch by lib/solver.py:293-302.  `FusedAdam` IS a torch.optim.Adam (same constructor
arguments, same `state_dict()` layout -- per parameter {"step", "exp_avg", "exp_avg_sq"} -- so
lib/solver.py:501-515's checkpoint.tar loads into it and its own loads into torch.optim.Adam); only
`step()` differs: the first / second moments of every parameter are views into two flat buffers, the
step counts into one small vector, and the update of all tensors is one `s2c_adam_multi` launch
(torch's fused multi-tensor Adam: three launches of ~40 us + a `_foreach_add_` for the counters).
Parameters the kernel does not take (not fp32, not on a GPU, not contiguous, amsgrad / maximize /
differentiable switched on) make the whole optimizer fall back to torch's own step -- loudly, once.
"""
import ctypes
import warnings

import torch

from . import _C

MAX_TENSORS = 128          # S2C_ADAM_MAX_TENSORS (include/s2c_fused.h)


class _AdamTensor(ctypes.Structure):
    """s2c_adam_tensor (include/s2c_fused.h)."""
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("numel", ctypes.c_int),
                ("offset", ctypes.c_int)]


class _AdamArgs(ctypes.Structure):
    """s2c_adam_args (include/s2c_fused.h)."""
    _fields_ = [("n_tensors", ctypes.c_int), ("pad_", ctypes.c_int),
                ("lr", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double),
                ("eps", ctypes.c_double), ("weight_decay", ctypes.c_double),
                ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("step", ctypes.c_void_p), ("counter", ctypes.c_void_p),
                ("first_block", ctypes.c_int * (MAX_TENSORS + 1)), ("pad3_", ctypes.c_int),
                ("t", _AdamTensor * MAX_TENSORS)]


_C.register("s2c_adam_multi", [ctypes.c_void_p, ctypes.c_void_p])


def _chunk():
    lib = _C.load()
    lib.s2c_adam_chunk.restype = ctypes.c_int
    lib.s2c_adam_chunk.argtypes = []
    return lib.s2c_adam_chunk()


class FusedAdam(torch.optim.Adam):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) whose `step()` is one kernel launch."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        # capturable: torch's own step (the fall-back, and what a state_dict is compared with) keeps the
        # step counts as device tensors too, so either path can sit inside a captured hipGraph
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                         capturable=True)
        self._flats = {}          # group index -> dict(m, v, steps, counter, params, offsets)
        self._fallback = None     # None: undecided, False: hand kernel, True: torch's step

    # ---------------------------------------------------------------------------------------
    def _supported(self):
        for g in self.param_groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("differentiable"):
                return False
            if torch.is_tensor(g["lr"]):
                return False
            devs = set()
            for p in g["params"]:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
                        and p.numel() < 2 ** 31):
                    return False
                devs.add(p.device)
            if len(devs) > 1:
                return False
        return True

    def _adopt(self):
        """Flat moment buffers for every group (first call), every parameter's state re-bound to its
        views; values a `load_state_dict` (or torch's lazy init) put there are carried over."""
        for gi, g in enumerate(self.param_groups):
            params = [p for p in g["params"] if p.requires_grad]
            fl = self._flats.get(gi)
            if fl is None or [id(p) for p in fl["params"]] != [id(p) for p in params]:
                if not params:
                    self._flats[gi] = dict(params=[], m=None)
                    continue
                dev = params[0].device
                offsets, total = [], 0
                for p in params:
                    offsets.append(total)
                    total += (p.numel() + 3) // 4 * 4
                if total >= 2 ** 31:
                    raise ValueError("FusedAdam: more than 2^31 parameters in one group")
                fl = dict(params=params, offsets=offsets,
                          m=torch.zeros(total, device=dev), v=torch.zeros(total, device=dev),
                          steps=torch.zeros(len(params), device=dev),
                          counter=torch.zeros(4, dtype=torch.int32, device=dev))
                self._flats[gi] = fl
            if not fl["params"]:
                continue
            for i, (p, off) in enumerate(zip(fl["params"], fl["offsets"])):
                n = p.numel()
                views = {"exp_avg": fl["m"][off:off + n].view_as(p),
                         "exp_avg_sq": fl["v"][off:off + n].view_as(p), "step": fl["steps"][i]}
                st = self.state[p]
                for k, view in views.items():
                    cur = st.get(k)
                    if torch.is_tensor(cur) and cur.data_ptr() == view.data_ptr():
                        continue
                    if cur is not None:
                        with torch.no_grad():
                            view.copy_(torch.as_tensor(cur, dtype=torch.float32).to(view.device).reshape(view.shape))
                    st[k] = view

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        if self._fallback is False:
            self._adopt()                    # the loaded tensors move into the flat buffers

    # ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self._fallback is None:
            self._fallback = not self._supported()
            if self._fallback:
                warnings.warn("scan2cap_amd.optim.FusedAdam: a parameter group is not taken by "
                              "s2c_adam_multi (fp32, contiguous, one GPU, no amsgrad / maximize); "
                              "using torch.optim.Adam's own step")
        if self._fallback:
            super().step()
            return loss
        self._adopt()
        chunk = _chunk()
        for gi, g in enumerate(self.param_groups):
            fl = self._flats[gi]
            params = fl["params"]
            for lo in range(0, len(params), MAX_TENSORS):
                part = params[lo:lo + MAX_TENSORS]
                a = _AdamArgs()
                a.n_tensors = len(part)
                a.lr, (a.beta1, a.beta2) = float(g["lr"]), g["betas"]
                a.eps, a.weight_decay = float(g["eps"]), float(g["weight_decay"])
                a.exp_avg, a.exp_avg_sq = fl["m"].data_ptr(), fl["v"].data_ptr()
                a.step = fl["steps"].data_ptr() + 4 * lo
                a.counter = fl["counter"].data_ptr()
                blocks = 0
                any_grad = False
                for i, p in enumerate(part):
                    gr = p.grad
                    if gr is not None:
                        if gr.is_sparse:
                            raise RuntimeError("Adam does not support sparse gradients")
                        if not (gr.is_contiguous() and gr.dtype == torch.float32 and gr.device == p.device):
                            gr = gr.contiguous().float()
                            p.grad = gr
                        any_grad = True
                    a.t[i] = _AdamTensor(p.data_ptr(), gr.data_ptr() if gr is not None else None,
                                         p.numel(), fl["offsets"][lo + i])
                    a.first_block[i] = blocks
                    blocks += (p.numel() + chunk - 1) // chunk
                a.first_block[len(part)] = blocks
                if not any_grad:
                    continue
                with torch.cuda.device(part[0].device):
                    _C.call("s2c_adam_multi", ctypes.byref(a), _C.stream_ptr())
        return loss

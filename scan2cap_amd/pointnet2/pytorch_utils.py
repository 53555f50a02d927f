"""Shared-MLP building blocks with the reference's module / state_dict naming
(lib/pointnet2/pytorch_utils.py:11-120): `layer{i}.conv.weight`,
`layer{i}.bn.bn.{weight,bias,running_mean,running_var,num_batches_tracked}`.

Only what the CapNet hot path instantiates is provided: SharedMLP (stacks of
1x1 Conv2d [no bias when followed by BN] -> BatchNorm2d -> ReLU), the Conv1d /
Conv2d / FC helpers it is made of, and the BN-momentum scheduler
(pytorch_utils.py:271-296).
"""
import torch.nn as nn

_BN_OF = {1: nn.BatchNorm1d, 2: nn.BatchNorm2d, 3: nn.BatchNorm3d}
_CONV_OF = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}


class _BatchNorm(nn.Sequential):
    """Wrapper that owns one `bn` child (hence the `bn.bn.*` key names);
    weight 1 / bias 0 as pytorch_utils.py:45-46."""

    def __init__(self, channels, dims, name=""):
        super().__init__()
        self.add_module(name + "bn", _BN_OF[dims](channels))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0.0)


class BatchNorm1d(_BatchNorm):
    def __init__(self, in_size, *, name=""):
        super().__init__(in_size, 1, name)


class BatchNorm2d(_BatchNorm):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, 2, name)


class _ConvUnit(nn.Sequential):
    """conv (+bn) (+activation), or the pre-activation order bn, act, conv.
    The conv has a bias only when no BN follows (pytorch_utils.py:87)."""

    def __init__(self, dims, in_size, out_size, kernel_size, stride, padding,
                 activation, bn, init, bias, preact, name):
        super().__init__()
        conv = _CONV_OF[dims](in_size, out_size, kernel_size=kernel_size,
                              stride=stride, padding=padding,
                              bias=bias and not bn)
        init(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0.0)
        norm = _BatchNorm(in_size if preact else out_size, dims) if bn else None
        order = ([("bn", norm), ("activation", activation), ("conv", conv)]
                 if preact else
                 [("conv", conv), ("bn", norm), ("activation", activation)])
        for key, mod in order:
            if mod is not None:
                self.add_module(name + key, mod)


class Conv1d(_ConvUnit):
    def __init__(self, in_size, out_size, *, kernel_size=1, stride=1, padding=0,
                 activation=nn.ReLU(inplace=True), bn=False,
                 init=nn.init.kaiming_normal_, bias=True, preact=False, name=""):
        super().__init__(1, in_size, out_size, kernel_size, stride, padding,
                         activation, bn, init, bias, preact, name)


class Conv2d(_ConvUnit):
    def __init__(self, in_size, out_size, *, kernel_size=(1, 1), stride=(1, 1),
                 padding=(0, 0), activation=nn.ReLU(inplace=True), bn=False,
                 init=nn.init.kaiming_normal_, bias=True, preact=False, name=""):
        super().__init__(2, in_size, out_size, kernel_size, stride, padding,
                         activation, bn, init, bias, preact, name)


class SharedMLP(nn.Sequential):
    """args = [C0, C1, ..., Ck]: k Conv2d units named layer0..layer{k-1}."""

    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True),
                 preact=False, first=False, name=""):
        super().__init__()
        for i in range(len(args) - 1):
            plain = first and preact and i == 0  # pytorch_utils.py:30-33
            self.add_module(
                name + "layer{}".format(i),
                Conv2d(args[i], args[i + 1], bn=bn and not plain,
                       activation=None if plain else activation,
                       preact=preact))


class FC(nn.Sequential):
    def __init__(self, in_size, out_size, *, activation=nn.ReLU(inplace=True),
                 bn=False, init=None, preact=False, name=""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0.0)
        norm = BatchNorm1d(in_size if preact else out_size) if bn else None
        order = ([("bn", norm), ("activation", activation), ("fc", fc)]
                 if preact else
                 [("fc", fc), ("bn", norm), ("activation", activation)])
        for key, mod in order:
            if mod is not None:
                self.add_module(name + key, mod)


def set_bn_momentum_default(bn_momentum):
    def fn(m):
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = bn_momentum
    return fn


class BNMomentumScheduler(object):
    """pytorch_utils.py:271-296: sets every BN's momentum to bn_lambda(epoch)."""

    def __init__(self, model, bn_lambda, last_epoch=-1,
                 setter=set_bn_momentum_default):
        if not isinstance(model, nn.Module):
            raise RuntimeError("Class '{}' is not a PyTorch nn Module".format(
                type(model).__name__))
        self.model, self.setter, self.lmbd = model, setter, bn_lambda
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        self.model.apply(self.setter(self.lmbd(epoch)))

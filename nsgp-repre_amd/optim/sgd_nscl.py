"""SGDNSCL / SGDNSCLNA -- interface mirror of mmdet/engine/optimizers/SGD_NSCL.py:15-51 and
SGD_NSCL_NoAdaptive.py:14-54, running on the HIP projected-step plan."""
import torch

from .. import _lib
from ..registry import OPTIMIZERS, register
from .base import NSCLOptimizerBase


@register(OPTIMIZERS)
class SGDNSCL(NSCLOptimizerBase):
    """SGD with momentum whose final update is right-multiplied by the layer's null-space
    projector.  Same constructor as the reference (SGD_NSCL.py:40-41)."""
    _kind = _lib.NSGP_OPT_SGD
    _threshold_rule = "sgd"

    def __init__(self, params, lr=1e-3, momentum=0, dampening=0, nesterov=False, svd=False, thres=1.001,
                 weight_decay=0):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, nesterov=nesterov,
                        weight_decay=weight_decay, svd=svd, thres=thres)
        super().__init__(params, defaults)

    def _init_state(self, p, state, group):
        state["step"] = 0
        state["previous_grad"] = torch.zeros_like(p.data)

    def _state_tensors(self, state, group):
        return state["previous_grad"], None, None

    def _fill_hyper(self, h, group, step):
        h.lr = group["lr"]
        h.momentum = group["momentum"]
        h.one_minus_dampening = 1 - group["dampening"]
        h.weight_decay = group["weight_decay"]
        h.nesterov = int(bool(group["nesterov"]))
        h.first_step = int(step == 1)


@register(OPTIMIZERS)
class SGDNSCLNA(SGDNSCL):
    """Non-adaptive variant: the null space is every direction whose singular value is
    ``<= sigma_min * thres`` (SGD_NSCL_NoAdaptive.py:157-158)."""

    def __init__(self, params, lr=1e-3, momentum=0, dampening=0, nesterov=False, svd=False, thres=1.001,
                 weight_decay=0):
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, nesterov=nesterov, svd=svd,
                         thres=thres, weight_decay=weight_decay)
        self.thres = thres

    def _null_space_start(self, group, n):
        sv = self.eigens[n]["eigen_value"]
        ind = sv <= sv[-1] * group["thres"]
        return int(ind.to(torch.int8).argmax().item())

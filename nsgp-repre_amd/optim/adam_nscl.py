"""AdamWNSCL / AdamNSCL -- interface mirror of mmdet/engine/optimizers/AdamW_NSCL.py:15-64 and
Adam_NSCL.py:15-64, running on the HIP projected-step plan."""
import math

import torch

from .. import _lib
from ..registry import OPTIMIZERS, register
from .base import NSCLOptimizerBase


class _AdamFamily(NSCLOptimizerBase):
    _kind = _lib.NSGP_OPT_ADAM
    _threshold_rule = "adam"
    _decoupled = False

    def _check(self, lr, betas, eps):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))

    def __setstate__(self, state):
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("amsgrad", False)

    def _init_state(self, p, state, group):
        state["step"] = 0
        state["exp_avg"] = torch.zeros_like(p.data)
        state["exp_avg_sq"] = torch.zeros_like(p.data)
        if group["amsgrad"]:
            state["max_exp_avg_sq"] = torch.zeros_like(p.data)

    def _state_tensors(self, state, group):
        return state["exp_avg"], state["exp_avg_sq"], state.get("max_exp_avg_sq")

    def _fill_hyper(self, h, group, step):
        beta1, beta2 = group["betas"]
        h.lr = group["lr"]
        h.beta1, h.beta2 = beta1, beta2
        h.one_minus_beta1, h.one_minus_beta2 = 1 - beta1, 1 - beta2
        h.eps = group["eps"]
        h.amsgrad = int(bool(group["amsgrad"]))
        # AdamW_NSCL.py:244-248, in double like the reference's Python floats
        h.step_size = group["lr"] * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
        if self._decoupled:
            h.weight_decay = 0.0
            h.decoupled_decay = group["lr"] * group["weight_decay"]   # AdamW_NSCL.py:87
        else:
            h.weight_decay = group["weight_decay"]                    # Adam_NSCL.py:229-230
            h.decoupled_decay = 0.0


@register(OPTIMIZERS)
class AdamWNSCL(_AdamFamily):
    """Adam moments + decoupled decay, the whole update projected (AdamW_NSCL.py:39-40, 87)."""
    _decoupled = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, svd=False, thres=1.001, weight_decay=0,
                 amsgrad=False):
        self._check(lr, betas, eps)
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, svd=svd, thres=thres)
        super().__init__(params, defaults)


@register(OPTIMIZERS)
class AdamNSCL(_AdamFamily):
    """Adam with L2 folded into the gradient; every projector is Frobenius-normalised
    (Adam_NSCL.py:39-40, 183, 229-230)."""
    _decoupled = False
    _normalise_all = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, svd=False, thres=0.99, weight_decay=0,
                 amsgrad=False):
        self._check(lr, betas, eps)
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, svd=svd, thres=thres)
        super().__init__(params, defaults)

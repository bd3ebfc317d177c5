"""EWC regulariser on the BatchNorm parameters -- mirror of
``register_params`` / ``load_importance`` / ``calculate_save_importance`` / ``EWCHook``
(mmdet/engine/runner/nsrunner_roi_replay.py:946-1073).

Per step the reference evaluates ``1000 * sum(F * (theta - theta*)**2)`` parameter by parameter
(~8 tiny launches x ~106 BN tensors).  Here the loss is ONE multi-tensor HIP launch (+ a one-block
finish) and its gradient one more, wired into autograd with a custom Function.
"""
import ctypes as C
import os.path as osp
from collections import defaultdict
from typing import Dict

import torch

from .. import _lib
from .safe_load import load_handoff

EWC_WEIGHT = 1000.0            # runner:1068
IGNORE_NAMES = ["teacher_model"]
MUST_NAMES = ["bn"]


def register_params(model) -> Dict[str, torch.nn.Parameter]:
    """runner:1010-1031: parameters whose name contains "bn" and not "teacher_model"."""
    reg = {}
    for n, p in model.named_parameters():
        if any(k in n for k in IGNORE_NAMES):
            continue
        if len(MUST_NAMES) == 0 or any(k in n for k in MUST_NAMES):
            reg[n] = p
    return reg


def load_importance(path: str, device):
    """``ewc_reg_terms_ewc.pth``: {'importance': {name: [T x (1,*shape)]}, 'task_param': {...}} (runner:996-999)."""
    return load_handoff(path, device)      # the reference writes defaultdict(list) containers (runner:957-958)


class _EwcFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, reg, *thetas):
        lib = _lib.load_library()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        loss = torch.empty((), dtype=torch.float32, device=thetas[0].device)
        _lib.check(lib.nsgp_ewc_loss(C.c_void_p(reg.table.data_ptr()), reg.n, reg.weight,
                                     C.c_void_p(reg.partials.data_ptr()), C.c_void_p(loss.data_ptr()), stream), "nsgp_ewc_loss")
        ctx.reg = reg
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        reg = ctx.reg
        lib = _lib.load_library()
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        go = grad_out.detach().reshape(()).float().contiguous()
        _lib.check(lib.nsgp_ewc_grad(C.c_void_p(reg.table.data_ptr()), reg.n, reg.weight, C.c_void_p(go.data_ptr()), stream),
                   "nsgp_ewc_grad")
        return (None,) + tuple(reg.grad_views)


class EWCRegulariser:
    """``loss = 1000 * sum_n sum(F_n * (theta_n - theta*_n)**2)`` over the registered parameters that
    require grad, differentiable w.r.t. the parameters."""

    def __init__(self, reg_params: Dict[str, torch.nn.Parameter], ewc_reg_terms: dict, weight: float = EWC_WEIGHT):
        self.weight = float(weight)
        self.names = [n for n, p in reg_params.items() if p.requires_grad]
        self.params = [reg_params[n] for n in self.names]
        self.n = len(self.params)
        if self.n == 0:
            return
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("EWCRegulariser runs on the GPU only (no CPU fallback)")
        self._keep = []
        total = sum(p.numel() for p in self.params)
        self.grad_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad_views, rows, off = [], [], 0
        for n, p in zip(self.names, self.params):
            imp = torch.cat(ewc_reg_terms["importance"][n], dim=0).to(dev).float().contiguous()
            old = torch.cat(ewc_reg_terms["task_param"][n], dim=0).to(dev).float().contiguous()
            if imp.shape != old.shape or tuple(imp.shape[1:]) != tuple(p.shape):
                raise ValueError(f"{n}: importance/task_param shapes {tuple(imp.shape)}/{tuple(old.shape)} vs param {tuple(p.shape)}")
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError(f"{n}: parameters must be contiguous fp32")
            gv = self.grad_flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            self._keep += [imp, old]
            self.grad_views.append(gv)
            rows.append([p.data_ptr(), imp.data_ptr(), old.data_ptr(), gv.data_ptr(), p.numel(), imp.shape[0]])
        self.table = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.partials = torch.empty(self.n, dtype=torch.float64, device=dev)

    def __call__(self) -> torch.Tensor:
        if self.n == 0:
            return 0
        return _EwcFunction.apply(self, *self.params)


class EWCHook:
    """Wraps ``module.loss`` and adds ``ewc_loss`` to the loss dict (runner:1038-1073)."""

    def __init__(self, module, reg_params, ewc_reg_terms):
        self.module = module
        self.reg_params = reg_params
        self.ewc_reg_terms = ewc_reg_terms
        self.ori_loss = module.loss
        self.reg = EWCRegulariser(reg_params, ewc_reg_terms)

    def __call__(self, *args, **kwargs):
        result = self.ori_loss(*args, **kwargs)
        if self.reg.n > 0:
            result.update({"ewc_loss": self.reg()})
        return result


@torch.no_grad()
def accumulate_importance(importance: Dict[str, torch.Tensor], reg_params: Dict[str, torch.nn.Parameter],
                          batch_len: int, loader_len: int) -> None:
    """One batch of ``calculate_save_importance`` (runner:978-981): F += grad**2 * len(batch)/len(loader)."""
    names = [n for n in importance if reg_params[n].grad is not None]
    if names:
        torch._foreach_addcmul_([importance[n] for n in names], [reg_params[n].grad for n in names],
                                [reg_params[n].grad for n in names], value=batch_len / loader_len)


def save_importance(work_dir: str, ewc_reg_terms: dict, importance, reg_params) -> dict:
    """runner:985-989: append this task's (importance, parameters) and write ``ewc_reg_terms_ewc.pth``."""
    if len(ewc_reg_terms) == 0:
        ewc_reg_terms = {"importance": defaultdict(list), "task_param": defaultdict(list)}
    for n, p in reg_params.items():
        ewc_reg_terms["importance"][n].append(importance[n].unsqueeze(0))
        ewc_reg_terms["task_param"][n].append(p.unsqueeze(0).clone().detach())
    torch.save({k: dict(v) for k, v in ewc_reg_terms.items()}, osp.join(work_dir, "ewc_reg_terms_ewc.pth"))
    return ewc_reg_terms

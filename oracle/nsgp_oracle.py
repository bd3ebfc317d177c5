"""CPU oracle for the NSGP-RePRE hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

A from-scratch CPU restatement (torch-CPU fp32 + numpy) of what the reference
(yyl404/NSGP-RePRE) computes on the path SURVEY.md section 8 scopes.  Every
function cites the reference file:line it follows (paths relative to the
reference repo root).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file; the product package
(``nsgp-repre_amd/``) never does, and fails loudly without its HIP library.

Parity pin: the reference ships no tests/golden vectors for this path
(SURVEY.md section 4), so this oracle is pinned against outputs of the
reference itself, produced in the build container by
``tests/golden/make_golden.py`` (which executes the reference's own files) and
committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays
them.

Arithmetic type: fp32 for everything the reference does in fp32; integer /
bool for ranks, masks and prototype indices.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# a2 / a3 / a4: per-tensor optimizer updates
# ----------------------------------------------------------------------------


def sgd_get_update(grad: torch.Tensor, p: torch.Tensor, state: dict, *, lr: float,
                   momentum: float = 0.0, dampening: float = 0.0,
                   weight_decay: float = 0.0, nesterov: bool = False) -> torch.Tensor:
    """mmdet/engine/optimizers/SGD_NSCL.py:387-415 (``get_update``).

    Mutates ``grad`` in place (weight decay, and the Nesterov add) and the
    momentum buffer ``state['previous_grad']`` exactly like the reference.
    Step 1 stores the raw (decayed) gradient in the buffer (``:406``); later
    steps do ``buf = m*buf + (1-dampening)*grad`` (``:404``).  The momentum
    buffer therefore holds UN-projected gradients.
    """
    if len(state) == 0:
        state["step"] = 0
        state["previous_grad"] = torch.zeros_like(p)
    buf = state["previous_grad"]
    state["step"] += 1
    if weight_decay != 0:
        grad.add_(p, alpha=weight_decay)
    if momentum != 0:
        if state["step"] > 1:
            buf.mul_(momentum).add_(grad, alpha=1 - dampening)
        else:
            buf.add_(grad)
        if nesterov:
            grad.add_(buf, alpha=momentum)
        else:
            grad = buf
    return -(lr * grad)


def adam_moments_update(grad: torch.Tensor, p: torch.Tensor, state: dict, *, lr: float,
                        betas=(0.9, 0.999), eps: float = 1e-8, amsgrad: bool = False,
                        l2_weight_decay: float = 0.0) -> torch.Tensor:
    """mmdet/engine/optimizers/AdamW_NSCL.py:212-250 / Adam_NSCL.py:207-247.

    ``l2_weight_decay`` is the Adam_NSCL-only ``grad += wd*p`` placed after
    ``step += 1`` (Adam_NSCL.py:229-230); AdamW passes 0 here and applies the
    decoupled decay in ``adamw_step_tensor``.
    """
    if len(state) == 0:
        state["step"] = 0
        state["exp_avg"] = torch.zeros_like(p)
        state["exp_avg_sq"] = torch.zeros_like(p)
        if amsgrad:
            state["max_exp_avg_sq"] = torch.zeros_like(p)
    exp_avg, exp_avg_sq = state["exp_avg"], state["exp_avg_sq"]
    beta1, beta2 = betas
    state["step"] += 1
    if l2_weight_decay != 0:
        grad.add_(p, alpha=l2_weight_decay)
    exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    if amsgrad:
        torch.max(state["max_exp_avg_sq"], exp_avg_sq, out=state["max_exp_avg_sq"])
        denom = state["max_exp_avg_sq"].sqrt().add_(eps)
    else:
        denom = exp_avg_sq.sqrt().add_(eps)
    bc1 = 1 - beta1 ** state["step"]
    bc2 = 1 - beta2 ** state["step"]
    step_size = lr * math.sqrt(bc2) / bc1
    return -step_size * exp_avg / denom


class HeadBasis:
    """A projector given by the r REMOVED directions instead of the dense matrix: ``P = c (I - U U^T)``, ``U = V[:, :r]``
    (SGD_NSCL.py:270-285 builds the same matrix from the other side, ``V[:, r:] V[:, r:]^T``, ``/ ||P||_F`` for backbone
    layers; the two are equal for an orthonormal V).  ``project_update`` applies it as ``c (u - (u U) U^T)``, the form the
    north star names (g <- g - U (U^T g)) and the product's default step runs."""

    def __init__(self, U: torch.Tensor, normalise: bool):
        self.U = U
        D = U.shape[0]
        self.c = 1.0 / float(torch.norm(torch.eye(D, dtype=U.dtype) - U @ U.t())) if normalise else 1.0


def project_update(update: torch.Tensor, P) -> torch.Tensor:
    """mmdet/engine/optimizers/SGD_NSCL.py:82-94: right-multiply the final
    update by the projector; 4-D weights are viewed as ``[Cout, Cin*kh*kw]``."""
    if P is None:
        return update
    if isinstance(P, HeadBasis):
        u2 = update.reshape(update.size(0), -1)
        return (P.c * (u2 - torch.mm(torch.mm(u2, P.U), P.U.t()))).view_as(update)
    if update.dim() == 4:
        return torch.mm(update.view(update.size(0), -1), P).view_as(update)
    return torch.mm(update, P)


def sgd_nscl_step(names: Sequence[str], params: Sequence[torch.Tensor],
                  grads: Sequence[torch.Tensor], states: Sequence[dict],
                  transforms: Dict[str, torch.Tensor], *, lr, momentum=0.0, dampening=0.0,
                  weight_decay=0.0, nesterov=False, svd=True) -> None:
    """mmdet/engine/optimizers/SGD_NSCL.py:59-96 (``SGDNSCL.step``)."""
    for n, p, g, st in zip(names, params, grads, states):
        u = sgd_get_update(g, p, st, lr=lr, momentum=momentum, dampening=dampening,
                           weight_decay=weight_decay, nesterov=nesterov)
        P = transforms.get(n) if (svd and len(transforms) > 0) else None
        p.add_(project_update(u, P))


def adamw_nscl_step(names, params, grads, states, transforms, *, lr, betas=(0.9, 0.999),
                    eps=1e-8, weight_decay=0.0, amsgrad=False, svd=True) -> None:
    """mmdet/engine/optimizers/AdamW_NSCL.py:66-103: the decoupled decay term
    ``- lr*wd*p`` is part of the update BEFORE projection (``:87``)."""
    for n, p, g, st in zip(names, params, grads, states):
        u = adam_moments_update(g, p, st, lr=lr, betas=betas, eps=eps, amsgrad=amsgrad)
        u = u - lr * weight_decay * p
        P = transforms.get(n) if (svd and len(transforms) > 0) else None
        p.add_(project_update(u, P))


def adam_nscl_step(names, params, grads, states, transforms, *, lr, betas=(0.9, 0.999),
                   eps=1e-8, weight_decay=0.0, amsgrad=False, svd=True) -> None:
    """mmdet/engine/optimizers/Adam_NSCL.py:66-102 (L2 decay folded into grad)."""
    for n, p, g, st in zip(names, params, grads, states):
        u = adam_moments_update(g, p, st, lr=lr, betas=betas, eps=eps, amsgrad=amsgrad,
                                l2_weight_decay=weight_decay)
        P = transforms.get(n) if (svd and len(transforms) > 0) else None
        p.add_(project_update(u, P))


# ----------------------------------------------------------------------------
# a5 / a6 / a7: spectrum -> rank -> projector
# ----------------------------------------------------------------------------


def eigens(C: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """mmdet/engine/optimizers/SGD_NSCL.py:377: ``_, s, V = torch.svd(C, some=False)``
    (descending singular values, right singular vectors in columns)."""
    _, s, V = torch.svd(C, some=False)
    return s, V


def gaussian_filter1d_reflect(x: np.ndarray, sigma: float = 10.0, truncate: float = 4.0) -> np.ndarray:
    """What ``scipy.ndimage.gaussian_filter1d(points, sigma=10)`` computes at the
    reference call site SGD_NSCL.py:139 (scipy defaults: order 0,
    ``mode='reflect'`` = half-sample symmetric ``d c b a | a b c d | d c b a``,
    ``truncate=4`` -> radius 40).  scipy builds the kernel in float64,
    accumulates in float64 and writes the input dtype (fp32 in -> fp32 out)."""
    x = np.asarray(x)
    radius = int(truncate * float(sigma) + 0.5)
    t = np.arange(-radius, radius + 1, dtype=np.float64)
    w = np.exp(-0.5 / (sigma * sigma) * t * t)
    w /= w.sum()
    n = x.shape[0]
    idx = np.arange(-radius, n + radius)
    period = 2 * n
    idx = np.mod(idx, period)
    idx = np.where(idx >= n, period - 1 - idx, idx)
    xp = x.astype(np.float64)[idx]
    out = np.empty(n, dtype=np.float64)
    # symmetric kernel: correlation == convolution; sum in scipy's order
    # (centre first, then pairs outwards) so float64 rounding matches.
    for i in range(n):
        c = i + radius
        acc = xp[c] * w[radius]
        for k in range(1, radius + 1):
            acc += (xp[c - k] + xp[c + k]) * w[radius + k]
        out[i] = acc
    return out.astype(x.dtype)


def elbow_index(points: np.ndarray, offset: float = 0.0, rule: str = "sgd") -> int:
    """mmdet/engine/optimizers/SGD_NSCL.py:134-170 (rule ``'sgd'``; identical
    copy in standard_roi_replay_head.py:301-330) and AdamW_NSCL.py:105-127
    (rule ``'adam'``): returns ``i_thres``; the mask is True for ``i >= i_thres``.
    """
    points = np.asarray(points)
    assert points.ndim == 1
    n = len(points)
    if n >= 128:
        fil = gaussian_filter1d_reflect(points, sigma=10)
        d1 = fil[:-1] - fil[1:]
        d2 = d1[:-1] - d1[1:]
        drop = int(n * 0.03 / 2)
        assert n - drop >= 10
        valid = d2[drop:-drop]
        thres_val = points[int(np.argmax(valid)) + int((n - len(valid)) / 2)]
    else:
        d1 = points[:-1] - points[1:]
        d2 = d1[:-1] - d1[1:]
        thres_val = points[int(np.argmax(d2)) + int((n - len(d2)) / 2)]
    i_thres = int(np.arange(n)[points >= thres_val].max())
    if rule == "sgd":
        if -1 <= offset <= 1:
            i_thres = max(0, min(i_thres + int(offset * i_thres), n - 1))
        else:
            i_thres = max(min(i_thres + int(offset), n - 1), 0)
    elif rule == "adam":
        if -1 < offset < 1:
            i_thres = max(0, min(i_thres + int(offset * (n - i_thres)), n - 1))
        else:
            i_thres = max(min(i_thres + int(offset), n - 1), 0)
    else:
        raise ValueError(rule)
    return i_thres


def adaptive_threshold(svals: torch.Tensor, offset: float = 0.0, rule: str = "sgd") -> torch.Tensor:
    """Bool mask ``[D]``, True for the small-sigma tail (SGD_NSCL.py:172-177)."""
    i = elbow_index(svals.cpu().numpy(), offset, rule)
    m = torch.zeros(svals.shape[0], dtype=torch.bool)
    m[i:] = True
    return m


def na_threshold(svals: torch.Tensor, thres: float) -> torch.Tensor:
    """mmdet/engine/optimizers/SGD_NSCL_NoAdaptive.py:157-158: ``s <= s_min*thres``."""
    return svals <= svals[-1] * thres


def build_projector(V: torch.Tensor, mask: torch.Tensor, normalise: bool) -> torch.Tensor:
    """mmdet/engine/optimizers/SGD_NSCL.py:270-285: ``P = V[:,mask] V[:,mask]^T``;
    divided by its Frobenius norm when ``normalise`` (name contains 'backbone'
    for SGD/AdamW/NA; always for Adam_NSCL.py:183)."""
    basis = V[:, mask]
    P = torch.mm(basis, basis.transpose(1, 0))
    if normalise:
        P = P / torch.norm(P)
    return P


def get_transforms(names: Sequence[str], fea_in: Dict[str, torch.Tensor], offset: float = 0.0,
                   rule: str = "sgd", normalise_all: bool = False) -> Tuple[dict, dict]:
    """get_eigens + get_transforms (SGD_NSCL.py:203-290, 360-380)."""
    eig, tr = {}, {}
    for n in names:
        if n not in fea_in:
            continue
        s, V = eigens(fea_in[n])
        eig[n] = dict(eigen_value=s, eigen_vector=V)
        mask = adaptive_threshold(s, offset, rule)
        tr[n] = build_projector(V, mask, normalise_all or ("backbone" in n))
    return eig, tr


# ----------------------------------------------------------------------------
# a8 / a9: covariance accumulation
# ----------------------------------------------------------------------------


def unfold_mean_batch(x: torch.Tensor, kernel_size, stride, padding) -> torch.Tensor:
    """mmdet/engine/runner/nsrunner_roi_replay.py:908-913: batch-mean first,
    then every receptive-field patch as a row ``[L, Cin*kh*kw]`` (channel-major,
    then kernel row, then kernel column -- ``F.unfold`` order; dilation/groups
    are ignored by the reference).  Written with explicit slicing, not
    ``F.unfold``, so the restatement is independent of the call it restates."""
    kh, kw = kernel_size
    sh, sw = stride
    ph, pw = padding
    xm = torch.mean(x, 0, True)[0]  # [Cin,H,W]
    cin, H, W = xm.shape
    xp = torch.zeros(cin, H + 2 * ph, W + 2 * pw, dtype=x.dtype)
    xp[:, ph:ph + H, pw:pw + W] = xm
    Ho = (H + 2 * ph - kh) // sh + 1
    Wo = (W + 2 * pw - kw) // sw + 1
    cols = torch.empty(Ho * Wo, cin, kh, kw, dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i:i + sh * (Ho - 1) + 1:sh, j:j + sw * (Wo - 1) + 1:sw]
            cols[:, :, i, j] = patch.reshape(cin, -1).t()
    return cols.reshape(Ho * Wo, cin * kh * kw)


def cov_conv2d(x, kernel_size, stride, padding) -> torch.Tensor:
    """compute_cov (conv branch) + update_cov: ``X^T X`` (runner:908-913, 930)."""
    X = unfold_mean_batch(x, kernel_size, stride, padding)
    return torch.mm(X.t(), X)


def cov_linear(x: torch.Tensor) -> torch.Tensor:
    """compute_cov (linear branch), runner:901-902: ``mean(x, 0, keepdim)`` -> X^T X.
    NOTE the mean is over dim 0 only, whatever the input rank."""
    X = torch.mean(x, 0, True)
    X = X.reshape(-1, X.shape[-1]) if X.dim() > 2 else X
    return torch.mm(X.t(), X)


def update_cov(fea_in: dict, key: str, cov: torch.Tensor) -> None:
    """runner:923-934: first call assigns, later calls add."""
    if key not in fea_in or len(fea_in[key]) == 0:
        fea_in[key] = cov
    else:
        fea_in[key] = fea_in[key] + cov


def filter_ignore(fea_in: dict, ignore_keys: Sequence[str]) -> dict:
    """runner:643-653: drop keys for which ``re.match(ignore_key, name)``."""
    import re
    return {k: v for k, v in fea_in.items()
            if not any(bool(re.match(ik, k)) for ik in ignore_keys)}


# ----------------------------------------------------------------------------
# a15: prototype bank
# ----------------------------------------------------------------------------


def prototype_select(Fc: torch.Tensor, max_proto: int = 10, thr: float = 0.6,
                     order: Optional[torch.Tensor] = None,
                     saved_masks: Optional[List[torch.Tensor]] = None, stable: bool = False):
    """mmdet/models/roi_heads/standard_roi_replay_head.py:411-446 for ONE class.

    ``Fc`` is ``[N, D]`` (the class's RoI features).  Returns
    ``(coarse[1,D], fine list of [1,D], masks list of bool[N], centre ids, counts)``.
    ``order``: the visiting order of rows.  The reference takes it from torch's
    UNSTABLE CPU ``sort(descending=True)`` (``:421``); the default here is that
    very call (third-party arithmetic the reference itself calls, same torch
    build on the GPU box), ``stable=True`` gives ties -> lowest row index, and a
    recorded order can be injected for replay.  ``saved_masks`` replays
    ``mask.pth`` (``:425-433``).
    """
    N = Fc.shape[0]
    coarse = torch.mean(Fc, dim=0, keepdim=True)
    flat = Fc.reshape(N, -1)
    nrm = flat / flat.norm(dim=-1, keepdim=True)
    sim = nrm @ nrm.t()
    sim_mask = sim >= thr
    counts = sim_mask.long().sum(dim=-1)
    if order is None:
        order = (torch.sort(counts, descending=True, stable=True).indices if stable
                 else counts.sort(dim=-1, descending=True).indices)
    cnt_sorted = counts[order]
    thr_cnt = cnt_sorted[-N // 3]  # python precedence: (-N)//3 -> position N-ceil(N/3)
    covered = counts <= thr_cnt
    masks: List[torch.Tensor] = list(saved_masks) if saved_masks is not None else []
    fine, centres = [], []
    for pc in range(max_proto - 1):
        for id_ in order.tolist():
            if pc < len(masks):
                m = masks[pc]
                cid = -1
            else:
                if bool(covered[id_]):
                    continue
                m = sim_mask[id_]
                masks.append(m)
                cid = id_
            covered = torch.logical_or(covered, m)
            fine.append(torch.mean(Fc[m], dim=0, keepdim=True))
            centres.append(cid)
            break
    return coarse, fine, masks, centres, counts


def build_bank(feats: torch.Tensor, cls_targets: torch.Tensor, task_split: Sequence[int],
               task_id: int, max_proto: int = 10, orders: Optional[dict] = None,
               saved: Optional[list] = None):
    """standard_roi_replay_head.py:397-452: loop over old classes
    ``range(task_split[0], task_split[task_id-1])``; bank rows are
    [coarse_c, fine_c...] per class; labels repeat the class id."""
    bank, labels, all_masks, all_centres = [], [], [], []
    for c in range(task_split[0], task_split[task_id - 1]):
        Fc = feats[cls_targets == c]
        sm = saved[c] if (saved is not None and c < len(saved)) else None
        coarse, fine, masks, centres, _ = prototype_select(
            Fc, max_proto, order=None if orders is None else orders.get(c), saved_masks=sm)
        bank.append(coarse)
        labels.append(c)
        for f in fine:
            bank.append(f)
            labels.append(c)
        all_masks.append(masks)
        all_centres.append(centres)
    return torch.cat(bank, 0), torch.tensor(labels, dtype=torch.long), all_masks, all_centres


# ----------------------------------------------------------------------------
# a16 / a17: task head forward and the replay loss
# ----------------------------------------------------------------------------


def task_head_forward(x: torch.Tensor, shared_fcs: Sequence[Tuple[torch.Tensor, torch.Tensor]],
                      fc_cls: Sequence[Tuple[torch.Tensor, torch.Tensor]],
                      fc_reg: Sequence[Tuple[torch.Tensor, torch.Tensor]],
                      task_id: int, n_tasks_plus_one: int, reg_class_agnostic: bool = False):
    """convfc_bbox_head_task.py:209-288 for the Shared2FC shape: flatten ->
    (Linear+ReLU)x2 -> per-task fc_cls (future tasks: input detached, output
    -inf; the last entry is the background head and always live) || per-task
    fc_reg (future tasks -> 0) -> cat.  ``n_tasks_plus_one == len(task_split)``.
    Each weight pair is ``(W[out,in], b[out])``."""
    h = x.flatten(1)
    for W, b in shared_fcs:
        h = torch.relu(F.linear(h, W, b))
    preds = []
    for i, (W, b) in enumerate(fc_cls):
        future = (i + 1 > task_id) and (i + 1 != n_tasks_plus_one)
        o = F.linear(h.detach() if future else h, W, b)
        if future:
            o = torch.full_like(o, float("-inf"))
        preds.append(o)
    cls = torch.cat(preds, dim=-1)
    preds = []
    for i, (W, b) in enumerate(fc_reg):
        future = (i + 1 > task_id) and not reg_class_agnostic
        o = F.linear(h.detach() if future else h, W, b)
        if future:
            o = torch.zeros_like(o)
        preds.append(o)
    reg = torch.cat(preds, dim=-1)
    return cls, reg


def replay_loss_from_scores(cls_score: torch.Tensor, labels: torch.Tensor, pre_idx: int) -> torch.Tensor:
    """standard_roi_replay_head.py:496-499: keep columns ``[:pre_idx]`` + the last
    (background) column, then ``CE(softmax(.), labels)`` -- the double softmax is
    the reference's behaviour and is reproduced."""
    s = torch.cat([cls_score[:, :pre_idx], cls_score[:, -1:]], dim=-1)
    return F.cross_entropy(s.softmax(dim=-1), labels)


# ----------------------------------------------------------------------------
# 8f-1: EWC regulariser on the BatchNorm parameters
# ----------------------------------------------------------------------------


def ewc_registered(names: Sequence[str]) -> List[str]:
    """runner:1010-1031 ``register_params``: names containing "bn" and not "teacher_model"."""
    return [n for n in names if ("bn" in n) and ("teacher_model" not in n)]


def ewc_loss(params: Dict[str, torch.Tensor], importance: Dict[str, List[torch.Tensor]],
             task_param: Dict[str, List[torch.Tensor]], weight: float = 1000.0) -> torch.Tensor:
    """runner:1055-1073 ``EWCHook.__call__``: sum_n weight * sum((cat(F_n) * (theta_n - cat(theta*_n))**2))
    over parameters that require grad, accumulated in that order."""
    total = 0
    for n, p in params.items():
        if not p.requires_grad:
            continue
        F_ = torch.cat(importance[n], dim=0)
        old = torch.cat(task_param[n], dim=0)
        new = p.unsqueeze(0).expand(old.shape)
        total = total + weight * (F_ * (new - old) ** 2).sum()
    return total


# ----------------------------------------------------------------------------
# 8f-2: teacher pseudo-label filter  -- PARITY UNPINNED: the reference calls torchvision.ops.box_iou
# and mmengine's InstanceData, both absent here, so no golden could be produced; this restates the
# loop of faster_rcnn_roi_replay.py:78-108 with torchvision's published box_iou formula.
# ----------------------------------------------------------------------------


def box_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torchvision.ops.box_iou: xyxy boxes, inter / (area_a + area_b - inter), no +1."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None, :] - inter)


def pseudo_label_filter(boxes: torch.Tensor, scores: torch.Tensor, gt: torch.Tensor, rpn_thresh: float,
                        roi_thresh: float, iou_thresh: float = 0.7):
    """faster_rcnn_roi_replay.py:78-108 for one image: the RoI-side ground truth GROWS inside the loop,
    so box k is also tested against earlier accepted pseudo boxes."""
    add_rpn = torch.zeros(len(boxes), dtype=torch.bool)
    add_roi = torch.zeros(len(boxes), dtype=torch.bool)
    cur = gt.clone()
    for k in range(len(boxes)):
        max_iou = box_iou(boxes[k:k + 1], cur).max().item() if len(cur) > 0 else 0.0
        if max_iou > iou_thresh:
            continue
        if scores[k] > rpn_thresh:
            add_rpn[k] = True
        if scores[k] > roi_thresh:
            add_roi[k] = True
            cur = torch.cat([cur, boxes[k:k + 1]])
    return add_rpn, add_roi


def nms_greedy(boxes: torch.Tensor, scores: torch.Tensor, iou_thresh: float, idxs: torch.Tensor = None,
               max_keep: int = None) -> torch.Tensor:
    """mmcv.ops.nms / batched_nms as the teacher's ``predict`` uses them (faster_rcnn_roi_replay.py:72-74).
    mmcv (>=2.0.0rc4,<2.2.0) is not vendored in the reference tree; this is the published greedy algorithm:
    descending (stable) score order, a box is kept unless IoU with an already kept box > thr; batched = boxes
    of different groups shifted apart by ``idx * (max coordinate + 1)``.  Parity unpinned by reference
    fixtures (the reference holds none for NMS)."""
    if len(boxes) == 0:
        return torch.zeros(0, dtype=torch.int64)
    b = boxes.float().cpu()
    if idxs is not None:
        b = b + (idxs.cpu().to(b) * (b.max() + 1))[:, None]
    order = torch.sort(scores.cpu(), descending=True, stable=True).indices
    kept = []
    for i in order.tolist():
        if kept and (box_iou(b[i:i + 1], b[kept]) > iou_thresh).any():
            continue
        kept.append(i)
        if max_keep is not None and len(kept) >= max_keep:
            break
    return torch.tensor(kept, dtype=torch.int64)


# ----------------------------------------------------------------------------
# C1 / C2 collectives, stated as plain list-of-ranks arithmetic
# ----------------------------------------------------------------------------


def all_reduce_dict_sum(per_rank: Sequence[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """runner:746-749 (``mmengine.dist.all_reduce_dict`` default op 'sum'): the
    element-wise SUM over ranks of every value, keys in sorted order."""
    keys = sorted(per_rank[0].keys())
    return {k: sum(d[k] for d in per_rank) for k in keys}


def all_gather_different_shape(per_rank: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """runner:73-105: every rank ends with the list [t_rank0, t_rank1, ...]."""
    return [t.clone() for t in per_rank]


# ----------------------------------------------------------------------------
# synthetic R-50/R-101-FPN layer table
# ----------------------------------------------------------------------------


def resnet_fpn_projected_layers(depth: int = 50) -> List[Tuple[str, int, int]]:
    """(name, Cout, D=Cin*kh*kw) of every conv that is projected at the V15/C40
    configs: backbone layer2-4 (frozen_stages=1 drops conv1/layer1,
    cl_faster_rcnn_nsgp_repre_15_5_2.py:39) + FPN lateral/fpn convs
    (ignore_keys=['rpn','roi_head'], :18).  Arithmetic of
    _base_/models/faster-rcnn_r50_fpn.py (Bottleneck, style='pytorch')."""
    blocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}[depth]
    out = []
    inplanes = 256
    for li, (nb, planes) in enumerate(zip(blocks[1:], (128, 256, 512)), start=2):
        for b in range(nb):
            pre = f"backbone.layer{li}.{b}"
            out.append((f"{pre}.conv1.weight", planes, inplanes))
            out.append((f"{pre}.conv2.weight", planes, planes * 9))
            out.append((f"{pre}.conv3.weight", planes * 4, planes))
            if b == 0:
                out.append((f"{pre}.downsample.0.weight", planes * 4, inplanes))
            inplanes = planes * 4
    for i, cin in enumerate((256, 512, 1024, 2048)):
        out.append((f"neck.lateral_convs.{i}.conv.weight", 256, cin))
    for i in range(4):
        out.append((f"neck.fpn_convs.{i}.conv.weight", 256, 256 * 9))
    return out

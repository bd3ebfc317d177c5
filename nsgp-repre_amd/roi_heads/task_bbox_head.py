"""Per-task classification / regression heads (``ConvFCBBoxHeadTask`` / ``Shared2FCBBoxHeadTask``,
mmdet/models/roi_heads/bbox_heads/convfc_bbox_head_task.py:14-288, 516-529).

The task-specific part is small: one ``fc_cls`` Linear per task + one background Linear, one
``fc_reg`` Linear per task; heads of FUTURE tasks see a detached input and their logits are
forced to -inf (regression to 0) so that ``softmax`` ignores unseen classes (``:259-272``).  The
GEMMs stay on PyTorch-ROCm (hipBLASLt) -- SURVEY K8.

When mmdet is importable the class also inherits its ``BBoxHead`` (losses, targets, bbox coder)
so the reference configs build it unchanged; without mmdet it is a plain ``nn.Module`` that
carries exactly the forward path the replay loss needs.
"""
from typing import Sequence

import torch
import torch.nn as nn

from ..registry import MODELS, register

try:  # pragma: no cover - mmdet is absent in this image
    from mmdet.models.roi_heads.bbox_heads.bbox_head import BBoxHead as _Base
    _HAVE_MMDET = True
except Exception:
    _Base = nn.Module
    _HAVE_MMDET = False


class TaskHeadMixin:
    """Builds and runs the per-task predictors on top of ``x_cls`` / ``x_reg`` features."""

    def _build_task_predictors(self, cls_in: int, reg_in: int, num_classes: int, task_split: Sequence[int],
                               task_id: int, reg_class_agnostic: bool = False, box_dim: int = 4,
                               extra_channels: int = 1):
        self.task_split = list(task_split)
        self.task_id = task_id
        self.reg_class_agnostic = reg_class_agnostic
        self.fc_cls = nn.ModuleList()
        for i in range(1, len(self.task_split)):
            self.fc_cls.append(nn.Linear(cls_in, self.task_split[i] - self.task_split[i - 1]))
        self.fc_cls.append(nn.Linear(cls_in, extra_channels))            # background (:108-112)
        self.background_nums = extra_channels
        self.fc_reg = nn.ModuleList()
        if reg_class_agnostic:
            self.fc_reg.append(nn.Linear(reg_in, box_dim))
        else:
            for i in range(1, len(self.task_split)):
                self.fc_reg.append(nn.Linear(reg_in, box_dim * (self.task_split[i] - self.task_split[i - 1])))
        # freeze the heads of tasks that have not arrived yet (:130-144); background stays live
        for i, m in enumerate(self.fc_cls):
            m.requires_grad_(i + 1 <= self.task_id or i + 1 == len(self.fc_cls))
        for i, m in enumerate(self.fc_reg):
            m.requires_grad_(i + 1 <= self.task_id or self.reg_class_agnostic)

    def _task_predict(self, x_cls: torch.Tensor, x_reg: torch.Tensor):
        preds = []
        for i, module in enumerate(self.fc_cls):
            future = (i + 1 > self.task_id) and (i + 1 != len(self.task_split))
            o = module(x_cls.detach() if future else x_cls)
            if future:
                o = torch.full_like(o, float("-inf"))     # exp(-inf) = 0: unseen classes vanish (:262-264)
            preds.append(o)
        cls_score = torch.cat(preds, dim=-1)
        preds = []
        for i, module in enumerate(self.fc_reg):
            future = (i + 1 > self.task_id) and not self.reg_class_agnostic
            o = module(x_reg.detach() if future else x_reg)
            if future:
                o = torch.zeros_like(o)
            preds.append(o)
        return cls_score, torch.cat(preds, dim=-1)


class _ConvReLU(nn.Module):
    """Stand-in for mmcv's ``ConvModule(cin, cout, 3, padding=1)`` without a norm layer: same sub-module name (``conv``),
    bias on, ReLU after -- so ``named_parameters()`` yields the reference's names (``cls_convs.0.conv.weight``)."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 3, padding=1)
        self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.activate(self.conv(x))


@register(MODELS)
class ConvFCBBoxHeadTask(TaskHeadMixin, _Base):
    """The general task head (convfc_bbox_head_task.py:14-288)::

                                    /-> cls convs -> cls fcs -> per-task cls
        shared convs -> shared fcs
                                    \\-> reg convs -> reg fcs -> per-task reg

    Constructor keywords, asserts, layer names and dimension rules follow the reference (``:60-128`` and
    ``_add_conv_fc_branch`` ``:146-189``); the per-task predictors are ``TaskHeadMixin``'s.  Without mmdet it is a plain
    ``nn.Module`` (no losses / targets / bbox coder -- those are stock ``BBoxHead``) and rejects keywords it cannot honour
    instead of swallowing them."""

    def __init__(self, num_shared_convs: int = 0, num_shared_fcs: int = 0, num_cls_convs: int = 0, num_cls_fcs: int = 0,
                 num_reg_convs: int = 0, num_reg_fcs: int = 0, conv_out_channels: int = 256, fc_out_channels: int = 1024,
                 conv_cfg=None, norm_cfg=None, init_cfg=None, task_split: Sequence[int] = (0, 10, 20), task_id: int = 1,
                 in_channels: int = 256, roi_feat_size: int = 7, num_classes: int = 80, reg_class_agnostic: bool = False,
                 with_avg_pool: bool = False, **kwargs):
        if _HAVE_MMDET:  # pragma: no cover
            super().__init__(in_channels=in_channels, roi_feat_size=roi_feat_size, num_classes=num_classes,
                             reg_class_agnostic=reg_class_agnostic, with_avg_pool=with_avg_pool, init_cfg=init_cfg, **kwargs)
        else:
            if kwargs:
                raise TypeError(f"{type(self).__name__}: unsupported keyword(s) {sorted(kwargs)} (mmdet's BBoxHead is not "
                                "available in this environment to take them)")
            if conv_cfg is not None or norm_cfg is not None:
                raise NotImplementedError("conv_cfg / norm_cfg need mmcv's ConvModule")
            nn.Module.__init__(self)
            self.in_channels, self.num_classes, self.with_avg_pool = in_channels, num_classes, with_avg_pool
            self.roi_feat_area = roi_feat_size * roi_feat_size
            if with_avg_pool:
                self.avg_pool = nn.AvgPool2d(roi_feat_size)
        assert num_shared_convs + num_shared_fcs + num_cls_convs + num_cls_fcs + num_reg_convs + num_reg_fcs > 0      # :87-88
        if num_cls_convs > 0 or num_reg_convs > 0:
            assert num_shared_fcs == 0                                                                                  # :89-90
        self.num_shared_convs, self.num_shared_fcs = num_shared_convs, num_shared_fcs
        self.num_cls_convs, self.num_cls_fcs = num_cls_convs, num_cls_fcs
        self.num_reg_convs, self.num_reg_fcs = num_reg_convs, num_reg_fcs
        self.conv_out_channels, self.fc_out_channels = conv_out_channels, fc_out_channels
        self.shared_convs, self.shared_fcs, last = self._add_conv_fc_branch(num_shared_convs, num_shared_fcs, self.in_channels, True)
        self.shared_out_channels = last
        self.cls_convs, self.cls_fcs, self.cls_last_dim = self._add_conv_fc_branch(num_cls_convs, num_cls_fcs, last)
        self.reg_convs, self.reg_fcs, self.reg_last_dim = self._add_conv_fc_branch(num_reg_convs, num_reg_fcs, last)
        if num_shared_fcs == 0 and not self.with_avg_pool:                                                              # :119-123
            if num_cls_fcs == 0:
                self.cls_last_dim *= self.roi_feat_area
            if num_reg_fcs == 0:
                self.reg_last_dim *= self.roi_feat_area
        self.relu = nn.ReLU(inplace=True)
        self._build_task_predictors(self.cls_last_dim, self.reg_last_dim, num_classes, task_split, task_id, reg_class_agnostic)
        self.null_space = False

    def _add_conv_fc_branch(self, num_convs: int, num_fcs: int, in_channels: int, is_shared: bool = False):
        """convs -> avg pool (optional) -> fcs  (convfc_bbox_head_task.py:146-189)."""
        last = in_channels
        convs = nn.ModuleList()
        for i in range(num_convs):
            convs.append(_ConvReLU(last if i == 0 else self.conv_out_channels, self.conv_out_channels))
        if num_convs > 0:
            last = self.conv_out_channels
        fcs = nn.ModuleList()
        if num_fcs > 0:
            if (is_shared or self.num_shared_fcs == 0) and not self.with_avg_pool:
                last *= self.roi_feat_area
            for i in range(num_fcs):
                fcs.append(nn.Linear(last if i == 0 else self.fc_out_channels, self.fc_out_channels))
            last = self.fc_out_channels
        return convs, fcs, last

    def get_mid_features(self, x: torch.Tensor) -> torch.Tensor:
        """What the RoI dump stores: the features after the shared convs, flattened, BEFORE the shared FCs (:290-323)."""
        for conv in self.shared_convs:
            x = conv(x)
        if self.num_shared_fcs > 0 and self.with_avg_pool:
            x = self.avg_pool(x)
        return x.flatten(1) if x.dim() > 2 else x

    def _branch(self, x, convs, fcs):
        for conv in convs:
            x = conv(x)
        if x.dim() > 2:
            if self.with_avg_pool:
                x = self.avg_pool(x)
            x = x.flatten(1)
        for fc in fcs:
            x = self.relu(fc(x))
        return x

    def forward(self, x):
        # the prototype bank replays FLATTENED mid features ([K x 12544], head:490-499): convs are skipped for 2-D input
        if x.dim() > 2:
            for conv in self.shared_convs:
                x = conv(x)
        if self.num_shared_fcs > 0:
            if x.dim() > 2:
                if self.with_avg_pool:
                    x = self.avg_pool(x)
                x = x.flatten(1)
            for fc in self.shared_fcs:
                x = self.relu(fc(x))
        return self._task_predict(self._branch(x, self.cls_convs, self.cls_fcs), self._branch(x, self.reg_convs, self.reg_fcs))


@register(MODELS)
class Shared2FCBBoxHeadTask(ConvFCBBoxHeadTask):
    """flatten -> (Linear + ReLU) x 2 -> per-task cls || per-task reg: ``ConvFCBBoxHeadTask`` with two shared FCs and nothing
    else, exactly as the reference defines it (convfc_bbox_head_task.py:516-529)."""

    def __init__(self, fc_out_channels: int = 1024, *args, **kwargs):
        super().__init__(*args, num_shared_convs=0, num_shared_fcs=2, num_cls_convs=0, num_cls_fcs=0, num_reg_convs=0,
                         num_reg_fcs=0, fc_out_channels=fc_out_channels, **kwargs)
